#!/usr/bin/env python3
"""In-kernel phase timing of csrc/mlp_fused.hip (diagnostic build, see MLP_STAMP there).
    hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-slp-vectorize -DPPT_MLP_STAMP -shared ppt_amd/csrc/mlp_fused.hip -o tools/_build/libmlp_stamp.so
    python tools/mlp_stamp.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ppt_amd import _lib

ctypes.CDLL(os.path.join(ROOT, "ppt_amd", "csrc", "libppt_hip.so"), mode=ctypes.RTLD_GLOBAL)
L = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "libmlp_stamp.so"))
L.ppt_vit_mlp_bf16.restype = ctypes.c_int
L.ppt_vit_mlp_bf16.argtypes = [ctypes.POINTER(_lib.VitMlpParams), ctypes.c_void_p]
g = torch.Generator().manual_seed(0)
for B in (32,):
    M = B * 513
    x = torch.randn(M, 384, generator=g).cuda()
    out = torch.empty_like(x)
    gam, bet = torch.ones(384).cuda(), torch.zeros(384).cuda()
    w1 = (torch.randn(1536, 384, generator=g) * 0.05).cuda().to(torch.bfloat16)
    w2 = (torch.randn(384, 1536, generator=g) * 0.02).cuda().to(torch.bfloat16)
    b1, b2 = torch.randn(1536, generator=g).cuda(), torch.randn(384, generator=g).cuda()
    from ppt_amd import ops
    w1, w2 = ops.vit_mlp_retile(w1, w2, variant=2)
    stamps = torch.zeros(256 * 8 * 14 * 8, dtype=torch.int64, device="cuda")
    p = _lib.VitMlpParams()
    p.x, p.out, p.W1, p.W2, p.ln_w, p.ln_b, p.ln_eps = x.data_ptr(), out.data_ptr(), w1.data_ptr(), w2.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-5
    p.b1, p.b2, p.M, p.D, p.hidden = b1.data_ptr(), b2.data_ptr(), M, 384, 1536
    p.residual2 = stamps.data_ptr()
    for _ in range(3):
        stamps.zero_()
        assert L.ppt_vit_mlp_bf16(ctypes.byref(p), None) == 0
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(256, 8, 14, 8)
    t0 = s[:, :, 12, 0]
    med = lambda a: int(np.median(a))
    print(f"B={B}: chunk start -> LN done {med(s[:, :, 12, 1] - t0)} | gemm1(0)+barrier {med(s[:, :, 12, 2] - s[:, :, 12, 1])}")
    print("  slab  gemm2   gemm1-mfma   gelu   barrier   total")
    for j in range(12):
        a = s[:, :, j]
        g1 = med(a[:, :, 4] - a[:, :, 1]) if j < 11 else 0
        ge = med(a[:, :, 2] - a[:, :, 4]) if j < 11 else 0
        print(f"  {j:3d}  {med(a[:, :, 1] - a[:, :, 0]):6d}  {g1:9d}  {ge:6d}  {med(a[:, :, 3] - a[:, :, 2]):7d}  {med(a[:, :, 3] - a[:, :, 0]):7d}")
    print(f"  epilogue {med(s[:, :, 13, 1] - s[:, :, 13, 0])} | chunk total {med(s[:, :, 13, 1] - t0)} cycles")
