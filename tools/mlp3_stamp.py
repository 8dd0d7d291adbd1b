#!/usr/bin/env python3
"""In-kernel phase timing of csrc/mlp_fused3.hip (diagnostic build, see MLP3_STAMP there): s_memtime stamps of the first chunk of
every workgroup, median over workgroups x waves.
    hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-slp-vectorize -DPPT_MLP3_STAMP -Iinclude -shared ppt_amd/csrc/mlp_fused3.hip -o tools/_build/libmlp3_stamp.so
    python3 tools/mlp3_stamp.py [workgroups] [proj]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ppt_amd import _lib, ops

ctypes.CDLL(os.path.join(ROOT, "ppt_amd", "csrc", "libppt_hip.so"), mode=ctypes.RTLD_GLOBAL)      # (ppt_stream, ppt_cu_count ... live there)
L = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", os.environ.get("MLP3_LIB", "libmlp3_stamp.so")))
L.ppt_vit_mlp3_bf16.restype = ctypes.c_int
L.ppt_vit_mlp3_bf16.argtypes = [ctypes.POINTER(_lib.VitMlpParams), ctypes.c_void_p]
wgs = int(sys.argv[1]) if len(sys.argv) > 1 else 206
with_proj = len(sys.argv) > 2 and sys.argv[2] == "proj"
g = torch.Generator().manual_seed(0)
M = int(os.environ.get("MLP3_ROWS", 32 * 513))
dt = torch.float16
x = torch.randn(M, 384, generator=g).cuda()
out = torch.empty_like(x)
gam, bet = torch.ones(384).cuda(), torch.zeros(384).cuda()
w1 = (torch.randn(1536, 384, generator=g) * 0.05).cuda().to(dt)
w2 = (torch.randn(384, 1536, generator=g) * 0.02).cuda().to(dt)
b1, b2 = torch.randn(1536, generator=g).cuda(), torch.randn(384, generator=g).cuda()
a = torch.randn(M, 384, generator=g).cuda().to(dt)
wp = ops.vit_proj_retile((torch.randn(384, 384, generator=g) * 0.05).cuda().to(dt))
w1, w2 = ops.vit_mlp_retile(w1, w2, variant=3)
stamps = torch.zeros(256 * 8 * 8 * 8, dtype=torch.int64, device="cuda")
p = _lib.VitMlpParams()
p.x, p.out, p.W1, p.W2, p.ln_w, p.ln_b, p.ln_eps = x.data_ptr(), out.data_ptr(), w1.data_ptr(), w2.data_ptr(), gam.data_ptr(), bet.data_ptr(), 1e-5
p.b1, p.b2, p.M, p.D, p.hidden, p.dtype, p.workgroups = b1.data_ptr(), b2.data_ptr(), M, 384, 1536, 2, wgs
if with_proj:
    p.proj_a, p.proj_W = a.data_ptr(), wp.data_ptr()
p.residual2 = stamps.data_ptr()
for _ in range(3):
    stamps.zero_()
    assert L.ppt_vit_mlp3_bf16(ctypes.byref(p), None) == 0
    torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(256, 8, 8, 8)[:wgs]
med = lambda v: int(np.median(v))
t0 = s[:, :, 6, 0]
print(f"workgroups {wgs}, rows per chunk {-(-M // wgs)}, proj prologue: {with_proj}")
print(f"chunk start -> prologue done {med(s[:, :, 6, 1] - t0)} | gemm1(0) + gelu(0) + barrier {med(s[:, :, 6, 2] - s[:, :, 6, 1])}")
pro = s[:, :, 5]
print("prologue: first barrier %d | loads + constants issued %d | a -> LDS + barrier %d | proj MFMAs %d | combine (waits for x) %d | rings issued + statistics + image %d | last barrier %d" % (
    med(pro[:, :, 1] - t0), med(pro[:, :, 2] - pro[:, :, 1]), med(pro[:, :, 3] - pro[:, :, 2]) if with_proj else 0, med(pro[:, :, 4] - pro[:, :, 3]) if with_proj else 0,
    med(pro[:, :, 5] - (pro[:, :, 4] if with_proj else pro[:, :, 2])), med(pro[:, :, 6] - pro[:, :, 5]), med(s[:, :, 6, 1] - pro[:, :, 6])))
print("  slab   gemm1(j+1)   gemm2(j)+gelu(j+1)   barrier   total")
for j in range(5):
    v = s[:, :, j]
    print(f"  {j:3d}  {med(v[:, :, 1] - v[:, :, 0]):10d}  {med(v[:, :, 2] - v[:, :, 1]):18d}  {med(v[:, :, 3] - v[:, :, 2]):8d}  {med(v[:, :, 3] - v[:, :, 0]):7d}")
print(f"    5  {'-':>10}  {med(s[:, :, 7, 0] - s[:, :, 5, 0]):18d}   (last slab: no GELU)")
print(f"epilogue {med(s[:, :, 7, 1] - s[:, :, 7, 0])} | chunk total {med(s[:, :, 7, 1] - t0)} cycles | spread of chunk totals over workgroups: "
      f"{int((s[:, 0, 7, 1] - s[:, 0, 6, 0]).min())} .. {int((s[:, 0, 7, 1] - s[:, 0, 6, 0]).max())}")
lo, hi = s[:, :, 6, 0].min(), s[:, :, 7, 1].max()
print(f"first chunk start -> last chunk end across the launch: {int(hi - lo)} shader cycles (s_memtime)")
