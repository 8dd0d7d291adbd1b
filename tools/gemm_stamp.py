#!/usr/bin/env python3
"""In-kernel phase timing of the 64x64 LDS-DMA GEMM (csrc/gemm.hip, GEMM_STAMP) on the prompt chain's shapes.

    hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-slp-vectorize -DPPT_GEMM_STAMP -shared ppt_amd/csrc/gemm.hip -o tools/_build/libgemm_stamp.so
    python tools/gemm_stamp.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ppt_amd import _lib

L = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "libgemm_stamp.so"))
L.ppt_gemm.restype = ctypes.c_int
L.ppt_gemm.argtypes = [ctypes.POINTER(_lib.GemmParams), ctypes.c_void_p]


def run(M, K, N, resid):
    g = torch.Generator().manual_seed(0)
    a = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
    b = torch.randn(N).cuda()
    x = torch.randn(M, N, generator=g).cuda()
    out = torch.empty(M, N, device="cuda")
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    stamps = torch.zeros(tiles * 4 * 8, dtype=torch.int64, device="cuda")
    p = _lib.GemmParams()
    p.A, p.lda, p.B, p.ldb, p.C, p.ldc = a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N
    p.M, p.N, p.K, p.dtype, p.c_dtype = M, N, K, 1, 0
    if resid:
        p.bias, p.residual, p.ld_res = b.data_ptr(), x.data_ptr(), N
    p.batch = 1
    p.pool_min = stamps.data_ptr()
    for _ in range(3):
        stamps.zero_()
        torch.cuda.synchronize()
        assert L.ppt_gemm(ctypes.byref(p), None) == 0
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(tiles, 4, 8).astype(np.int64)
    s = s[(s[:, :, 0] > 0).all(axis=1)]
    t0 = s[:, :, 0].min()
    med = lambda v: int(np.median(v))
    print(f"M={M} K={K} N={N} resid={resid}: {tiles} workgroups; entry spread median {med(s[:, :, 0] - t0)} max {int((s[:, :, 0] - t0).max())} | "
          f"issue {med(s[:, :, 1] - s[:, :, 0])} | first slab {med(s[:, :, 2] - s[:, :, 1])} | K loop {med(s[:, :, 3] - s[:, :, 2])} | "
          f"epilogue {med(s[:, :, 4] - s[:, :, 3])} | lifetime {med(s[:, :, 4] - s[:, :, 0])} | last exit {int(s[:, :, 4].max() - t0)} (s_memtime ticks)")


if __name__ == "__main__":
    for M, K, N, r in ((817, 512, 512, True), (817, 2048, 512, True), (817, 2048, 512, False), (817, 512, 2048, False), (817, 1536, 512, False)):
        run(M, K, N, r)
