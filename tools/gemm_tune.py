#!/usr/bin/env python3
"""Compile debug variants of gemm.hip on the GPU box and time them in one process (development tool)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ppt_amd import _lib, ops

VARIANTS = {"base": [], "noloads": ["-DPPT_DBG_SKIP_LOADS"], "noepi": ["-DPPT_DBG_SKIP_EPILOGUE"],
            "noloads_noepi": ["-DPPT_DBG_SKIP_LOADS", "-DPPT_DBG_SKIP_EPILOGUE"],
            "noloads_noepi_nowrite": ["-DPPT_DBG_SKIP_LOADS", "-DPPT_DBG_SKIP_EPILOGUE", "-DPPT_DBG_SKIP_LDS_WRITE"],
            "noxcd": ["-DPPT_DBG_NO_XCD_SWIZZLE"]}
if len(sys.argv) > 1:
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in sys.argv[1:]}
libs = {}
os.makedirs("/tmp/gt", exist_ok=True)
procs = []
for name, flags in VARIANTS.items():
    out = f"/tmp/gt/lib_{name}.so"
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-slp-vectorize", "-shared", "-I" + os.path.join(ROOT, "ppt_amd/csrc"), "-o", out,
           os.path.join(ROOT, "ppt_amd/csrc/gemm.hip")] + flags
    procs.append((name, out, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
for name, out, pr in procs:
    assert pr.wait() == 0, name
    L = ctypes.CDLL(out)
    L.ppt_gemm.restype = ctypes.c_int
    L.ppt_gemm.argtypes = [ctypes.POINTER(_lib.GemmParams), ctypes.c_void_p]
    libs[name] = L

class Shim:
    def __init__(self, L): self.L = L
    def __getattr__(self, k): return getattr(self.L, k)

def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

B = 32
shapes = [("proj", B * 513, 384, 384), ("fc2", B * 513, 384, 1536), ("txt c_fc", 1480, 2048, 512), ("txt out", 1480, 512, 512)]
real = _lib.lib()
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    o32 = torch.empty(M, N, device="cuda"); res = torch.randn(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
    row = []
    for rnd in range(2):
        for v, L in libs.items():
            _lib._lib = Shim(L); _lib._lib.ppt_strerror = real.ppt_strerror
            us = timeit(lambda: ops.gemm(A, W, out=o32, bias=bias, residual=res))
            row.append((v, us))
    _lib._lib = real
    print(f"{name:9s} M={M} N={N} K={K}: " + "  ".join(f"{v}={us:.1f}us({2*M*N*K/us/1e6:.0f}TF)" for v, us in row), flush=True)

