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
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", "-I" + os.path.join(ROOT, "ppt_amd/csrc"), "-o", out,
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
shapes = [("qkv", B * 513, 1152, 384), ("fc2", B * 513, 384, 1536), ("txt c_fc", 3080, 2048, 512), ("conv3", B * 16384, 512, 256)]
real = _lib.lib()
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    row = []
    for rnd in range(2):
        for v, L in libs.items():
            _lib._lib = Shim(L); _lib._lib.ppt_strerror = real.ppt_strerror
            us = timeit(lambda: ops.gemm(A, W, out=out))
            row.append((v, us))
    _lib._lib = real
    print(f"{name:9s} M={M} N={N} K={K}: " + "  ".join(f"{v}={us:.1f}us({2*M*N*K/us/1e6:.0f}TF)" for v, us in row), flush=True)

# mini-PointNet launches with their real prologues / epilogues
Mm = B * 512 * 32
pts = torch.randn(Mm, 3, device="cuda") * 0.1
w1 = torch.randn(128, 3, device="cuda"); b1 = torch.randn(128, device="cuda")
sc = torch.rand(128, device="cuda") + 0.5; sh = torch.randn(128, device="cuda")
W2 = torch.randn(256, 128, device="cuda").bfloat16(); y2 = torch.empty(Mm, 256, device="cuda", dtype=torch.bfloat16)
gmax = torch.empty(Mm // 32, 256, device="cuda", dtype=torch.bfloat16)
y3 = torch.randn(Mm, 512, device="cuda").bfloat16(); W4 = torch.randn(256, 512, device="cuda").bfloat16()
sc2 = torch.rand(512, device="cuda") + 0.5; sh2 = torch.randn(512, device="cuda")
tok = torch.empty(Mm // 32, 256, device="cuda", dtype=torch.bfloat16)
W3 = torch.randn(512, 256, device="cuda").bfloat16(); gt = torch.randn(Mm // 32, 512, device="cuda")
cs = torch.empty(Mm // 32, 512, device="cuda"); cq = torch.empty_like(cs); o3 = torch.empty(Mm, 512, device="cuda", dtype=torch.bfloat16)
cases = [("conv2 CONV1+pool", 2 * Mm * 256 * 128, lambda: ops.gemm(None, W2, out=y2, a_mode=ops.A_CONV1, pts=pts, w1=w1, b1=b1, a_scale=sc, a_shift=sh, pool_max=gmax, pool_rows=32)),
         ("conv3 gadd+stats", 2 * Mm * 512 * 256, lambda: ops.gemm(y2, W3, out=o3, group_add=gt, group_rows=32, col_stats=(cs, cq))),
         ("conv3 gadd only", 2 * Mm * 512 * 256, lambda: ops.gemm(y2, W3, out=o3, group_add=gt, group_rows=32)),
         ("conv3 plain", 2 * Mm * 512 * 256, lambda: ops.gemm(y2, W3, out=o3)),
         ("conv4 AFFINE+pool", 2 * Mm * 256 * 512, lambda: ops.gemm(y3, W4, a_mode=ops.A_AFFINE_RELU, a_scale=sc2, a_shift=sh2, pool_max=tok, pool_rows=32, want_out=False))]
for name, fl, fn in cases:
    row = []
    for v, L in libs.items():
        _lib._lib = Shim(L); _lib._lib.ppt_strerror = real.ppt_strerror
        us = timeit(fn, 10)
        row.append((v, us))
    _lib._lib = real
    print(f"{name:18s}: " + "  ".join(f"{v}={us:.1f}us({fl/us/1e6:.0f}TF)" for v, us in row), flush=True)
