#!/usr/bin/env python3
"""In-kernel phase timing of the 256-row macro-tile GEMM core (csrc/gemm256.hip, G256_STAMP): where a workgroup's lifetime goes
on the tower's shapes.  Builds its own library with -DPPT_GEMM_STAMP (the shipped one carries no stamps).

    python tools/gemm256_stamp.py [fc1 qkv fc2 sq4k ...]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ppt_amd import _lib

SO = os.path.join(ROOT, "tools", "_build", "libgemm256_stamp.so")
os.makedirs(os.path.dirname(SO), exist_ok=True)
src = [os.path.join(ROOT, "ppt_amd", "csrc", f) for f in ("gemm256.hip", "api.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-slp-vectorize", "-Wno-inline-asm",
                       "-DPPT_GEMM_STAMP", "-shared"] + src + ["-o", SO] + os.environ.get("STAMP_FLAGS", "").split())
L = ctypes.CDLL(SO)
L.ppt_gemm256.restype = ctypes.c_int
L.ppt_gemm256.argtypes = [ctypes.POINTER(_lib.GemmParams), ctypes.c_void_p]
SHAPES = {"qkv": (16416, 1152, 384, "plain"), "fc1": (16416, 1536, 384, "gelu"), "fc2": (16416, 384, 1536, "res"),
          "proj": (16416, 384, 384, "res"), "sq4k": (4096, 4096, 4096, "plain"), "sq8k": (8192, 8192, 8192, "plain"),
          "fc1p": (16416, 1536, 384, "plain")}


def run(name):
    M, N, K, kind = SHAPES[name]
    g = torch.Generator().manual_seed(0)
    a = (torch.rand(M, K, generator=g) * 2 - 1).cuda().to(torch.float16)
    w = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).cuda().to(torch.float16)
    b = torch.randn(N).cuda()
    x = torch.randn(M, N, generator=g).cuda()
    out = torch.empty(M, N, device="cuda", dtype=torch.float32 if kind == "res" else torch.float16)
    bn = int(os.environ.get("PPT_GEMM256_BN", "0")) or (256 if (N % 256 == 0 or (N >= 1024 and N % 256 >= 128)) else 128)
    tiles = ((M + 255) // 256) * ((N + bn - 1) // bn)
    stamps = torch.zeros(tiles * 8 * 8, dtype=torch.int64, device="cuda")
    p = _lib.GemmParams()
    p.A, p.lda, p.B, p.ldb, p.C, p.ldc = a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N
    p.M, p.N, p.K, p.dtype, p.c_dtype = M, N, K, 2, (0 if kind == "res" else 2)
    if kind == "gelu":
        p.bias, p.act = b.data_ptr(), 2
    if kind == "res":
        p.bias, p.residual, p.ld_res = b.data_ptr(), x.data_ptr(), N
    p.batch = 1
    p.pool_min = stamps.data_ptr()
    for _ in range(3):
        stamps.zero_()
        torch.cuda.synchronize()
        rc = L.ppt_gemm256(ctypes.byref(p), None)
        assert rc == 0, rc
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(tiles, 8, 8).astype(np.int64)
    t0 = s[:, :, 0].min()
    med = lambda v: int(np.median(v))
    first = s[:, :, 0].min(axis=1) - t0                  # workgroup start times: the second round starts when a first-round one exits
    print(f"{name}: M={M} N={N} K={K} {kind}; {tiles} workgroups of 256x{bn}; entry (median / max) {med(first)} / {int(first.max())} | "
          f"issue {med(s[:, :, 1] - s[:, :, 0])} | first stage readable {med(s[:, :, 2] - s[:, :, 1])} | K loop {med(s[:, :, 3] - s[:, :, 2])} "
          f"({med(s[:, :, 3] - s[:, :, 2]) / (K // 32):.0f} per 32-deep stage; MFMA-bound: 1024) | epilogue {med(s[:, :, 4] - s[:, :, 3])} | "
          f"lifetime {med(s[:, :, 4] - s[:, :, 0])} | last exit {int(s[:, :, 4].max() - t0)}   (s_memtime ticks = shader cycles)")


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["qkv", "fc1p", "fc1", "fc2", "proj", "sq4k"]):
        run(n)
