#!/usr/bin/env python3
"""split16 (fp32 operands as hi + lo half pairs, ppt_gemm_params.split16) against the fp32 MFMA and an fp64 product: error and time
on the prompt chain's and the tower's shapes.     python tools/split16_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppt_amd import ops

SHAPES = [("in_proj", 817, 1536, 512), ("out_proj", 817, 512, 512), ("c_fc", 817, 2048, 512), ("c_proj", 817, 512, 2048),
          ("qkv", 16416, 1152, 384), ("fc1", 16416, 1536, 384), ("fc2", 16416, 384, 1536), ("conv3", 131072, 512, 256),
          ("dec", 32768, 384, 1536), ("sq4k", 4096, 4096, 4096)]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for name, M, N, K in SHAPES:
    A = torch.randn(M, K, generator=g).to(dev)
    A[:, ::37] *= 30.0                                      # outlier channels
    B = (torch.randn(N, K, generator=g) * 0.03).to(dev)
    rows = slice(0, min(M, 2048))
    ref = (A[rows].double() @ B.double().t())
    res = {}
    for tag, split in (("fp32", False), ("split16", True), ("split16 a0b0", (0, 0))):
        out = ops.gemm(A, B, split=split)
        err = ((out[rows].double() - ref).abs().max() / ref.abs().max()).item()
        rel = ((out[rows].double() - ref).norm() / ref.norm()).item()
        for _ in range(3):
            ops.gemm(A, B, out=out, split=split)
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            st.record()
            for _ in range(10):
                ops.gemm(A, B, out=out, split=split)
            en.record(); en.synchronize()
            ts.append(st.elapsed_time(en) * 100)
        us = sorted(ts)[2]
        print(f"{name:9s} {M:6d}x{N:5d}x{K:5d} {tag:13s} {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  max-err/max {err:.2e}  rel-L2 {rel:.2e}", flush=True)
# gradient-like A (tiny magnitudes): the floor of half's subnormals, and what the scale buys
A = (torch.randn(817, 512, generator=g) * 1e-6).to(dev)
B = (torch.randn(512, 512, generator=g) * 0.03).to(dev)
ref = A.double() @ B.double().t()
for tag, split in (("fp32", False), ("split16 a0", (0, 4)), ("split16 a12", (12, 4)), ("split16 a20", (20, 4))):
    out = ops.gemm(A, B, split=split)
    print(f"tiny A (1e-6) {tag:12s} rel-L2 {((out.double() - ref).norm() / ref.norm()).item():.2e}")
