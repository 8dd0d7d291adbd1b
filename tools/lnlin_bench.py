#!/usr/bin/env python3
"""Dev tool (round 6): norm1 + qkv of a frozen block -- csrc/lnlin.hip (rows stationary, weight streamed) against csrc/rowgemm.hip
(weight stationary) at C2's and C3's row counts.   python3 tools/lnlin_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppt_amd import ops

g = torch.Generator().manual_seed(0)
dt = torch.float16


def timeit(fn, n=20, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * a.elapsed_time(b) / n)
    return best


print("| rows | rowgemm (us) | lnlin (us) | TFLOP/s |")
print("|---|---|---|---|")
for M in (16416, 32832, 8208):
    x = (torch.randn(M, 384, generator=g) * 2).cuda()
    gam, bet = torch.ones(384).cuda(), torch.zeros(384).cuda()
    w = (torch.randn(1152, 384, generator=g) * 384 ** -0.5).cuda().to(dt)
    wt = ops.lnlin_retile(w)
    t_r = timeit(lambda: ops.rowgemm(x, w, ln=(gam, bet)))
    t_l = timeit(lambda: ops.lnlin(x, wt, (gam, bet)))
    print(f"| {M} | {t_r:.1f} | {t_l:.1f} | {2.0 * M * 1152 * 384 / t_l / 1e6:.0f} |", flush=True)
