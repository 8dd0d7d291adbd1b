#!/usr/bin/env python3
"""Dev tool (round 6): the text tower's MLP half -- ppt_text_mlp_pair (one launch) against the two launches it replaces (ppt_gemm
with the QuickGELU epilogue, then the split-K c_proj product), forward and backward, at the prompt chain's sizes.
    python3 tools/text_mlp_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppt_amd import ops

dt = torch.float16
g = torch.Generator().manual_seed(0)


def timeit(fn, iters=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters


print("| rows | direction | two launches (us) | one launch (us) |")
print("|---|---|---|---|")
for M in (817, 317, 1037, 3080):
    a = torch.randn(M, 512, generator=g).cuda().to(dt)
    w1 = (torch.randn(2048, 512, generator=g) * 512 ** -0.5).cuda().to(dt)
    w2 = (torch.randn(512, 2048, generator=g) * 2048 ** -0.5).cuda().to(dt)
    b1 = torch.randn(2048, generator=g).cuda()
    pre = torch.empty((M, 2048), dtype=dt, device="cuda")
    w1t, w2t = ops.text_mlp_retile(w1, w2)
    w2T, w1T = w2.t().contiguous(), w1.t().contiguous()           # [2048, 512] / [512, 2048]: the dX operands
    b1t, b2t = ops.text_mlp_retile(w2T, w1T)

    def two_fwd():
        f = ops.gemm(a, w1, out_dtype=dt, bias=b1, act=ops.ACT_QUICKGELU, out2=pre, out2_pre=True)
        return ops.gemm_splitk(f, w2, 4)

    def two_bwd():
        d_pre = ops.gemm(a, w2T, out_dtype=dt, act=ops.ACT_QUICKGELU, dact_pre=pre)
        return ops.gemm_splitk(d_pre, w1T, 4)
    t2f, t1f = timeit(two_fwd), timeit(lambda: ops.text_mlp_pair(a, w1t, w2t, bias=b1, pre=pre))
    t2b, t1b = timeit(two_bwd), timeit(lambda: ops.text_mlp_pair(a, b1t, b2t, pre=pre, backward=True))
    print(f"| {M} | forward | {t2f:.1f} | {t1f:.1f} |")
    print(f"| {M} | backward | {t2b:.1f} | {t1b:.1f} |", flush=True)
