#!/usr/bin/env python3
"""Dev tool (round 6): the fused frozen-block kernel alone -- csrc/mlp_fused.hip (variant 2) against csrc/mlp_fused3.hip (variant 3),
with and without the proj prologue, at C2's (16 416 rows) and C3's (65 664 / 2 = 32 832 rows per ... ) sizes; HIP-event timing over
back-to-back launches, TFLOP/s on the model's FLOPs, and the largest difference between the two variants' outputs.
    python3 tools/vit_mlp_bench.py [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppt_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
D, Hd = 384, 1536
g = torch.Generator().manual_seed(0)
dt = torch.float16


def timeit(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / iters


print("| rows | form | variant | us | TFLOP/s | frac of 2.5 PF | max |v3 - v2| |")
print("|---|---|---|---|---|---|---|")
for M in (16416, 32832, 65664):
    x = (torch.randn(M, D, generator=g) * 2).cuda()
    a = torch.randn(M, D, generator=g).cuda().to(dt)
    wp, bp = (torch.randn(D, D, generator=g) * D ** -0.5).cuda().to(dt), (0.1 * torch.randn(D, generator=g)).cuda()
    gam, bet = (1 + 0.1 * torch.randn(D, generator=g)).cuda(), (0.1 * torch.randn(D, generator=g)).cuda()
    w1, b1 = (torch.randn(Hd, D, generator=g) * D ** -0.5).cuda().to(dt), (0.1 * torch.randn(Hd, generator=g)).cuda()
    w2, b2 = (torch.randn(D, Hd, generator=g) * Hd ** -0.5).cuda().to(dt), (0.1 * torch.randn(D, generator=g)).cuda()
    dp = (torch.floor(0.9 + torch.rand((M + 512) // 513, generator=g)) / 0.9).cuda()
    pos = torch.randn(M, D, generator=g).cuda()
    wpt = ops.vit_proj_retile(wp)
    outs = {}
    for form in ("mlp", "proj+mlp"):
        flops = 4.0 * M * D * Hd + (2.0 * M * D * D if form != "mlp" else 0.0)
        for variant in (2, 3):
            w1t, w2t = ops.vit_mlp_retile(w1, w2, variant=variant)
            out = torch.empty_like(x)
            kw = dict(out=out, row_scale=dp, row_scale_rows=513, residual2=pos)
            if form != "mlp":
                kw["proj"] = (a, wpt, bp, dp, 513)
            fn = lambda: ops.vit_mlp(x, w1t, b1, w2t, b2, (gam, bet), **kw)
            us = timeit(fn)
            outs[(form, variant)] = out.clone()
            diff = (outs[(form, 3)] - outs[(form, 2)]).abs().max().item() if variant == 3 else float("nan")
            print(f"| {M} | {form} | {variant} | {us:.1f} | {flops / us / 1e6:.0f} | {flops / us / 1e6 / 2500:.3f} | {diff:.2e} |", flush=True)
