#!/usr/bin/env python3
"""Time the weight-stationary short-K linears (csrc/rowgemm.hip) against the tile-loop path (LayerNorm kernel + ppt_gemm)
on the shapes of one C2 / C3 PointBERT block and of the text tower.   python tools/rowgemm_bench.py [walkers ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppt_amd import ops


def timeit(fn, n=20, reps=5):
    """GPU time per call in us: n calls captured in ONE hipGraph (eager Python launches of these kernels are host-bound)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        g.replay()
        en.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * st.elapsed_time(en) / n)
    return best


def main():
    walkers = [int(a) for a in sys.argv[1:]] or [0]
    g = torch.Generator().manual_seed(0)
    for B, name in ((32, "C2"), (64, "C3")):
        M = B * 513
        x = torch.randn(M, 384, generator=g).cuda()
        gam, bet = torch.ones(384).cuda(), torch.zeros(384).cuda()
        dp = torch.ones(B).cuda()
        for nm, N, act, resid in (("ln+qkv", 1152, ops.ACT_NONE, False), ("ln+fc1+gelu", 1536, ops.ACT_GELU, False),
                                  ("proj+res", 384, ops.ACT_NONE, True)):
            w = (torch.randn(N, 384, generator=g) * 0.05).cuda().to(torch.bfloat16)
            b = torch.randn(N, generator=g).cuda()
            a16 = torch.randn(M, 384, generator=g).cuda().to(torch.bfloat16)
            xo = x.clone()
            if resid:
                old = lambda: ops.gemm(a16, w, out=xo, bias=b, row_scale=dp, row_scale_rows=513, residual=xo)
                news = {wk: (lambda wk=wk: ops.rowgemm(a16, w, bias=b, residual=xo, out=xo, row_scale=dp, row_scale_rows=513, walkers=wk))
                        for wk in walkers}
            else:
                def old():
                    h, _, _ = ops.layernorm_fwd(x, gam, bet, torch.bfloat16)
                    return ops.gemm(h, w, out_dtype=torch.bfloat16, bias=b, act=act)
                news = {wk: (lambda wk=wk: ops.rowgemm(x, w, ln=(gam, bet), bias=b, act=act, walkers=wk)) for wk in walkers}
            t_old = timeit(old)
            fl = 2.0 * M * N * 384
            line = f"{name} {nm:12s} M={M} N={N}: tile path {t_old:7.1f} us ({fl / t_old / 1e6:6.0f} TF)"
            for wk, fn in news.items():
                t = timeit(fn)
                line += f" | rowgemm[w={wk}] {t:7.1f} us ({fl / t / 1e6:6.0f} TF)"
            print(line, flush=True)
    for B in (32, 64):                                             # the fused MLP against LayerNorm + fc1 + fc2
        M = B * 513
        x = torch.randn(M, 384, generator=g).cuda()
        gam, bet = torch.ones(384).cuda(), torch.zeros(384).cuda()
        w1 = (torch.randn(1536, 384, generator=g) * 0.05).cuda().to(torch.bfloat16)
        w2 = (torch.randn(384, 1536, generator=g) * 0.02).cuda().to(torch.bfloat16)
        b1, b2, dp = torch.randn(1536, generator=g).cuda(), torch.randn(384, generator=g).cuda(), torch.ones(B).cuda()
        xo = x.clone()

        def old():
            h, _, _ = ops.layernorm_fwd(xo, gam, bet, torch.bfloat16)
            f = ops.gemm(h, w1, out_dtype=torch.bfloat16, bias=b1, act=ops.ACT_GELU)
            ops.gemm(f, w2, out=xo, bias=b2, row_scale=dp, row_scale_rows=513, residual=xo)
        t_old = timeit(old)
        w1t, w2t = ops.vit_mlp_retile(w1, w2)
        t_new = timeit(lambda: ops.vit_mlp(xo, w1t, b1, w2t, b2, (gam, bet), row_scale=dp, row_scale_rows=513))
        fl = 4.0 * M * 384 * 1536
        print(f"MLP B={B} M={M}: LN + fc1 + fc2 {t_old:7.1f} us ({fl / t_old / 1e6:6.0f} TF) | fused {t_new:7.1f} us ({fl / t_new / 1e6:6.0f} TF)", flush=True)
    for M in (1480, 817):
        x = torch.randn(M, 512, generator=g).cuda()
        gam, bet = torch.ones(512).cuda(), torch.zeros(512).cuda()
        for nm, N, act in (("ln+in_proj", 1536, ops.ACT_NONE), ("ln+c_fc+qgelu", 2048, ops.ACT_QUICKGELU)):
            w = (torch.randn(N, 512, generator=g) * 0.05).cuda().to(torch.bfloat16)
            b = torch.randn(N, generator=g).cuda()

            def old():
                h, _, _ = ops.layernorm_fwd(x, gam, bet, torch.bfloat16)
                return ops.gemm(h, w, out_dtype=torch.bfloat16, bias=b, act=act)
            t_old, t_new = timeit(old), timeit(lambda: ops.rowgemm(x, w, ln=(gam, bet), bias=b, act=act))
            print(f"text {nm:14s} M={M} N={N}: tile path {t_old:6.1f} us | rowgemm {t_new:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
