#!/usr/bin/env python3
"""Stand-alone timing + check of the 256-row macro-tile GEMM core (csrc/gemm256.hip) against the tile loops of gemm.hip, on the
tower's shapes (C2: 16 416 token rows; C3: 32 832) and on square problems.  Interleaved rounds in ONE process (guide rule 24),
random operands (rule 25), HIP events around `reps` back-to-back launches, median over rounds.

    python tools/gemm256_bench.py [--rounds 7] [--reps 20] [--shapes fc1,fc2,qkv,proj,sq4k,sq8k] [--dtype f16]

Prints one line per (shape, path): us per launch, TFLOP/s, fraction of the 2.5 PFLOP/s dense 16-bit peak, max |err| against an
fp32 torch product of the same 16-bit operands."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppt_amd import _lib, ops  # noqa: E402

PEAK = 2500.0


def shapes(rows):
    return {
        "qkv": dict(M=rows, N=1152, K=384),
        "fc1": dict(M=rows, N=1536, K=384, bias=True, act=ops.ACT_GELU),
        "fc2": dict(M=rows, N=384, K=1536, bias=True, residual=True),
        "proj": dict(M=rows, N=384, K=384, bias=True, residual=True),
        "conv3": dict(M=rows * 32 // 1, N=512, K=256) if False else dict(M=524288, N=512, K=256),
        "dg1": dict(M=131072, N=512, K=768),                 # C5 decoder shapes (16 x 2048 points, k = 4 neighbours)
        "dg1b": dict(M=131072, N=768, K=512),
        "dg2": dict(M=131072, N=384, K=1024),
        "p1": dict(M=32768, N=384, K=1536, bias=True),
        "p0b": dict(M=32768, N=1536, K=384),
        "fc1c3": dict(M=32832, N=1536, K=384, bias=True, act=ops.ACT_GELU),
        "fc2c3": dict(M=32832, N=384, K=1536, bias=True, residual=True),
        "fc1keep": dict(M=32832, N=1536, K=384, bias=True, act=ops.ACT_GELU, out2=True),     # un-frozen block: GELU + saved pre-activation
        "dpre": dict(M=32832, N=1536, K=384, act=ops.ACT_GELU, dact=True),                  # its backward: dX with the GELU derivative
        "sq4k": dict(M=4096, N=4096, K=4096),
        "sq8k": dict(M=8192, N=8192, K=8192),
    }


def make(cfg, dtype, dev):
    g = torch.Generator(device="cpu").manual_seed(1)
    M, N, K = cfg["M"], cfg["N"], cfg["K"]
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(dtype).to(dev)
    B = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(dtype).to(dev)
    kw = {}
    if cfg.get("bias"):
        kw["bias"] = (torch.rand(N, generator=g) - 0.5).to(dev)
    if cfg.get("act"):
        kw["act"] = cfg["act"]
    if cfg.get("out2"):
        kw["out2"] = torch.empty(M, N, dtype=dtype, device=dev)
        kw["out2_pre"] = True
    if cfg.get("dact"):
        kw["dact_pre"] = torch.randn(M, N, generator=g).to(dtype).to(dev)
    out_dtype = dtype
    res = None
    if cfg.get("residual"):
        res = torch.randn(M, N, generator=g).to(dev)
        out_dtype = torch.float32
        kw["row_scale"] = torch.full((max(1, M // 513),), 1.0, device=dev)
        kw["row_scale_rows"] = 513
    return A, B, kw, out_dtype, res


def reference(A, B, kw, res):
    y = A.float() @ B.float().t()
    if "bias" in kw:
        y = y + kw["bias"]
    if kw.get("act") == ops.ACT_GELU and "dact_pre" not in kw:
        y = torch.nn.functional.gelu(y)
    if res is not None:
        y = y + res
    return y


def run(A, B, kw, out, res, core):
    if res is not None:
        return ops.gemm(A, B, out=out, residual=res, core=core, **kw)
    return ops.gemm(A, B, out=out, core=core, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rows", type=int, default=16416)
    ap.add_argument("--shapes", default="qkv,fc1,fc2,proj,sq4k,sq8k")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dtype = torch.float16 if a.dtype == "f16" else torch.bfloat16
    lib = _lib.lib()
    table = shapes(a.rows)
    results = []
    for name in a.shapes.split(","):
        cfg = table[name]
        A, B, kw, out_dtype, res = make(cfg, dtype, dev)
        M, N, K = cfg["M"], cfg["N"], cfg["K"]
        out = {c: torch.empty(M, N, dtype=out_dtype, device=dev) for c in ("old", "256")}
        # check (a 4096-row window of the reference keeps the fp32 product small)
        rows = slice(0, min(M, 4096))
        rows2 = slice(max(0, M - 300), M)
        for c in ("old", "256"):
            lib.ppt_set_gemm256(0)
            run(A, B, kw, out[c], res, "256" if c == "256" else None)
        torch.cuda.synchronize()
        errs = {}
        for c in ("old", "256"):
            e = 0.0
            for rs in (rows, rows2):
                if "dact_pre" in kw:
                    continue                                  # (the derivative epilogue is checked by the tests; here: timing + identity)
                ref = reference(A[rs], B, kw, None if res is None else res[rs])
                e = max(e, (out[c][rs].float() - ref).abs().max().item())
            errs[c] = e
        same = torch.equal(out["old"], out["256"])
        times = {"old": [], "256": []}
        for _ in range(a.rounds):
            for c in ("old", "256"):
                lib.ppt_set_gemm256(0)
                core = "256" if c == "256" else None
                for _ in range(3):
                    run(A, B, kw, out[c], res, core)
                st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                st.record()
                for _ in range(a.reps):
                    run(A, B, kw, out[c], res, core)
                en.record()
                en.synchronize()
                times[c].append(st.elapsed_time(en) * 1e3 / a.reps)
        lib.ppt_set_gemm256(-1)
        flop = 2.0 * M * N * K
        for c in ("old", "256"):
            t = sorted(times[c])
            med, best = t[len(t) // 2], t[0]
            tf = flop / (med * 1e-6) / 1e12
            r = dict(shape=name, M=M, N=N, K=K, path=c, us_median=round(med, 2), us_min=round(best, 2), tflops=round(tf, 1),
                     frac=round(tf / PEAK, 4), max_abs_err=errs[c], bit_identical_to_old=same)
            results.append(r)
            print(f"{name:6s} {M:6d}x{N:5d}x{K:5d} {c:4s} {med:8.2f} us (min {best:8.2f})  {tf:7.1f} TFLOP/s  {tf / PEAK:6.3f} of peak   "
                  f"err {errs[c]:.3e}  identical={same}", flush=True)
    if a.json:
        json.dump(results, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
