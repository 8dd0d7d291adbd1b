#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one GPU session (gpurun_out/) into the committed summaries under profiles/.

    python tools/make_profiles.py --round r01 --bench gpurun_out/bench_c2.json --trace gpurun_out/prof_stats \\
        [--fetch gpurun_out/prof_fetch --write gpurun_out/prof_write]

--trace: directory of `rocprofv3 --kernel-trace --stats --output-format csv`; --fetch / --write: directories of the two
`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (separate runs, MI355X_MICROARCH.md HBM section)."""
import argparse, glob, json, os, shutil
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEMM_BF16 = r"gemm_kernel.*<unsigned short|mpn[134]_kernel"


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    if not hits:
        raise SystemExit(f"no {pat} under {d}")
    return hits[0]


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r01")
    ap.add_argument("--bench", required=True)
    ap.add_argument("--trace", required=True)
    ap.add_argument("--steps-in-trace", type=int, default=55)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--pmc-steps", type=int, default=45)
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    line = [l for l in open(a.bench).read().splitlines() if l.startswith("{")][-1]
    bench = json.loads(line)
    json.dump(bench, open(os.path.join(out, f"{a.round}_bench_c2.json"), "w"), indent=1)
    shutil.copy(find(a.trace, "*kernel_stats.csv"), os.path.join(out, f"{a.round}_bench_c2_kernel_stats.csv"))
    tr = pd.read_csv(find(a.trace, "*kernel_trace.csv"))
    tr["dur"] = tr.End_Timestamp - tr.Start_Timestamp
    tr["nm"] = tr.Kernel_Name.map(short)
    n = a.steps_in_trace
    t = tr.groupby("nm").dur.agg(["count", "sum", "mean"]).sort_values("sum", ascending=False)
    total = t["sum"].sum()
    g = tr[tr.Kernel_Name.str.contains(GEMM_BF16)]
    rows = ["| kernel | launches/step | avg us | ms/step | % of kernel time |", "|---|---|---|---|---|"]
    for nm, r in t.head(20).iterrows():
        rows.append(f"| `{nm}` | {r['count'] / n:.1f} | {r['mean'] / 1e3:.1f} | {r['sum'] / n / 1e6:.3f} | {100 * r['sum'] / total:.1f} |")
    traffic = None
    if a.fetch and a.write:
        def pmc(d, name):
            c = pd.read_csv(find(d, "*counter_collection.csv"))
            c = c[(c.Counter_Name == name) & c.Kernel_Name.str.contains(GEMM_BF16)]
            return c.Counter_Value.sum(), len(c)
        f, nf = pmc(a.fetch, "FETCH_SIZE")
        w, nw = pmc(a.write, "WRITE_SIZE")
        traffic = {
            "kernel": "bf16 GEMM family (gemm_kernel / gemm_kernel_glds / gemm_kernel_glds_h <unsigned short>, mpn1 / mpn3 / mpn4_kernel)",
            "command": "PPT_HIP_GRAPHS=0 rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --output-format csv -- "
                       "python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline",
            "launches_counted": int(nf), "fetch_size_kb_sum": float(f), "write_size_kb_sum": float(w),
            "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; both in KiB",
            "fetch_bytes_per_launch": 2 * f * 1024 / nf, "write_bytes_per_launch": w * 1024 / nw,
        }
        traffic["hbm_bytes_per_launch"] = traffic["fetch_bytes_per_launch"] + traffic["write_bytes_per_launch"]
        json.dump(traffic, open(os.path.join(out, f"{a.round}_gemm_hbm_traffic.json"), "w"), indent=1)
    roof = bench.get("roofline") or {}
    md = [f"# profiles (round {a.round[1:].lstrip('0') or '0'})", "",
          f"`{a.round}_bench_c2.json` -- `python bench.py` (default: C2, 1 GPU, {bench['steps']} steps, {bench['warmup']} warm-up) on one MI355X.", "",
          f"`{a.round}_bench_c2_kernel_stats.csv` -- `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps "
          f"10 --warmup 5 --no-cpu-baseline --no-roofline` on the same box ({n} steps in the trace: 40 burn-in + 5 + 10).", "",
          f"bench: **{bench['value']} clouds/s**, {bench['ms_per_step']} ms/step; roofline (bf16 GEMM family): {roof.get('achieved')} TFLOP/s = "
          f"{100 * (roof.get('frac') or 0):.1f} % of 2.5 PFLOP/s, {roof.get('launches_per_step')} launches/step, avg launch {roof.get('avg_launch_us')} us "
          "(HIP-event brackets around every launch of an eager, un-graphed pass, minus the calibrated dispatch overhead per bracket).", "",
          f"rocprofv3, same command: bf16 GEMM family {len(g) / n:.0f} launches/step, average {g.dur.mean() / 1e3:.1f} us, "
          f"{g.dur.sum() / n / 1e6:.2f} ms/step.", ""] + rows + ["",
          f"total kernel time {total / n / 1e6:.2f} ms/step over two streams (the prompt side -- text tower forward / backward, AdamW -- "
          "runs beside the point tower; both towers are hipGraph replays in the timed run)."]
    if traffic:
        md += ["", f"`{a.round}_gemm_hbm_traffic.json` -- HBM bytes of the same kernels from two PMC passes: "
               f"{traffic['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch ({traffic['fetch_bytes_per_launch'] / 1e6:.1f} read + "
               f"{traffic['write_bytes_per_launch'] / 1e6:.1f} written)."]
    open(os.path.join(out, "README.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
