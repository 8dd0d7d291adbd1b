#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of one GPU session (gpurun_out/, written by tools/profile_all.sh) into the committed summaries
under profiles/: per configuration the bench line, the rocprofv3 kernel statistics and a table; for C2 also the HBM traffic
of the GEMM family from two PMC passes.

    bash tools/profile_all.sh                       # on the GPU box: bench + rocprofv3 per configuration -> gpurun_out/prof_r02/
    python tools/make_profiles.py --round r02 --dir gpurun_out/prof_r02

Per configuration <cfg> the directory holds bench_<cfg>.json (the bench line), trace_<cfg>/ (`rocprofv3 --kernel-trace --stats
--output-format csv`), and for c2 pmc_fetch/ and pmc_write/ (`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, separate runs, as
MI355X_MICROARCH.md's HBM section prescribes)."""
import argparse, glob, json, os, shutil
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the bf16 MFMA GEMM family (what bench.py's roofline object is about): tile GEMMs, the fused mini-PointNet kernels, the
# weight-stationary short-K linears and the fused ViT MLP
GEMM_BF16 = r"gemm_kernel.*<unsigned short|mpn[134]_kernel|rowgemm_kernel|vit_mlp_kernel"
CONFIGS = ("c2", "c3", "c4", "c5", "mlp")
TRACE_STEPS = 40 + 5 + 10                   # burn-in + warm-up + timed steps of the traced command


def find(d, pat, required=True):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    if not hits:
        if required:
            raise SystemExit(f"no {pat} under {d}")
        return None
    return hits[0]


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def section(rnd, cfg, d, out):
    line = [l for l in open(os.path.join(d, f"bench_{cfg}.json")).read().splitlines() if l.startswith("{")][-1]
    bench = json.loads(line)
    json.dump(bench, open(os.path.join(out, f"{rnd}_bench_{cfg}.json"), "w"), indent=1)
    tdir = os.path.join(d, f"trace_{cfg}")
    shutil.copy(find(tdir, "*kernel_stats.csv"), os.path.join(out, f"{rnd}_bench_{cfg}_kernel_stats.csv"))
    tr = pd.read_csv(find(tdir, "*kernel_trace.csv"))
    tr["dur"] = tr.End_Timestamp - tr.Start_Timestamp
    tr["nm"] = tr.Kernel_Name.map(short)
    n = TRACE_STEPS
    t = tr.groupby("nm").dur.agg(["count", "sum", "mean"]).sort_values("sum", ascending=False)
    total = t["sum"].sum()
    g = tr[tr.Kernel_Name.str.contains(GEMM_BF16)]
    at = tr[tr.Kernel_Name.str.contains(r"at::native|rocclr")]
    roof = bench.get("roofline") or {}
    par = bench.get("parity_mode") or {}
    md = [f"## {cfg.upper()} -- {bench['config']['workload']}", "",
          f"`{rnd}_bench_{cfg}.json`: `python bench.py --config {cfg.upper()}` -> **{bench['value']} clouds/s**, {bench['ms_per_step']} ms/step "
          f"(median step {bench.get('ms_per_step_median')} ms; {bench['steps']} steps after {bench.get('burn_in')} burn-in + {bench['warmup']} warm-up)"
          + (f"; fp32 parity mode {par.get('value')} clouds/s ({par.get('ms_per_step')} ms/step)" if par else "") + ".", "",
          f"roofline (bf16 GEMM family, HIP-event brackets in an eager pass): {roof.get('achieved')} TFLOP/s = {100 * (roof.get('frac') or 0):.1f} % of 2.5 "
          f"PFLOP/s, {roof.get('launches_per_step')} launches/step, avg {roof.get('avg_launch_us')} us.", "",
          f"`{rnd}_bench_{cfg}_kernel_stats.csv`: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --config {cfg.upper()} --steps 10 "
          f"--warmup 5 --no-cpu-baseline --no-roofline --no-parity-mode` ({n} steps in the trace).  {len(tr) / n:.0f} kernels/step, "
          f"{total / n / 1e6:.2f} ms/step of kernel time over all streams; bf16 GEMM family {len(g) / n:.0f} launches/step, avg {g.dur.mean() / 1e3:.1f} us, "
          f"{g.dur.sum() / n / 1e6:.2f} ms/step; ATen / runtime-copy kernels {len(at) / n:.1f} launches/step, {at.dur.sum() / n / 1e6:.3f} ms/step.", "",
          "| kernel | launches/step | avg us | ms/step | % of kernel time |", "|---|---|---|---|---|"]
    for nm, r in t.head(16).iterrows():
        md.append(f"| `{nm[:90]}` | {r['count'] / n:.1f} | {r['mean'] / 1e3:.1f} | {r['sum'] / n / 1e6:.3f} | {100 * r['sum'] / total:.1f} |")
    if cfg == "c2" and os.path.isdir(os.path.join(d, "pmc_fetch")) and os.path.isdir(os.path.join(d, "pmc_write")):
        def pmc(sub, name):
            c = pd.read_csv(find(os.path.join(d, sub), "*counter_collection.csv"))
            c = c[(c.Counter_Name == name) & c.Kernel_Name.str.contains(GEMM_BF16)]
            return c.Counter_Value.sum(), len(c)
        f, nf = pmc("pmc_fetch", "FETCH_SIZE")
        w, nw = pmc("pmc_write", "WRITE_SIZE")
        traffic = {
            "kernel": "bf16 GEMM family (gemm_kernel* <unsigned short>, mpn1 / mpn3 / mpn4_kernel, rowgemm_kernel, vit_mlp_kernel)",
            "command": "PPT_HIP_GRAPHS=0 rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --output-format csv -- "
                       "python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode",
            "launches_counted": int(nf), "fetch_size_kb_sum": float(f), "write_size_kb_sum": float(w),
            "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; both in KiB",
            "fetch_bytes_per_launch": 2 * f * 1024 / nf, "write_bytes_per_launch": w * 1024 / nw,
        }
        traffic["hbm_bytes_per_launch"] = traffic["fetch_bytes_per_launch"] + traffic["write_bytes_per_launch"]
        json.dump(traffic, open(os.path.join(out, f"{rnd}_gemm_hbm_traffic.json"), "w"), indent=1)
        md += ["", f"`{rnd}_gemm_hbm_traffic.json`: HBM bytes of the GEMM family from two PMC passes -- {traffic['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch "
               f"({traffic['fetch_bytes_per_launch'] / 1e6:.1f} read + {traffic['write_bytes_per_launch'] / 1e6:.1f} written) over {nf} launches."]
    return md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r02")
    ap.add_argument("--dir", required=True)
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    md = [f"# profiles (round {a.round[1:].lstrip('0') or '0'})", "",
          "One MI355X, one process; every file below comes from ONE gpurun session (`bash tools/profile_all.sh`, then "
          f"`python tools/make_profiles.py --round {a.round} --dir <that session's output>`).  Round-1 files (`r01_*`) are kept for comparison.", ""]
    for cfg in CONFIGS:
        if os.path.exists(os.path.join(a.dir, f"bench_{cfg}.json")):
            md += section(a.round, cfg, a.dir, out) + [""]
    extra = os.path.join(out, f"{a.round}_notes.md")
    if os.path.exists(extra):
        md += open(extra).read().splitlines()
    open(os.path.join(out, "README.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
