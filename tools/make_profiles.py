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
GEMM_BF16 = r"gemm_kernel.*<(unsigned short|f16_t)|gemm256_kernel|gemm_tn_kernel|mpn[134]_kernel|rowgemm_kernel|vit_mlp_kernel|vit_mlp3_kernel|text_mlp_kernel|lnlin_kernel"
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
          f"roofline (16-bit MFMA GEMM family, HIP-event brackets in an eager pass): {roof.get('achieved')} TFLOP/s = {100 * (roof.get('frac') or 0):.1f} % of 2.5 "
          f"PFLOP/s, {roof.get('launches_per_step')} launches/step, avg {roof.get('avg_launch_us')} us.", "",
          f"`{rnd}_bench_{cfg}_kernel_stats.csv`: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --config {cfg.upper()} --steps 10 "
          f"--warmup 5 --no-cpu-baseline --no-roofline --no-parity-mode` ({n} steps in the trace).  {len(tr) / n:.0f} kernels/step, "
          f"{total / n / 1e6:.2f} ms/step of kernel time over all streams; 16-bit GEMM family {len(g) / n:.0f} launches/step, avg {g.dur.mean() / 1e3:.1f} us, "
          f"{g.dur.sum() / n / 1e6:.2f} ms/step; ATen / runtime-copy kernels {len(at) / n:.1f} launches/step, {at.dur.sum() / n / 1e6:.3f} ms/step.", "",
          "| kernel | launches/step | avg us | ms/step | % of kernel time |", "|---|---|---|---|---|"]
    for nm, r in t.head(16).iterrows():
        md.append(f"| `{nm[:90]}` | {r['count'] / n:.1f} | {r['mean'] / 1e3:.1f} | {r['sum'] / n / 1e6:.3f} | {100 * r['sum'] / total:.1f} |")
    pf, pw = os.path.join(d, f"pmc_{cfg}_fetch"), os.path.join(d, f"pmc_{cfg}_write")
    if os.path.isdir(pf) and os.path.isdir(pw):
        import re
        def summ(sub):
            return pd.read_csv(find(sub, "*counter_summary.csv"))
        f, w = summ(pf), summ(pw)
        f = f[f.Counter_Name == "FETCH_SIZE"].set_index("Kernel_Name")
        w = w[w.Counter_Name == "WRITE_SIZE"].set_index("Kernel_Name")
        fam_f, fam_w = f[f.index.str.contains(GEMM_BF16)], w[w.index.str.contains(GEMM_BF16)]
        nf, nw = int(fam_f["count"].sum()), int(fam_w["count"].sum())
        if cfg == "c2" and nf:
            traffic = {
                "kernel": "16-bit MFMA GEMM family (gemm_kernel* <unsigned short | f16_t>, gemm_tn_kernel, mpn1 / mpn3 / mpn4_kernel, rowgemm_kernel, lnlin_kernel, vit_mlp3_kernel, text_mlp_kernel)",
                "command": "PPT_HIP_GRAPHS=0 rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --output-format csv -- "
                           "python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-parity-mode --no-secondary",
                "launches_counted": nf, "fetch_size_kb_sum": float(fam_f["sum"].sum()), "write_size_kb_sum": float(fam_w["sum"].sum()),
                "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; both in KiB",
                "fetch_bytes_per_launch": 2 * float(fam_f["sum"].sum()) * 1024 / nf, "write_bytes_per_launch": float(fam_w["sum"].sum()) * 1024 / nw,
            }
            traffic["hbm_bytes_per_launch"] = traffic["fetch_bytes_per_launch"] + traffic["write_bytes_per_launch"]
            # the family's rate from the rocprofv3 KERNEL TRACE of the same command (kernel-only durations; bench.py's live figure
            # comes from HIP-event brackets and reads ~8 % higher): executed FLOPs per step from the bench line's own accounting
            ex = (roof.get("executed_gflop_per_launch") or 0) * (roof.get("launches_per_step") or 0)
            fam_ms = g.dur.sum() / n / 1e6
            if ex and fam_ms:
                traffic["rocprof_family"] = {"launches_per_step": round(len(g) / n, 1), "avg_us": round(g.dur.mean() / 1e3, 2),
                                             "ms_per_step": round(fam_ms, 3), "executed_gflop_per_step": round(ex, 1),
                                             "achieved_tflops": round(ex / fam_ms, 1), "frac": round(ex / fam_ms / 2500.0, 4),
                                             "source": f"profiles/{rnd}_bench_c2_kernel_stats.csv (rocprofv3 --kernel-trace --stats)"}
            json.dump(traffic, open(os.path.join(out, f"{rnd}_gemm_hbm_traffic.json"), "w"), indent=1)
            md += ["", f"`{rnd}_gemm_hbm_traffic.json`: HBM bytes of the GEMM family from two PMC passes -- {traffic['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch "
                   f"({traffic['fetch_bytes_per_launch'] / 1e6:.1f} read + {traffic['write_bytes_per_launch'] / 1e6:.1f} written) over {nf} launches."]
        # per-kernel HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes), joined with the kernel-trace average duration
        dur = tr.groupby("nm").dur.mean()
        rows = []
        for k in f.index:
            if k not in w.index:
                continue
            fb, wb = 2 * f.loc[k, "sum"] * 1024 / f.loc[k, "count"], w.loc[k, "sum"] * 1024 / w.loc[k, "count"]
            us = dur.get(k, float("nan")) / 1e3
            rows.append((k, int(f.loc[k, "count"]), fb, wb, us, (fb + wb) / (us * 1e-6) / 1e9 if us == us and us > 0 else float("nan")))
        rows.sort(key=lambda r: -(r[2] + r[3]) * r[1])
        pd.DataFrame(rows, columns=["kernel", "launches_counted", "fetch_bytes_per_launch(x2)", "write_bytes_per_launch", "avg_us(kernel trace)", "GB_per_s"]) \
            .to_csv(os.path.join(out, f"{rnd}_{cfg}_hbm_per_kernel.csv"), index=False)
        md += ["", f"`{rnd}_{cfg}_hbm_per_kernel.csv`: HBM bytes per launch of every kernel (PMC FETCH_SIZE x 2 + WRITE_SIZE, separate passes, eager launches) beside its "
               "kernel-trace average duration.  Index kernels and the largest movers:", "",
               "| kernel | read MB / launch | written MB / launch | avg us | GB/s | % of 8 TB/s |", "|---|---|---|---|---|---|"]
        want = [r for r in rows if re.search(r"fps_kernel|knn_group|ball_query", r[0])] + [r for r in rows if not re.search(r"fps_kernel|knn_group|ball_query", r[0])][:8]
        for k, n_, fb, wb, us, gbs in want:
            md.append(f"| `{k[:80]}` | {fb / 1e6:.3f} | {wb / 1e6:.3f} | {us:.1f} | {gbs:.0f} | {gbs / 80:.2f} |")
    psq, pg = os.path.join(d, f"pmc_{cfg}_sq"), os.path.join(d, f"pmc_{cfg}_grbm")
    if os.path.isdir(psq) and os.path.isdir(pg):
        sq = pd.read_csv(find(psq, "*counter_summary.csv")).pivot(index="Kernel_Name", columns="Counter_Name", values="sum")
        cnt = pd.read_csv(find(psq, "*counter_summary.csv")).groupby("Kernel_Name")["count"].first()
        gr = pd.read_csv(find(pg, "*counter_summary.csv"))
        gr = gr[gr.Counter_Name == "GRBM_GUI_ACTIVE"].set_index("Kernel_Name")
        rows = []
        for k in sq.index:
            if k not in gr.index or not re.search(GEMM_BF16 + r"|attn_", k):
                continue
            cyc = gr.loc[k, "sum"] / gr.loc[k, "count"] / 8.0              # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            mf = sq.loc[k, "SQ_VALU_MFMA_BUSY_CYCLES"] / cnt[k]
            rows.append((k, int(cnt[k]), cyc, mf, mf / (cyc * 1024.0) if cyc > 0 else float("nan"), sq.loc[k, "SQ_BUSY_CYCLES"] / cnt[k],
                         sq.loc[k, "SQ_WAVE_CYCLES"] / cnt[k], sq.loc[k, "SQ_WAIT_ANY"] / cnt[k], sq.loc[k, "SQ_WAIT_INST_ANY"] / cnt[k],
                         sq.loc[k, "SQ_ACTIVE_INST_ANY"] / cnt[k], sq.loc[k, "SQ_LDS_BANK_CONFLICT"] / cnt[k], sq.loc[k, "SQ_LDS_IDX_ACTIVE"] / cnt[k]))
        rows.sort(key=lambda r: -r[2] * r[1])
        cols = ["kernel", "launches_counted", "gpu_cycles_per_launch(GRBM_GUI_ACTIVE/8)", "SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy_frac(=busy/(cycles*1024 SIMDs))",
                "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
        pd.DataFrame(rows, columns=cols).to_csv(os.path.join(out, f"{rnd}_{cfg}_mfma_util.csv"), index=False)
        md += ["", f"`{rnd}_{cfg}_mfma_util.csv`: matrix-pipe utilisation of the MFMA kernels from the SQ counters (one `--pmc` pass of eight SQ counters, one of "
               "GRBM_GUI_ACTIVE): MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1 024 SIMDs); a fully issued bf16 / fp16 "
               "MFMA stream reads 1.0 = 2.5 PFLOP/s.", "", "| kernel | launches | cycles / launch | MFMA-busy fraction | wave cycles waiting (WAIT_ANY / WAVE_CYCLES) | LDS conflict / LDS active |",
               "|---|---|---|---|---|---|"]
        for r in rows[:10]:
            md.append(f"| `{r[0][:80]}` | {r[1]} | {r[2]:.0f} | {r[4]:.3f} | {r[7] / r[6] if r[6] else float('nan'):.2f} | {r[10] / r[11] if r[11] else float('nan'):.3f} |")
    return md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r03")
    ap.add_argument("--dir", required=True)
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    md = [f"# profiles (round {a.round[1:].lstrip('0') or '0'})", "",
          "One MI355X, one process; every file below comes from ONE gpurun session (`bash tools/profile_all.sh`, then "
          f"`python tools/make_profiles.py --round {a.round} --dir <that session's output>`).  Earlier rounds' files (`r01_*`, `r02_*`) are kept for comparison.", ""]
    for cfg in CONFIGS:
        if os.path.exists(os.path.join(a.dir, f"bench_{cfg}.json")):
            md += section(a.round, cfg, a.dir, out) + [""]
    extra = os.path.join(out, f"{a.round}_notes.md")
    if os.path.exists(extra):
        md += open(extra).read().splitlines()
    # the hand-written part of the README (documents that are not generated from a trace) survives a regeneration
    readme = os.path.join(out, "README.md")
    keep = ""
    if os.path.exists(readme):
        old = open(readme).read()
        i = old.find("## Round-5 documents beside the generated tables")
        keep = "\n" + old[i:] if i >= 0 else ""
    open(readme, "w").write("\n".join(md) + "\n" + keep)
    print("\n".join(md))


if __name__ == "__main__":
    main()
