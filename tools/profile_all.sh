#!/bin/bash
# On the GPU box (gpurun): the bench line and a rocprofv3 kernel trace of every configuration, and the two PMC passes of the
# headline configuration -> gpurun_out/prof_r06/ (tools/make_profiles.py turns that into profiles/).
#   gpurun --timeout 2400 -- 'bash tools/profile_all.sh [configs ...]'
export PPT_BENCH_BURN_IN_S=0      # (the traces count on the 40-step burn-in: steps = 40 + warmup + K)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${PPT_PROF_DIR:-prof_r06}
mkdir -p "$OUT"
CFGS=${@:-C2 C3 C4 C5 MLP}
cd /tmp && export TMPDIR=/tmp
for C in $CFGS; do
    c=$(echo "$C" | tr 'A-Z' 'a-z')
    # C5 is profiled in the MIXED mode, as in rounds 2-5 (comparable tables): by default the Trainer's first-batch gradient self-check
    # moves part segmentation to split16 (round 6), and that number is in the C2 line's `secondary.C5` / `C5_split16`.  (exported
    # here, in the shell: nothing but the program itself may follow rocprofv3's `--`)
    if [ "$C" = C5 ]; then export PPT_GRAD_CHECK=off; else unset PPT_GRAD_CHECK; fi
    python3 "$ROOT/bench.py" --config "$C" --no-secondary > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err"
    tail -c 400 "$OUT/bench_$c.json"; echo
    rm -rf "$OUT/trace_$c"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$c" -o p -- python3 "$ROOT/bench.py" --config "$C" --steps 10 --warmup 5 \
        --no-cpu-baseline --no-roofline --no-parity-mode --no-secondary > "$OUT/trace_$c.log" 2>&1
    # keep the two summaries, drop the rest (agent info, per-domain stats): gpurun_out is capped at 64 MiB
    find "$OUT/trace_$c" -type f ! -name '*kernel_stats.csv' ! -name '*kernel_trace.csv' -delete
done
# PMC passes (each counter set in a run of its own, --pmc only: MI355X_MICROARCH.md, HBM / rocprofv3 sections): HBM bytes
# (FETCH_SIZE, WRITE_SIZE) for C2 and C4 -- the GEMM family, FPS, kNN, ball query -- and the SQ busy / MFMA-busy cycles for C2.
# Eager launches (PPT_HIP_GRAPHS=0): kernels replayed from a hipGraph carry no per-dispatch counters.
for C in C2 C4; do
    if echo "$CFGS" | grep -qw $C; then
        c=$(echo "$C" | tr 'A-Z' 'a-z')
        SETS="fetch:FETCH_SIZE write:WRITE_SIZE"
        if [ $C = C2 ]; then SETS="$SETS sq:SQ_BUSY_CYCLES,SQ_VALU_MFMA_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE grbm:GRBM_GUI_ACTIVE"; fi
        for P in $SETS; do
            d=${P%%:*}; ctr=$(echo "${P##*:}" | tr ',' ' ')
            rm -rf "$OUT/pmc_${c}_$d"
            PPT_HIP_GRAPHS=0 rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_${c}_$d" -o p -- python3 "$ROOT/bench.py" --config $C --steps 3 --warmup 2 \
                --no-cpu-baseline --no-roofline --no-parity-mode --no-secondary > "$OUT/pmc_${c}_$d.log" 2>&1
            find "$OUT/pmc_${c}_$d" -type f ! -name '*counter_collection.csv' -delete
            # keep only what the summaries need: kernel name, counter name, value (the full CSV is ~100 MB per pass)
            python3 - "$OUT/pmc_${c}_$d" <<'PY'
import glob, sys, pandas as pd
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    c = pd.read_csv(f, usecols=["Kernel_Name", "Counter_Name", "Counter_Value"])
    c["Kernel_Name"] = c.Kernel_Name.str.replace("(anonymous namespace)::", "", regex=False).str.replace("void ", "", regex=False).str.split("(").str[0]
    g = c.groupby(["Kernel_Name", "Counter_Name"]).Counter_Value.agg(["count", "sum"]).reset_index()
    g.to_csv(f.replace("counter_collection.csv", "counter_summary.csv"), index=False)
    import os; os.remove(f)
PY
        done
    fi
done
du -sh "$OUT"
