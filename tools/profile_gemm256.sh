#!/bin/bash
# On the GPU box: the stand-alone numbers of the 256-row macro-tile GEMM core (csrc/gemm256.hip) -- HIP-event A/B table, rocprofv3
# kernel trace of the same tool, SQ counters (MFMA-busy, waits, LDS conflicts) and in-kernel stamps -> gpurun_out/prof_r05/gemm256_*.
export PPT_BENCH_BURN_IN_S=0      # (the traces count on the 40-step burn-in: steps = 40 + warmup + K)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${PPT_PROF_DIR:-prof_r05}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SH=qkv,fc1,fc2,proj,fc1c3,fc2c3,p0b,p1,dg1,dg1b,dg2,sq4k,sq8k
python3 "$ROOT/tools/gemm256_bench.py" --rounds 5 --reps 10 --shapes $SH --json "$OUT/gemm256_bench.json" > "$OUT/gemm256_bench.log" 2>&1
rm -rf "$OUT/gemm256_trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/gemm256_trace" -o p -- python3 "$ROOT/tools/gemm256_bench.py" --rounds 2 --reps 10 --shapes $SH > "$OUT/gemm256_trace.log" 2>&1
find "$OUT/gemm256_trace" -type f ! -name '*kernel_stats.csv' -delete
for P in "sq:SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "grbm:GRBM_GUI_ACTIVE"; do
    d=${P%%:*}; ctr=${P##*:}
    rm -rf "$OUT/gemm256_pmc_$d"
    rocprofv3 --pmc $ctr --output-format csv -d "$OUT/gemm256_pmc_$d" -o p -- python3 "$ROOT/tools/gemm256_bench.py" --rounds 1 --reps 3 --shapes $SH > "$OUT/gemm256_pmc_$d.log" 2>&1
    python3 - "$OUT/gemm256_pmc_$d" <<'PY'
import glob, sys, os, pandas as pd
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    c = pd.read_csv(f, usecols=["Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
    c["Kernel_Name"] = c.Kernel_Name.str.replace("(anonymous namespace)::", "", regex=False).str.replace("void ", "", regex=False).str.split("(").str[0]
    c = c[c.Kernel_Name.str.contains("gemm")]
    g = c.groupby(["Kernel_Name", "Grid_Size", "Counter_Name"]).Counter_Value.agg(["count", "sum"]).reset_index()
    g.to_csv(os.path.join(sys.argv[1], "counter_summary.csv"), index=False)
    os.remove(f)
PY
    find "$OUT/gemm256_pmc_$d" -type f ! -name 'counter_summary.csv' -delete
done
python3 "$ROOT/tools/gemm256_stamp.py" qkv fc1p fc1 fc2 proj sq4k > "$OUT/gemm256_stamps.log" 2>&1
du -sh "$OUT"; tail -30 "$OUT/gemm256_bench.log"
