#!/bin/bash
# On the GPU box (gpurun): the bench line and a rocprofv3 kernel trace of every configuration, and the two PMC passes of the
# headline configuration -> gpurun_out/prof_r02/ (tools/make_profiles.py turns that into profiles/).
#   gpurun --timeout 2400 -- 'bash tools/profile_all.sh [configs ...]'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_r02
mkdir -p "$OUT"
CFGS=${@:-C2 C3 C4 C5 MLP}
cd /tmp && export TMPDIR=/tmp
for C in $CFGS; do
    c=$(echo "$C" | tr 'A-Z' 'a-z')
    python3 "$ROOT/bench.py" --config "$C" > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err"
    tail -c 400 "$OUT/bench_$c.json"; echo
    rm -rf "$OUT/trace_$c"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$c" -o p -- python3 "$ROOT/bench.py" --config "$C" --steps 10 --warmup 5 \
        --no-cpu-baseline --no-roofline --no-parity-mode > "$OUT/trace_$c.log" 2>&1
    # keep the two summaries, drop the rest (agent info, per-domain stats): gpurun_out is capped at 64 MiB
    find "$OUT/trace_$c" -type f ! -name '*kernel_stats.csv' ! -name '*kernel_trace.csv' -delete
done
if echo "$CFGS" | grep -qw C2; then
    for P in fetch:FETCH_SIZE write:WRITE_SIZE; do
        d=${P%%:*}; ctr=${P##*:}
        rm -rf "$OUT/pmc_$d"
        PPT_HIP_GRAPHS=0 rocprofv3 --pmc "$ctr" --output-format csv -d "$OUT/pmc_$d" -o p -- python3 "$ROOT/bench.py" --steps 3 --warmup 2 \
            --no-cpu-baseline --no-roofline --no-parity-mode > "$OUT/pmc_$d.log" 2>&1
        find "$OUT/pmc_$d" -type f ! -name '*counter_collection.csv' -delete
    done
fi
du -sh "$OUT"
