#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the C2 step on checkpoint-like weights (the text tower on split16 products) ->
# gpurun_out/r6_c2_ckpt_like_kernel_stats.csv (copied to profiles/r06_c2_ckpt_like_kernel_stats.csv).
export PPT_BENCH_BURN_IN_S=0 PPT_BENCH_WEIGHTS=ckpt_like
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ck
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ck -o p -- python3 $ROOT/bench.py --config C2 --steps 10 --warmup 5 \
    --no-cpu-baseline --no-roofline --no-parity-mode --no-secondary > /tmp/ck.log 2>&1
cp $(find /tmp/ck -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/r6_c2_ckpt_like_kernel_stats.csv
tail -1 /tmp/ck.log | cut -c1-300
head -12 $ROOT/gpurun_out/r6_c2_ckpt_like_kernel_stats.csv | cut -c1-160
