#!/bin/bash
# Dev tool (GPU box): ordered kernel sequence of one steady-state step of a bench configuration, per stream, with gaps.
#   bash tools/step_sequence.sh C5 [marker=adamw]
export PPT_BENCH_BURN_IN_S=0      # (the traces count on the 40-step burn-in: steps = 40 + warmup + K)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
C=${1:-C5}; M=${2:-adamw}
rm -rf /tmp/ss
rocprofv3 --kernel-trace --output-format csv -d /tmp/ss -o p -- python3 $ROOT/bench.py --config $C --steps 10 --warmup 5 \
    --no-cpu-baseline --no-roofline --no-parity-mode --no-secondary > /tmp/ss.log 2>&1
python3 $ROOT/tools/chain_sequence.py /tmp/ss $M > $ROOT/gpurun_out/r5_seq_$C.log 2>&1
tail -2 $ROOT/gpurun_out/r5_seq_$C.log
