# gpurun -- 'bash tools/eval_trace.sh C2 [lines]': per-kernel table of validate()'s forward (bench.py --eval) in the mode PPT_BENCH_MODE says
export PPT_BENCH_BURN_IN_S=0      # (the traces count on the 40-step burn-in: steps = 40 + warmup + K)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/et
rocprofv3 --kernel-trace --output-format csv -d /tmp/et -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config $1 --eval --steps 20 --warmup 5 > /tmp/et.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/chain_kernels.py /tmp/et 65 2>&1 | head -${2:-16} | cut -c1-140
