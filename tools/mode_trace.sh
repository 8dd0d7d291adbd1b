# gpurun -- 'bash tools/mode_trace.sh C2 fp32': per-kernel table of a configuration's step in a precision mode
cd /tmp && export TMPDIR=/tmp
C=$1; MODE=$2; rm -rf /tmp/mt
rocprofv3 --kernel-trace --output-format csv -d /tmp/mt -o p -- python3 $GRAFT_REPO_ROOT/tools/mode_steps.py $C $MODE 12 > /tmp/mt.log 2>&1
grep "ms/step" /tmp/mt.log
python3 $GRAFT_REPO_ROOT/tools/chain_kernels.py /tmp/mt 18 2>&1 | head -${3:-32} | cut -c1-130
