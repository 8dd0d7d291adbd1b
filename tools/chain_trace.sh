cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ct
rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o p -- python3 $GRAFT_REPO_ROOT/tools/chain_only.py 40 > /tmp/ct.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/chain_sequence.py /tmp/ct > $GRAFT_REPO_ROOT/gpurun_out/r5_chain_seq.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/chain_kernels.py /tmp/ct 40 > $GRAFT_REPO_ROOT/gpurun_out/r5_chain_kernels.log 2>&1
