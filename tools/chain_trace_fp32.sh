cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ct
export PPT_TEXT_PRECISION=fp32
rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o p -- python3 $GRAFT_REPO_ROOT/tools/chain_only.py 40 > /tmp/ct.log 2>&1
tail -3 /tmp/ct.log
python3 $GRAFT_REPO_ROOT/tools/chain_kernels.py /tmp/ct 40 > $GRAFT_REPO_ROOT/gpurun_out/r5_chain32_kernels.log 2>&1
head -40 $GRAFT_REPO_ROOT/gpurun_out/r5_chain32_kernels.log
