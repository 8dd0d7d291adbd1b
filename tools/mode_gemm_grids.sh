cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mt
rocprofv3 --kernel-trace --output-format csv -d /tmp/mt -o p -- python3 $GRAFT_REPO_ROOT/tools/mode_steps.py $1 split16 12 > /tmp/mt.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/kernel_grids.py /tmp/mt gemm_kernel 18 | head -${2:-60}
