#!/bin/bash
# Dev tool: rocprofv3 mean duration of the text tower's attention kernels in tools/attn_short_bench.py, per variant library.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
    if [ "$v" = base ]; then unset PPT_HIP_LIB; else export PPT_HIP_LIB=$ROOT/tools/_build/libppt_$v.so; fi
    for both in 1 0; do
        export PPT_ATTN_SHORT_BOTH=$both
        rm -rf /tmp/asp
        rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/asp -o p -- python3 $ROOT/tools/attn_short_bench.py > /tmp/asp.log 2>&1
        echo "== $v both=$both"
        python3 - <<'PY'
import pandas as pd, glob
s = pd.read_csv(glob.glob("/tmp/asp/**/*kernel_stats.csv", recursive=True)[0])
s = s[s.Name.str.contains("attn")]
for _, r in s.iterrows():
    print(f"   {r.Name.split('(')[0][-60:]:60s} calls {r.Calls:5d}  avg {r.AverageNs / 1e3:7.2f} us  min {r.MinNs / 1e3:7.2f}")
PY
    done
done
