#!/bin/bash
# Dev tool: kernel-trace durations of the attention forward under the launch switches (XCD map of the streaming kernel, the
# LDS-resident ViT kernel).
export PPT_BENCH_BURN_IN_S=0      # (the traces count on the 40-step burn-in: steps = 40 + warmup + K)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for V in "0 0" "1 0" "1 1"; do
    set -- $V
    export PPT_ATTN_XCD_MAP=$1 PPT_ATTN_RESIDENT=$2
    rm -rf /tmp/attnv
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/attnv -o p -- python3 $ROOT/tools/attn_bench.py > /dev/null 2>&1
    python3 - <<PY
import pandas as pd, glob
t = pd.read_csv(glob.glob("/tmp/attnv/**/*kernel_trace.csv", recursive=True)[0])
t["dur"] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
a = t[t.Kernel_Name.str.contains("attn_fwd")]
print("xcd_map=$1 resident=$2:", a.groupby(["Grid_Size_X", "Grid_Size_Y"]).dur.agg(["count", "min", "median"]).round(1).to_string().replace("\n", " | "))
PY
done
