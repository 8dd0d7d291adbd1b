#!/bin/bash
# Dev tool: kernel-trace table of one configuration's bench run (per stream: launches / step, avg us, ms / step).
#   bash tools/quick_trace.sh C2
export PPT_BENCH_BURN_IN_S=0      # (the traces count on the 40-step burn-in: steps = 40 + warmup + K)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
C=${1:-C2}
rm -rf /tmp/qt
rocprofv3 --kernel-trace --output-format csv -d /tmp/qt -o p -- python3 $ROOT/bench.py --config $C --steps 10 --warmup 5 \
    --no-cpu-baseline --no-roofline --no-parity-mode --no-secondary > /tmp/qt.log 2>&1
python3 - <<'PY'
import pandas as pd, glob
pd.set_option('display.width', 250); pd.set_option('display.max_colwidth', 64)
tr = pd.read_csv(glob.glob("/tmp/qt/**/*kernel_trace.csv", recursive=True)[0])
tr["dur"] = (tr.End_Timestamp - tr.Start_Timestamp) / 1e3
tr["nm"] = tr.Kernel_Name.str.replace("(anonymous namespace)::", "", regex=False).str.replace("void ", "", regex=False).str.split("(").str[0]
n = 55
t = tr.groupby(["nm", "Stream_Id"]).agg(per_step=("dur", lambda x: len(x) / n), mean=("dur", "mean"), mn=("dur", "min"), gx=("Grid_Size_X", "max"), gy=("Grid_Size_Y", "max"), tot=("dur", "sum"))
t["ms_step"] = t.tot / n / 1e3
print(t.sort_values("tot", ascending=False).head(26).drop(columns="tot").round(2).to_string())
print("per stream ms/step:", (tr.groupby("Stream_Id").dur.sum() / n / 1e3).round(3).to_dict())
PY
tail -1 /tmp/qt.log | cut -c1-200
