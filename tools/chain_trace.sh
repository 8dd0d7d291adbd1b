#!/bin/bash
# Dev tool: kernel-trace table of the prompt chain ALONE (tools/chain_only.py): launches per iteration, avg us, ms per iteration.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ct
rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o p -- python3 $ROOT/tools/chain_only.py 40 > /tmp/ct.log 2>&1
python3 - <<'PY'
import pandas as pd, glob
pd.set_option('display.width', 250); pd.set_option('display.max_colwidth', 70)
tr = pd.read_csv(glob.glob("/tmp/ct/**/*kernel_trace.csv", recursive=True)[0])
tr = tr.sort_values("Start_Timestamp")
tr["dur"] = (tr.End_Timestamp - tr.Start_Timestamp) / 1e3
tr["nm"] = tr.Kernel_Name.str.replace("(anonymous namespace)::", "", regex=False).str.replace("void ", "", regex=False).str.split("(").str[0]
n = 60
t = tr.groupby(["nm"]).agg(per_it=("dur", lambda x: len(x) / n), mean=("dur", "mean"), mn=("dur", "min"), gx=("Grid_Size_X", "max"), wg=("Workgroup_Size_X", "max"), tot=("dur", "sum"))
t["ms_it"] = t.tot / n / 1e3
print(t.sort_values("tot", ascending=False).head(40).drop(columns="tot").round(2).to_string())
print("launches / iteration:", len(tr) / n, " busy ms / iteration:", tr.dur.sum() / n / 1e3)
# gaps between consecutive kernels in the last 20 iterations
last = tr.tail(int(len(tr) / n * 20))
gap = (last.Start_Timestamp.values[1:] - last.End_Timestamp.values[:-1]) / 1e3
import numpy as np
it = tr.tail(int(round(len(tr) / n)) + 4)
g = np.concatenate([[0], (it.Start_Timestamp.values[1:] - it.End_Timestamp.values[:-1]) / 1e3])
for (nm, d, gx, st), gg in zip(it[["nm", "dur", "Grid_Size_X", "Stream_Id"]].values, g):
    print(f"  +{gg:6.2f}  {d:7.2f} us  s{st} {gx:8d}  {nm[:90]}")
print("gap us: median %.2f mean %.2f  sum/iter %.3f ms" % (np.median(gap), gap.mean(), gap.sum() / 20 / 1e3))
PY
tail -1 /tmp/ct.log
