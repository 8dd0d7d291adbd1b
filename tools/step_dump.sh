#!/bin/bash
# Dev tool: every kernel of ONE step of a configuration's bench run, ordered by start time, all streams side by side
# (start offset within the step, duration, gap to the previous kernel of the SAME stream).
#   bash tools/step_dump.sh C2
export PPT_BENCH_BURN_IN_S=0
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
C=${1:-C2}
rm -rf /tmp/sd
rocprofv3 --kernel-trace --output-format csv -d /tmp/sd -o p -- python3 $ROOT/bench.py --config $C --steps 10 --warmup 5 \
    --no-cpu-baseline --no-roofline --no-parity-mode --no-secondary > /tmp/sd.log 2>&1
python3 - <<'PY'
import pandas as pd, glob, numpy as np
tr = pd.read_csv(glob.glob("/tmp/sd/**/*kernel_trace.csv", recursive=True)[0]).sort_values("Start_Timestamp").reset_index(drop=True)
tr["nm"] = tr.Kernel_Name.str.replace("(anonymous namespace)::", "", regex=False).str.replace("void ", "", regex=False).str.split("(").str[0].str.slice(0, 40)
ad = tr.index[tr.nm.str.startswith("adamw_step")].values
a, b = ad[-4], ad[-3]                      # one whole step between two optimizer launches
w = tr.iloc[a:b + 1].copy()
t0 = w.Start_Timestamp.iloc[0]
print("step length %.3f ms" % ((w.Start_Timestamp.iloc[-1] - t0) / 1e6))
last = {}
streams = sorted(w.Stream_Id.unique())
for _, r in w.iterrows():
    s = r.Stream_Id
    gap = (r.Start_Timestamp - last[s]) / 1e3 if s in last else 0.0
    last[s] = r.End_Timestamp
    col = streams.index(s)
    print(f"{(r.Start_Timestamp - t0) / 1e3:8.1f} {' ' * (46 * col)}s{s} +{gap:6.1f} {(r.End_Timestamp - r.Start_Timestamp) / 1e3:6.1f} {r.nm}")
PY
tail -1 /tmp/sd.log | cut -c1-200
