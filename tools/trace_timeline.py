#!/usr/bin/env python3
"""Per-stream timeline of the benchmark step from a rocprofv3 kernel trace (which stream is the critical path, how busy it is,
how large the gaps between its kernels are).
    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-roofline --no-parity-mode
    python tools/trace_timeline.py DIR [steps_in_window=8]"""
import glob, sys
import numpy as np
import pandas as pd

d = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tr = pd.read_csv(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])
tr["name"] = tr.Kernel_Name.str.replace("(anonymous namespace)::", "").str.replace("void ", "").str.split("(").str[0].str.slice(0, 60)
tr = tr.sort_values("Start_Timestamp").reset_index(drop=True)
# steps are delimited by the FPS kernel of the point tower (one per step)
fps = tr[tr.name.str.contains("fps_kernel")]
starts = fps.Start_Timestamp.values
lo, hi = starts[-nsteps - 1], starts[-1]
w = tr[(tr.Start_Timestamp >= lo) & (tr.Start_Timestamp < hi)].copy()
span = (hi - lo) / nsteps / 1e3
print(f"window: {nsteps} steps, {span:.1f} us per step")
key = "Stream_Id" if "Stream_Id" in w.columns else "Queue_Id"
for sid, g in w.groupby(key):
    busy = (g.End_Timestamp - g.Start_Timestamp).sum() / nsteps / 1e3
    g = g.sort_values("Start_Timestamp")
    gaps = (g.Start_Timestamp.values[1:] - g.End_Timestamp.values[:-1]) / 1e3
    gaps = gaps[(gaps > 0)]
    print(f"  {key} {sid}: {len(g) / nsteps:.0f} kernels/step, busy {busy:.0f} us/step ({100 * busy / span:.0f} %), "
          f"gaps: median {np.median(gaps):.1f} us, sum {gaps.sum() / nsteps:.0f} us/step; top kernels:")
    top = g.assign(dur=(g.End_Timestamp - g.Start_Timestamp) / 1e3).groupby("name").dur.agg(["count", "sum", "mean"]).sort_values("sum", ascending=False).head(6)
    for n, r in top.iterrows():
        print(f"      {n:60s} {r['count'] / nsteps:6.1f}/step  {r['mean']:7.1f} us  {r['sum'] / nsteps:7.0f} us/step")
