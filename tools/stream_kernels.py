"""Dev tool: per-stream busy time of the benchmark step from a rocprofv3 kernel trace, and the kernel table of the busiest stream
(the step's critical chain when it is bound by one stream).   python tools/stream_kernels.py DIR [steps_in_window=8]"""
import glob, sys
import pandas as pd

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tr = pd.read_csv(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])
tr["dur"] = (tr.End_Timestamp - tr.Start_Timestamp) / 1e3
tr = tr.sort_values("Start_Timestamp")
fps = tr[tr.Kernel_Name.str.contains("fps_kernel")].Start_Timestamp.values
per_step = max(1, round(len(fps) / max(1, len(tr[tr.Kernel_Name.str.contains("adamw|multi_tensor_apply")]) / 8)))
lo, hi = fps[-n * per_step - 1], fps[-1]
w = tr[(tr.Start_Timestamp >= lo) & (tr.Start_Timestamp < hi)]
key = "Stream_Id" if "Stream_Id" in w.columns else "Queue_Id"
for sid, g in w.groupby(key):
    print(f"stream {sid}: {len(g) / n:.0f} kernels/step, busy {g.dur.sum() / n:.0f} us/step")
main = w[w[key] == w.groupby(key).dur.sum().idxmax()]
main = main.assign(nm=main.Kernel_Name.str.replace("(anonymous namespace)::", "").str.replace("void ", "").str.split("(").str[0].str.slice(0, 70))
t = main.groupby("nm").dur.agg(["count", "sum", "mean"]).sort_values("sum", ascending=False)
for k, r in t.head(28).iterrows():
    print(f"{r['count'] / n:6.1f}/step {r['mean']:7.1f} us {r['sum'] / n:7.0f} us/step  {k}")
