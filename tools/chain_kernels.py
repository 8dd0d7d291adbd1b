"""Dev tool: per-kernel table of a rocprofv3 kernel trace of tools/chain_only.py: launches per step, mean duration, workgroups,
and workgroup-time (workgroups x duration: an upper bound of the CU time a kernel holds).   python tools/chain_kernels.py DIR steps"""
import glob, sys
import pandas as pd

d, steps = sys.argv[1], int(sys.argv[2])
tr = pd.read_csv(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])
tr["name"] = tr.Kernel_Name.str.replace("(anonymous namespace)::", "").str.replace("void ", "").str.slice(0, 70)
tr = tr.sort_values("Start_Timestamp").reset_index(drop=True)
tr = tr.iloc[len(tr) // 2:]                     # steady state (graph replays)
steps = steps / 2
tr["dur"] = (tr.End_Timestamp - tr.Start_Timestamp) / 1e3
tr["wgs"] = (tr.Grid_Size_X * tr.Grid_Size_Y * tr.Grid_Size_Z) / (tr.Workgroup_Size_X * tr.Workgroup_Size_Y * tr.Workgroup_Size_Z)
tr["cu_us"] = tr.dur * tr.wgs.clip(upper=256)
g = tr.groupby("name").agg(n=("dur", "count"), dur=("dur", "mean"), wgs=("wgs", "mean"), tot=("dur", "sum"), cu=("cu_us", "sum")).sort_values("tot", ascending=False)
print(f"{'kernel':70s} {'n/step':>7s} {'us':>7s} {'WGs':>6s} {'us/step':>8s} {'CU-us/step':>10s}")
for n, r in g.iterrows():
    print(f"{n:70s} {r.n / steps:7.1f} {r.dur:7.1f} {r.wgs:6.0f} {r.tot / steps:8.0f} {r.cu / steps:10.0f}")
print(f"total: {len(tr) / steps:.0f} kernels/step, {tr.dur.sum() / steps:.0f} us/step busy, {tr.cu_us.sum() / steps / 256:.0f} us/step machine-equivalent (workgroups <= 256 CUs x duration)")
