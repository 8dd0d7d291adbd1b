"""Dev tool: launches of the kernels whose name contains SUBSTR in a rocprofv3 kernel trace, grouped by grid: launches per step, mean
duration, workgroups.    python tools/kernel_grids.py DIR SUBSTR steps"""
import glob, sys
import pandas as pd

d, sub, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
tr = pd.read_csv(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]).sort_values("Start_Timestamp").reset_index(drop=True)
tr = tr.iloc[len(tr) // 2:]
steps /= 2
tr = tr[tr.Kernel_Name.str.contains(sub, regex=False)].copy()
tr["dur"] = (tr.End_Timestamp - tr.Start_Timestamp) / 1e3
tr["name"] = tr.Kernel_Name.str.replace("(anonymous namespace)::", "", regex=False).str.replace("void ", "", regex=False).str.slice(0, 44)
tr["gx"] = tr.Grid_Size_X // tr.Workgroup_Size_X
tr["gy"] = tr.Grid_Size_Y // tr.Workgroup_Size_Y
tr["gz"] = tr.Grid_Size_Z // tr.Workgroup_Size_Z
g = tr.groupby(["name", "gx", "gy", "gz"]).dur.agg(["count", "mean", "sum"]).reset_index().sort_values("sum", ascending=False)
print(f"{'kernel':44s} {'grid (N tiles, M tiles, batch)':>32s} {'n/step':>7s} {'us':>8s} {'us/step':>8s}")
for _, r in g.iterrows():
    print(f"{r['name']:44s} {str((int(r.gx), int(r.gy), int(r.gz))):>32s} {r['count'] / steps:7.1f} {r['mean']:8.1f} {r['sum'] / steps:8.0f}")
