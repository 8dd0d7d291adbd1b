"""Dev tool: the ORDERED kernel sequence of one steady-state iteration of the prompt chain (tools/chain_only.py under
rocprofv3 --kernel-trace): name, stream, duration, gap to the previous kernel of the same stream.  Shows which launches are
graph plumbing (ATen copies / fills) and where the dependent chain idles.
    python tools/chain_sequence.py DIR [marker-substring=adamw]"""
import glob, sys
import pandas as pd

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "adamw"
tr = pd.read_csv(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]).sort_values("Start_Timestamp").reset_index(drop=True)
tr["name"] = tr.Kernel_Name.str.replace("(anonymous namespace)::", "", regex=False).str.replace("void ", "", regex=False).str.slice(0, 90)
idx = tr.index[tr.name.str.contains(marker)].tolist()
a, b = idx[-3], idx[-2]                               # one iteration between two optimizer launches, late in the run
it = tr.iloc[a + 1:b + 1].copy()
it["dur"] = (it.End_Timestamp - it.Start_Timestamp) / 1e3
last_end = {}
rows = []
for _, r in it.iterrows():
    gap = (r.Start_Timestamp - last_end[r.Stream_Id]) / 1e3 if r.Stream_Id in last_end else float("nan")
    last_end[r.Stream_Id] = r.End_Timestamp
    wgs = (r.Grid_Size_X * r.Grid_Size_Y * r.Grid_Size_Z) // max(1, r.Workgroup_Size_X * r.Workgroup_Size_Y * r.Workgroup_Size_Z)
    rows.append((r.Stream_Id, r["name"], r.dur, gap, wgs))
print(f"{len(rows)} kernels in the iteration, {(it.End_Timestamp.max() - it.Start_Timestamp.min()) / 1e3:.0f} us wall, {it.dur.sum():.0f} us busy")
for s, n, du, gap, wgs in rows:
    print(f"s{s:<3d} {du:7.1f} us  gap {gap:6.1f}  wgs {wgs:5d}  {n}")
aten = it[it.name.str.contains("at::native|rocclr|Memcpy|elementwise|fill", regex=True)]
print(f"ATen / runtime kernels: {len(aten)} launches, {aten.dur.sum():.0f} us")
