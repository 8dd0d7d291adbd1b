"""Dev tool: where the C2 step's wall time goes on the caller's stream (HIP events, no profiler)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ppt_amd.train import Trainer
from ppt_amd import weights as W

torch.cuda.set_device(0)
cfg = bench.CONFIGS["C2"]
model = bench.build_model(cfg["dataset"], cfg["head_type"], torch.bfloat16, "ULIP_PointBERT", "cls")
model.train()
tr = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=False)
pc = torch.from_numpy(W.synth_clouds(32, 1024, seed=1)[0]).cuda()
label = torch.randint(0, 40, (32,), device="cuda")
orig = model.point_encoder.forward
marks = []
def tower(x, *a):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record(); y = orig(x, *a); e.record()
    marks.append((s, e))
    return y
model.point_encoder.forward = tower
for _ in range(10): tr.step(pc, label)
torch.cuda.synchronize(); marks.clear()
t0 = time.perf_counter()
N = 40
for _ in range(N): tr.step(pc, label)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
tower = sum(s.elapsed_time(e) for s, e in marks) / N
between = sum(marks[i][1].elapsed_time(marks[i + 1][0]) for i in range(N - 1)) / (N - 1)
print(f"step {wall:.3f} ms = point tower on its stream {tower:.3f} ms + tower-end -> next tower-start {between:.3f} ms")
