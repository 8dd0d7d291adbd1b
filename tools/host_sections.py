"""Dev tool: host (enqueue) time of the sections of a part-seg step -- forward, loss, backward, optimizer -- and the top Python
functions by cumulative time (cProfile).   python tools/host_sections.py"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import bench
from ppt_amd import graphs, weights as W
from ppt_amd.train import Trainer
torch.cuda.set_device(0)
graphs.shared_text_stream()
cfg = bench.CONFIGS["C5"]
model = bench.build_model(cfg["dataset"], 0, model=cfg["model"], task="partseg"); model.train()
tr = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=False)
B, N = cfg["batch"], cfg["npoints"]
pc = torch.from_numpy(W.synth_clouds(B, N, seed=1)[0]).cuda()
label = torch.from_numpy(np.random.default_rng(0).integers(0, 50, size=(B, N))).cuda()
onehot = torch.zeros(B, 16, device="cuda"); onehot[:, 0] = 1
tr.extra_inputs = (onehot,)
for _ in range(20): tr.step(pc, label)
tr.finish(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(20): tr.step(pc, label)
t1 = time.perf_counter()
pr.disable(); tr.finish(); torch.cuda.synchronize()
print(f"host {1e3 * (t1 - t0) / 20:.2f} ms/step")
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
