"""Dev tool: host (enqueue) time per step against the synchronised step time, per configuration: is the step host-bound?
    python tools/host_time.py C5"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
import bench
from ppt_amd import graphs, weights as W
from ppt_amd.train import Trainer

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
cfg = bench.CONFIGS[name]
torch.cuda.set_device(0)
graphs.shared_text_stream(priority=-1 if cfg["head_type"] == 0 and cfg.get("model", "ULIP_PointBERT") == "ULIP_PointBERT" else 0)
model = bench.build_model(cfg["dataset"], cfg["head_type"], model=cfg.get("model", "ULIP_PointBERT"), task=cfg.get("task", "cls"))
model.train()
tr = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=False)
B, N = cfg["batch"], cfg["npoints"]
pc = torch.from_numpy(W.synth_clouds(B, N, seed=1)[0]).cuda()
partseg = cfg.get("task") == "partseg"
ncls = len(model.prompt_learner.classnames)
label = torch.from_numpy(np.random.default_rng(0).integers(0, ncls, size=(B, N) if partseg else (B,))).cuda()
if partseg:
    onehot = torch.zeros(B, 16, device="cuda"); onehot[:, 0] = 1
    tr.extra_inputs = (onehot,)
for _ in range(30):
    tr.step(pc, label)
tr.finish(); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(40):
        tr.step(pc, label)
    t1 = time.perf_counter()
    tr.finish(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e3 * (t1 - t0) / 40:.3f} ms/step, synchronised {1e3 * (t2 - t0) / 40:.3f} ms/step", flush=True)
