"""Dev tool: which side of the two-stream C2 step is the critical path?  Times the full step with the text tower cut to half
its layers and with the point tower cut to half its blocks (timing experiments only: the results are not the model's).
    python tools/critical_path.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from ppt_amd import graphs, weights as W
from ppt_amd.train import Trainer

torch.cuda.set_device(0)
cfg = bench.CONFIGS["C2"]
graphs.shared_text_stream(priority=-1)
B, N = cfg["batch"], cfg["npoints"]
pc = torch.from_numpy(W.synth_clouds(B, N, seed=1)[0]).cuda()
label = torch.randint(0, 40, (B,), device="cuda")


def run(text_layers, depth, reps=3):
    model = bench.build_model(cfg["dataset"], cfg["head_type"], torch.bfloat16, "ULIP_PointBERT", "cls")
    model.train()
    model.transformer.layers = text_layers
    model.point_encoder.depth = depth
    model.point_encoder.dpr = model.point_encoder.dpr[:depth]
    tr = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=False)
    for _ in range(30):
        tr.step(pc, label)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            tr.step(pc, label)
        tr.finish()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
    return best


for tl, dp in ((12, 12), (6, 12), (12, 6), (6, 6), (1, 12), (12, 1), (12, 12)):
    print(f"text layers {tl:2d}, point blocks {dp:2d}: {run(tl, dp):.3f} ms/step", flush=True)
