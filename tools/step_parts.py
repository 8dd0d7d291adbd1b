"""Dev tool: the C2 step in parts -- the point tower alone, the prompt side alone (text tower forward + head + backward + AdamW
with a cached point feature), and the full two-stream step -- to see how much of each hides under the other.
    python tools/step_parts.py [C2|C3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from ppt_amd import graphs, weights as W
from ppt_amd.train import Trainer

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
torch.cuda.set_device(0)
cfg = bench.CONFIGS[name]
graphs.shared_text_stream(priority=-1 if cfg["head_type"] == 0 else 0)
model = bench.build_model(cfg["dataset"], cfg["head_type"], torch.bfloat16, "ULIP_PointBERT", "cls")
model.train()
tr = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=False)
tr.inputs_ready = os.environ.get("PPT_INPUTS_READY", "0") == "1"
B, N = cfg["batch"], cfg["npoints"]
pc = torch.from_numpy(W.synth_clouds(B, N, seed=1)[0]).cuda()
label = torch.randint(0, len(model.prompt_learner.classnames), (B,), device="cuda")


def timed(fn, n=40, warm=12):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full = timed(lambda: tr.step(pc, label))
tr.finish()
with torch.no_grad():
    model.train()
    tower = timed(lambda: model.point_encoder(pc))
# prompt side alone: replace the point tower by a cached feature
feat = model.point_encoder(pc).detach()
orig = model.point_encoder.forward
model.point_encoder.forward = lambda x: feat
prompt = timed(lambda: tr.step(pc, label))
tr.finish()
model.point_encoder.forward = orig
print(f"{name}: full step {full:.3f} ms | point tower alone {tower:.3f} ms | prompt side alone {prompt:.3f} ms | "
      f"sum {tower + prompt:.3f} ms | hidden {tower + prompt - full:.3f} ms")
