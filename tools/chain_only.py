"""Dev tool: N iterations of the prompt chain alone (text forward -> head -> text backward -> AdamW with a cached point
feature) -- the thing to put under rocprofv3 --kernel-trace to see the chain's kernels without the tower's.
    python3 tools/chain_only.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ppt_amd import graphs, weights as W
from ppt_amd.train import Trainer

torch.cuda.set_device(0)
cfg = bench.CONFIGS["C2"]
graphs.shared_text_stream(priority=-1)
model = bench.build_model(cfg["dataset"], cfg["head_type"], torch.bfloat16, "ULIP_PointBERT", "cls")
if os.environ.get("PPT_BENCH_MODE") == "split16":
    model.set_precision("split16")
model.train()
tr = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=False)
B, N = cfg["batch"], cfg["npoints"]
pc = torch.from_numpy(W.synth_clouds(B, N, seed=1)[0]).cuda()
label = torch.randint(0, 40, (B,), device="cuda")
with torch.no_grad():
    feat = model.point_encoder(pc).detach()
model.point_encoder.forward = lambda x: feat
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    tr.step(pc, label)
tr.finish()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    tr.step(pc, label)
tr.finish()
torch.cuda.synchronize()
print(f"chain alone: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per iteration")
