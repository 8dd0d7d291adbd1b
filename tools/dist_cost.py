"""Dev tool: cost of the two per-step collectives (single-rank RCCL) in the C2 step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
import bench
from ppt_amd.train import Trainer
from ppt_amd import weights as W
torch.cuda.set_device(0)
if os.environ.get("EARLY_STREAMS") == "1":
    torch.zeros(1, device="cuda")
    _early = [torch.cuda.Stream() for _ in range(int(os.environ.get("NEARLY", "1")))]
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = bench.CONFIGS["C2"]
model = bench.build_model(cfg["dataset"], cfg["head_type"], torch.bfloat16, "ULIP_PointBERT", "cls")
model.train()
if os.environ.get("EARLY_STREAMS") == "1":
    model._text_stream = _early[-1]
DIST = os.environ.get("DIST", "1") == "1"
tr = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=DIST)
pc = torch.from_numpy(W.synth_clouds(32, 1024, seed=1)[0]).cuda()
label = torch.randint(0, 40, (32,), device="cuda")
def run(tag):
    for _ in range(10): tr.step(pc, label)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): tr.step(pc, label)
    torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)
run("bcast + all_reduce" if DIST else "process group initialised, Trainer(distributed=False)")
print("graphs:", list(model._graphs.entries), list(model.point_encoder._graphs.entries))
if not DIST:
    dist.destroy_process_group(); sys.exit(0)
b = tr.bcast.broadcast; tr.bcast.broadcast = lambda: None
run("all_reduce only")
a = tr.sync.all_reduce; tr.sync.all_reduce = lambda: None
run("no collectives")
tr.bcast.broadcast = b
run("bcast only")
x = torch.zeros(16384, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): dist.all_reduce(x)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"100 all_reduce(64 KiB): host {1e4 * (t1 - t0):.1f} us each, total {1e4 * (t2 - t0):.1f} us each")
dist.destroy_process_group()
