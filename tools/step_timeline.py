"""Dev tool: GPU timeline of the two-stream C2 step from HIP events on the streams the work is queued on -- when the point
tower, the text forward, the head, the text backward and the optimizer of iterations i .. i+n start and end, and when the
host issued each step() call.
    python tools/step_timeline.py [C2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from ppt_amd import graphs, weights as W
from ppt_amd.models import ULIP_models as U
from ppt_amd.train import Trainer

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
torch.cuda.set_device(0)
cfg = bench.CONFIGS[name]
graphs.shared_text_stream(priority=-1 if cfg["head_type"] == 0 else 0)
model = bench.build_model(cfg["dataset"], cfg["head_type"], torch.bfloat16, "ULIP_PointBERT", "cls")
model.train()
force_dist = os.environ.get("PPT_FORCE_DIST") == "1"          # the RCCL path with a single rank
if force_dist:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
tr = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=force_dist)
tr.inputs_ready = os.environ.get("PPT_INPUTS_READY", "0") == "1"
B, N = cfg["batch"], cfg["npoints"]
pc = torch.from_numpy(W.synth_clouds(B, N, seed=1)[0]).cuda()
label = torch.randint(0, len(model.prompt_learner.classnames), (B,), device="cuda")

log = []          # (step, what, start event, end event)
on = [False]
it = [0]


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record(torch.cuda.current_stream())
    return e


def wrap(fn, what):
    def f(*a, **k):
        if not on[0]:
            return fn(*a, **k)
        s = ev()
        r = fn(*a, **k)
        log.append((it[0], what, s, ev()))
        return r
    return f


pe_fwd = model.point_encoder.forward
model.point_encoder.forward = wrap(pe_fwd, "tower")
model._text_raw = wrap(model._text_raw, "text_fwd")
tr._optimizer_step = wrap(tr._optimizer_step, "adamw")
head = U._HeadLossFn


class HeadProxy:
    apply = staticmethod(wrap(head.apply, "head"))


U._HeadLossFn = HeadProxy
orig_backward = torch.Tensor.backward
torch.Tensor.backward = wrap(orig_backward, "text_bwd")

for _ in range(30):
    tr.step(pc, label)
torch.cuda.synchronize()
on[0] = True
t_ref = ev()
host = []
h0 = time.perf_counter()
for i in range(8):
    it[0] = i
    a = time.perf_counter()
    tr.step(pc, label)
    host.append((a - h0, time.perf_counter() - h0))
tr.finish()
torch.cuda.synchronize()
print("host: step() issued at / returned at (ms):", " ".join(f"{a * 1e3:.2f}/{b * 1e3:.2f}" for a, b in host))
for i, what, s, e in log:
    print(f"step {i} {what:9s} {t_ref.elapsed_time(s):8.3f} -> {t_ref.elapsed_time(e):8.3f}  ({s.elapsed_time(e):.3f} ms)")
