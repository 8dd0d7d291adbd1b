"""Dev tool: does running the point tower's stream at high priority help against the text stream?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ppt_amd
import torch
import bench
from ppt_amd.train import Trainer
from ppt_amd import weights as W
torch.cuda.set_device(0)
print("priority range", torch.cuda.Stream.priority_range())
cfg = bench.CONFIGS["C2"]
pc = torch.from_numpy(W.synth_clouds(32, 1024, seed=1)[0]).cuda()
label = torch.randint(0, 40, (32,), device="cuda")
def run(tag, main_stream, side_prio):
    model = bench.build_model(cfg["dataset"], cfg["head_type"], torch.bfloat16, "ULIP_PointBERT", "cls")
    model.train()
    if side_prio != 0: model._text_stream = torch.cuda.Stream(priority=side_prio)
    tr = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=False)
    ctx = torch.cuda.stream(main_stream) if main_stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(40): tr.step(pc, label)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): tr.step(pc, label)
        torch.cuda.synchronize()
    from ppt_amd import graphs
    lead = graphs._lead_over(torch.cuda.current_stream(), model._text_stream)
    print(f"{tag}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms/step; lead of the text stream over the caller's now {lead:.2f}", flush=True)
lo, hi = torch.cuda.Stream.priority_range()
junk = []
for i in range(6):
    run(f"default main, probed side, run {i} ({len(junk)} unrelated streams alive)", None, 0)
    junk += [torch.cuda.Stream() for _ in range(3)]
