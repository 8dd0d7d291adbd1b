#!/usr/bin/env python3
"""Dev tool: long deterministic runs of the step with and without the ahead stage (grouping / tokenizer on its own stream), RNG
draws injected as static tensors so that both runs consume identical randomness: parameters and losses must be bit-identical
after hundreds of steps with the host running far ahead of the GPU (no synchronisation inside the loop).
    python tools/soak_ahead.py [head_type] [steps]        (PPT_SOAK_NODP=1: without DropPath -- the in-place hand-over path)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from ppt_amd import graphs, weights as W
from ppt_amd.train import Trainer

h = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
torch.cuda.set_device(0)
graphs.shared_text_stream(priority=-1 if h == 0 else 0)
graphs.shared_group_stream()
DIST = os.environ.get("PPT_SOAK_DIST") == "1"          # under a single-rank RCCL process group: the N-GPU code path of the Trainer
if DIST:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
B, N = 32, 1024
pcs = [torch.from_numpy(W.synth_clouds(B, N, seed=100 + i)[0]).cuda() for i in range(4)]
rng = np.random.default_rng(0)
label = torch.from_numpy(rng.integers(0, 40, size=(B,))).cuda()
start = torch.from_numpy(rng.integers(0, N, size=(B,))).cuda()
dp = torch.from_numpy((np.floor(0.9 + rng.random((12, 2, B))) / 0.9).astype(np.float32)).cuda()
res = []
for ahead in (False, True):
    model = bench.build_model("modelnet40", h, torch.bfloat16, "ULIP_PointBERT", "cls")
    model.train()
    pe = model.point_encoder
    pe.fps_start, pe.drop_path_factors = start, dp
    if os.environ.get("PPT_SOAK_NODP") == "1":       # no DropPath: the stage's outputs are then handed to the tower graph IN PLACE
        pe.drop_path_factors, pe.drop_path_rate = None, 0.0
    tr = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=DIST)
    tr.inputs_ready = ahead
    losses = []
    for it in range(steps):
        loss, _ = tr.step(pcs[it % 4], label)
        losses.append(loss)
    tr.finish()
    torch.cuda.synchronize()
    res.append((torch.stack(losses).cpu(), {n: p.detach().cpu().clone() for n, p in model.named_parameters() if p.requires_grad},
                {n: b.detach().cpu().clone() for n, b in pe.named_buffers()}))
    print(f"ahead={ahead}: final loss {losses[-1].item():.6f}, graphs {sorted({k[0] for k in pe._graphs.entries})}", flush=True)
(l0, p0, b0), (l1, p1, b1) = res
ok = torch.equal(l0, l1) and all(torch.equal(p0[k], p1[k]) for k in p0) and all(torch.equal(b0[k], b1[k]) for k in b0)
if DIST:
    dist.destroy_process_group()
print("IDENTICAL" if ok else f"MISMATCH: first differing step {int((l0 != l1).nonzero()[0]) if (l0 != l1).any() else -1}")
sys.exit(0 if ok else 1)
