"""Helper of tests/test_model_gpu.py::test_two_ranks_real_model_on_one_gpu: the REAL ULIP_PointBERT head_type 3
training step under a TWO-rank process group, both ranks on cuda:0 (the builder has one GPU at a time), each rank its own process
from the start (nothing touches the GPU before the rank's process exists).

    python tests/dist_two_ranks.py rank <rank> <outdir>     one rank of the 2-rank job (RANK / WORLD_SIZE / MASTER_* from the env)
    python tests/dist_two_ranks.py ref <outdir>             single-process references on the same inputs

Rank r: seeds with 0 + r (main_cls.py:39) and builds the model from ITS OWN synthetic state dict (seed r) -- so the un-frozen last
block and the prompt tokens differ between the ranks until train.Trainer's DDP-constructor broadcast (main_cls.py:47-49) -- then
runs 3 steps of train.Trainer(distributed=True) on its half of a B = 8 batch: ONE all-reduce of the flat gradient per step,
BatchNorm running statistics broadcast from rank 0 in finish().  It saves: the reduced gradients of step 1, every trained parameter
after step 3, the tokenizer's BatchNorm running statistics after finish().

Backend: RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the job runs over gloo with device tensors when `nccl`
cannot be initialised -- the collectives' ARITHMETIC (SUM, then x 1/W) is the same; what this test adds over the gloo CPU test is the
real model: real train-mode forwards on two ranks, real gradients, the real buffer broadcast.
"""
import contextlib
import io
import os
import sys
from types import SimpleNamespace

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

B, N, HEAD, STEPS = 8, 1024, 3, 3


def inputs():
    from ppt_amd import weights as W
    pc_np, start = W.synth_clouds(B, N, seed=13)
    rng = np.random.default_rng(5)
    dp = (np.floor(0.9 + rng.random((12, 2, B))) / 0.9).astype(np.float32)
    labels = rng.integers(0, 15, size=(B,))
    return torch.from_numpy(pc_np), torch.from_numpy(start), torch.from_numpy(dp), torch.from_numpy(labels)


def build(seed):
    from ppt_amd import weights as W
    from ppt_amd.models import ULIP_models as M
    names = M.dataset_classnames("scanobjectnn")
    args = SimpleNamespace(classnames=names, template_init='', class_name_position='middle', num_learnable_prompt_tokens=32,
                           gpu=0, task='cls', head_type=HEAD, evaluate_3d=False, ulip2=False, synthetic_weights=True)
    torch.manual_seed(seed)                                          # main_cls.py:39: seed = args.seed + rank
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.ULIP_PointBERT(args)
    sd = W.ulip_pointbert_state_dict(seed=0)
    if seed:
        # the frozen backbone is the same checkpoint on every rank; what differs per rank is what the reference initialises
        # randomly per process: the prompt tokens and the un-frozen last block
        other = W.ulip_pointbert_state_dict(seed=seed)
        for k in sd:
            if k.startswith("point_encoder.blocks.blocks.11."):
                sd[k] = other[k]
    m.load_state_dict(sd, strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(len(names), seed=0)
    m.cuda().set_precision(torch.bfloat16)
    m.train()
    return m


def bn_stats(m):
    enc = m.point_encoder.encoder
    return {"rm1": enc.first_conv[1].running_mean.detach().cpu().clone(), "rv1": enc.first_conv[1].running_var.detach().cpu().clone(),
            "rm2": enc.second_conv[1].running_mean.detach().cpu().clone(), "rv2": enc.second_conv[1].running_var.detach().cpu().clone()}


def trained(m):
    return {n: p.detach().cpu().clone() for n, p in m.named_parameters() if p.requires_grad}


def run_rank(rank, outdir):
    torch.cuda.set_device(0)
    from ppt_amd import graphs
    from ppt_amd.train import Trainer
    graphs.shared_text_stream()
    backend = "gloo"
    if os.environ.get("PPT_TWO_RANK_BACKEND", "gloo") == "nccl":
        backend = "nccl"
    dist.init_process_group(backend, rank=rank, world_size=2)
    pc, start, dp, labels = inputs()
    half = slice(rank * B // 2, (rank + 1) * B // 2)
    m = build(rank)
    before = trained(m)
    m.point_encoder.fps_start = start[half].cuda()
    m.point_encoder.drop_path_factors = dp[:, :, half].contiguous()
    tr = Trainer(m, lr=3e-3, distributed=True)
    after_bcast = trained(m)
    out = {"backend": backend, "before": before, "after_bcast": after_bcast, "init_broadcasts": tr.init_broadcasts}
    x, y = pc[half].cuda(), labels[half].cuda()
    losses = []
    for it in range(STEPS):
        loss, _ = tr.step(x, y)
        if it == 0:
            tr.finish()
            torch.cuda.synchronize()
            out["grads_step1"] = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}
        losses.append(float(loss))
    tr.finish()
    torch.cuda.synchronize()
    out.update(losses=losses, params=trained(m), bn=bn_stats(m), skipped=tr.nonfinite_grad_elements())
    torch.save(out, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
    print(f"RANK {rank} done ({backend})", flush=True)


def run_ref(outdir):
    """single process: (a) the gradients of step 1 on each half from rank 0's initial state (train.Trainer with lr = 0: the same
    criterion kernel as the ranks use, parameters untouched); (b) 3 Trainer steps on half 0 alone -> the BatchNorm statistics
    rank 0 must end with (they depend on its inputs only)."""
    torch.cuda.set_device(0)
    from ppt_amd.train import Trainer
    pc, start, dp, labels = inputs()
    out = {"grads": []}
    for r in range(2):
        half = slice(r * B // 2, (r + 1) * B // 2)
        m = build(0)
        m.point_encoder.fps_start = start[half].cuda()
        m.point_encoder.drop_path_factors = dp[:, :, half].contiguous()
        tr = Trainer(m, lr=0.0, wd=0.0, distributed=False)
        tr.step(pc[half].cuda(), labels[half].cuda())
        tr.finish()
        torch.cuda.synchronize()
        out["grads"].append({n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None})
    m = build(0)
    half = slice(0, B // 2)
    m.point_encoder.fps_start = start[half].cuda()
    m.point_encoder.drop_path_factors = dp[:, :, half].contiguous()
    tr = Trainer(m, lr=3e-3, distributed=False)
    for _ in range(STEPS):
        tr.step(pc[half].cuda(), labels[half].cuda())
    tr.finish()
    torch.cuda.synchronize()
    out["bn_half0"] = bn_stats(m)
    torch.save(out, os.path.join(outdir, "ref.pt"))
    print("REF done", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "rank":
        run_rank(int(sys.argv[2]), sys.argv[3])
    else:
        run_ref(sys.argv[2])
