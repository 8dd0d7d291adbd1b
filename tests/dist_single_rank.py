"""Helper of tests/test_model_gpu.py::test_single_rank_rccl_step_is_identical (run as a child process: a process group and
RCCL's streams should not leak into the pytest process).  Runs the same seeded training steps twice on cuda:0 --
Trainer(distributed=False), then Trainer(distributed=True) under a ONE-rank `nccl` (= RCCL) process group, i.e. the code
path of an N-GPU job (flat-gradient all-reduce on the text stream, re-bound BatchNorm buffers, deferred broadcast) -- and
prints one JSON line: bit-equality of losses and trained parameters, and the step times of both.

    python tests/dist_single_rank.py <head_type> <steps>
"""
import contextlib
import io
import json
import os
import sys
import time
from types import SimpleNamespace

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def run(head_type, steps, distributed):
    from ppt_amd import weights as W
    from ppt_amd.models import ULIP_models as M
    from ppt_amd.train import Trainer
    names = M.dataset_classnames("scanobjectnn" if head_type else "modelnet40")
    args = SimpleNamespace(classnames=names, template_init='', class_name_position='middle', num_learnable_prompt_tokens=32,
                           gpu=0, task='cls', head_type=head_type, evaluate_3d=False, ulip2=False, synthetic_weights=True)
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.ULIP_PointBERT(args)
    m.load_state_dict(W.ulip_pointbert_state_dict(seed=0), strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(len(names), seed=0)
    m.cuda().set_precision(torch.bfloat16)
    m.train()
    B = 16
    pc_np, start = W.synth_clouds(B, 1024, seed=3)
    m.point_encoder.fps_start = torch.from_numpy(start).cuda()
    rng = np.random.default_rng(1)
    m.point_encoder.drop_path_factors = torch.from_numpy((np.floor(0.9 + rng.random((12, 2, B))) / 0.9).astype(np.float32))
    pc = torch.from_numpy(pc_np).cuda()
    labels = torch.from_numpy(rng.integers(0, len(names), size=(B,))).cuda()
    tr = Trainer(m, lr=3e-3, distributed=distributed)
    assert (tr.bcast is not None) == distributed
    losses = []
    for _ in range(steps):
        loss, _ = tr.step(pc, labels)
        losses.append(loss.detach().clone())
    tr.finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(pc, labels)
    tr.finish()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    params = {n: p.detach().cpu().clone() for n, p in m.named_parameters() if p.requires_grad}
    rm = m.point_encoder.encoder.first_conv[1].running_mean.detach().cpu().clone()
    return [float(l) for l in losses], params, rm, ms


def main():
    head_type, steps = int(sys.argv[1]), int(sys.argv[2])
    torch.cuda.set_device(0)
    from ppt_amd import graphs
    graphs.shared_text_stream()                      # created before RCCL's streams, as bench.py does
    l0, p0, rm0, ms0 = run(head_type, steps, False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    l1, p1, rm1, ms1 = run(head_type, steps, True)
    dist.barrier()
    dist.destroy_process_group()
    out = {"losses_equal": l0 == l1, "params_equal": all(torch.equal(p0[k], p1[k]) for k in p0), "bn_equal": bool(torch.equal(rm0, rm1)),
           "n_params": len(p0), "finite": bool(np.isfinite(l0).all()), "ms_plain": round(ms0, 3), "ms_dist": round(ms1, 3)}
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
