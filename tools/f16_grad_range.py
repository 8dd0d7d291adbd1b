#!/usr/bin/env python3
"""How far the fp16 operand stages of the performance mode are from gradient underflow: the golden train step (B = 4) with
the loss multiplied by s = 1, 1/8, ... 1/32768 before backward (a batch of 4/s clouds has per-element gradients of that size:
mean-reduced cross entropy), gradients / s against the oracle's.  fp16 keeps 11 bits down to 6.1e-5 and flushes below 6e-8;
bf16 has fp32's range.  The first column is the operating point of the tests; C2 (B = 32) sits at s = 1/8.

    python tools/f16_grad_range.py [head_types ...]        # default: 0 3
"""
import contextlib
import io
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                    # noqa: E402
import torch                                          # noqa: E402
from oracle import oracle as O                        # noqa: E402
from ppt_amd import engine, weights as W              # noqa: E402
from ppt_amd.models import ULIP_models as M           # noqa: E402
from ppt_amd.train import Trainer                     # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
SCALES = [1.0, 1 / 8, 1 / 64, 1 / 512, 1 / 4096, 1 / 32768]


def run(h, g, s, all_bf16):
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=h, evaluate_3d=False, ulip2=False,
                           synthetic_weights=True)
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.ULIP_PointBERT(args)
    m.load_state_dict(W.ulip_pointbert_state_dict(seed=0), strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(40, seed=0)
    m.cuda().set_precision(torch.bfloat16)
    m.use_hip_graphs = m.point_encoder.use_hip_graphs = False
    m.overlap_text_tower = False
    engine.STAGE_DTYPE.clear()
    if all_bf16:
        m.text_precision = torch.bfloat16
        for st in ("tokenizer", "blocks", "last_block"):
            engine.STAGE_DTYPE[st] = torch.bfloat16
    m.train()
    pc, _ = W.synth_clouds(4, 1024, seed=77)
    m.point_encoder.fps_start = torch.from_numpy(g["fps_start"]).cuda()
    m.point_encoder.drop_path_factors = torch.from_numpy(g["dp_masks"]).cuda()
    tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
    tr.fused_head = False
    inner = tr._loss
    tr._loss = lambda a, b: inner(a, b) * s
    tr.step(torch.from_numpy(pc).cuda(), torch.from_numpy(g["labels"]).cuda())
    torch.cuda.synchronize()
    engine.STAGE_DTYPE.clear()
    return {k: p.grad.detach().cpu().double() / s for k, p in m.named_parameters() if p.grad is not None}


heads = [int(a) for a in sys.argv[1:] if a.isdigit()] or [0, 3]
for h in heads:
    g = np.load(os.path.join(G, f"g_step_h{h}.npz"))
    pc, _ = W.synth_clouds(4, 1024, seed=77)
    masks = [(torch.from_numpy(a[0]), torch.from_numpy(a[1])) for a in g["dp_masks"]]
    ids, name_lengths = M.tokenize_prompts(M.dataset_classnames("modelnet40"), 32)
    res = O.train_step(W.ulip_pointbert_state_dict(seed=0), torch.from_numpy(pc), torch.from_numpy(g["labels"]), g["fps_start"],
                       W.synth_prompt_embedding(40, 0), name_lengths, g["eot"].astype(np.int64), head_type=h, dp_masks=masks)
    gkeys = ["prompt_learner.learnable_tokens"] + (["point_encoder.blocks.blocks.11.mlp.fc2.weight", "point_encoder.blocks.blocks.11.attn.qkv.weight",
                                                    "point_encoder.cls_head_finetune.0.weight"] if h >= 3 else [])
    gkeys = [k for k in gkeys if k in res["grads"]]
    print(f"\nhead_type {h}: rel-L2 error of the gradients against the oracle, loss scaled by s before backward"
          f" (max |grad| of the tokens at s = 1: {res['grads'][gkeys[0]].abs().max().item():.2e})")
    print(f"{'operands':10s} {'s':>10s}" + "".join(f" {k.split('.')[-2] + '.' + k.split('.')[-1]:>26s}" for k in gkeys))
    for name, bf in (("f16 (default)", False), ("all bf16", True)):
        for s in SCALES:
            gr = run(h, g, s, bf)
            print(f"{name:13s} {s:10.2e}" + "".join(f" {((gr[k] - res['grads'][k].double()).norm() / res['grads'][k].double().norm()).item():26.5f}" for k in gkeys))
