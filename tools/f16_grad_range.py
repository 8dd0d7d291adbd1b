#!/usr/bin/env python3
"""How far the fp16 operand stages of the performance mode are from gradient underflow: the golden train step (B = 4) with
the loss multiplied by s = 1, 1/8, ... 1/32768 before backward (a batch of 4/s clouds has per-element gradients of that size:
mean-reduced cross entropy), gradients / s against the oracle's.  fp16 keeps 11 bits down to 6.1e-5 and flushes below 6e-8;
bf16 has fp32's range.  The first column is the operating point of the tests; C2 (B = 32) sits at s = 1/8.

    python tools/f16_grad_range.py [head_types ...] [partseg] [s=<scale> ...]       # default: 0 3
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
SCALES = [float(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('s=')] or [2.0 ** k for k in (15, 12, 9, 6, 3, 0, -3, -6, -9, -12, -15)]


def run(h, g, s, all_bf16):
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=h, evaluate_3d=False, ulip2=False,
                           synthetic_weights=True)
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.ULIP_PointBERT(args)
    m.load_state_dict(W.ulip_pointbert_state_dict(seed=0), strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(40, seed=0)
    m.cuda().set_precision("mixed16")
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
    # (the sweep measures the UN-scaled half backward: the nodes' own gradient scale, ppt_amd/gradscale.py, is switched off)
    from ppt_amd import gradscale
    old_policy, gradscale.POLICY = gradscale.POLICY, "off"
    try:
        tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
        tr.fused_head = False
        inner = tr._loss
        tr._loss = lambda a, b: inner(a, b) * s
        tr.step(torch.from_numpy(pc).cuda(), torch.from_numpy(g["labels"]).cuda())
        torch.cuda.synchronize()
    finally:
        gradscale.POLICY = old_policy
    engine.STAGE_DTYPE.clear()
    return {k: p.grad.detach().cpu().double() / s for k, p in m.named_parameters() if p.grad is not None}


def run_partseg(g, s, all_bf16):
    """the golden part-seg step (B = 2 x 2048 points; the loss is a mean over 4096 rows) -> (logits, gradients / s)"""
    args = SimpleNamespace(classnames=M.dataset_classnames("shapenetpart"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='partseg', head_type=0, evaluate_3d=False, ulip2=False,
                           synthetic_weights=True)
    engine.DECODER_F16 = not all_bf16
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.ULIP_PointBERT_partseg(args)
    m.load_state_dict(W.ulip_partseg_state_dict(seed=0), strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(50, seed=0)
    m.cuda().set_precision("mixed16")
    m.overlap_text_tower = False
    engine.STAGE_DTYPE.clear()
    if all_bf16:
        m.text_precision = torch.bfloat16
        for st in ("tokenizer", "blocks", "last_block"):
            engine.STAGE_DTYPE[st] = torch.bfloat16
    m.train()
    pe = m.point_encoder
    pe.fps_start = tuple(torch.from_numpy(g[k]).cuda() for k in ("s0", "s1", "s2"))
    pe.drop_path_factors = torch.from_numpy(g["dp_masks"]).cuda()
    pe.dropout_mask = torch.from_numpy(np.unpackbits(g["drop"]).reshape(2, 2048, 128).astype(np.float32) * 2.0)
    pc_np, _ = W.synth_clouds(2, 2048, seed=55, duplicates=True)
    labels = torch.from_numpy(g["labels"].astype(np.int64)).cuda()
    pred = m(torch.from_numpy(pc_np).cuda(), torch.from_numpy(g["onehot"]).cuda())
    loss = torch.nn.CrossEntropyLoss(label_smoothing=0.2)(pred.reshape(-1, 50), labels.reshape(-1))
    (loss * s).backward()
    torch.cuda.synchronize()
    engine.STAGE_DTYPE.clear()
    engine.DECODER_F16 = True
    return pred.detach().cpu().numpy(), {k: p.grad.detach().cpu().double() / s for k, p in m.named_parameters() if p.grad is not None}


if "partseg" in sys.argv[1:]:
    g = np.load(os.path.join(G, "g_partseg.npz"), allow_pickle=False)
    top = ("point_encoder.conv1.weight", "prompt_learner.learnable_tokens")
    deep = [k for k in g["trainable"].tolist() if "gradnorm_" + k in g and float(g["gradnorm_" + k]) >= 1e-4 and k not in top
            and g["gradsub_" + k].size > 8 and ("mlp_convs" in k or "layer" in k) and k.endswith("weight") and "gradsub_" + k in g]
    print(f"\npart segmentation (golden step, B = 2 x 2048 points): logits max |err| (|logits| <= {np.abs(g['logits_sub']).max():.0f}) and rel-L2 "
          f"error of gradient samples, loss scaled by s before backward; deep = worst of {len(deep)} decoder weight matrices")
    print(f"{'operands':22s} {'s':>10s} {'logits':>8s} {'conv1.weight':>13s} {'tokens':>9s} {'deep worst':>11s} {'deep median':>12s}")
    for name, bf in (("f16 (default)", False), ("all bf16", True)):
        for s in SCALES:
            lg, gr = run_partseg(g, s, bf)
            rel = lambda k: float(np.linalg.norm(gr[k].flatten()[::211].numpy() - g["gradsub_" + k]) / np.linalg.norm(g["gradsub_" + k]))
            d = sorted(rel(k) for k in deep)
            print(f"{name:22s} {s:10.2e} {np.abs(lg[:, ::16] - g['logits_sub']).max():8.3f} {rel(top[0]):13.4f} {rel(top[1]):9.4f} {d[-1]:11.4f} {d[len(d) // 2]:12.4f}")
    sys.exit(0)

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
