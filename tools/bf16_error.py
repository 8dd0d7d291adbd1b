#!/usr/bin/env python3
"""Where the bf16 performance mode's error comes from: the golden train step (B = 4, 1024 points) against the
oracle with ONE stage at a time flipped to fp32 operands -- tokenizer (mini-PointNet + reduce_dim + pos_embed), blocks 0-10,
block 11, text tower -- and all of them / none of them.  For each run: logits max |err| and rms err, loss err, rel-L2 of the
gradient of the learnable tokens (and of the last block's fc2 for head_type 3), and each stage's SHARE of the all-bf16 squared
error (1 - err_with_stage_in_fp32^2 / err_all_bf16^2; shares of independent errors add up to ~1).

    python tools/bf16_error.py [head_types ...]        # default: 0 3
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
STAGES = ("tokenizer", "blocks", "last_block", "text")
TEXT_PARTS = ("text_attn", "text_mlp")          # finer: one half of every text layer in fp32 (engine.STAGE_DTYPE)
extra = [a for a in sys.argv[1:] if "=" in a]                      # e.g. PPT-style experiment switches: key=value -> engine attr


def run(h, g, f32_stages):
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=h, evaluate_3d=False, ulip2=False,
                           synthetic_weights=True)
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.ULIP_PointBERT(args)
    sd = W.ulip_pointbert_state_dict(seed=0)
    m.load_state_dict(sd, strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(40, seed=0)
    m.cuda().set_precision("mixed16")
    m.use_hip_graphs = m.point_encoder.use_hip_graphs = False
    m.overlap_text_tower = False
    engine.STAGE_DTYPE.clear()
    for st in f32_stages:
        if st == "text":
            m.text_precision = torch.float32
        elif st == "text_bf16":
            m.text_precision = torch.bfloat16
        elif st == "tokenizer_bf16":
            engine.STAGE_DTYPE["tokenizer"] = torch.bfloat16
        elif st == "blocks_bf16":
            engine.STAGE_DTYPE["blocks"] = engine.STAGE_DTYPE["last_block"] = torch.bfloat16
        else:
            engine.STAGE_DTYPE[st] = torch.float32
    m.train()
    pc, _ = W.synth_clouds(4, 1024, seed=77)
    m.point_encoder.fps_start = torch.from_numpy(g["fps_start"]).cuda()
    m.point_encoder.drop_path_factors = torch.from_numpy(g["dp_masks"]).cuda()
    tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
    loss, pred = tr.step(torch.from_numpy(pc).cuda(), torch.from_numpy(g["labels"]).cuda())
    torch.cuda.synchronize()
    engine.STAGE_DTYPE.clear()
    grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None}
    return loss.item(), pred.detach().cpu().float(), grads


heads = [int(a) for a in sys.argv[1:] if a.isdigit()] or [0, 3]
for h in heads:
    g = np.load(os.path.join(G, f"g_step_h{h}.npz"))
    pc, _ = W.synth_clouds(4, 1024, seed=77)
    masks = [(torch.from_numpy(a[0]), torch.from_numpy(a[1])) for a in g["dp_masks"]]
    sd = W.ulip_pointbert_state_dict(seed=0)
    names = M.dataset_classnames("modelnet40")
    ids, name_lengths = M.tokenize_prompts(names, 32)
    res = O.train_step(sd, torch.from_numpy(pc), torch.from_numpy(g["labels"]), g["fps_start"], W.synth_prompt_embedding(40, 0),
                       name_lengths, g["eot"].astype(np.int64), head_type=h, dp_masks=masks)
    ref_logits = res["logits"]
    gkeys = ["prompt_learner.learnable_tokens"] + (["point_encoder.blocks.blocks.11.mlp.fc2.weight", "point_encoder.blocks.blocks.11.attn.qkv.weight"] if h >= 3 else [])
    rows = {}
    configs = [("all bf16", ("text_bf16", "blocks_bf16", "tokenizer_bf16")), ("text f16 only", ("blocks_bf16", "tokenizer_bf16")),
               ("text + blocks f16", ("tokenizer_bf16",)), ("default: all f16", ())] + \
        [(f"{st} in fp32", (st,)) for st in STAGES] + [("all fp32", STAGES)]
    for name, f32 in configs:
        loss, lg, grads = run(h, g, f32)
        e = (lg - ref_logits)
        row = dict(lmax=e.abs().max().item(), lrms=e.pow(2).mean().sqrt().item(), loss=abs(loss - res["loss"].item()))
        for k in gkeys:
            row[k] = ((grads[k] - res["grads"][k]).norm() / res["grads"][k].norm()).item()
        rows[name] = row
    base = rows["all bf16"]
    print(f"\nhead_type {h}: error of the golden train step against the oracle (|logits| max {ref_logits.abs().max().item():.1f})")
    hdr = f"{'configuration':22s} {'logits max':>10s} {'rms':>8s} {'share':>6s} {'loss':>8s}" + "".join(f" {('grad ' + k.split('.')[-2] + '.' + k.split('.')[-1])[:24]:>24s} {'share':>6s}" for k in gkeys)
    print(hdr)
    for name, row in rows.items():
        sh = lambda a, b: (1.0 - (a / b) ** 2) if b > 0 else 0.0
        line = f"{name:22s} {row['lmax']:10.4f} {row['lrms']:8.4f} {sh(row['lrms'], base['lrms']):6.2f} {row['loss']:8.5f}"
        for k in gkeys:
            line += f" {row[k]:24.5f} {sh(row[k], base[k]):6.2f}"
        print(line)
