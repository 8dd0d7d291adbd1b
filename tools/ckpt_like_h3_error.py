#!/usr/bin/env python3
"""on checkpoint-LIKE weights, head_type 3, WHERE does the un-frozen block's gradient error (0.09-0.117 rel-L2 in the mixed
mode) come from -- the block's own 16-bit forward / backward, or the frozen stages in front of it whose output it differentiates at?
The golden step of tests/golden/g_step_h3_ckpt.npz (reference) with single stages on fp32 operands (engine.STAGE_DTYPE, eager).
    python3 tools/ckpt_like_h3_error.py"""
import contextlib, io, os, sys, warnings
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ppt_amd import engine, weights as W
from ppt_amd.models import ULIP_models as M
from ppt_amd.train import Trainer

g = np.load(os.path.join(ROOT, "tests", "golden", "g_step_h3_ckpt.npz"))


def run(f32_stages=(), precision="mixed16", split=False):
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=3, evaluate_3d=False, ulip2=False, synthetic_weights=True)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = M.ULIP_PointBERT(args)
        sd = W.checkpoint_like(W.ulip_pointbert_state_dict(seed=0), seed=0)
        m.load_state_dict(sd, strict=False)
        m.prompt_learner.embedding = W.synth_prompt_embedding(40, seed=0)
        m.cuda().set_precision(precision)
        m.use_hip_graphs = m.point_encoder.use_hip_graphs = False
        m.overlap_text_tower = False
        engine.STAGE_DTYPE.clear()
        for st in f32_stages:
            engine.STAGE_DTYPE[st] = torch.float32
        try:
            m.train()
            pc, _ = W.synth_clouds(4, 1024, seed=77)
            m.point_encoder.fps_start = torch.from_numpy(g["fps_start"]).cuda()
            m.point_encoder.drop_path_factors = torch.from_numpy(g["dp_masks"]).cuda()
            tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
            loss, pred = tr.step(torch.from_numpy(pc).cuda(), torch.from_numpy(g["labels"]).cuda())
            tr.finish()
            torch.cuda.synchronize()
        finally:
            engine.STAGE_DTYPE.clear()
    worst2, worst1, wk = 0.0, 0.0, ""
    for k, q in m.named_parameters():
        if not q.requires_grad:
            continue
        gg = q.grad.detach().cpu()
        if "grad_" + k in g.files:
            gr = torch.from_numpy(g["grad_" + k]); rel = ((gg - gr).norm() / gr.norm()).item()
        else:
            gr = torch.from_numpy(g["gradsub_" + k]); rel = ((gg.flatten()[::97] - gr).norm() / gr.norm()).item()
        if gg.dim() > 1:
            if rel > worst2: worst2, wk = rel, k
        else:
            worst1 = max(worst1, rel)
    return float(np.abs(pred.detach().float().cpu().numpy() - g["logits"]).max()), worst2, wk, worst1


rows = [("fp32 mode", dict(precision="fp32")), ("split16 mode", dict(precision="split16")), ("mixed16 (text tower calibrated)", {}),
        ("+ last block fp32 operands", dict(f32_stages=("last_block",))),
        ("+ blocks 0-10 fp32", dict(f32_stages=("blocks",))),
        ("+ tokenizer fp32", dict(f32_stages=("tokenizer",))),
        ("+ blocks 0-10 + last block fp32", dict(f32_stages=("blocks", "last_block"))),
        ("+ tokenizer + all blocks fp32", dict(f32_stages=("tokenizer", "blocks", "last_block")))]
print("| configuration | logits max abs err | worst matrix / token gradient rel-L2 (which) | worst 1-D gradient rel-L2 |")
print("|---|---|---|---|")
for name, kw in rows:
    e, w2, wk, w1 = run(**kw)
    print(f"| {name} | {e:.3f} | {w2:.4f} ({wk.replace('point_encoder.blocks.blocks.11.', '')}) | {w1:.4f} |", flush=True)
