#!/usr/bin/env python3
"""On checkpoint-LIKE weights the CLIP text tower fails its self-check on IEEE-half operands (ULIP_WITH_IMAGE.calibrate_text_precision)
and runs on split16 products.  Does it need them in BOTH halves of a layer?  The golden step g_step_h0_ckpt.npz with the attention half
and the MLP half of the text layers on different operand formats (engine.STAGE_DTYPE "text_attn" / "text_mlp", eager):
text features against the fp32-operand tower, logits and token gradient against the reference fixture.
    python3 tools/ckpt_like_text_halves.py"""
import contextlib, io, os, sys, warnings
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PPT_GRAD_CHECK"] = "off"
import numpy as np
import torch
from ppt_amd import engine, weights as W
from ppt_amd.models import ULIP_models as M
from ppt_amd.train import Trainer

g = np.load(os.path.join(ROOT, "tests", "golden", "g_step_h0_ckpt.npz"))


def run(text_precision, stages):
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=0, evaluate_3d=False, ulip2=False, synthetic_weights=True)
    with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = M.ULIP_PointBERT(args)
        sd = W.checkpoint_like(W.ulip_pointbert_state_dict(seed=0), seed=0)
        m.load_state_dict(sd, strict=False)
        m.prompt_learner.embedding = W.synth_prompt_embedding(40, seed=0)
        m.cuda().set_precision("mixed16")
        m.use_hip_graphs = m.point_encoder.use_hip_graphs = False
        m.overlap_text_tower = False
        m._text_calibrated = True                         # (no self-check: the formats are set here)
        m.text_precision = text_precision
        engine.STAGE_DTYPE.clear()
        engine.STAGE_DTYPE.update(stages)
        try:
            with torch.no_grad():
                feat = m._text_raw().float()
                feat = feat / feat.norm(dim=-1, keepdim=True)
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
    tok = m.prompt_learner.learnable_tokens.grad.detach().cpu()
    k = "grad_prompt_learner.learnable_tokens"
    gr = torch.from_numpy(g[k]) if k in g.files else None
    rel = ((tok - gr).norm() / gr.norm()).item() if gr is not None else float("nan")
    return feat, float(np.abs(pred.detach().float().cpu().numpy() - g["logits"]).max()), rel


f32 = torch.float32
rows = [("text tower fp32 operands as split16 products (what the self-check selects)", f32, {}),
        ("text tower IEEE half (the performance mode's own choice)", None, {}),
        ("attention half split16, MLP half IEEE half", f32, {"text_mlp": torch.float16}),
        ("attention half IEEE half, MLP half split16", f32, {"text_attn": torch.float16})]
ref = None
print("| text tower operands | text features vs the split16 tower (rel-L2) | logits max abs err (|logits| <= 72) | token gradient rel-L2 |")
print("|---|---|---|---|")
for name, tp, st in rows:
    feat, e, rel = run(tp, st)
    if ref is None:
        ref = feat
    print(f"| {name} | {((feat - ref).norm() / ref.norm()).item():.2e} | {e:.3f} | {rel:.4f} |", flush=True)
