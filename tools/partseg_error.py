"""Where does the 16-bit mode's part-segmentation gradient error come from?  

One C5-sized step (B x 2048 points, the literal main_partseg.py:204-215 loop) in fp32, then the default 16-bit mode with ONE stage at
a time switched back to fp32 (and the converse: fp32 everywhere except one stage), reporting the logits error and the worst
decoder weight-matrix gradient (rel-L2 against the fp32 step).  Stages: backbone (tokenizer + blocks), text tower, decoder GEMMs,
per-point head.  Usage: python tools/partseg_error.py [B]"""
import contextlib
import io
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppt_amd import engine, weights as W                    # noqa: E402
from ppt_amd.models import ULIP_models as M                 # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = 2048
F32, H = torch.float32, torch.bfloat16
pc_np, s0 = W.synth_clouds(B, N, seed=5, duplicates=True)
_, s1 = W.synth_clouds(B, N, seed=6)
_, s2 = W.synth_clouds(B, N, seed=7)
pc = torch.from_numpy(pc_np).cuda()
labels = torch.from_numpy(np.random.default_rng(2).integers(0, 50, size=(B, N))).cuda()
onehot = torch.nn.functional.one_hot(torch.arange(B) % 16, 16).float().cuda()
drop = (torch.rand(B, N, 128, generator=torch.Generator().manual_seed(3)) >= 0.5).float() * 2.0
STAGES = ("tokenizer", "blocks", "text", "fp", "dgcnn", "conv1", "head")
NOISE = [0.0]            # relative Gaussian noise put on the backbone's three feature taps (conditioning experiment)
_pef = engine.point_encoder_forward


def _noisy_pef(*a, **k):
    out = _pef(*a, **k)
    if k.get("fetch") is not None and NOISE[0]:
        g = torch.Generator(device="cuda").manual_seed(11)
        feats, ctr = out
        feats = [f * (1 + NOISE[0] * torch.randn(f.shape, generator=g, device=f.device)) for f in feats]
        return feats, ctr
    return out


engine.point_encoder_forward = _noisy_pef


def run(prec):
    """prec: {stage: torch.float32 | torch.bfloat16 (= the performance mode's format for that stage)}"""
    args = SimpleNamespace(classnames=M.dataset_classnames("shapenetpart"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='partseg', head_type=0, evaluate_3d=False, ulip2=False,
                           synthetic_weights=True)
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.ULIP_PointBERT_partseg(args)
    m.load_state_dict(W.ulip_partseg_state_dict(seed=0), strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(50, seed=0)
    m.cuda().set_precision(H)
    m.train()
    pe = m.point_encoder
    engine.STAGE_DTYPE.clear()
    if prec["tokenizer"] == F32:
        engine.STAGE_DTYPE.update(tokenizer=F32)
    if prec["blocks"] == F32:
        engine.STAGE_DTYPE.update(blocks=F32, last_block=F32)
    if prec["text"] == F32:
        m.text_precision = F32
    if prec["fp"] == F32:
        for mod in (pe.propagation_0, pe.propagation_1, pe.propagation_2):
            mod.precision = F32
    if prec["dgcnn"] == F32:
        for mod in (pe.dgcnn_pro_1, pe.dgcnn_pro_2):
            mod.precision = F32
    if prec["conv1"] == F32:
        pe._dec_precision = F32
    if prec["head"] == F32:
        m._head_precision = lambda: F32
    pe.fps_start = tuple(torch.from_numpy(s).cuda() for s in (s0, s1, s2))
    pe.drop_path_factors = torch.ones(12, 2, B)
    pe.dropout_mask = drop
    pred = m(pc, onehot)
    loss = torch.nn.CrossEntropyLoss(label_smoothing=0.2)(pred.reshape(-1, 50), labels.reshape(-1))
    loss.backward()
    torch.cuda.synchronize()
    engine.STAGE_DTYPE.clear()
    return pred.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def report(name, ref, got):
    (p0, g0), (p1, g1) = ref, got
    worst2, worst1 = ("", 0.0), ("", 0.0)
    for n in g0:
        if g0[n].norm().item() < 1e-7:
            continue
        r = ((g1[n].double() - g0[n].double()).norm() / g0[n].double().norm()).item()
        if g0[n].dim() >= 2 and r > worst2[1]:
            worst2 = (n, r)
        if g0[n].dim() == 1 and r > worst1[1]:
            worst1 = (n, r)
    tok = "prompt_learner.learnable_tokens"
    rt = ((g1[tok].double() - g0[tok].double()).norm() / g0[tok].double().norm()).item()
    print(f"{name:34s} logits abs err {(p1 - p0).abs().max().item():8.4f}  worst matrix {worst2[1]:.4f} ({worst2[0].replace('point_encoder.', '')})"
          f"  worst 1-D {worst1[1]:.4f} ({worst1[0].replace('point_encoder.', '')})  tokens {rt:.4f}", flush=True)


ref = run({s: F32 for s in STAGES})
print(f"B = {B}: |logits| max {ref[0].abs().max().item():.1f}")
report("all 16-bit", ref, run({s: H for s in STAGES}))
for s in STAGES:
    report(f"16-bit, {s} in fp32", ref, run({t: (F32 if t == s else H) for t in STAGES}))
for s in STAGES:
    report(f"fp32, only {s} 16-bit", ref, run({t: (H if t == s else F32) for t in STAGES}))
report("fp32 again (run-to-run)", ref, run({s: F32 for s in STAGES}))
for nz in (1e-4, 3e-4, 1e-3, 3e-3):
    NOISE[0] = nz
    report(f"fp32 + {nz:g} rel. noise on the taps", ref, run({s: F32 for s in STAGES}))
NOISE[0] = 0.0
