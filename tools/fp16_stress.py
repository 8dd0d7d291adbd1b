"""fp16 stress harness (VERDICT r3 #5a): how far are the IEEE-half stages of the mixed 16-bit mode from 65 504 on weights of
checkpoint-like magnitude?  Real ULIP / SLIP checkpoints are not available offline; their known hazards are emulated on the
synthetic state dict:
    gain g       every LayerNorm / BatchNorm weight of both towers x g           (trained norms: up to ~10)
    outliers     4 channels of every LayerNorm get another x 30                  (the "massive activation" channels of transformers)
    stream       cls_token, pos_embed output layer, token/positional embeddings x 10  (a residual stream of O(10-100))
For each level the script runs one C2-shaped training step (B = 8) with every kernel UNFUSED (so that every 16-bit intermediate
-- the MLP hidden layer too -- passes through a probed wrapper) and prints, per tower and kernel, max |x| of the 16-bit activations
as a fraction of half's largest finite value, in the default formats (half) and with everything demoted to bf16; then it runs the
REAL (fused, graphed) Trainer on the harshest level and shows the run-time defence: flag -> skipped step -> demotion -> finite loss.

    python tools/fp16_stress.py
"""
import contextlib
import io
import os
import sys
import warnings
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppt_amd import engine, health, ops, weights as W          # noqa: E402
from ppt_amd.models import ULIP_models as M                     # noqa: E402
from ppt_amd.train import Trainer                               # noqa: E402

B, N = 8, 1024
HALF_MAX = 65504.0
LEVELS = [("synthetic (std 0.02, gains 1)", 1.0, False, False), ("gain 3", 3.0, False, False), ("gain 10", 10.0, False, False),
          ("gain 3 + outlier channels", 3.0, True, False), ("gain 10 + outlier channels", 10.0, True, False),
          ("gain 10 + outliers + stream x 10", 10.0, True, True)]


def stressed_state(gain, outliers, stream):
    sd = W.ulip_pointbert_state_dict(seed=0)
    g = torch.Generator().manual_seed(7)
    for k, v in sd.items():
        norm_w = (k.endswith("norm1.weight") or k.endswith("norm2.weight") or k.endswith("ln_1.weight") or k.endswith("ln_2.weight")
                  or k.endswith("ln_final.weight") or k.endswith("point_encoder.norm.weight")
                  or (("first_conv.1." in k or "second_conv.1." in k) and k.endswith(".weight")))
        if norm_w:
            v = v * gain
            if outliers and v.numel() >= 256 and "conv" not in k:
                idx = torch.randperm(v.numel(), generator=g)[:4]
                v = v.clone()
                v[idx] *= 30.0
            sd[k] = v
        if stream and (k.endswith("cls_token") or k.endswith("pos_embed.2.weight") or k == "positional_embedding" or k == "token_embedding.weight"):
            sd[k] = v * 10.0
    return sd


def build(sd):
    names = M.dataset_classnames("modelnet40")
    args = SimpleNamespace(classnames=names, template_init='', class_name_position='middle', num_learnable_prompt_tokens=32, gpu=0,
                           task='cls', head_type=0, evaluate_3d=False, ulip2=False, synthetic_weights=True)
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.ULIP_PointBERT(args)
    m.load_state_dict(sd, strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(len(names), seed=0) * (10.0 if sd["positional_embedding"].abs().max() > 0.5 else 1.0)
    m.cuda().set_precision("mixed16")
    m.train()
    return m


pc = torch.from_numpy(W.synth_clouds(B, N, seed=3)[0]).cuda()
labels = torch.from_numpy(np.random.default_rng(0).integers(0, 40, size=(B,))).cuda()


def probed_step(sd, demoted):
    """one forward + backward with every intermediate exposed; -> ({(tower, kernel): (max |x|, non-finite?)}, loss)"""
    saved = (engine.FUSED_MLP, engine.FUSED_PROJ, engine.ROWGEMM_MIN_ROWS, engine.FUSED_CONV12)
    engine.FUSED_MLP, engine.FUSED_PROJ, engine.ROWGEMM_MIN_ROWS = False, False, 1 << 30
    m = build(sd)
    if demoted:                                # (the model's own set: ppt_amd/health.py demote() fills it at run time)
        m.demoted.update(("tokenizer", "blocks", "last_block"))
    m.text_f16 = not demoted
    m.use_hip_graphs = m.point_encoder.use_hip_graphs = False
    m.overlap_text_tower = False
    rec, side = {}, ["point"]
    verbose = [os.environ.get("PPT_STRESS_SHAPES") == "1"]

    def probe(name, t):
        key = (side[0], f"{name} {tuple(t.shape)}" if verbose[0] else name)
        if key not in rec:
            rec[key] = (torch.zeros(1, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda"))
        ops.health_check(t.contiguous(), rec[key][1], 1, rec[key][0])
    ops.probe = probe
    try:
        side[0] = "text"
        te = m._text_embed()
        side[0] = "point"
        emb = m.encode_pc(pc)
        logits = m.logit_scale.exp() * emb @ te.t()
        loss = torch.nn.functional.cross_entropy(logits, labels, label_smoothing=0.2)
        side[0] = "backward"
        loss.backward()
        torch.cuda.synchronize()
    finally:
        ops.probe = None
        engine.FUSED_MLP, engine.FUSED_PROJ, engine.ROWGEMM_MIN_ROWS, engine.FUSED_CONV12 = saved
    g = m.prompt_learner.learnable_tokens.grad
    return {k: (v[0].item(), bool(v[1].item())) for k, v in rec.items()}, loss.item(), bool(torch.isfinite(g).all()) if g is not None else False


print(f"C2-shaped step, B = {B}; max |x| of every 16-bit activation tensor as a fraction of half's maximum (65 504); '!' = non-finite values present")
for name, gain, outl, stream in LEVELS:
    sd = stressed_state(gain, outl, stream)
    for demoted in (False, True):
        rec, loss, gfin = probed_step(sd, demoted)
        fmt = "bf16 (demoted)" if demoted else "half (default)"
        worst = {}
        for (tower, kern), (mx, bad) in rec.items():
            w_ = worst.setdefault(tower, [0.0, "", False])
            if mx > w_[0]:
                w_[0], w_[1] = mx, kern
            w_[2] = w_[2] or bad
        cells = " | ".join(f"{t}: {w_[0]:9.1f} = {w_[0] / HALF_MAX:7.4f} of max ({w_[1]}){' !' if w_[2] else ''}" for t, w_ in sorted(worst.items()))
        print(f"  {name:34s} {fmt:15s} loss {loss:10.4f} grad finite {gfin!s:5s} | {cells}", flush=True)
        bad = sorted(f"{t}/{k}" for (t, k), (_, b) in rec.items() if b)
        if bad:
            print(f"      non-finite values in: {', '.join(bad)}")

print("\nrun-time defence on the harshest level (the real fused / graphed Trainer, poll every step):")
engine.DEMOTED.clear()
os.environ["PPT_HEALTH_EVERY"] = "1"
m = build(stressed_state(*LEVELS[-1][1:]))
tr = Trainer(m, distributed=False)
with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter("always")
    for it in range(8):
        loss, _ = tr.step(pc, labels)
        tr.finish()
        print(f"  step {it}: loss {loss.item():10.4f} | skipped gradient elements so far {tr.nonfinite_grad_elements()} | demotions {len(tr.demotions)}"
              f" | half stages left: text {m.text_f16}, point {'blocks' not in engine.DEMOTED}", flush=True)
for w_ in caught:
    if issubclass(w_.category, RuntimeWarning):
        print("  warning:", str(w_.message)[:300])
print("  parameters finite:", all(bool(torch.isfinite(p).all()) for p in m.parameters()))
