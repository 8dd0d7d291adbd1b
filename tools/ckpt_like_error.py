#!/usr/bin/env python3
"""Which stage of the mixed 16-bit mode loses accuracy on checkpoint-LIKE weight magnitudes (ppt_amd.weights.checkpoint_like): the
golden train step of tests/golden/g_step_h0_ckpt.npz (captured from the reference) with ONE stage at a time on fp32 operands, and
with single kernel choices switched (fused MLP / rowgemm / fused proj off).  Prints logits max |err| against the fixture.
    python tools/ckpt_like_error.py"""
import contextlib, io, os, sys
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ppt_amd import engine, weights as W
from ppt_amd.models import ULIP_models as M
from ppt_amd.train import Trainer

g = np.load(os.path.join(ROOT, "tests", "golden", "g_step_h0_ckpt.npz"))


def run(f32_stages=(), knobs=None, precision="mixed16"):
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=0, evaluate_3d=False, ulip2=False, synthetic_weights=True)
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.ULIP_PointBERT(args)
    sd = W.checkpoint_like(W.ulip_pointbert_state_dict(seed=0), seed=0)
    m.load_state_dict(sd, strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(40, seed=0)
    m.cuda().set_precision(precision)
    m.use_hip_graphs = m.point_encoder.use_hip_graphs = False
    m.overlap_text_tower = False
    engine.STAGE_DTYPE.clear()
    saved = {}
    for k, v in (knobs or {}).items():
        saved[k] = getattr(engine, k)
        setattr(engine, k, v)
    for st in f32_stages:
        if st == "text":
            m.text_precision = torch.float32
        else:
            engine.STAGE_DTYPE[st] = torch.float32
    try:
        m.train()
        pc, _ = W.synth_clouds(4, 1024, seed=77)
        m.point_encoder.fps_start = torch.from_numpy(g["fps_start"]).cuda()
        m.point_encoder.drop_path_factors = torch.from_numpy(g["dp_masks"]).cuda()
        tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
        loss, pred = tr.step(torch.from_numpy(pc).cuda(), torch.from_numpy(g["labels"]).cuda())
        torch.cuda.synchronize()
    finally:
        engine.STAGE_DTYPE.clear()
        for k, v in saved.items():
            setattr(engine, k, v)
    err = np.abs(pred.detach().float().cpu().numpy() - g["logits"])
    gt = m.prompt_learner.learnable_tokens.grad.detach().cpu().numpy()
    rel = np.linalg.norm(gt - g["grad_prompt_learner.learnable_tokens"]) / np.linalg.norm(g["grad_prompt_learner.learnable_tokens"])
    return err.max(), float(np.sqrt((err ** 2).mean())), abs(loss.item() - float(g["loss"])), rel


rows = [("fp32 mode", dict(precision="fp32")), ("all mixed16", {}),
        ("tokenizer fp32", dict(f32_stages=("tokenizer",))), ("blocks fp32", dict(f32_stages=("blocks", "last_block"))),
        ("text fp32", dict(f32_stages=("text",))), ("tokenizer + blocks fp32", dict(f32_stages=("tokenizer", "blocks", "last_block"))),
        ("blocks + text fp32", dict(f32_stages=("blocks", "last_block", "text"))),
        ("text attention half fp32", dict(f32_stages=("text_attn",))), ("text MLP half fp32", dict(f32_stages=("text_mlp",))),
        ("text bf16", None)]
print(f"|logits| <= {np.abs(g['logits']).max():.1f}, loss {float(g['loss']):.2f}")
for name, kw in rows:
    if kw is None:
        st = name.split()[0]
        engine_kw = {}
        if st == "text":
            def run_text_bf16():
                return run()
            # text tower on bf16 operands: ULIP_WITH_IMAGE.text_f16 = False
            import ppt_amd.models.ULIP_models as U
            old = os.environ.get("PPT_TEXT_F16")
            os.environ["PPT_TEXT_F16"] = "0"
            try:
                r = run()
            finally:
                if old is None:
                    os.environ.pop("PPT_TEXT_F16")
                else:
                    os.environ["PPT_TEXT_F16"] = old
        else:
            key = {"tokenizer": "TOKENIZER_F16", "blocks": "BLOCKS_F16"}[st]
            r = run(knobs={key: False})
    else:
        r = run(**kw)
    print(f"{name:32s} logits max {r[0]:9.4f} rms {r[1]:9.4f} | loss err {r[2]:9.4f} | token-grad rel-L2 {r[3]:8.4f}", flush=True)
