#!/usr/bin/env python3
"""Per-tensor error of the bf16 performance mode against the oracle / golden fixtures of the golden train step (B = 4,
1024 points, head_type 0..3): logits (abs), every gradient (relative L2), BatchNorm statistics.  The numbers behind the
tolerances stated in tests/test_model_gpu.py.      python tools/bf16_error.py [f32]"""
import contextlib, io, os, sys
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import oracle as O
from ppt_amd import weights as W
from ppt_amd.models import ULIP_models as M
from ppt_amd.train import Trainer

prec = torch.float32 if "f32" in sys.argv[1:] else torch.bfloat16
G = os.path.join(ROOT, "tests", "golden")
for h in ((0, 3) if len(sys.argv) > 1 else (0, 1, 2, 3)):
    g = np.load(os.path.join(G, f"g_step_h{h}.npz"))
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=h, evaluate_3d=False, ulip2=False,
                           synthetic_weights=True)
    with contextlib.redirect_stdout(io.StringIO()):
        m = M.ULIP_PointBERT(args)
    sd = W.ulip_pointbert_state_dict(seed=0)
    m.load_state_dict(sd, strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(40, seed=0)
    m.cuda().set_precision(prec)
    if "text=f32" in sys.argv[1:]:
        m.text_precision = torch.float32
    if "point=f32" in sys.argv[1:]:
        m.set_precision(torch.float32)
        m.text_precision = torch.bfloat16
    m.overlap_text_tower = False
    m.train()
    pc, start = W.synth_clouds(4, 1024, seed=77)
    m.point_encoder.fps_start = torch.from_numpy(g["fps_start"]).cuda()
    m.point_encoder.drop_path_factors = torch.from_numpy(g["dp_masks"]).cuda()
    tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
    loss, pred = tr.step(torch.from_numpy(pc).cuda(), torch.from_numpy(g["labels"]).cuda())
    torch.cuda.synchronize()
    masks = [(torch.from_numpy(a[0]), torch.from_numpy(a[1])) for a in g["dp_masks"]]
    res = O.train_step(sd, torch.from_numpy(pc), torch.from_numpy(g["labels"]), g["fps_start"], W.synth_prompt_embedding(40, 0),
                       m.prompt_learner.name_lengths, g["eot"].astype(np.int64), head_type=h, dp_masks=masks)
    lg = pred.detach().cpu().float()
    print(f"head_type {h} ({prec}): loss {loss.item():.5f} (golden {float(g['loss']):.5f}); logits max|err| vs golden "
          f"{np.abs(lg.numpy() - g['logits']).max():.4f}, vs oracle {(lg - res['logits']).abs().max().item():.4f} "
          f"(|logits| max {np.abs(g['logits']).max():.1f}); argmax agree {(lg.argmax(1).numpy() == g['logits'].argmax(1)).mean():.2f}")
    live = dict(m.named_parameters())
    for k, go in res["grads"].items():
        gg = live[k].grad.detach().cpu()
        rel = ((gg - go).norm() / go.norm()).item()
        cos = (gg.flatten() @ go.flatten() / (gg.norm() * go.norm())).item()
        print(f"    grad {k:55s} rel-L2 {rel:.4f}  cos {cos:.5f}  |g| {go.norm().item():.3e}")
