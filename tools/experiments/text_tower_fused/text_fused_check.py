"""Dev tool (GPU): the fused text tower (csrc/text_tower.hip) against the per-layer launches -- error vs the fp32 tower and time.
    python tools/text_fused_check.py"""
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppt_amd import weights as W                     # noqa: E402
from ppt_amd.models import ULIP_models as M          # noqa: E402

torch.cuda.set_device(0)
args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                       num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=0, evaluate_3d=False, ulip2=False, synthetic_weights=True)
res = {}
for mode in ("f32", "unfused", "fused"):
    m = M.ULIP_PointBERT(args)
    m.load_state_dict(W.ulip_pointbert_state_dict(seed=0), strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding_from_tokens(m.tokenized_prompts, seed=0)
    m.cuda().set_precision(torch.float32 if mode == "f32" else torch.bfloat16)
    m.overlap_text_tower = False
    m.use_hip_graphs = False
    m.fused_text_tower = mode == "fused"
    cot = torch.randn(40, 512, generator=torch.Generator().manual_seed(1)).cuda()

    def step():
        m.zero_grad()
        te = m._text_raw()
        (te * cot).sum().backward()
        return te
    te = step()
    torch.cuda.synchronize()
    res[mode] = (te.detach().float().clone(), m.prompt_learner.learnable_tokens.grad.detach().clone())
    print(mode, "finite:", bool(torch.isfinite(res[mode][0]).all()), bool(torch.isfinite(res[mode][1]).all()), flush=True)
    m.use_hip_graphs = True
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    print(f"{mode}: fwd+bwd {1e3 * (time.perf_counter() - t0) / 30:.3f} ms per iteration (graph replay)", flush=True)


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


for k in ("unfused", "fused"):
    print(k, "vs f32: features", rel(res[k][0], res["f32"][0]), "token grad", rel(res[k][1], res["f32"][1]))
print("fused vs unfused:", rel(res["fused"][0], res["unfused"][0]), rel(res["fused"][1], res["unfused"][1]))
