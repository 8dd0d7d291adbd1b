"""Dev tool (GPU): where a layer of the fused text tower spends its cycles -- shader-clock stamps of workgroup 0 / wave 0 at the
phase boundaries of layer 1 (ppt_text_tower_params.dbg).   python tools/text_tower_stamps.py"""
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppt_amd import engine, ops, weights as W            # noqa: E402
from ppt_amd.models import ULIP_models as M               # noqa: E402

torch.cuda.set_device(0)
args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                       num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=0, evaluate_3d=False, ulip2=False, synthetic_weights=True)
m = M.ULIP_PointBERT(args)
m.load_state_dict(W.ulip_pointbert_state_dict(seed=0), strict=False)
m.prompt_learner.embedding = W.synth_prompt_embedding_from_tokens(m.tokenized_prompts, seed=0)
m.cuda().set_precision(torch.bfloat16)
m.overlap_text_tower = False
m.use_hip_graphs = False
dbg = torch.zeros(2, 64, dtype=torch.int64, device="cuda")
real = ops.text_tower_params


def patched(**kw):
    p = real(**kw)
    p.dbg = dbg[0 if kw["g"] is None else 1].data_ptr()
    return p


ops.text_tower_params = patched
cot = torch.randn(40, 512, generator=torch.Generator().manual_seed(1)).cuda()
for _ in range(3):
    m.zero_grad()
    te = m._text_raw()
    (te * cot).sum().backward()
torch.cuda.synchronize()
d = dbg.cpu().numpy()
names_f = ["params->LDS", "barrier", "LN1", "in_proj x3", "vm_barrier", "attention", "barrier", "out_proj", "vm_barrier", "LN2"]
f = d[0]
print("forward, layer 1, cycles per phase (100 MHz memtime ticks x ~24 = shader cycles at 2.4 GHz; memtime is the 100 MHz counter):")
for i in range(9):
    print(f"  {i}->{i+1} {names_f[i] if i < len(names_f) else ''}: {int(f[i + 1] - f[i])}")
print("  LN2 -> first slab:", int(f[9] - f[8]))
for j in range(4):
    print(f"  slab {j}: c_fc {int(f[10 + 3 * j] - (f[9] if j == 0 else f[12 + 3 * (j - 1)]))}  barrier {int(f[11 + 3 * j] - f[10 + 3 * j])}  c_proj {int(f[12 + 3 * j] - f[11 + 3 * j])}")
print("  epilogue:", int(f[22] - f[21]), " vm_barrier:", int(f[23] - f[22]), " LAYER:", int(f[23] - f[0]))
b = d[1]
names_b = ["MLP bwd (8 units)", "LN2 bwd", "out_proj dX", "barrier", "attention bwd", "vm_barrier", "in_proj dX x3 + staging", "LN1 bwd"]
print("backward, layer 1:")
for i in range(8):
    print(f"  {names_b[i]}: {int(b[i + 1] - b[i])}")
print("  LAYER:", int(b[8] - b[0]))
