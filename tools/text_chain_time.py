"""Dev tool (GPU): forward + backward of the text tower alone (graph replay), operand format from argv[1] (bf16 | f16 | f32)."""
import os, sys, time
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppt_amd import weights as W
from ppt_amd.models import ULIP_models as M
torch.cuda.set_device(0)
fmt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "f16"]
args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                       num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=0, evaluate_3d=False, ulip2=False, synthetic_weights=True)
m = M.ULIP_PointBERT(args)
m.load_state_dict(W.ulip_pointbert_state_dict(seed=0), strict=False)
m.prompt_learner.embedding = W.synth_prompt_embedding_from_tokens(m.tokenized_prompts, seed=0)
m.cuda().set_precision("mixed16")
m.text_precision = fmt
m.overlap_text_tower = False
cot = torch.randn(40, 512, generator=torch.Generator().manual_seed(1)).cuda()
def step():
    m.zero_grad()
    te = m._text_raw()
    (te * cot).sum().backward()
for _ in range(6):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print(f"{sys.argv[1] if len(sys.argv) > 1 else 'f16'}: text tower fwd+bwd {1e3 * (time.perf_counter() - t0) / 50:.3f} ms per iteration")
