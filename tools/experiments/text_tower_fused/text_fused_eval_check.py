"""Dev tool (GPU): run-to-run reproducibility of the fused text tower's forward -- the saved activations of two identical calls,
layer by layer (the first tensor that differs names the phase with the hazard)."""
import os, sys
from types import SimpleNamespace
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppt_amd import engine, ops, weights as W
from ppt_amd.models import ULIP_models as M
torch.cuda.set_device(0)
args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                       num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=0, evaluate_3d=False, ulip2=False, synthetic_weights=True)
m = M.ULIP_PointBERT(args)
m.load_state_dict(W.ulip_pointbert_state_dict(seed=0), strict=False)
m.prompt_learner.embedding = W.synth_prompt_embedding_from_tokens(m.tokenized_prompts, seed=0)
m.cuda().set_precision(torch.bfloat16)
m.text_precision = torch.bfloat16
sd, wc = m._live_state(), m._cache()
pl = m.prompt_learner
C, L, P = 40, m._text_len(), pl.shared_prefix()
NP = engine.text_group_size(C, L, P)
base, slot, pos_rows, rows_of, Mr, eot_rows = pl.row_layout(sd["positional_embedding"], L, P, group=NP)
x0 = ops.prompt_rows(base, slot, pl.learnable_tokens.detach().float().contiguous(), pos_rows)
RW = P + NP * (L - P)
runs = []
for rep in range(4):
    out, s = engine.text_tower_forward_fused(sd, wc, x0, C, L, P, 8, 12, True, eot_rows)
    torch.cuda.synchronize()
    runs.append({k: s[k].clone() for k in ("X", "XM", "QKV", "A", "PRE", "LSE", "ST")} | {"out": out.clone()})
valid = torch.zeros(Mr, dtype=torch.bool, device="cuda")
for g in range(Mr // RW):
    valid[g * RW:g * RW + P + NP * (L - P)] = True
for rep in range(1, 4):
    print(f"--- run {rep} vs run 0: out max diff {float((runs[rep]['out'] - runs[0]['out']).abs().max()):.3e}")
    found = False
    for l in range(12):
        for k in ("QKV", "A", "XM", "PRE", "X"):
            a, b = runs[rep][k][l][valid].float(), runs[0][k][l][valid].float()
            d = (a - b).abs()
            if float(d.max()) > 0:
                rows = torch.nonzero(d.reshape(d.shape[0], -1).amax(1) > 0).flatten()
                vr = torch.nonzero(valid).flatten()[rows]
                cols = torch.nonzero(d.reshape(d.shape[0], -1).amax(0) > 0).flatten()
                print(f"   first difference: layer {l} tensor {k}: max {float(d.max()):.3e}; rows (global) {vr[:12].tolist()} (in-group {[int(r) % RW for r in vr[:12]]}), "
                      f"cols {cols[:8].tolist()} .. {cols[-4:].tolist()} ({len(cols)} cols)")
                found = True
                break
        if found:
            break

# ---- what is wrong in the differing entries of x_mid (layer 0): expected = A(l0) @ W_out^T + b_out + x0
w_out = sd["transformer.resblocks.0.attn.out_proj.weight"].detach().to(torch.bfloat16).float()
b_out = sd["transformer.resblocks.0.attn.out_proj.bias"].detach().float()
for rep in range(4):
    A0 = runs[rep]["A"][0].float()
    acc = A0 @ w_out.t()
    exp = acc + b_out + x0
    got = runs[rep]["XM"][0]
    d = (got - exp).abs() * valid.view(-1, 1)
    bad = torch.nonzero(d > 0.05)
    print(f"run {rep}: {bad.shape[0]} entries of x_mid(l0) off by > 0.05 from A @ W^T + b + x0")
    for r, c in bad[:6].tolist():
        print(f"    row {r} (in-group {r % RW}) col {c}: got {got[r, c]:.4f} expected {exp[r, c]:.4f}; acc {acc[r, c]:.4f} bias {b_out[c]:.4f} x0 {x0[r, c]:.4f}; got - x0 - bias = {got[r, c] - x0[r, c] - b_out[c]:.4f}; "
              f"x0 at col+1..3: {x0[r, c + 1]:.3f} {x0[r, c + 2]:.3f} {x0[r, c + 3]:.3f}, x0[r+16,c] {x0[min(r + 16, Mr - 1), c]:.3f}")
