"""GPU: the CALLABLE sub-modules (ppt_amd/blocks.py behind Mlp / Attention / Block / TransformerEncoder / ResidualAttentionBlock /
Transformer .forward) against fixtures G5 / G6 made from the reference modules (tests/golden/make_golden.py: gen_blocks).
Reference: models/pointbert/point_encoder.py:24-30, 46-58, 76-79, 99-110; models/ULIP_models.py:49-67, 203-222.

Tolerances: fp32 parity mode 2e-4 relative (L2) on outputs and gradients; bf16 mode 3e-2."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from ppt_amd import weights as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "g_blocks.npz"))


def block_inputs():
    """the seeded inputs of make_golden.block_inputs"""
    g = torch.Generator().manual_seed(2024)
    x = torch.randn(2, 513, 384, generator=g) * 0.5
    pos = torch.randn(2, 513, 384, generator=g) * 0.1
    cot = torch.randn(2, 513, 384, generator=g)
    tcot = torch.randn(40, 512, generator=g)
    xt = torch.randn(37, 3, 512, generator=g) * 0.3
    xtcot = torch.randn(37, 3, 512, generator=g)
    return x, pos, cot, tcot, xt, xtcot


def _sub(t, n=4099):
    f = t.detach().float().flatten().cpu()
    step = max(1, f.numel() // n) | 1
    return f[::step].numpy(), float(f.double().norm().item())


def _model(precision):
    from ppt_amd.models import ULIP_models as M
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=3, evaluate_3d=False, ulip2=False,
                           synthetic_weights=True)
    m = M.ULIP_PointBERT(args)
    m.load_state_dict(W.ulip_pointbert_state_dict(seed=0), strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(len(args.classnames), seed=0)
    m.cuda().set_precision(precision)
    m.eval()
    for p in m.parameters():
        p.requires_grad_(True)
    return m


def _close(name, got, want_sub, want_norm, tol):
    sub, norm = _sub(got)
    rel = np.linalg.norm(sub - want_sub) / (np.linalg.norm(want_sub) + 1e-30)
    print(f"PARITY blocks {name}: rel {rel:.3g} (bound {tol:.3g}), norm {norm:.5g} vs {float(want_norm):.5g}")
    assert rel < tol, (name, rel)
    assert abs(norm - float(want_norm)) < 2 * tol * float(want_norm) + 1e-12, (name, norm, float(want_norm))


@pytest.mark.parametrize("precision,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
def test_point_modules_are_callable_and_match_the_reference(precision, tol):
    m = _model(precision)
    x, pos, cot, _, _, _ = block_inputs()
    x, pos, cot = x.cuda(), pos.cuda(), cot.cuda()
    blk = m.point_encoder.blocks.blocks[11]
    calls = {"block": lambda t: blk(t), "mlp": lambda t: blk.mlp(t), "attn": lambda t: blk.attn(t),
             "encoder": lambda t: m.point_encoder.blocks(t, pos)}
    for name, fn in calls.items():
        m.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = fn(xi)
        assert y.shape == x.shape and y.dtype == torch.float32
        (y * cot).sum().backward()
        etol = tol * (4 if name == "encoder" else 1)             # twelve blocks deep
        _close(f"{name}.y", y, G[f"{name}_y"], G[f"{name}_ynorm"], etol)
        _close(f"{name}.dx", xi.grad, G[f"{name}_dx"], G[f"{name}_dxnorm"], etol)
        if name != "encoder":
            for k, p_ in blk.named_parameters():
                if f"{name}_g_{k}" in G.files:
                    assert p_.grad is not None, (name, k)
                    _close(f"{name}.grad[{k}]", p_.grad, G[f"{name}_g_{k}"], G[f"{name}_gn_{k}"], etol)
    with torch.no_grad():
        part = m.point_encoder.blocks(x, pos, task='partseg')
    assert len(part) == 3
    sub, _ = _sub(part[0])
    assert np.linalg.norm(sub - G["encoder_partseg_y3"]) / np.linalg.norm(G["encoder_partseg_y3"]) < 4 * tol


@pytest.mark.parametrize("precision,tol", [(torch.float32, 2e-4), (torch.bfloat16, 3e-2)])
def test_text_modules_are_callable_and_match_the_reference(precision, tol):
    """G6: encode_text(prompt_learner()) [40,512] + the gradient of learnable_tokens, through the PUBLIC encode_text; the text
    Transformer called as a module on [L, N, D] gives the same features; one ResidualAttentionBlock forward / backward."""
    m = _model(precision)
    _, _, _, tcot, xt, xtcot = block_inputs()
    tcot, xt, xtcot = tcot.cuda(), xt.cuda(), xtcot.cuda()
    m.zero_grad()
    te = m.encode_text(m.prompt_learner(), m.tokenized_prompts)
    (te * tcot).sum().backward()
    ref = torch.from_numpy(G["text_feat"]).cuda()
    rel = ((te.detach() - ref).norm() / ref.norm()).item()
    gref = torch.from_numpy(G["text_gtok"]).cuda()
    grel = ((m.prompt_learner.learnable_tokens.grad - gref).norm() / gref.norm()).item()
    print(f"PARITY blocks text: feat rel {rel:.3g}, token-grad rel {grel:.3g} (bounds {tol:.3g} / {2 * tol:.3g})")
    assert rel < tol and grel < 2 * tol
    # the module path: x = prompts + pos -> [L, N, D] -> transformer -> ln_final -> EOT rows @ text_projection (ULIP_models.py:210-222)
    with torch.no_grad():
        p = m.prompt_learner() + m.positional_embedding
        h = m.transformer(p.permute(1, 0, 2)).permute(1, 0, 2)
        h = m.ln_final(h)
        eot = m.tokenized_prompts.argmax(-1).cuda()
        te2 = h[torch.arange(h.shape[0], device=h.device), eot] @ m.text_projection
    assert ((te2 - ref).norm() / ref.norm()).item() < 2 * tol
    rb = m.transformer.resblocks[0]
    m.zero_grad()
    xi = xt.clone().requires_grad_(True)
    y = rb(xi)
    (y * xtcot).sum().backward()
    _close("resblock.y", y, G["resblock_y"], G["resblock_ynorm"], tol)
    _close("resblock.dx", xi.grad, G["resblock_dx"], G["resblock_dxnorm"], tol)
    for k, p_ in rb.named_parameters():
        _close(f"resblock.grad[{k}]", p_.grad, G[f"resblock_g_{k}"], G[f"resblock_gn_{k}"], 2 * tol)
