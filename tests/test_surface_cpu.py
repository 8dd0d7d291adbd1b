"""CPU: the drop-in module surface -- module paths, factory, attribute names,
state-dict keys and shapes, trainable sets per head_type, prompt construction -- without any compute."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from ppt_amd import weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make(head_type=0, position="middle", ds="modelnet40"):
    import models.ULIP_models as models          # the reference's import line (main_cls.py:25)
    args = SimpleNamespace(classnames=models.dataset_classnames(ds), template_init='', class_name_position=position,
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=head_type, evaluate_3d=False, synthetic_weights=True,
                           ulip2=False, model='ULIP_PointBERT')
    return getattr(models, args.model)(args), models       # main_cls.py:44


def test_state_dict_keys_and_shapes_match_reference_layout():
    m, _ = make()
    sd = m.state_dict()
    spec = dict(W.ulip_spec(768, True) + W.pointbert_spec())
    assert set(sd) == set(spec)
    for k, shape in spec.items():
        assert tuple(sd[k].shape) == tuple(shape), k


@pytest.mark.parametrize("head_type,count", [(0, 16384), (1, 607360), (2, 1199488), (3, 1789696)])
def test_trainable_sets(head_type, count):
    """SURVEY.md §2.5 / BASELINE.md: trainable parameter counts measured on the reference."""
    m, models = make(head_type)
    names = {n for n, p in m.named_parameters() if p.requires_grad}
    assert names == {"prompt_learner.learnable_tokens"} | set(models.unfreeze_list(head_type))
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == count
    assert m.point_encoder._tier() == head_type


@pytest.mark.parametrize("head_type,prio", [(0, 1), (1, 0), (3, 0)])
def test_chain_priority_follows_the_trainable_set(head_type, prio, monkeypatch):
    """The prompt chain's kernels raise their wave priority only when nothing but the PromptLearner trains (the chain is then the
    step's critical path; with a training point side the tower is).  PPT_CHAIN_PRIO overrides."""
    monkeypatch.delenv("PPT_CHAIN_PRIO", raising=False)
    m, _ = make(head_type)
    assert m.chain_priority() == prio
    monkeypatch.setenv("PPT_CHAIN_PRIO", str(1 - prio))
    m.reset_caches()
    assert m.chain_priority() == 1 - prio


def test_surface_attributes():
    m, models = make(3)
    assert models.get_metric_names() == ['loss', 'acc']
    assert list(m.prompt_learner.state_dict()) == ["learnable_tokens"]                       # main_cls.py:124
    assert "mlp.fc2.weight" in m.point_encoder.blocks.blocks[-1].state_dict()                # main_cls.py:127
    assert m.logit_scale.shape == () and m.token_embedding.weight.shape == (49408, 512)
    assert m.tokenized_prompts.shape == (40, 77)
    from models.pointbert.point_encoder import PointTransformer                              # noqa: F401
    from models.pointbert.dvae import Group, Encoder, knn_point, square_distance             # noqa: F401
    from models.pointbert.misc import fps, farthest_point_sample, index_points               # noqa: F401
    # a reference checkpoint's last block loads (save_recog_feats.py:33-35)
    blk = {'point_encoder.blocks.blocks.11.' + k: v for k, v in m.point_encoder.blocks.blocks[-1].state_dict().items()}
    missing, unexpected = m.load_state_dict(blk, strict=False)
    assert not unexpected


def test_tokenised_prompts_match_golden_eot():
    m, _ = make()
    g = np.load(os.path.join(ROOT, "tests", "golden", "g_step_h0.npz"))
    assert np.array_equal(m.tokenized_prompts.argmax(-1).numpy(), g["eot"].astype(np.int64))


@pytest.mark.parametrize("position", ["front", "middle", "end"])
def test_prompt_splice_matches_oracle(position):
    from oracle import oracle as O
    m, _ = make(position=position, ds="shapenetpart")
    pl = m.prompt_learner
    ref = O.splice_prompts(pl.embedding, pl.learnable_tokens, pl.name_lengths, position)
    out = pl()
    assert torch.equal(out, ref)
    out.sum().backward()                          # every learnable token appears once per class
    assert torch.allclose(pl.learnable_tokens.grad, torch.full_like(pl.learnable_tokens, 50.0))


def test_bad_position_raises_valueerror():
    m, _ = make()
    m.prompt_learner.class_name_position = "sideways"
    m.prompt_learner._index = None
    with pytest.raises(ValueError):
        m.prompt_learner()


def test_product_does_not_import_the_oracle():
    """the product path must never route through the oracle (or any CPU fallback)."""
    import subprocess
    import sys
    code = ("import sys; import ppt_amd, ppt_amd.ops, ppt_amd.engine, ppt_amd.train, ppt_amd.models.ULIP_models, "
            "ppt_amd.models.pointbert.point_encoder; "
            "bad=[m for m in sys.modules if m.split('.')[0]=='oracle']; assert not bad, bad")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ppt_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_cpu_tensor_is_rejected_loudly():
    from ppt_amd import ops
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.fps(torch.zeros(1, 8, 3), 4, torch.zeros(1, dtype=torch.long))


def test_other_factories_state_dicts():
    """ULIP_PN_MSG (C4) and ULIP_PointBERT_partseg (C5): reference state-dict layouts and trainable sets."""
    import models.ULIP_models as models
    a = SimpleNamespace(classnames=models.dataset_classnames("shapenetpart"), template_init='', class_name_position='middle',
                        num_learnable_prompt_tokens=32, gpu=0, task='partseg', head_type=0, evaluate_3d=False, ulip2=False, synthetic_weights=True)
    m = models.ULIP_PointBERT_partseg(a)
    spec = dict(W.ulip_spec(128, True) + W.pointbert_spec() + W.partseg_decoder_spec())
    assert set(m.state_dict()) == set(spec)
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 5242152
    a.task, a.classnames = 'cls', models.dataset_classnames("modelnet40")
    m = models.ULIP_PN_MSG(a)
    spec = dict(W.ulip_spec(256, True) + W.pointnet2_msg_spec())
    sd = m.state_dict()
    assert set(sd) == set(spec) and all(tuple(sd[k].shape) == tuple(v) for k, v in spec.items())
    assert [n for n, p in m.named_parameters() if p.requires_grad] == ["prompt_learner.learnable_tokens"]
    m = models.ULIP_PN_SSG(a)
    spec = dict(W.ulip_spec(256, True) + W.pointnet2_ssg_spec())
    sd = m.state_dict()
    assert set(sd) == set(spec) and all(tuple(sd[k].shape) == tuple(v) for k, v in spec.items())
    assert [n for n, p in m.named_parameters() if p.requires_grad] == ["prompt_learner.learnable_tokens"]
    m = models.ULIP_PN_MLP(a)
    spec = dict(W.ulip_spec(256, True) + W.pointmlp_spec())
    sd = m.state_dict()
    assert set(sd) == set(spec) and all(tuple(sd[k].shape) == tuple(v) for k, v in spec.items())
    assert [n for n, p in m.named_parameters() if p.requires_grad] == ["prompt_learner.learnable_tokens"]
    from models.pointmlp.pointMLP import pointMLP, pointMLPElite
    assert len(pointMLP().state_dict()) == len(W.pointmlp_spec())
    assert len(pointMLPElite().state_dict()) > 0
    from models.pointnet2.pointnet2 import Pointnet2_Msg, Pointnet2_Ssg                    # noqa: F401
    from models.pointbert.pointnet2_utils import PointNetFeaturePropagation, DGCNN_Propagation   # noqa: F401


def test_checkpoint_payload_roundtrip(tmp_path):
    """reference checkpoint_best.pt layout (main_cls.py:132-137) written and read back."""
    from ppt_amd.train import checkpoint_payload, load_prompt_checkpoint
    m, _ = make(3)
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    payload = checkpoint_payload(m, opt, epoch=4, best_acc=91.5, args={"model": "ULIP_PointBERT"}, head_type=3)
    # the reference's keys, plus ONE of ours that its readers never look at: the precision mode the run was in (round 6)
    assert set(payload) == {'epoch', 'state_dict', 'optimizer', 'best_acc', 'args', 'last_block', 'ppt_precision'}
    assert payload['ppt_precision'] == m.precision_name
    assert list(payload['state_dict']) == ['learnable_tokens'] and 'attn.qkv.weight' in payload['last_block']
    f = tmp_path / "checkpoint_best.pt"
    torch.save(payload, f)
    m2, _ = make(3)
    load_prompt_checkpoint(m2, torch.load(f, weights_only=False))
    assert torch.equal(m2.prompt_learner.learnable_tokens, m.prompt_learner.learnable_tokens)
    assert torch.equal(m2.point_encoder.blocks.blocks[-1].mlp.fc2.weight, m.point_encoder.blocks.blocks[-1].mlp.fc2.weight)


def test_partseg_checkpoint_payload_uses_the_reference_keys(tmp_path):
    """main_partseg.py:134-143 writes 'state_dict_prompt' / 'state_dict_partseg' / 'best_test_acc' / 'best_mean_class_iou' /
    'best_mean_inst_iou'; notebook/show_balls.py:219 reads saved_data['state_dict_prompt']."""
    from types import SimpleNamespace
    from ppt_amd.models import ULIP_models as models
    from ppt_amd.train import checkpoint_payload, load_prompt_checkpoint
    a = SimpleNamespace(classnames=models.dataset_classnames("shapenetpart"), template_init='', class_name_position='middle',
                        num_learnable_prompt_tokens=32, gpu=0, task='partseg', head_type=0, evaluate_3d=False, ulip2=False,
                        synthetic_weights=True)
    m = models.ULIP_PointBERT_partseg(a)
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    payload = checkpoint_payload(m, opt, epoch=2, best_acc=93.0, args={}, partseg=True, best_mean_class_iou=81.0,
                                 best_mean_inst_iou=84.5)
    assert set(payload) == {'epoch', 'state_dict_prompt', 'state_dict_partseg', 'optimizer', 'best_test_acc',
                            'best_mean_class_iou', 'best_mean_inst_iou', 'args', 'ppt_precision'}
    assert list(payload['state_dict_prompt']) == ['learnable_tokens'] and payload['best_test_acc'] == 93.0
    f = tmp_path / "checkpoint_best.pt"
    torch.save(payload, f)
    m2 = models.ULIP_PointBERT_partseg(a)
    load_prompt_checkpoint(m2, torch.load(f, weights_only=False))
    assert torch.equal(m2.prompt_learner.learnable_tokens, m.prompt_learner.learnable_tokens)
    assert torch.equal(m2.point_encoder.conv1.weight, m.point_encoder.conv1.weight)


# ---- N2: the read side of the checkpoint formats (ULIP_models.py:472-507, point_encoder.py:206-232) ----------------
def _fake_pretrained(tmp_path, head_type=2):
    """Synthetic pointbert.pt / slip_base_100ep.pt in the reference's on-disk layout: {'state_dict': {'module.<key>': t}}.
    The point checkpoint holds point_encoder.* and pc_projection; everything else (text tower, logit_scale ...) is only
    in the SLIP checkpoint, which ALSO holds a (different) pc_projection: the point checkpoint must win."""
    m, models = make(head_type)
    g = torch.Generator().manual_seed(7)
    point, slip = {}, {}
    for name, p in m.named_parameters():
        if name == 'prompt_learner.learnable_tokens':
            continue
        t = torch.randn(p.shape, generator=g)
        if name.startswith('point_encoder.') or name == 'pc_projection':
            point['module.' + name] = torch.nn.Parameter(t) if name.endswith('norm.weight') else t   # both value kinds occur
        else:
            slip['module.' + name] = t
    slip['module.pc_projection'] = torch.full_like(m.pc_projection, 9.0)
    slip['module.visual.cls_token'] = torch.zeros(1, 1, 768)              # image-tower entries are simply never asked for
    os.makedirs(tmp_path / "data" / "pretrained_models")
    os.makedirs(tmp_path / "data" / "initialize_models")
    torch.save({'state_dict': point, 'epoch': 3}, tmp_path / "data" / "pretrained_models" / "pointbert.pt")
    torch.save({'state_dict': slip}, tmp_path / "data" / "initialize_models" / "slip_base_100ep.pt")
    return point, slip


def test_ulip_checkpoints_load_and_freeze_as_the_reference(tmp_path, monkeypatch):
    import models.ULIP_models as models
    point, slip = _fake_pretrained(tmp_path, head_type=2)
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("PPT_SYNTHETIC_WEIGHTS", raising=False)
    args = SimpleNamespace(classnames=models.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=2, evaluate_3d=False, ulip2=False)
    torch.manual_seed(0)
    m = models.ULIP_PointBERT(args)
    skip = set(models.unfreeze_list(2))
    for name, p in m.named_parameters():
        if name == 'prompt_learner.learnable_tokens':
            assert p.requires_grad
        elif name in skip:                                      # un-frozen tier keys keep their random init
            assert p.requires_grad
            assert not torch.equal(p.data, point['module.' + name].data), name
        else:
            assert not p.requires_grad, name
            src = point.get('module.' + name, slip.get('module.' + name))
            assert torch.equal(p.data, src.data), name
    assert torch.equal(m.pc_projection.data, point['module.pc_projection'])       # point checkpoint first, SLIP second


def test_missing_checkpoint_raises_unless_opted_in(tmp_path, monkeypatch):
    """ULIP_models.py:472-485: torch.load raises on a missing file; a silently random frozen backbone must not happen."""
    import models.ULIP_models as models
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("PPT_SYNTHETIC_WEIGHTS", raising=False)
    base = dict(classnames=models.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=0, evaluate_3d=False, ulip2=False)
    with pytest.raises(FileNotFoundError):
        models.ULIP_PointBERT(SimpleNamespace(**base))
    with pytest.raises(FileNotFoundError):
        models.ULIP_PN_MSG(SimpleNamespace(**base))
    with pytest.raises(FileNotFoundError):
        models.ULIP_PointBERT_partseg(SimpleNamespace(**{**base, "task": "partseg"}))
    # one file present, the other absent: still an error (and nothing half-loaded goes unnoticed)
    _fake_pretrained(tmp_path / "w", head_type=0)
    os.makedirs(tmp_path / "data" / "pretrained_models")
    os.replace(tmp_path / "w" / "data" / "pretrained_models" / "pointbert.pt", tmp_path / "data" / "pretrained_models" / "pointbert.pt")
    with pytest.raises(FileNotFoundError):
        models.ULIP_PointBERT(SimpleNamespace(**base))
    # explicit opt-in: the file that IS present is loaded, the missing one is tolerated
    m = models.ULIP_PointBERT(SimpleNamespace(**base, synthetic_weights=True))
    pt = torch.load(tmp_path / "data" / "pretrained_models" / "pointbert.pt", weights_only=False)['state_dict']
    assert torch.equal(m.point_encoder.reduce_dim.weight.data, pt['module.point_encoder.reduce_dim.weight'])
    assert not m.point_encoder.reduce_dim.weight.requires_grad


def test_load_model_from_ckpt_key_translation(tmp_path):
    """point_encoder.py:206-232: 'base_model' table, module. prefix, transformer_q.* / base_model.* renaming, cls_head and
    foreign entries dropped, non-strict load."""
    from models.pointbert.point_encoder import PointTransformer
    import models.ULIP_models as models
    pe = PointTransformer(models.POINTBERT_CONFIG)
    g = torch.Generator().manual_seed(3)
    want = {k: torch.randn(v.shape, generator=g) if v.dtype.is_floating_point else v.clone() for k, v in pe.state_dict().items()}
    table = {}
    for i, (k, v) in enumerate(want.items()):
        if k.startswith("blocks.blocks.11.mlp"):
            continue                                           # left out: reported as missing, keeps its value
        table[("module.transformer_q." if i % 2 else "module.base_model.") + k] = v
    table["module.transformer_q.cls_head.0.weight"] = torch.zeros(3)      # must be ignored, not reported
    table["module.dvae.encoder.conv.weight"] = torch.zeros(3)             # neither prefix: dropped
    torch.save({"base_model": table}, tmp_path / "Point-BERT.pth")
    before = pe.blocks.blocks[11].mlp.fc1.weight.detach().clone()
    res = pe.load_model_from_ckpt(str(tmp_path / "Point-BERT.pth"))
    assert not res.unexpected_keys
    assert sorted(res.missing_keys) == sorted(k for k in want if k.startswith("blocks.blocks.11.mlp"))
    assert torch.equal(pe.blocks.blocks[11].mlp.fc1.weight, before)
    assert torch.equal(pe.encoder.first_conv[0].weight, want["encoder.first_conv.0.weight"])
    assert torch.equal(pe.blocks.blocks[0].attn.qkv.weight, want["blocks.blocks.0.attn.qkv.weight"])


def test_optimizer_state_in_the_checkpoint_uses_the_reference_indexing():
    """main_cls.py:58 builds AdamW over model.parameters(): the saved state must be loadable by such an optimizer."""
    from ppt_amd.train import reference_optimizer_state
    m, _ = make(1)
    trainable = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(trainable, lr=1e-3)
    for p in trainable:
        p.grad = torch.ones_like(p)
    opt.step()
    sd = reference_optimizer_state(m, opt)
    ref_opt = torch.optim.AdamW(m.parameters(), lr=5.0)
    ref_opt.load_state_dict(sd)                                         # group sizes match, no error
    assert ref_opt.param_groups[0]['lr'] == 1e-3
    allp = list(m.parameters())
    for p in trainable:
        i = next(j for j, q in enumerate(allp) if q is p)
        assert torch.equal(ref_opt.state[allp[i]]['exp_avg'], opt.state[p]['exp_avg'])
    assert len(ref_opt.state) == len(trainable)


def test_top_level_load_state_dict_resets_the_point_encoders_caches():
    """nn.Module.load_state_dict loads children through _load_from_state_dict -- the point encoder's operand
    copies and captured graphs must be dropped by the TOP-LEVEL load too."""
    m, _ = make(1)
    pe = m.point_encoder
    pe._wc, pe._sd = object(), (None, None, None)
    pe._graphs.entries["stale"] = object()
    m._graphs.entries["stale"] = object()
    m.load_state_dict({'point_encoder.blocks.blocks.11.mlp.fc2.bias': torch.zeros(384)}, strict=False)
    assert pe._wc is None and pe._sd is None and not pe._graphs.entries and not m._graphs.entries


def test_gradient_scale_rule():
    """ppt_amd/gradscale.py: a 16-bit backward stage's scale is fixed at forward time -- "auto" = the number of rows the caller's
    criterion averages over (announced by the model's forward through `rows`), rounded down to a power of two; fp32 stages are
    never scaled; a number fixes it; "off" / "0" / "1" turn it off; outside any `rows` context a node falls back to its own rows."""
    import torch
    from ppt_amd import gradscale as gs
    old = gs.POLICY
    try:
        gs.POLICY = "auto"
        with gs.rows(32):
            assert gs.current(torch.float16) == 32.0 and gs.current(torch.bfloat16) == 32.0
            assert gs.current(torch.float32) == 1.0                   # parity mode: never scaled
            with gs.rows(48):
                assert gs.current(torch.float16) == 32.0
            assert gs.current(torch.float16) == 32.0
        with gs.rows(16 * 2048):
            assert gs.current(torch.float16) == 32768.0
        with gs.rows(1):
            assert gs.current(torch.float16) == 1.0
        assert gs.current(torch.float16) == 1.0                       # no context, no default
        assert gs.current(torch.float16, default_rows=4096) == 4096.0
        for off in ("0", "1", "off", "none", ""):
            gs.POLICY = off
            with gs.rows(32):
                assert gs.current(torch.float16) == 1.0
        gs.POLICY = "4096"
        with gs.rows(32):
            assert gs.current(torch.float16) == 4096.0 and gs.current(torch.float32) == 1.0
    finally:
        gs.POLICY = old


def test_scaled_backward_wrapper_is_exact():
    """gradscale.scaled_backward: incoming gradients x S, outgoing x 1 / S -- the caller sees what the raw backward would have
    produced for a linear backward (CPU: the wrapper itself is plain torch)."""
    import torch
    from ppt_amd import gradscale as gs

    class Ctx:
        grad_scale = 64.0

    seen = {}

    def raw(ctx, d):
        seen["in"] = d.clone()
        return d * 3.0, None, d.sum()

    wrapped = gs.scaled_backward(raw)
    d = torch.randn(5, 7)
    a, b, c = wrapped(Ctx(), d)
    assert torch.equal(seen["in"], d * 64.0) and b is None
    assert torch.equal(a, d * 3.0) and torch.allclose(c, d.sum(), rtol=1e-6)
    assert wrapped.raw is raw
    Ctx.grad_scale = 1.0
    a2, _, _ = wrapped(Ctx(), d)
    assert torch.equal(a2, d * 3.0) and torch.equal(seen["in"], d)


def test_set_precision_names():
    """the performance mode is a MIXED 16-bit mode (IEEE half in the normalised stages, bf16 in PointNet++ /
    PointMLP) and is named so -- set_precision("mixed16") / "fp32"; torch.bfloat16 stays as a deprecated alias that warns."""
    import warnings
    from types import SimpleNamespace
    import pytest
    import torch
    from ppt_amd.models import ULIP_models as M
    args = SimpleNamespace(classnames=["chair", "table"], template_init='', class_name_position='middle', num_learnable_prompt_tokens=4,
                           gpu=0, task='cls', head_type=0, evaluate_3d=False, ulip2=False, synthetic_weights=True)
    m = M.ULIP_PointBERT(args)
    assert m.precision_name == "mixed16"
    assert m.set_precision("fp32") is m and m.precision == torch.float32 and m.precision_name == "fp32"
    assert m.set_precision("mixed16").precision_name == "mixed16" and m.precision == torch.bfloat16
    assert m.set_precision(torch.float32).precision_name == "fp32"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.set_precision(torch.bfloat16)
    assert m.precision_name == "mixed16" and any(issubclass(x.category, DeprecationWarning) for x in w)
    with pytest.raises(ValueError):
        m.set_precision("bf16")
    with pytest.raises(ValueError):
        m.set_precision(torch.int8)


def test_split16_range_fit_lowers_the_weight_prescale_for_huge_weights():
    """The split16 pre-scale of the weight operand is fitted per MODEL and per WEIGHT SET (ULIP_WITH_IMAGE._fit_split16_range): 4 x
    max|w| x 2^b must fit IEEE half, b is lowered with a warning otherwise; the result is the model's own `split_pow2` (the
    process-wide default in ops stays what it was -- ADVICE r5: one model's fit must not change another's), every _cache() call hands
    it to ops with the flag, and load_state_dict / reset_caches re-arm the fit."""
    import warnings
    from types import SimpleNamespace
    import torch
    from ppt_amd import ops
    from ppt_amd.models import ULIP_models as M
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=0, evaluate_3d=False, ulip2=False,
                           synthetic_weights=True)
    m = M.ULIP_PointBERT(args)
    other = M.ULIP_PointBERT(args)
    old = ops.SPLIT16_POW2
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            m._fit_split16_range()                       # freshly initialised weights: nothing to do, no warning
        assert m.split_pow2 == old and not m._split_fit_pending
        with torch.no_grad():
            m.text_projection[0, 0] = 3000.0
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            m._fit_split16_range()
        assert any("split16" in str(x.message) for x in w)
        b = m.split_pow2[1]
        assert b < old[1] and 4.0 * 3000.0 * 2.0 ** b < 32768.0 <= 4.0 * 3000.0 * 2.0 ** (b + 1)
        assert ops.SPLIT16_POW2 == old and other.split_pow2 == old          # nobody else's pre-scale moved
        # the flag and the pair travel together: whoever fetched its cache last decides what an un-annotated ops.gemm uses
        ops.set_split16(True, m.split_pow2)
        assert ops.SPLIT16_POW2 == m.split_pow2
        ops.set_split16(True, other.split_pow2)
        assert ops.SPLIT16_POW2 == old
        ops.set_split16(False)
        # new weights re-arm the fit (a checkpoint loaded AFTER set_precision("split16") used never to be looked at)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        sd["text_projection"][0, 0] = 0.01
        m.load_state_dict(sd)
        assert m._split_fit_pending
        m._fit_split16_range()
        assert m.split_pow2 == old
    finally:
        ops.SPLIT16_POW2 = old
        ops.set_split16(False)


def test_ahead_stage_orders_itself_behind_every_input():
    """the copy-complete event rides on the batch OBJECT (DevicePrefetcher), but an ahead stage reads what the
    encoder derives from it -- `pc.contiguous().float()` is the same object for a contiguous fp32 batch and a NEW tensor, produced on
    the caller's stream and covered by no event, for anything else.  graphs.wait_inputs: event -> wait for it; no event and no
    promise from the caller -> fall back to the caller's stream (in-order, never a race); promise (Trainer.inputs_ready) -> nothing."""
    from ppt_amd import graphs

    class FakeStream:
        def __init__(self):
            self.calls = []

        def wait_event(self, e):
            self.calls.append(("event", e))

        def wait_stream(self, s):
            self.calls.append(("stream", s))

    batch = torch.zeros(2, 8, 3)
    batch._ppt_ready = "copy-done"
    same = batch.contiguous().float()
    assert same is batch and graphs.ready_event(same) == "copy-done"
    derived = batch.double().float()                                  # what a non-fp32 / sliced batch turns into
    assert graphs.ready_event(derived) is None
    s = FakeStream()
    graphs.wait_inputs(s, [same], main="caller")
    assert s.calls == [("event", "copy-done")]
    s = FakeStream()
    graphs.wait_inputs(s, [derived], main="caller")
    assert s.calls == [("stream", "caller")], "a tensor without an event must send the stage behind the caller's stream"
    s = FakeStream()
    graphs.wait_inputs(s, [same, derived, derived], main="caller")
    assert s.calls == [("event", "copy-done"), ("stream", "caller")]
    s = FakeStream()
    graphs.wait_inputs(s, [derived], main="caller", vouched=True)     # Trainer.inputs_ready: the caller's promise covers it
    assert s.calls == []
