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
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=head_type, evaluate_3d=False,
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
                        num_learnable_prompt_tokens=32, gpu=0, task='partseg', head_type=0, evaluate_3d=False, ulip2=False)
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
    assert set(payload) == {'epoch', 'state_dict', 'optimizer', 'best_acc', 'args', 'last_block'}
    assert list(payload['state_dict']) == ['learnable_tokens'] and 'attn.qkv.weight' in payload['last_block']
    f = tmp_path / "checkpoint_best.pt"
    torch.save(payload, f)
    m2, _ = make(3)
    load_prompt_checkpoint(m2, torch.load(f, weights_only=False))
    assert torch.equal(m2.prompt_learner.learnable_tokens, m.prompt_learner.learnable_tokens)
    assert torch.equal(m2.point_encoder.blocks.blocks[-1].mlp.fc2.weight, m.point_encoder.blocks.blocks[-1].mlp.fc2.weight)
