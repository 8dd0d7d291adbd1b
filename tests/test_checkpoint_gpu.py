"""GPU: SURVEY §8(f) N2 -- `checkpoint_best.pt` written from a GPU-TRAINED model and resumed / evaluated from.

The reference writes the file from the model it has just trained on the GPU (main_cls.py:118-137, main_partseg.py:127-143) and its
readers build a fresh model + `AdamW(model.parameters())` and load it (main_cls.py:58, save_recog_feats.py:29-35).  Here the
parameters are updated through raw pointers by ppt_adamw_multi on the model's text stream and the optimizer state lives in the
torch optimizer in torch's own layout: these tests show that both survive save -> load -> identical validate() logits and an
identical next step, bit for bit.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from ppt_amd import weights as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
LR = 3e-3


def _cls_model(head_type):
    from ppt_amd.models import ULIP_models as M
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=head_type, evaluate_3d=False,
                           synthetic_weights=True, ulip2=False)
    m = M.ULIP_PointBERT(args)
    m.load_state_dict(W.ulip_pointbert_state_dict(seed=0), strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(len(args.classnames), seed=0)
    m.cuda().set_precision(torch.bfloat16)
    return m


def _eval_logits(m, *inputs):
    m.eval()
    with torch.no_grad():
        out = m(*inputs).float().clone()
    m.train()
    return out


@pytest.mark.parametrize("head_type", [0, 3])
def test_cls_checkpoint_written_on_the_gpu_resumes_bit_identically(head_type, tmp_path):
    """3 steps of head_type 0 / 3 under Trainer (two streams, fused AdamW, hipGraphs) -> checkpoint_payload -> torch.save /
    torch.load -> fresh model, `torch.optim.AdamW(model.parameters())` as main_cls.py:58 loads the optimizer entry,
    load_prompt_checkpoint loads the rest -> eval logits and step 4 are the uninterrupted run's, bit for bit."""
    from ppt_amd.train import Trainer, checkpoint_payload, load_prompt_checkpoint, load_reference_optimizer_state
    pc_np, start = W.synth_clouds(4, 1024, seed=77)
    pc = torch.from_numpy(pc_np).cuda()
    label = torch.tensor([3, 17, 0, 39]).cuda()

    def prepare(m):
        m.train()
        m.point_encoder.fps_start = torch.from_numpy(start).cuda()
        m.point_encoder.drop_path_factors = torch.ones(12, 2, 4)
        return Trainer(m, lr=LR, label_smoothing=0.2, distributed=False)

    a = _cls_model(head_type)
    tr = prepare(a)
    for it in range(3):
        tr.step(torch.roll(pc, it, 0), label)
    tr.finish()
    payload = checkpoint_payload(a, tr.optimizer, epoch=0, best_acc=12.5, args={"model": "ULIP_PointBERT"}, head_type=head_type)
    f = tmp_path / "checkpoint_best.pt"
    torch.save(payload, f)
    # the reference's checkpoint holds the prompt (+ last block) only: the frozen tokenizer's BatchNorm running statistics, which
    # train() keeps updating, are not in it -- carried over by hand so that the continuation can be compared
    bn = {n: b.detach().clone() for n, b in a.named_buffers()}
    logits_a = _eval_logits(a, pc)
    loss_a, pred_a = tr.step(torch.roll(pc, 3, 0), label)
    tr.finish()
    torch.cuda.synchronize()
    after_a = {n: p.detach().clone() for n, p in a.named_parameters() if p.requires_grad}

    ckpt = torch.load(f, weights_only=False)
    # (the reference's keys + 'ppt_precision': the mode the run was in, which a resumed run continues in -- round 6)
    assert set(ckpt) == {'epoch', 'state_dict', 'optimizer', 'best_acc', 'args', 'last_block', 'ppt_precision'} and ckpt['epoch'] == 1
    assert ckpt['ppt_precision'] == a.precision_name
    assert (ckpt['last_block'] is not None) == (head_type > 0)
    b = _cls_model(head_type)
    if head_type:                                   # (Q4: the un-frozen block is random per construction; the checkpoint overrides it)
        with torch.no_grad():
            b.point_encoder.blocks.blocks[-1].mlp.fc2.weight.add_(1.0)
    # the reference's reader: an optimizer over ALL parameters takes the saved entry as it is
    ref_opt = torch.optim.AdamW(b.parameters(), lr=5.0, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
    ref_opt.load_state_dict(ckpt['optimizer'])
    assert ref_opt.param_groups[0]['lr'] == LR
    n_train = sum(p.requires_grad for p in b.parameters())
    assert len(ref_opt.state) == n_train
    assert all(st['exp_avg'].is_cuda and float(st['step']) == 3.0 for st in ref_opt.state.values())
    load_prompt_checkpoint(b, ckpt)
    with torch.no_grad():
        for n, buf in b.named_buffers():
            buf.copy_(bn[n])
    b.reset_caches()
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if p.requires_grad:
            assert q.is_cuda and q.requires_grad, n
    tr_b = prepare(b)                                # (also injects the FPS start indices: validate() draws them at random otherwise)
    assert torch.equal(_eval_logits(b, pc), logits_a)
    load_reference_optimizer_state(b, tr_b.optimizer, ckpt['optimizer'])
    tr_b.it = 3
    loss_b, pred_b = tr_b.step(torch.roll(pc, 3, 0), label)
    tr_b.finish()
    torch.cuda.synchronize()
    assert loss_b.item() == loss_a.item() and torch.equal(pred_a, pred_b)
    for n, q in b.named_parameters():
        if q.requires_grad:
            assert torch.equal(q, after_a[n]), n
    # ... and the literal main_cls.py:194-198 continuation (model(pc) -> criterion -> backward -> ref_opt.step()) from the same file
    c = _cls_model(head_type)
    load_prompt_checkpoint(c, ckpt)
    with torch.no_grad():
        for n, buf in c.named_buffers():
            buf.copy_(bn[n])
    c.reset_caches()
    c.train()
    c.point_encoder.fps_start = torch.from_numpy(start).cuda()
    c.point_encoder.drop_path_factors = torch.ones(12, 2, 4)
    opt_c = torch.optim.AdamW(c.parameters(), lr=5.0, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1, foreach=False)
    opt_c.load_state_dict(ckpt['optimizer'])
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.2)
    out = c(torch.roll(pc, 3, 0))
    loss_c = crit(out, label)
    opt_c.zero_grad()
    loss_c.backward()
    opt_c.step()
    torch.cuda.synchronize()
    assert abs(loss_c.item() - loss_a.item()) < 1e-5 * abs(loss_a.item())       # (ATen's criterion vs the fused head: fp32 rounding)
    for n, q in c.named_parameters():
        if q.requires_grad:
            d = (q - after_a[n]).abs().max().item()
            # (the literal loop's criterion / head kernels round differently from the fused head's; Adam's normalised update turns a
            # 1e-3 relative difference of a small gradient element into lr x 1e-3)
            assert d <= 1e-4 * max(1.0, after_a[n].abs().max().item()), (n, d)


def test_partseg_checkpoint_written_on_the_gpu_resumes_bit_identically(tmp_path):
    """main_partseg.py:127-143: 'state_dict_prompt' + 'state_dict_partseg' (the whole point encoder, buffers included) + the
    optimizer over model.parameters().  3 steps -> save -> load into a fresh model -> identical eval logits, identical step 4
    (41 trained tensors through ppt_adamw_multi; conv2 never gets a gradient and has no optimizer state)."""
    from ppt_amd.models import ULIP_models as M
    from ppt_amd.train import Trainer, checkpoint_payload, load_prompt_checkpoint, load_reference_optimizer_state
    g = np.load(os.path.join(G, "g_partseg.npz"), allow_pickle=False)
    pc_np, _ = W.synth_clouds(2, 2048, seed=55, duplicates=True)
    pc = torch.from_numpy(pc_np).cuda()
    labels = torch.from_numpy(g["labels"].astype(np.int64)).cuda()
    onehot = torch.from_numpy(g["onehot"]).cuda()

    def make():
        args = SimpleNamespace(classnames=M.dataset_classnames("shapenetpart"), template_init='', class_name_position='middle',
                               num_learnable_prompt_tokens=32, gpu=0, task='partseg', head_type=0, evaluate_3d=False, ulip2=False,
                               synthetic_weights=True)
        m = M.ULIP_PointBERT_partseg(args)
        m.load_state_dict(W.ulip_partseg_state_dict(seed=0), strict=False)
        m.prompt_learner.embedding = W.synth_prompt_embedding(50, seed=0)
        m.cuda().set_precision(torch.bfloat16)
        return m

    def prepare(m):
        m.train()
        pe = m.point_encoder
        pe.fps_start = tuple(torch.from_numpy(g[k]).cuda() for k in ("s0", "s1", "s2"))
        pe.drop_path_factors = torch.from_numpy(g["dp_masks"]).cuda()
        pe.dropout_mask = (torch.from_numpy(np.unpackbits(g["drop"]).reshape(2, 2048, 128).astype(np.float32)) * 2.0).cuda()
        pe._graph_injected = True
        tr = Trainer(m, lr=1e-3, label_smoothing=0.2, distributed=False)
        tr.extra_inputs = (onehot,)
        return tr

    a = make()
    tr = prepare(a)
    for _ in range(3):
        tr.step(pc, labels)
    tr.finish()
    payload = checkpoint_payload(a, tr.optimizer, epoch=0, best_acc=80.0, args={}, partseg=True, best_mean_class_iou=70.0,
                                 best_mean_inst_iou=75.0)
    f = tmp_path / "checkpoint_best.pt"
    torch.save(payload, f)
    logits_a = _eval_logits(a, pc, onehot)
    loss_a, _ = tr.step(pc, labels)
    tr.finish()
    torch.cuda.synchronize()
    after_a = {n: p.detach().clone() for n, p in a.named_parameters() if p.requires_grad}

    ckpt = torch.load(f, weights_only=False)
    b = make()
    ref_opt = torch.optim.AdamW(b.parameters(), lr=5.0)                      # main_partseg.py:62
    ref_opt.load_state_dict(ckpt['optimizer'])
    trained = [n for n, p in b.named_parameters() if p.requires_grad]
    assert len(ref_opt.state) == len(trained) - 2                           # conv2.{weight,bias} (unused in forward) never stepped
    load_prompt_checkpoint(b, ckpt)
    b.reset_caches()
    tr_b = prepare(b)
    assert torch.equal(_eval_logits(b, pc, onehot), logits_a)
    load_reference_optimizer_state(b, tr_b.optimizer, ckpt['optimizer'])
    tr_b.it = 3
    loss_b, _ = tr_b.step(pc, labels)
    tr_b.finish()
    torch.cuda.synchronize()
    assert loss_b.item() == loss_a.item()
    for n, q in b.named_parameters():
        if q.requires_grad:
            assert torch.equal(q, after_a[n]), n
