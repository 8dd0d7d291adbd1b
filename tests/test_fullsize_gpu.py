"""GPU: BASELINE.json's configurations at their FULL per-GPU sizes.  The oracle's dense math is too
slow for these batches, so the checks are the size-independent ones the domain offers:
  * index work (FPS, kNN, ball query) against the C oracle -- that part of the oracle is fast at any size -- bit for bit;
  * bf16 performance mode against the fp32 parity mode of the same model (the fp32 mode is pinned to the reference by the
    golden fixtures at small sizes: tests/test_model_gpu.py);
  * run-to-run bit reproducibility (no atomics, fixed summation orders);
  * one real training step per configuration: finite loss, only the expected parameters receive gradients."""
import contextlib
import io
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import oracle as O
from ppt_amd import weights as W

pytestmark = pytest.mark.gpu


def _model(factory, ds, head_type=0, task='cls', sd_fn=None, precision=torch.bfloat16):
    from ppt_amd.models import ULIP_models as M
    args = SimpleNamespace(classnames=M.dataset_classnames(ds), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task=task, head_type=head_type, evaluate_3d=False, ulip2=False,
                           synthetic_weights=True)
    with contextlib.redirect_stdout(io.StringIO()):
        m = getattr(M, factory)(args)
    m.load_state_dict(sd_fn(seed=0), strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(len(args.classnames), seed=0)
    m.cuda().set_precision(precision)
    return m


def test_c2_full_batch_32x1024():
    """C2: B = 32, N = 1024.  Group.forward (FPS + kNN) against the oracle over the whole batch; tower features bf16 vs fp32;
    the training step twice from the same state: identical bits."""
    from ppt_amd.train import Trainer
    B, N = 32, 1024
    pc_np, start = W.synth_clouds(B, N, seed=1234)
    pc = torch.from_numpy(pc_np).cuda()
    labels = torch.from_numpy(np.random.default_rng(0).integers(0, 40, size=(B,))).cuda()
    m = _model("ULIP_PointBERT", "modelnet40", sd_fn=W.ulip_pointbert_state_dict)
    # ---- index work, whole batch, through the module surface
    nbhd, center = m.point_encoder.group_divider(pc, torch.from_numpy(start).cuda())
    cidx = O.fps(pc_np, 512, start)
    _, nb_ref, ce_ref = O.group(pc_np, cidx, 32)
    assert np.array_equal(center.cpu().numpy(), ce_ref)
    # both sides order a centre's neighbours by (distance, index): identical arrays (the oracle's tie rule is the kernel's)
    assert np.array_equal(nbhd.cpu().numpy(), nb_ref)
    # ---- features: bf16 against the fp32 parity mode, eval BatchNorm, no DropPath
    m.eval()
    m.point_encoder.fps_start = torch.from_numpy(start).cuda()
    feats = {}
    for prec in (torch.float32, torch.bfloat16):
        m.set_precision(prec)
        with torch.no_grad():
            feats[prec] = m.point_encoder(pc).float().cpu()
    rel = ((feats[torch.bfloat16] - feats[torch.float32]).norm() / feats[torch.float32].norm()).item()
    assert rel < 2e-2, rel
    # ---- the eval tokenizer's ONE-kernel second half (csrc/mpn34.hip) against the two-kernel path it replaces
    # (conv3 written to HBM, BatchNorm + ReLU applied while conv4 reads it back), whole batch, and no further from fp32
    from ppt_amd import engine
    saved = engine.FUSED_CONV34
    try:
        engine.FUSED_CONV34 = False
        m.point_encoder._graphs.clear()
        with torch.no_grad():
            unfused = m.point_encoder(pc).float().cpu()
    finally:
        engine.FUSED_CONV34 = saved
        m.point_encoder._graphs.clear()
    rel_f = ((feats[torch.bfloat16] - unfused).norm() / unfused.norm()).item()
    rel_u = ((unfused - feats[torch.float32]).norm() / feats[torch.float32].norm()).item()
    print(f"PARITY C2 eval features, 16-bit vs fp32: fused conv3+conv4 {rel:.4g}, unfused {rel_u:.4g}; fused vs unfused {rel_f:.4g}")
    assert rel_f < 5e-3 and rel < rel_u * 1.25 + 1e-4, (rel, rel_u, rel_f)
    # ---- the training step is bit-reproducible
    runs = []
    for _ in range(2):
        mm = _model("ULIP_PointBERT", "modelnet40", sd_fn=W.ulip_pointbert_state_dict)
        mm.train()
        mm.point_encoder.fps_start = torch.from_numpy(start).cuda()
        mm.point_encoder.drop_path_factors = torch.ones(12, 2, B)
        tr = Trainer(mm, distributed=False)
        for _ in range(4):                                          # eager calls, then hipGraph replays
            loss, pred = tr.step(pc, labels)
        tr.finish()
        torch.cuda.synchronize()
        runs.append((loss.item(), pred.float().cpu().clone(), mm.prompt_learner.learnable_tokens.detach().cpu().clone()))
        assert [n for n, p in mm.named_parameters() if p.grad is not None] == ["prompt_learner.learnable_tokens"]
    assert np.isfinite(runs[0][0]) and runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])


def test_c3_full_batch_64x2048_head3():
    """C3: B = 64, N = 2048 (clouds with duplicate points, as ScanObjectNN resampling gives), head_type 3: FPS indices of the
    whole batch against the oracle, a finite reproducible step, gradients exactly on the tier's 12 tensors."""
    from ppt_amd import ops
    from ppt_amd.train import Trainer
    B, N = 64, 2048
    pc_np, start = W.synth_clouds(B, N, seed=77, duplicates=True)
    pc = torch.from_numpy(pc_np).cuda()
    idx, _ = ops.fps(pc, 512, torch.from_numpy(start).cuda())
    assert np.array_equal(idx.cpu().numpy(), O.fps(pc_np, 512, start))
    labels = torch.from_numpy(np.random.default_rng(1).integers(0, 15, size=(B,))).cuda()
    losses = []
    for _ in range(2):
        m = _model("ULIP_PointBERT", "scanobjectnn", head_type=3, sd_fn=W.ulip_pointbert_state_dict)
        m.train()
        m.point_encoder.fps_start = torch.from_numpy(start).cuda()
        m.point_encoder.drop_path_factors = torch.ones(12, 2, B)
        tr = Trainer(m, distributed=False)
        for _ in range(3):
            loss, _ = tr.step(pc, labels)
        tr.finish()
        torch.cuda.synchronize()
        losses.append(loss.item())
        assert sum(1 for p in m.parameters() if p.grad is not None) == 12
    assert np.isfinite(losses[0]) and losses[0] == losses[1]


def test_c4_full_batch_32x8192_pointnet2_msg():
    """C4: B = 32, N = 8192 through Pointnet2_Msg (the multi-scale FPS / ball-query stress): both levels' FPS picks and all six
    ball queries of the grouping stage against the oracle, bit for bit; bf16 features against fp32; two runs identical."""
    from ppt_amd import engine
    from ppt_amd.models.pointnet2.pointnet2 import Pointnet2_Msg
    B, N = 32, 8192
    pc_np, s1 = W.synth_clouds(B, N, seed=99)
    _, s2 = W.synth_clouds(B, 512, seed=98)
    pc = torch.from_numpy(pc_np).cuda()
    starts = (torch.from_numpy(s1).cuda(), torch.from_numpy(s2).cuda())
    g = engine.pointnet2_group(pc, starts, engine.PN2_MSG_LEVELS)
    l1_idx = O.fps(pc_np, 512, s1)
    l1_xyz = np.take_along_axis(pc_np, l1_idx[:, :, None], axis=1)
    assert np.array_equal(g[0].cpu().numpy(), l1_xyz)
    for i, (r, K) in enumerate(engine.PN2_MSG_LEVELS[0][1]):                     # level 1: grouped, centred coordinates
        bi = O.ball_query(pc_np, l1_xyz, r, K)
        want = np.take_along_axis(pc_np[:, None], bi[..., None], axis=2) - l1_xyz[:, :, None]
        assert np.array_equal(g[1 + i].cpu().numpy(), want.astype(np.float32)), (r, K)
    l2_idx = O.fps(l1_xyz, 128, s2)
    l2_xyz = np.take_along_axis(l1_xyz, l2_idx[:, :, None], axis=1)
    assert np.array_equal(g[4].cpu().numpy(), l2_xyz)
    for i, (r, K) in enumerate(engine.PN2_MSG_LEVELS[1][1]):                     # level 2: neighbour indices
        assert np.array_equal(g[5 + i].cpu().numpy(), O.ball_query(l1_xyz, l2_xyz, r, K)), (r, K)
    m = Pointnet2_Msg()
    m.load_state_dict(W.synth_state_dict(W.pointnet2_msg_spec(prefix=""), seed=0))
    m.cuda().eval()
    m.fps_start = starts
    outs = {}
    for prec in (torch.float32, torch.bfloat16, torch.bfloat16):
        m.precision, m._wc = prec, None
        with torch.no_grad():
            outs.setdefault(prec, []).append(m(pc).float().cpu())
    a, b = outs[torch.float32][0], outs[torch.bfloat16][0]
    assert torch.isfinite(a).all() and ((a - b).norm() / a.norm()).item() < 3e-2
    assert torch.equal(outs[torch.bfloat16][0], outs[torch.bfloat16][1])


def test_c5_full_batch_16x2048_partseg_step():
    """C5: B = 16, N = 2048 part segmentation: one training step (forward, per-point CE, backward through the whole decoder),
    finite, every decoder parameter that takes part in the forward gets a gradient, two runs give identical bits."""
    from ppt_amd.models import ULIP_models as M
    B, N = 16, 2048
    pc_np, s0 = W.synth_clouds(B, N, seed=5, duplicates=True)
    _, s1 = W.synth_clouds(B, N, seed=6)
    _, s2 = W.synth_clouds(B, N, seed=7)
    pc = torch.from_numpy(pc_np).cuda()
    labels = torch.from_numpy(np.random.default_rng(2).integers(0, 50, size=(B, N))).cuda()
    onehot = torch.nn.functional.one_hot(torch.arange(B) % 16, 16).float().cuda()
    res = []
    for _ in range(2):
        m = _model("ULIP_PointBERT_partseg", "shapenetpart", task='partseg', sd_fn=W.ulip_partseg_state_dict)
        m.overlap_text_tower = False
        m.train()
        pe = m.point_encoder
        pe.fps_start = tuple(torch.from_numpy(s).cuda() for s in (s0, s1, s2))
        pe.drop_path_factors = torch.ones(12, 2, B)
        pe.dropout_mask = torch.ones(B, N, 128)
        pred = m(pc, onehot)
        loss = torch.nn.CrossEntropyLoss(label_smoothing=0.2)(pred.reshape(-1, 50), labels.reshape(-1))
        loss.backward()
        torch.cuda.synchronize()
        missing = [n for n, p in m.named_parameters() if p.requires_grad and p.grad is None]
        assert missing == ["point_encoder.conv2.weight", "point_encoder.conv2.bias"], missing      # unused by forward, as in the reference
        res.append((loss.item(), m.point_encoder.propagation_0.mlp_convs[0].weight.grad.cpu().clone(),
                    m.point_encoder.dgcnn_pro_1.layer1[0].weight.grad.cpu().clone()))
    assert np.isfinite(res[0][0]) and res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _zero_by_construction(n):
    """A conv bias in front of a BatchNorm has a mathematically zero gradient (the mean subtraction removes it): what any
    implementation -- the reference included -- leaves there is rounding noise, not a quantity to compare."""
    return n.endswith(".bias") and (".mlp_convs." in n or n.endswith("conv1.bias"))


def test_c5_plain_partseg_loop_16bit_gradients_agree_with_fp32():
    """The LITERAL loop of main_partseg.py:204-215 -- `pred = model(pc, onehot); loss = criterion(...); loss.backward();
    optimizer.step()` with a stock torch.optim.AdamW, no train.Trainer -- at C5's full size (B = 16 x 2048 points: the criterion
    averages over 32 768 rows, d loss / d logits ~ 3e-5 per row) in the default 16-bit mode, against the fp32 parity mode of the
    same step (pinned to the reference by the goldens at B = 2).  The gradient scale lives in the autograd nodes
    (ppt_amd/gradscale.py), so the unchanged caller gets it; with it switched off the head-side gradients are 2.3x further from fp32
    (asserted), the deep ones equally far -- their error is the forward's, see below.

    Bounds.  Head-side tensors (conv1, bn1, prompt tokens) are tight: 3e-2 / 5e-3.  The DEEP decoder matrices and norm parameters
    are bounded at 0.14 rel-L2 (measured: <= 0.114 / 0.117) -- not the 5e-2 one would like, and NOT because of the backward's
    format: the decoder is ill-conditioned in its input features.  The third leg of this test measures that in fp32: a relative
    Gaussian perturbation of 1e-4 on the backbone's three feature taps -- a fifth of IEEE half's unit roundoff (4.9e-4), an
    eightieth of bf16's -- already moves the deepest matrices' gradients by > 3 % (measured 4.8 %), growing like the square root
    of the perturbation (3e-4: 8.6 %, 1e-3: 16 %: the signature of arg-max / LeakyReLU flips in the k = 4 DGCNN max-pools and of
    a softmax over logits up to 49).  No 16-bit operand format in the frozen backbone can therefore reach 5e-2 here; the fp32
    parity mode is the mode for that (tools/partseg_error.py prints the full attribution: every single stage put back to fp32
    moves the worst matrix from 0.114 to no better than 0.10)."""
    from ppt_amd import engine, gradscale
    B, N = 16, 2048
    pc_np, s0 = W.synth_clouds(B, N, seed=5, duplicates=True)
    _, s1 = W.synth_clouds(B, N, seed=6)
    _, s2 = W.synth_clouds(B, N, seed=7)
    pc = torch.from_numpy(pc_np).cuda()
    labels = torch.from_numpy(np.random.default_rng(2).integers(0, 50, size=(B, N))).cuda()
    onehot = torch.nn.functional.one_hot(torch.arange(B) % 16, 16).float().cuda()
    drop = (torch.rand(B, N, 128, generator=torch.Generator().manual_seed(3)) >= 0.5).float() * 2.0
    noise = [0.0]
    pef = engine.point_encoder_forward

    def noisy_pef(*a, **k):
        out = pef(*a, **k)
        if k.get("fetch") is not None and noise[0]:
            gen = torch.Generator(device="cuda").manual_seed(11)
            return [f * (1 + noise[0] * torch.randn(f.shape, generator=gen, device=f.device)) for f in out[0]], out[1]
        return out

    def run(precision, policy="auto", tap_noise=0.0):
        old = gradscale.POLICY
        gradscale.POLICY, noise[0], engine.point_encoder_forward = policy, tap_noise, noisy_pef
        try:
            m = _model("ULIP_PointBERT_partseg", "shapenetpart", task='partseg', sd_fn=W.ulip_partseg_state_dict, precision=precision)
            m.train()
            pe = m.point_encoder
            pe.fps_start = tuple(torch.from_numpy(s).cuda() for s in (s0, s1, s2))
            pe.drop_path_factors = torch.ones(12, 2, B)
            pe.dropout_mask = drop
            criterion = torch.nn.CrossEntropyLoss(label_smoothing=0.2)                      # main_partseg.py:213
            optimizer = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
            optimizer.zero_grad()
            pred = m(pc, onehot)                                                              # main_partseg.py:210
            loss = criterion(pred.reshape(-1, 50), labels.reshape(-1))
            loss.backward()                                                                   # main_partseg.py:214
            grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
            optimizer.step()
            torch.cuda.synchronize()
            assert all(torch.isfinite(p).all() for p in m.parameters())
            return loss.item(), grads
        finally:
            gradscale.POLICY, noise[0], engine.point_encoder_forward = old, 0.0, pef

    l32, g32 = run(torch.float32)
    l16, g16 = run(torch.bfloat16)
    assert abs(l16 - l32) < 5e-3, (l16, l32)
    assert sorted(g16) == sorted(g32) and len(g16) == 41            # 40 decoder tensors (conv2 is unused by forward) + the prompt tokens
    head_side = {"point_encoder.conv1.weight": 3e-2, "point_encoder.bn1.weight": 5e-3, "point_encoder.bn1.bias": 5e-3,
                 "prompt_learner.learnable_tokens": 5e-3}
    worst, over = {1: ("", 0.0), 2: ("", 0.0)}, []
    for n in g32:
        if _zero_by_construction(n):
            continue
        r = _rel(g16[n], g32[n])
        d = 1 if g32[n].dim() == 1 else 2
        if n not in head_side and r > worst[d][1]:
            worst[d] = (n, r)
        bound = head_side.get(n, 0.14)
        print(f"PARITY C5 plain loop 16-bit vs fp32 grad {n} rel-L2: {r:.4g} (bound {bound})")
        if not r < bound:
            over.append((n, r, bound))
    print(f"PARITY C5 plain loop worst deep matrix {worst[2]}, worst deep 1-D {worst[1]}")
    assert not over, over
    # split16 (round 5: fp32 storage, every GEMM and the attention forward from hi + lo half pairs on the 16-bit matrix pipe): the
    # mode for callers who need these deep gradients right without the fp32 MFMA's cost -- bound 2e-2 on the deep matrices
    # (what is left is the decoder's conditioning acting on ~1e-6 relative differences), head side 1e-3
    ls, gs = run("split16")
    assert abs(ls - l32) < 1e-4, (ls, l32)
    deep_s, head_s, over = ("", 0.0), ("", 0.0), []
    for n in g32:
        if _zero_by_construction(n):
            continue
        r = _rel(gs[n], g32[n])
        if n in head_side:
            head_s = max(head_s, (n, r), key=lambda t: t[1])
        elif r > deep_s[1]:
            deep_s = (n, r)
        if not r < (5e-3 if n in head_side else 2e-2):
            over.append((n, r))
    print(f"PARITY C5 plain loop split16 vs fp32: worst deep tensor {deep_s}, worst head-side {head_s}, loss diff {abs(ls - l32):.3g}")
    assert not over, over
    # the same loop with the nodes' scale switched off: what the unchanged caller got before round 4
    _, goff = run(torch.bfloat16, policy="off")
    ratios = {n: (_rel(goff[n], g32[n]), _rel(g16[n], g32[n])) for n in g32 if not _zero_by_construction(n)}
    for n, (ro, rn) in ratios.items():
        print(f"PARITY C5 plain loop, gradient scale off: {n} rel-L2 {ro:.4g} (with the scale: {rn:.4g})")
    # where the forward's conditioning noise does not drown it (head side), the un-scaled half backward is visibly worse: measured
    # 2.3x on the prompt tokens and on bn1.weight (at this size the seed is 1/32768 per row; at 1/8192 x smaller still, the
    # gradient is lost altogether: tests/test_model_gpu.py::test_loss_scaling_keeps_fp16_gradients_out_of_the_subnormals)
    for n in ("prompt_learner.learnable_tokens", "point_encoder.bn1.weight"):
        assert ratios[n][0] > 1.5 * ratios[n][1], (n, ratios[n])
    # conditioning of the decoder in its input features, measured in fp32 (see the docstring)
    _, gn = run(torch.float32, tap_noise=1e-4)
    deep = max(_rel(gn[n], g32[n]) for n in g32 if g32[n].dim() >= 2 and n not in head_side)
    print(f"PARITY C5 conditioning: fp32 step, 1e-4 relative noise on the backbone's feature taps -> worst deep matrix rel-L2 {deep:.4g}")
    assert deep > 3e-2


@pytest.mark.parametrize("head_type,ds,N", [(0, "modelnet40", 1024), (3, "scanobjectnn", 2048)])
def test_plain_cls_loop_16bit_gradients_agree_with_fp32(head_type, ds, N):
    """The literal loop of main_cls.py:194-198 (`outputs = model(pc); loss = criterion(outputs, target); loss.backward();
    optimizer.step()`, stock AdamW, no train.Trainer) at B = 32 in the default 16-bit mode against the fp32 mode of the same
    step: every gradient within 1.5e-2 rel-L2 (the bound of the golden step, tests/test_model_gpu.py), with the scale coming
    from the nodes (B = 32 rows -> S = 32)."""
    B = 32
    pc_np, start = W.synth_clouds(B, N, seed=21, duplicates=(N == 2048))
    pc = torch.from_numpy(pc_np).cuda()
    C = 40 if ds == "modelnet40" else 15
    labels = torch.from_numpy(np.random.default_rng(4).integers(0, C, size=(B,))).cuda()

    def run(precision):
        m = _model("ULIP_PointBERT", ds, head_type=head_type, sd_fn=W.ulip_pointbert_state_dict, precision=precision)
        m.train()
        m.point_encoder.fps_start = torch.from_numpy(start).cuda()
        m.point_encoder.drop_path_factors = torch.ones(12, 2, B)
        criterion = torch.nn.CrossEntropyLoss(label_smoothing=0.2)                          # main_cls.py:52
        optimizer = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1)
        optimizer.zero_grad()
        outputs = m(pc)                                                                       # main_cls.py:194
        loss = criterion(outputs, labels)
        loss.backward()                                                                       # main_cls.py:197
        grads = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
        optimizer.step()
        torch.cuda.synchronize()
        assert all(torch.isfinite(p).all() for p in m.parameters())
        return loss.item(), grads

    l32, g32 = run(torch.float32)
    l16, g16 = run(torch.bfloat16)
    assert abs(l16 - l32) < 1e-3 * max(1.0, abs(l32)) and sorted(g16) == sorted(g32) and len(g16) == (1 if head_type == 0 else 12)
    for n in g32:
        r = _rel(g16[n], g32[n])
        print(f"PARITY plain cls loop h{head_type} 16-bit vs fp32 grad {n} rel-L2: {r:.4g} (bound 0.015)")
        assert r < 1.5e-2, (n, r)
