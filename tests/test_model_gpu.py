"""GPU: the drop-in model surface (ppt_amd.models.*) against the oracle / golden fixtures.

Tolerances (stated, per north_star "logits and gradients within a stated fp32 tolerance"):
  parity mode (fp32 MFMA):  logits |err| <= 2e-3 absolute on |logits| <= ~45 (4e-5 relative),
                            loss 1e-4, gradients 1e-3 relative (L2), BN running stats 1e-5;
  performance mode (set_precision(torch.bfloat16): 16-bit MFMA operands, fp32 accumulate -- IEEE half for the PointBERT
                            tokenizer, the transformer blocks and the text tower since round 3, bf16 only where a
                            tensor is not bounded by a normalisation: PointNet2 / PointMLP / the part-seg decoder):
                            logits 0.1 absolute (measured 0.049; 0.28 with bf16 operands), loss 0.01 (0.0027),
                            every gradient 1.5e-2 relative L2 (measured <= 0.85e-2; 5-6e-2 with bf16 operands);
                            tools/bf16_error.py attributes the error stage by stage.
FPS indices / kNN neighbour sets are bit-exact in both modes (they never leave fp32).
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import oracle as O
from ppt_amd import weights as W

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def build(head_type, precision, n_classes_ds="modelnet40"):
    from ppt_amd.models import ULIP_models as M
    args = SimpleNamespace(classnames=M.dataset_classnames(n_classes_ds), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=head_type, evaluate_3d=False, synthetic_weights=True,
                           ulip2=False)
    m = M.ULIP_PointBERT(args)
    sd = W.ulip_pointbert_state_dict(seed=0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert missing == ["token_embedding.weight"] and not unexpected
    m.prompt_learner.embedding = W.synth_prompt_embedding(len(args.classnames), seed=0)
    m.cuda()
    m.set_precision(precision)
    m.overlap_text_tower = False
    return m, sd


def _bound(what, value, bound):
    """assert value < bound, and print the pair (pytest -s): the stated bf16 tolerances are kept at ~1.5x what is measured"""
    print(f"PARITY {what}: {value:.4g} (bound {bound:.4g})")
    assert value < bound, (what, value, bound)


def oracle_inputs():
    pc, start = W.synth_clouds(4, 1024, seed=77)
    return torch.from_numpy(pc), start


FP32_GRADE = (torch.float32, "split16")       # the modes held to the fp32-level bounds (split16: hi + lo half products, fp32 storage)


@pytest.mark.parametrize("precision,ltol,gtol", [(torch.float32, 2e-3, 1e-3), ("split16", 2e-3, 1e-3), (torch.bfloat16, 0.1, 1.5e-2)])
@pytest.mark.parametrize("head_type", [0, 1, 2, 3])
def test_train_step_matches_oracle_and_golden(head_type, precision, ltol, gtol):
    from ppt_amd.train import Trainer
    g = np.load(os.path.join(G, f"g_step_h{head_type}.npz"))
    m, sd = build(head_type, precision)
    assert m.precision_name == {torch.float32: "fp32", "split16": "split16", torch.bfloat16: "mixed16"}[precision]
    pc, start = oracle_inputs()
    m.train()
    m.point_encoder.fps_start = torch.from_numpy(g["fps_start"]).cuda()
    m.point_encoder.drop_path_factors = torch.from_numpy(g["dp_masks"]).cuda()
    tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
    loss, pred = tr.step(pc.cuda(), torch.from_numpy(g["labels"]).cuda())
    torch.cuda.synchronize()
    # ---- against the golden fixture captured from the reference
    err = np.abs(pred.detach().cpu().numpy() - g["logits"]).max()
    _bound(f"step h{head_type} {precision} logits abs err", err, ltol)
    _bound(f"step h{head_type} {precision} loss abs err", abs(loss.item() - float(g["loss"])), 1e-3 if precision in FP32_GRADE else 0.01)
    # ---- gradients against the oracle (full tensors)
    masks = [(torch.from_numpy(a[0]), torch.from_numpy(a[1])) for a in g["dp_masks"]]
    nl = m.prompt_learner.name_lengths
    res = O.train_step(sd, pc, torch.from_numpy(g["labels"]), g["fps_start"], W.synth_prompt_embedding(40, 0), nl,
                       g["eot"].astype(np.int64), head_type=head_type, dp_masks=masks)
    live = dict(m.named_parameters())
    assert sorted(res["grads"]) == sorted(k for k, q in live.items() if q.requires_grad)     # the tier's trainable set
    for k, go in res["grads"].items():
        gg = live[k].grad.detach().cpu()
        rel = ((gg - go).norm() / go.norm()).item()
        _bound(f"step h{head_type} {precision} grad {k} rel-L2", rel, gtol)
        # ---- and against the gradient the REFERENCE produced (full tensor, or a strided sample + the norm of big ones)
        if "grad_" + k in g.files:
            gr = torch.from_numpy(g["grad_" + k])
            assert ((gg - gr).norm() / gr.norm()).item() < gtol, k
        else:
            gr = torch.from_numpy(g["gradsub_" + k])
            assert ((gg.flatten()[::97] - gr).norm() / gr.norm()).item() < gtol, k
            assert abs(gg.double().norm().item() / float(g["gradnorm_" + k]) - 1.0) < gtol, k
    # BN running statistics were updated exactly as nn.BatchNorm1d does in train()
    msd = m.state_dict()
    for k, v in res["new_stats"].items():
        tol = 1e-5 if precision in FP32_GRADE else 2e-3
        assert (msd[k].float().cpu() - v.float()).abs().max().item() < tol * max(1.0, v.float().abs().max().item()), k
    if precision in FP32_GRADE:
        # post-AdamW parameters: the update is ~lr*sign(g), so compare where |g| is well above noise
        for k, newp in res["new_params"].items():
            go = res["grads"][k]
            big = go.abs() > 1e-3 * go.abs().max()
            d = (live[k].detach().cpu() - newp).abs()[big].max().item()
            assert d < 5e-5, (k, d)


@pytest.mark.parametrize("precision", [torch.float32, "split16", torch.bfloat16])
@pytest.mark.parametrize("head_type", [0, 3])
def test_train_step_on_checkpoint_like_weights(head_type, precision):
    """VERDICT r4 weak #10 / #4b: every other parity number of this repository is on std-0.02 synthetic weights.  Here the SAME train
    step runs on checkpoint-LIKE magnitudes (ppt_amd.weights.checkpoint_like: LayerNorm gains log-normal around 1 with four 5-10x
    outlier channels per in-block LayerNorm, norm biases N(0, 0.1), 3 x larger weight matrices: |logits| up to 72, loss 50, token
    gradient norm 4e5) against fixtures the REFERENCE produced on those weights (tests/golden/make_golden.py ckpt ->
    g_step_h{0,3}_ckpt.npz).  Stated bounds: fp32 mode -- logits 5e-2 abs (7e-4 of the range; reference vs oracle differ by 1.2e-2
    themselves), loss 1e-2, gradients 1e-2 rel-L2; mixed 16-bit mode -- logits 1.0 abs (1.4 % of the range; measured 0.49), loss 0.5 (1 %;
    0.06), gradients 0.12 rel-L2 for matrices / tokens (tokens 0.009; the un-frozen block's fc1.weight 0.091) and 0.15 for the 1-D norm
    parameters (0.116) -- AFTER the mode's
    self-check moved the text tower to fp32 operands (without it: 14.3 / 4.3 / 1.2, tools/ckpt_like_error.py).  The health monitor must
    not have demoted anything: no half stage overflows at these magnitudes; what fails is accuracy, which is what the self-check sees."""
    from ppt_amd.train import Trainer
    g = np.load(os.path.join(G, f"g_step_h{head_type}_ckpt.npz"))
    m, sd0 = build(head_type, precision)
    sd = W.checkpoint_like(sd0, seed=0)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert missing == ["token_embedding.weight"] and not unexpected
    m.reset_caches()
    pc, _ = oracle_inputs()
    m.train()
    m.point_encoder.fps_start = torch.from_numpy(g["fps_start"]).cuda()
    m.point_encoder.drop_path_factors = torch.from_numpy(g["dp_masks"]).cuda()
    tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
    loss, pred = tr.step(pc.cuda(), torch.from_numpy(g["labels"]).cuda(), check_finite=True)
    tr.finish()
    torch.cuda.synchronize()
    # Round 6: with a trainable point side (head_type 3) the mixed mode's first-batch GRADIENT self-check (Trainer.calibrate_gradients)
    # finds its gradients 0.10 away from the fp32-grade ones on these weights -- inherited from the frozen tower's 16-bit forward,
    # tools/ckpt_like_h3_error.py -- and continues the run in split16: the step below is held to the fp32-grade bounds then.
    switched = precision == torch.bfloat16 and head_type == 3
    if precision == torch.bfloat16:
        cal = tr.grad_calibration
        print("PARITY ckpt-like gradient self-check:", cal)
        if head_type == 3:
            assert cal["checked"] and cal["over"] and cal["worst_rel_l2"] > 3 * cal["threshold"] and cal["mode_after"] == "split16"
            assert m.precision_name == "split16"
        else:
            assert not cal["checked"] and m.precision_name == "mixed16"          # only the PromptLearner trains: the text tower's own check covers it
    f32 = precision in FP32_GRADE or switched
    rng_ = float(np.abs(g["logits"]).max())
    err = np.abs(pred.detach().cpu().numpy() - g["logits"]).max()
    _bound(f"ckpt-like h{head_type} {precision} logits abs err (|logits| <= {rng_:.0f})", err, 5e-2 if f32 else 1.0)
    _bound(f"ckpt-like h{head_type} {precision} loss abs err (loss {float(g['loss']):.1f})", abs(loss.item() - float(g["loss"])), 1e-2 if f32 else 0.5)
    live = dict(m.named_parameters())
    worst = 0.0
    for k, q in live.items():
        if not q.requires_grad:
            continue
        gg = q.grad.detach().cpu()
        if "grad_" + k in g.files:
            gr = torch.from_numpy(g["grad_" + k])
            rel = ((gg - gr).norm() / gr.norm()).item()
        else:
            gr = torch.from_numpy(g["gradsub_" + k])
            rel = ((gg.flatten()[::97] - gr).norm() / gr.norm()).item()
            assert abs(gg.double().norm().item() / float(g["gradnorm_" + k]) - 1.0) < (1e-2 if f32 else 8e-2), k
        # (1-D norm parameters of the un-frozen block are sums with heavy cancellation: measured 0.116 on norm1.weight in the mixed
        # mode, 0.01-0.03 on the weight matrices and the tokens)
        gb = (1e-2 if precision == torch.float32 else 2e-2) if f32 else (0.15 if gg.dim() == 1 else 0.12)
        if gg.dim() > 1 or f32:
            worst = max(worst, rel)
        assert rel < gb, (k, rel)
    _bound(f"ckpt-like h{head_type} {precision} worst gradient rel-L2 (matrices / tokens)", worst,
           (1e-2 if precision == torch.float32 else 2e-2) if f32 else 0.12)
    print("PARITY ckpt-like demotions:", tr.demotions, "skipped gradient elements:", tr.nonfinite_grad_elements(),
          "text calibration:", m.text_calibration)
    assert not tr.demotions and tr.nonfinite_grad_elements() == 0 and not m.demoted
    if precision == torch.bfloat16 and not switched:
        # the mode's self-check (ULIP_WITH_IMAGE.calibrate_text_precision) found the half text tower too coarse for THESE weights
        # (tools/ckpt_like_error.py: 14.3 of |logits| <= 72 with it, 0.49 without) and moved it to fp32 operands; on the std-0.02
        # synthetic weights it stays on half (test_text_calibration_keeps_half_on_the_synthetic_weights)
        assert m.text_calibration and m.text_calibration["demoted"] and m.text_precision is torch.float32
        assert m.text_calibration["rel_l2"] > 5 * m.text_calibration["threshold"]


def test_gradient_self_check_leaves_no_trace_when_it_passes(monkeypatch):
    """Trainer.calibrate_gradients (round 6): on the synthetic weights, head_type 3, the mixed mode's gradients are within 2e-2 of the
    split16 mode's on the first batch (measured ~5e-3), so the run stays in mixed16 -- and the two dry passes leave NOTHING behind:
    three steps with the check are bit-identical (losses, trained parameters, BatchNorm running statistics, the RNG draws of FPS /
    DropPath) to three steps with PPT_GRAD_CHECK=off."""
    from ppt_amd.train import Trainer
    pc, _ = oracle_inputs()
    labels = torch.tensor([1, 2, 3, 4]).cuda()
    outs = {}
    for policy in ("off", "switch"):
        monkeypatch.setenv("PPT_GRAD_CHECK", policy)
        torch.manual_seed(123)
        torch.cuda.manual_seed(123)
        m, _ = build(3, torch.bfloat16)
        m.train()
        tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
        losses = []
        for _ in range(3):
            loss, _ = tr.step(pc.cuda(), labels)
            losses.append(loss.item())
        tr.finish()
        torch.cuda.synchronize()
        cal = tr.grad_calibration
        if policy == "switch":
            print("PARITY gradient self-check, synthetic weights h3:", cal)
            # (B = 4: the worst tensor is a 1-D LayerNorm gain, 1.0e-2; at C3's batch of 64 every gradient is within 5e-3)
            assert cal["checked"] and not cal["over"] and cal["worst_rel_l2"] < cal["threshold"] and cal["mode_after"] == "mixed16"
        else:
            assert cal == {"checked": False}
        assert m.precision_name == "mixed16"
        outs[policy] = (losses, {k: v.detach().clone() for k, v in m.state_dict().items()})
    assert outs["off"][0] == outs["switch"][0], (outs["off"][0], outs["switch"][0])
    for k, v in outs["off"][1].items():
        assert torch.equal(v, outs["switch"][1][k]), k


def test_text_calibration_keeps_half_on_the_synthetic_weights():
    """The half-vs-fp32 self-check of the text tower passes with a wide margin on the weights every other parity number was measured
    on, costs two text forwards once, and re-arms on load_state_dict / set_precision."""
    m, sd = build(0, torch.bfloat16)
    pc, start = oracle_inputs()
    m.eval()
    m.point_encoder.fps_start = torch.from_numpy(start).cuda()
    assert m.text_calibration is None
    with torch.no_grad():
        m(pc.cuda())
    cal = m.text_calibration
    print("PARITY synthetic-weights text tower, half vs fp32 operands, rel-L2 of the normalised features:", cal)
    assert cal and not cal["demoted"] and cal["rel_l2"] < 0.2 * cal["threshold"] and m.text_precision is None
    assert m._cache().dtype == torch.float16
    m.load_state_dict(W.checkpoint_like(sd, seed=0), strict=False)
    assert m.text_calibration is None and not m._text_calibrated
    import warnings
    with warnings.catch_warnings(record=True) as caught, torch.no_grad():
        warnings.simplefilter("always")
        m(pc.cuda())
    assert m.text_calibration["demoted"] and m.text_precision is torch.float32 and m._cache().dtype == torch.float32
    assert any("fp32 operands" in str(c.message) for c in caught)
    m.load_state_dict(sd, strict=False)                    # back to the synthetic weights: half again
    with torch.no_grad():
        m(pc.cuda())
    assert not m.text_calibration["demoted"] and m.text_precision is None


def test_loss_scaling_keeps_fp16_gradients_out_of_the_subnormals():
    """The performance mode's fp16 stages carry activation gradients in IEEE half.  With the criterion's mean over a batch
    2048x the golden one (emulated: the golden step's loss x 1/2048) an un-scaled backward loses the token gradient to fp16
    subnormals; with the nodes' own gradient scale (ppt_amd/gradscale.py) the gradient an UNCHANGED caller finds in .grad after a
    plain `loss.backward()` (main_cls.py:194-198) stays at the golden accuracy.  The default ("auto" = rows of the mean, here 4) is
    covered by test_train_step_matches_oracle_and_golden."""
    from ppt_amd import gradscale
    g = np.load(os.path.join(G, "g_step_h3.npz"))
    pc, _ = oracle_inputs()
    shrink = 1.0 / 2048
    errs = {}
    old = gradscale.POLICY
    try:
        for scale in ("off", str(4 * 2048)):
            gradscale.POLICY = scale
            m, sd = build(3, torch.bfloat16)
            m.train()
            m.point_encoder.fps_start = torch.from_numpy(g["fps_start"]).cuda()
            m.point_encoder.drop_path_factors = torch.from_numpy(g["dp_masks"]).cuda()
            criterion = torch.nn.CrossEntropyLoss(label_smoothing=0.2)
            pred = m(pc.cuda())
            loss = criterion(pred, torch.from_numpy(g["labels"]).cuda()) * shrink
            loss.backward()
            torch.cuda.synchronize()
            live = dict(m.named_parameters())
            for k in ("prompt_learner.learnable_tokens", "point_encoder.blocks.blocks.11.attn.qkv.weight"):
                gr = torch.from_numpy(g["grad_" + k] if "grad_" + k in g.files else g["gradsub_" + k])
                gg = live[k].grad.detach().cpu() / shrink
                gg = gg if gg.shape == gr.shape else gg.flatten()[::97]
                errs[(scale, k)] = ((gg - gr).norm() / gr.norm()).item()
    finally:
        gradscale.POLICY = old
    for (scale, k), e in errs.items():
        print(f"PARITY gradient scale {scale} (loss x 1/2048) grad {k} rel-L2: {e:.4g}")
        if scale == "off":
            assert e > 0.03, (k, e)            # the failure the scale is there for
        else:
            assert e < 1.5e-2, (k, e)


@pytest.mark.parametrize("precision,tol", [(torch.float32, 2e-3), ("split16", 2e-3), (torch.bfloat16, 0.1)])
def test_eval_forward_matches_golden(precision, tol):
    g = np.load(os.path.join(G, "g_eval.npz"))
    f0 = np.load(os.path.join(G, "g_step_h0.npz"))
    m, sd = build(0, precision)
    m.eval()
    m.point_encoder.fps_start = torch.from_numpy(f0["fps_start"]).cuda()
    pc, _ = oracle_inputs()
    with torch.no_grad():
        feat = m.point_encoder(pc.cuda())
        logits = m(pc.cuda())
    ftol = 5e-4 if precision in FP32_GRADE else 6e-3
    _bound(f"eval {precision} feature abs err", np.abs(feat.cpu().numpy() - g["pc_feat"]).max(), ftol)
    _bound(f"eval {precision} logits abs err", np.abs(logits.cpu().numpy() - g["logits"]).max(), tol)
    # argmax agreement is what validate() (main_cls.py:266-270) consumes
    assert (logits.argmax(1).cpu().numpy() == g["logits"].argmax(1)).all()


def test_c3_shape_head3_2048pts_15classes_parity():
    """BASELINE config C3 in miniature: ScanObjectNN class list (15 classes), 2048-point clouds with duplicate
    points (resampled with replacement), head_type=3, parity mode vs the oracle."""
    from ppt_amd.train import Trainer
    m, sd = build(3, torch.float32, "scanobjectnn")
    emb = W.synth_prompt_embedding(15, seed=0)
    m.prompt_learner.embedding = emb.clone().cuda()
    pc_np, start = W.synth_clouds(2, 2048, seed=5, duplicates=True)
    labels = torch.tensor([3, 14])
    m.train()
    m.point_encoder.fps_start = torch.from_numpy(start).cuda()
    rng = np.random.default_rng(0)
    dpf = torch.from_numpy((np.floor(0.9 + rng.random((12, 2, 2))) / 0.9).astype(np.float32))
    m.point_encoder.drop_path_factors = dpf.cuda()
    tr = Trainer(m, distributed=False)
    loss, pred = tr.step(torch.from_numpy(pc_np).cuda(), labels.cuda())
    eot = m.tokenized_prompts.argmax(-1).numpy()
    res = O.train_step(sd, torch.from_numpy(pc_np), labels, start, emb, m.prompt_learner.name_lengths, eot, head_type=3,
                       dp_masks=[(dpf[l, 0], dpf[l, 1]) for l in range(12)])
    assert (pred.detach().cpu() - res["logits"]).abs().max().item() < 2e-3
    live = dict(m.named_parameters())
    for k, go in res["grads"].items():
        rel = ((live[k].grad.cpu() - go).norm() / go.norm()).item()
        assert rel < 1e-3, (k, rel)


def test_text_truncation_is_exact():
    """evaluating the causal text tower only up to the last EOT position changes nothing (fp32 mode: bit-level
    differences can only come from nothing at all -- every surviving row sees identical operands)."""
    m, _ = build(0, torch.float32)
    m.train()
    pc, start = oracle_inputs()
    m.point_encoder.fps_start = torch.from_numpy(start).cuda()
    m.point_encoder.drop_path_factors = torch.ones(12, 2, 4)
    outs = []
    for trunc in (False, True):
        m.truncate_text_to_eot = trunc
        m.prompt_learner.learnable_tokens.grad = None
        logits = m(pc.cuda())
        logits.square().mean().backward()
        outs.append((logits.detach().clone(), m.prompt_learner.learnable_tokens.grad.clone()))
    assert m._text_len() == 37          # SOT + 32 ctx + <=2 name tokens + "." + EOT
    assert (outs[0][0] - outs[1][0]).abs().max().item() < 1e-4
    assert ((outs[0][1] - outs[1][1]).norm() / outs[0][1].norm()).item() < 1e-5


def test_overlapped_text_tower_is_identical():
    m, _ = build(0, torch.bfloat16)
    m.eval()
    pc, start = oracle_inputs()
    m.point_encoder.fps_start = torch.from_numpy(start).cuda()
    with torch.no_grad():
        a = m(pc.cuda())
        m.overlap_text_tower = True
        b = m(pc.cuda())
    torch.cuda.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize("head_type", [0, 3])
def test_run_ahead_training_is_identical(head_type):
    """Trainer.step with the prompt side queued on the text stream (and, for head_type 0, the next iteration's
    point tower running ahead of the optimizer), without and with the text tower replayed from hipGraphs, and with
    FPS + kNN + the frozen tokenizer of an iteration running ahead on the grouping stream, must produce exactly the losses and
    parameters of the single-stream eager step."""
    from ppt_amd.train import Trainer
    pc, start = oracle_inputs()
    label = torch.tensor([3, 17, 0, 39]).cuda()
    results = []
    for run_ahead, hip_graphs, group_ahead in ((False, False, False), (True, False, False), (True, True, False), (True, True, True)):
        m, _ = build(head_type, torch.bfloat16)
        m.use_hip_graphs = m.point_encoder.use_hip_graphs = hip_graphs
        m.train()
        torch.manual_seed(5)                   # DropPath factors are drawn on the device
        m.point_encoder.fps_start = torch.from_numpy(start).cuda()
        m.overlap_text_tower = run_ahead
        tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
        tr.run_ahead = run_ahead
        # (the clouds below are complete on the device when step() is called: pageable host-to-device copies are synchronous)
        tr.inputs_ready = tr.group_ahead_when_frozen = group_ahead
        assert tr._point_side_frozen == (head_type == 0)
        losses = []
        for it in range(6):
            loss, pred = tr.step(torch.roll(pc, it, 0).cuda(), label)
            losses.append(loss)
        tr.finish()
        torch.cuda.synchronize()
        results.append(([l.item() for l in losses], pred.clone(),
                        {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}))
        assert bool(m._graphs.entries) == hip_graphs          # the text tower really was replayed from a hipGraph
        # ... and the point tower: all of it for head_type 0, the frozen prefix in front of the last block otherwise
        tower = "point_fwd" if head_type == 0 else "point_prefix"
        kinds = sorted(k[0] + ("+" + k[-1] if k[-1] in ("grouped", "tokens") else "") for k in m.point_encoder._graphs.entries)
        # (with the grouping stage ahead the whole tokenizer runs there -- PointTransformer.tokenize_ahead -- and the tower graph
        # starts at the blocks)
        stage = "tokens" if m.point_encoder.tokenize_ahead else "group"
        assert kinds == ([] if not hip_graphs else [tower] if not group_ahead
                         else sorted([stage, stage, tower + ("+tokens" if stage == "tokens" else "+grouped")]))
    la, pa, wa = results[0]
    for lb, pb, wb in results[1:]:
        assert la == lb
        assert torch.equal(pa, pb)
        for n in wa:
            assert torch.equal(wa[n], wb[n]), n


@pytest.mark.parametrize("head_type", [0, 3])
def test_device_prefetcher_feeds_the_ahead_stage_bit_identically(head_type):
    """an unchanged main_cls.py:171-194 loop whose loader is wrapped in ppt_amd.data.DevicePrefetcher.  The batches
    come from pinned HOST memory every step (a different one each step), nothing is vouched for (Trainer.inputs_ready False): the
    tokenizer stage of a step still runs ahead on the grouping stream, behind the EVENT of the batch's own copy.  Losses, logits
    and trained parameters are the in-order run's, bit for bit; eval (validate()) likewise."""
    from ppt_amd import graphs
    from ppt_amd.data import DevicePrefetcher
    from ppt_amd.train import Trainer
    pc, start = oracle_inputs()
    label = torch.tensor([3, 17, 0, 39])
    host = [(torch.roll(pc, it, 0).contiguous().pin_memory(), label.pin_memory()) for it in range(7)]
    results = []
    for fed in (False, True):
        m, _ = build(head_type, torch.bfloat16)
        m.overlap_text_tower = True
        m.train()
        m.point_encoder.fps_start = torch.from_numpy(start).cuda()
        m.point_encoder.drop_path_factors = torch.ones(12, 2, 4)
        tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
        assert tr.inputs_ready is False
        losses = []
        if fed:
            for pc_d, lab_d in DevicePrefetcher(host):
                assert graphs.ready_event(pc_d) is not None
                pc_d = pc_d.cuda(non_blocking=True)                 # main_cls.py:188: a no-op on a device tensor
                loss, pred = tr.step(pc_d, lab_d)
                losses.append(loss)
        else:
            for pc_h, lab_h in host:
                loss, pred = tr.step(pc_h.cuda(), lab_h.cuda())
                losses.append(loss)
        tr.finish()
        torch.cuda.synchronize()
        kinds = {k[0] for k in m.point_encoder._graphs.entries}
        assert ("tokens" in kinds or "group" in kinds) == fed, kinds       # the ahead stage ran exactly when the copies carried events
        m.eval()
        with torch.no_grad():
            if fed:
                ev = [m(b[0]).float().clone() for b in DevicePrefetcher(host[:4])]
            else:
                ev = [m(b[0].cuda()).float().clone() for b in host[:4]]
        results.append(([l.item() for l in losses], pred.clone(), {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}, ev))
    (la, pa, wa, ea), (lb, pb, wb, eb) = results
    assert la == lb and torch.equal(pa, pb)
    for n in wa:
        assert torch.equal(wa[n], wb[n]), n
    for x, y in zip(ea, eb):
        assert torch.equal(x, y)


def test_saved_activations_do_not_live_in_the_ahead_stage_buffers():
    """head_type 3 with every RNG draw on the device and the tokenizer running ahead: the last block keeps activations for its
    backward, and the stage of a later step (which waits for the forward only) rewrites its ping-pong buffers -- so the prefix graph
    must read the tokens into buffers of its own.  No tensor the autograd node saves may alias a stage output."""
    from ppt_amd.train import Trainer
    pc, _ = oracle_inputs()
    label = torch.tensor([3, 17, 0, 39]).cuda()
    m, _ = build(3, torch.bfloat16)
    m.overlap_text_tower = True
    m.train()
    tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
    tr.inputs_ready = True
    pe = m.point_encoder
    for it in range(6):
        loss, _ = tr.step(torch.roll(pc, it, 0).cuda(), label)
    tr.finish()
    torch.cuda.synchronize()
    assert torch.isfinite(loss)
    stage = [g for k, g in pe._graphs.entries.items() if k[0] == "tokens"]
    prefix = [g for k, g in pe._graphs.entries.items() if k[0] == "point_prefix"]
    assert len(stage) == 2 and len(prefix) == 1
    stage_ptrs = {t.data_ptr() for g in stage for t in g.outputs}
    assert not any(t.data_ptr() in stage_ptrs for t in prefix[0].static_in)
    assert not any(t.data_ptr() in stage_ptrs for t in prefix[0].outputs if torch.is_tensor(t))


@pytest.mark.parametrize("head_type", [1, 2, 3])
def test_validation_between_training_epochs_reads_current_weights(head_type):
    """validate() under no_grad after further training must see the CURRENT last-block weights.  A
    whole-tower hipGraph would bake in the pointers of the bf16 operand copies, which are re-made at every optimizer
    step; only the frozen prefix may be replayed.  train -> eval -> train -> eval, graphs on, against eager execution."""
    from ppt_amd.train import Trainer
    pc, start = oracle_inputs()
    pc = pc.cuda()
    labels = torch.tensor([1, 7, 30, 12]).cuda()
    outs = {}
    for graphs_on in (False, True):
        m, _ = build(head_type, torch.bfloat16)
        m.use_hip_graphs = m.point_encoder.use_hip_graphs = graphs_on
        m.point_encoder.fps_start = torch.from_numpy(start).cuda()
        m.point_encoder.drop_path_factors = torch.ones(12, 2, 4)
        tr = Trainer(m, lr=3e-2, distributed=False)
        seq = []
        for epoch in range(3):
            m.train()
            for _ in range(4):                              # > graphs.WARMUP_CALLS, so replay is active where allowed
                tr.step(pc, labels)
            tr.finish()
            m.eval()
            with torch.no_grad():
                for _ in range(4):                          # several validate() batches: the third and later would replay
                    lg = m(pc)
            seq.append(lg.float().cpu().clone())
        outs[graphs_on] = seq
        if graphs_on:
            assert any(k[0] == "point_prefix" for k in m.point_encoder._graphs.entries), "the frozen prefix is replayed"
            assert not any(k[0] == "point_fwd" for k in m.point_encoder._graphs.entries), "never the trainable block"
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b)
    assert not torch.equal(outs[True][0], outs[True][2]), "training changed the logits between the two validations"


@pytest.mark.parametrize("head_type", [0, 1])
def test_fused_adamw_keeps_eval_caches_current(head_type):
    """ppt_adamw_step writes parameters through a raw pointer; without a version bump the eval text-feature
    cache (_te_cache) and the operand copies of a trained last-block weight (engine.WeightCache) would keep the values of the
    first step.  train -> eval -> train -> eval with the fused AdamW against torch.optim.AdamW: same logits at every validation
    (the two optimizers are bit-identical per step, test_adamw_step_matches_torch), and the validations differ from each other."""
    from ppt_amd.train import Trainer
    pc, start = oracle_inputs()
    pc = pc.cuda()
    labels = torch.tensor([1, 7, 30, 12]).cuda()
    outs = {}
    for fused in (False, True):
        m, _ = build(head_type, torch.bfloat16)
        m.point_encoder.fps_start = torch.from_numpy(start).cuda()
        m.point_encoder.drop_path_factors = torch.ones(12, 2, 4)
        tr = Trainer(m, lr=3e-2, distributed=False)
        tr.fused_adamw = fused
        seq = []
        for epoch in range(3):
            m.train()
            for _ in range(3):
                tr.step(pc, labels)
            tr.finish()
            m.eval()
            with torch.no_grad():
                for _ in range(2):
                    lg = m(pc)
            seq.append(lg.float().cpu().clone())
        outs[fused] = seq
    moved = min((outs[False][0] - outs[False][1]).abs().max().item(), (outs[False][1] - outs[False][2]).abs().max().item())
    assert moved > 0.5, moved                                 # training moves the validation logits (|logits| ~ 45)
    for a, b in zip(outs[False], outs[True]):                 # ... and the fused optimizer's validations follow, every epoch
        assert (a - b).abs().max().item() < 0.05 * moved, ((a - b).abs().max().item(), moved)


def test_replayed_activations_are_guarded_against_a_second_forward():
    """head_type 3 keeps the prefix's activations in the graph's buffers: a backward through a forward that a later
    forward has overwritten must fail loudly (and work with use_hip_graphs = False)."""
    m, _ = build(3, torch.bfloat16)
    m.train()
    pc, start = oracle_inputs()
    m.point_encoder.fps_start = torch.from_numpy(start).cuda()
    x = pc.cuda()
    for _ in range(3):                                    # two eager calls, then the capture
        m(x).sum().backward()
    assert any(k[0] == "point_prefix" for k in m.point_encoder._graphs.entries)
    first = m(x).sum()
    m(x)
    with pytest.raises(RuntimeError, match="overwritten by a later forward"):
        first.backward()
    m.use_hip_graphs = m.point_encoder.use_hip_graphs = False
    first = m(x).sum()
    m(x)
    first.backward()


def test_device_rng_draws_replay_like_eager():
    """FPS start indices and DropPath factors drawn on the device INSIDE the replayed hipGraph consume the generator
    exactly as the eager launches do: same seed, same losses."""
    from ppt_amd.train import Trainer
    pc, _ = oracle_inputs()
    label = torch.tensor([3, 17, 0, 39]).cuda()
    runs = []
    for hip_graphs in (False, True):
        m, _ = build(0, torch.bfloat16)
        m.train()
        m.use_hip_graphs = m.point_encoder.use_hip_graphs = hip_graphs
        m.overlap_text_tower = True
        tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
        torch.manual_seed(123)
        losses = [tr.step(pc.cuda(), label)[0] for _ in range(6)]
        tr.finish()
        torch.cuda.synchronize()
        runs.append([l.item() for l in losses])
        assert bool(m.point_encoder._graphs.entries) == hip_graphs
    assert runs[0] == runs[1], runs


def test_prompt_tuning_converges_on_a_fixed_batch():
    """60 iterations of the full step (two streams, hipGraph replay, fused head) on one fixed batch: the label-smoothed
    loss falls to a quarter of its start and the prompt stays finite -- the optimisation really uses the
    gradients the kernels produce."""
    from ppt_amd.train import Trainer
    m, _ = build(0, torch.bfloat16)
    m.train()
    m.overlap_text_tower = True
    pc, start = oracle_inputs()
    m.point_encoder.fps_start = torch.from_numpy(start).cuda()
    tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
    label = torch.tensor([3, 17, 0, 39]).cuda()
    losses = []
    torch.manual_seed(11)
    for _ in range(60):
        loss, _ = tr.step(pc.cuda(), label)
        losses.append(loss)
    tr.finish()
    torch.cuda.synchronize()
    v = [l.item() for l in losses]
    assert all(np.isfinite(v)) and torch.isfinite(m.prompt_learner.learnable_tokens).all()
    # synthetic weights start at a loss of ~26 (|logits| up to 45); DropPath and batch-of-4 BatchNorm keep it noisy
    assert np.mean(v[-5:]) < 0.25 * np.mean(v[:5]), (v[:5], v[-5:])


def test_group_and_encoder_modules():
    from ppt_amd.models.pointbert.dvae import Encoder, Group, knn_point
    from ppt_amd.models.pointbert import misc
    pc, start = W.synth_clouds(2, 2048, seed=3)
    xyz = torch.from_numpy(pc).cuda()
    st = torch.from_numpy(start).cuda()
    nb, ce = Group(512, 32)(xyz, start_idx=st)
    cidx = O.fps(pc, 512, start)
    _, nb_ref, ce_ref = O.group(pc, cidx, 32)
    assert np.array_equal(nb.cpu().numpy(), nb_ref) and np.array_equal(ce.cpu().numpy(), ce_ref)
    assert np.array_equal(misc.farthest_point_sample(xyz, 512, st).cpu().numpy(), cidx)
    assert np.array_equal(misc.index_points(xyz, torch.from_numpy(cidx).cuda()).cpu().numpy(), ce_ref)
    assert np.array_equal(np.sort(knn_point(32, xyz, ce).cpu().numpy(), -1), np.sort(O.knn(pc, ce_ref, 32)[0], -1))
    sd = {k[len("point_encoder.encoder."):]: v for k, v in W.ulip_pointbert_state_dict(0).items()
          if k.startswith("point_encoder.encoder.")}
    enc = Encoder(256)
    enc.load_state_dict(sd)
    enc.cuda()
    for prec, tol in ((torch.float32, 1e-4), (torch.bfloat16, 0.01)):
        enc.precision = prec
        for train in (False, True):
            enc.load_state_dict(sd)
            enc.train(train)
            out = enc(nb)
            with torch.no_grad():
                ref = O.mini_pointnet({"e." + k: v for k, v in sd.items()}, torch.from_numpy(nb_ref), train, prefix="e.")
            err = (out.cpu() - ref).abs().max().item()
            _bound(f"mini-PointNet {prec} train={train} abs err / max(1, |ref|)", err / max(1.0, ref.abs().max().item()), tol)


# ------------------------------------------------------------------ PointNet2-MSG (BASELINE config C4)
def _pn2_inputs():
    g = np.load(os.path.join(G, "g_pn2msg.npz"))
    pc_np, s1 = W.synth_clouds(2, 1024, seed=31)
    assert np.array_equal(s1, g["start1"])
    return g, torch.from_numpy(pc_np)


@pytest.mark.parametrize("precision,tol", [(torch.float32, 2e-3), (torch.bfloat16, 1e-3)])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_pointnet2_msg_matches_golden(mode, precision, tol):
    """Pointnet2_Msg forward vs the reference output captured in g_pn2msg.npz (tolerance relative to max|ref|;
    the train-mode FC head normalises over a batch of 2, which amplifies rounding)."""
    from ppt_amd.models.pointnet2.pointnet2 import Pointnet2_Msg
    g, pc = _pn2_inputs()
    m = Pointnet2_Msg()
    sd = W.synth_state_dict(W.pointnet2_msg_spec(prefix=""), seed=0)
    m.load_state_dict(sd)
    m.cuda()
    m.precision = precision
    m.train(mode == "train")
    m.fps_start = (torch.from_numpy(g["start1"]).cuda(), torch.from_numpy(g["start2"]).cuda())
    if mode == "train":
        m.dropout_masks = (torch.from_numpy(g["drop1"]), torch.from_numpy(g["drop2"]))
    out = m(pc.cuda())
    ref = torch.from_numpy(g[mode])
    err = (out.cpu() - ref).abs().max().item()
    if mode == "train" and precision == torch.bfloat16:
        # the golden batch has 2 clouds: BatchNorm1d of the FC head then normalises every channel to +-1 and divides by
        # a near-zero variance wherever the two samples almost agree -- ill-conditioned for 8-bit significands.  The bf16
        # train path is therefore checked against the fp32 parity path (itself pinned to the golden above) on a batch of 8.
        pc8, s8 = W.synth_clouds(8, 1024, seed=41)
        _, t8 = W.synth_clouds(8, 512, seed=42)
        rng = np.random.default_rng(3)
        dm = (torch.from_numpy((rng.random((8, 512)) > 0.4).astype(np.float32) / 0.6),
              torch.from_numpy((rng.random((8, 256)) > 0.5).astype(np.float32) / 0.5))
        outs = []
        for prec in (torch.float32, torch.bfloat16):
            m.load_state_dict(sd)
            m.precision, m._wc = prec, None
            m.fps_start = (torch.from_numpy(s8).cuda(), torch.from_numpy(t8).cuda())
            m.dropout_masks = dm
            outs.append(m(torch.from_numpy(pc8).cuda()).cpu())
        rel = ((outs[0] - outs[1]).norm() / outs[0].norm()).item()
        err8 = (outs[0] - outs[1]).abs().max().item()
        # three BatchNorm'd set-abstraction levels + two batch-of-8 BatchNorm1d layers on bf16 operands
        _bound("pn2msg bf16 train (batch 8) rel-L2 vs fp32 path", rel, 0.12)
        _bound("pn2msg bf16 train (batch 8) max err / max|out|", err8 / outs[0].abs().max().item(), 0.25)
        return
    _bound(f"pn2msg {mode} {precision} abs err / max|ref|", err / max(ref.abs().max().item(), 0.05), tol)
    if mode == "train":
        msd = m.state_dict()
        for k in ("sa1.bn_blocks.2.2.running_var", "sa2.bn_blocks.1.0.running_mean", "sa3.mlp_bns.2.running_var"):
            r = torch.from_numpy(g["stat_" + k])
            rtol = 1e-4 if precision == torch.float32 else 3e-2
            assert (msd[k].cpu() - r).abs().max().item() < rtol * max(1.0, r.abs().max().item()), k


@pytest.mark.parametrize("precision,tol", [(torch.float32, 2e-3), (torch.bfloat16, 1e-3)])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_pointnet2_ssg_matches_golden(mode, precision, tol):
    """N4: Pointnet2_Ssg forward vs the reference output captured in g_pn2ssg.npz (same conventions as the MSG test)."""
    from ppt_amd.models.pointnet2.pointnet2 import Pointnet2_Ssg
    g = np.load(os.path.join(G, "g_pn2ssg.npz"))
    pc_np, s1 = W.synth_clouds(2, 1024, seed=41)
    assert np.array_equal(s1, g["start1"])
    m = Pointnet2_Ssg()
    sd = W.synth_state_dict(W.pointnet2_ssg_spec(prefix=""), seed=0)
    m.load_state_dict(sd)
    m.cuda()
    m.precision = precision
    m.train(mode == "train")
    m.fps_start = (torch.from_numpy(g["start1"]).cuda(), torch.from_numpy(g["start2"]).cuda())
    if mode == "train":
        m.dropout_masks = (torch.from_numpy(g["drop1"]), torch.from_numpy(g["drop2"]))
    if mode == "train" and precision == torch.bfloat16:
        # (batch of 2 through BatchNorm1d: ill-conditioned for bf16 -- compare with the fp32 parity path on a batch of 8)
        pc8, s8 = W.synth_clouds(8, 1024, seed=43)
        _, t8 = W.synth_clouds(8, 512, seed=44)
        rng = np.random.default_rng(5)
        dm = (torch.from_numpy((rng.random((8, 512)) > 0.4).astype(np.float32) / 0.6),
              torch.from_numpy((rng.random((8, 256)) > 0.4).astype(np.float32) / 0.6))
        outs = []
        for prec in (torch.float32, torch.bfloat16):
            m.load_state_dict(sd)
            m.precision, m._wc = prec, None
            m.fps_start = (torch.from_numpy(s8).cuda(), torch.from_numpy(t8).cuda())
            m.dropout_masks = dm
            outs.append(m(torch.from_numpy(pc8).cuda()).cpu())
        rel = ((outs[0] - outs[1]).norm() / outs[0].norm()).item()
        _bound("pn2ssg bf16 train (batch 8) rel-L2 vs fp32 path", rel, 0.2)              # three BatchNorm'd levels + two batch-of-8 BatchNorm1d layers on bf16 operands
        return
    out = m(torch.from_numpy(pc_np).cuda())
    for _ in range(3):                                   # later calls replay the hipGraph: same result
        m.load_state_dict(sd) if mode == "train" else None
        again = m(torch.from_numpy(pc_np).cuda())
    ref = torch.from_numpy(g[mode])
    err = (out.cpu() - ref).abs().max().item()
    _bound(f"pn2ssg {mode} {precision} abs err / max|ref|", err / max(ref.abs().max().item(), 0.05), tol)
    assert (again.cpu() - ref).abs().max().item() < tol * max(ref.abs().max().item(), 0.05)
    if mode == "train":
        m.load_state_dict(sd)
        m(torch.from_numpy(pc_np).cuda())
        msd = m.state_dict()
        for k in ("sa1.mlp_bns.2.running_var", "sa2.mlp_bns.0.running_mean", "sa3.mlp_bns.2.running_var"):
            r = torch.from_numpy(g["stat_" + k])
            rtol = 1e-4 if precision == torch.float32 else 3e-2
            assert (msd[k].cpu() - r).abs().max().item() < rtol * max(1.0, r.abs().max().item()), k


@pytest.mark.parametrize("precision,tol", [(torch.float32, 2e-3), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_pointmlp_matches_golden(mode, precision, tol):
    """N4: pointMLP() forward vs the reference output captured in g_pointmlp.npz (FPS starts and Dropout masks injected)."""
    from ppt_amd.models.pointmlp.pointMLP import pointMLP
    g = np.load(os.path.join(G, "g_pointmlp.npz"))
    pc_np, s1 = W.synth_clouds(2, 1024, seed=61)
    assert np.array_equal(s1, g["start1"])
    m = pointMLP()
    sd = W.synth_state_dict(W.pointmlp_spec(prefix=""), seed=0)
    m.load_state_dict(sd)
    m.cuda()
    m.precision = precision
    m.train(mode == "train")
    m.fps_start = tuple(torch.from_numpy(g[f"start{i}"]).cuda() for i in (1, 2, 3, 4))
    if mode == "train":
        m.dropout_masks = (torch.from_numpy(g["drop1"]), torch.from_numpy(g["drop2"]))
    if mode == "train" and precision == torch.bfloat16:
        # Synthetic uniform clouds give nearly the same global feature for every sample, so the classifier's BatchNorm1d
        # over the batch divides by a variance that is mostly rounding noise: the OUTPUT of a bf16 run is ill-conditioned
        # by construction.  What is well-conditioned, and covers every layer, is the batch statistics each BatchNorm
        # folds into its running buffers: compare those with the fp32 parity path on a batch of 8.
        pc8, s8 = W.synth_clouds(8, 1024, seed=63)
        st8 = [s8] + [W.synth_clouds(8, n, seed=64 + i)[1] for i, n in enumerate((512, 256, 128))]
        stats = []
        for prec in (torch.float32, torch.bfloat16):
            m.load_state_dict(sd)
            m.precision, m._wc = prec, None
            m.fps_start = tuple(torch.from_numpy(s).cuda() for s in st8)
            m.dropout_masks = (torch.ones(8, 512), torch.ones(8, 256))
            out = m(torch.from_numpy(pc8).cuda())
            assert torch.isfinite(out).all()
            stats.append({k: v.cpu().clone() for k, v in m.state_dict().items() if "running_" in k})
        for k in stats[0]:
            if k.startswith("classifier.") and not k.endswith("1.running_mean"):
                continue
            moved = (stats[0][k] - sd[k]).norm().item()
            assert (stats[0][k] - stats[1][k]).norm().item() < 0.12 * moved, k
        return
    out = m(torch.from_numpy(pc_np).cuda())
    for _ in range(3):                                   # later calls replay the hipGraph: same result
        m.load_state_dict(sd) if mode == "train" else None
        again = m(torch.from_numpy(pc_np).cuda())
    ref = torch.from_numpy(g[mode])
    err = (out.cpu() - ref).abs().max().item()
    _bound(f"pointmlp {mode} {precision} abs err / max|ref|", err / max(ref.abs().max().item(), 0.05), tol)
    assert (again.cpu() - ref).abs().max().item() < tol * max(ref.abs().max().item(), 0.05)
    if mode == "train":
        m.load_state_dict(sd)
        m(torch.from_numpy(pc_np).cuda())
        msd = m.state_dict()
        for k in ("embedding.net.1.running_var", "pre_blocks_list.0.transfer.net.1.running_mean",
                  "pre_blocks_list.2.operation.1.net2.1.running_var", "pos_blocks_list.3.operation.0.net1.1.running_mean"):
            r = torch.from_numpy(g["stat_" + k])
            rtol = 1e-4 if precision == torch.float32 else 3e-2
            assert (msd[k].cpu() - r).abs().max().item() < rtol * max(1.0, r.abs().max().item()), k


@pytest.mark.parametrize("precision", [torch.float32, torch.bfloat16])
def test_pointmlp_fused_normalisation_is_the_aten_expression(precision):
    """ppt_pointmlp_cloud_rstd + ppt_pointmlp_pq (two launches per stage, csrc/pointmlp.hip) against the ~25 ATen launches they
    replace (pointMLP.py:170-175 + the transfer conv's operands by linearity): P and Q are the same operations in the same order --
    the same bits given the same r; r itself is an fp64 sum in another order, rounded to fp32 once -- equal or one ulp apart."""
    from ppt_amd import engine
    from ppt_amd.models.pointmlp.pointMLP import pointMLP
    pc_np, s1 = W.synth_clouds(4, 1024, seed=61)
    starts = [s1] + [W.synth_clouds(4, n, seed=64 + i)[1] for i, n in enumerate((512, 256, 128))]
    sd = W.synth_state_dict(W.pointmlp_spec(prefix=""), seed=0)
    outs = []
    for fused in (False, True):
        engine.POINTMLP_FUSED_NORM = fused
        try:
            m = pointMLP()
            m.load_state_dict(sd)
            m.cuda()
            m.precision = precision
            m.eval()
            m.fps_start = tuple(torch.from_numpy(s).cuda() for s in starts)
            outs.append(m(torch.from_numpy(pc_np).cuda()).float().cpu())
        finally:
            engine.POINTMLP_FUSED_NORM = True
    d = (outs[0] - outs[1]).abs().max().item()
    print(f"PARITY pointmlp fused normalisation vs ATen ({precision}): max abs diff {d:.3g} (|out| <= {outs[0].abs().max().item():.3g})")
    # (bf16 operands: a one-ulp difference in r flips bf16 roundings downstream -- the golden test above bounds that mode)
    assert d <= (2e-5 if precision == torch.float32 else 2e-2) * max(1.0, outs[0].abs().max().item())
    # the kernels themselves, on the same r: bit for bit
    from ppt_amd import ops
    g = torch.Generator().manual_seed(5)
    B, N, S, C = 3, 200, 50, 64
    PQ = torch.randn(B * N, 2 * C, generator=g).cuda()
    r = (torch.rand(B, generator=g) + 0.5).cuda()
    cidx = torch.randint(0, N, (B, S), generator=g).cuda()
    c0 = torch.randn(C, generator=g).cuda()
    P, Q = ops.pointmlp_pq(PQ, r, cidx, c0, B, N)
    Pw = (PQ[:, :C].reshape(B, N, C) * r.view(B, 1, 1)).reshape(B * N, C)
    a = (torch.arange(B, device="cuda").view(B, 1) * N + cidx).view(-1)
    Qw = (c0.view(1, C) + PQ[:, C:][a] - Pw[a]).contiguous()
    assert torch.equal(P, Pw) and torch.equal(Q, Qw)
    st = torch.rand(B, S, 2, generator=g).cuda() * torch.tensor([1.0, 40.0]).cuda()
    n = float(S * 24 * 64)
    rr = ops.pointmlp_cloud_rstd(st, n)
    sd_ = st.double().sum(1)
    var = ((sd_[:, 1] - sd_[:, 0] * sd_[:, 0] / n) / (n - 1.0)).clamp_min(0.0)
    want = (1.0 / (var.sqrt() + 1e-5)).float()
    assert ((rr - want).abs() <= 1.2e-7 * want.abs()).all()


def test_pointmlp_elite_matches_oracle():
    """pointMLPElite() (pointMLP.py:366-370: res_expansion 0.25, uneven block counts) against the oracle on the same inputs."""
    from ppt_amd.models.pointmlp.pointMLP import pointMLPElite
    from oracle import oracle as O
    torch.manual_seed(3)
    m = pointMLPElite()
    for n, b in m.named_buffers():
        if n.endswith("running_var"):
            b.uniform_(0.8, 1.4)
        elif n.endswith("running_mean"):
            b.normal_(0, 0.1)
    for n, p_ in m.named_parameters():
        if p_.dim() == 1 or "affine" in n:
            p_.data.add_(0.05 * torch.randn_like(p_))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    B = 4
    pc_np, s0 = W.synth_clouds(B, 1024, seed=71)
    starts = [s0] + [W.synth_clouds(B, n, seed=72 + i)[1] for i, n in enumerate((512, 256, 128))]
    with torch.no_grad():
        ref = O.pointmlp(sd, torch.from_numpy(pc_np), starts, train=False, prefix="", cfg=O.POINTMLP_ELITE)
    m.cuda().eval()
    m.precision = torch.float32
    m.fps_start = tuple(torch.from_numpy(s).cuda() for s in starts)
    out = m(torch.from_numpy(pc_np).cuda()).cpu()
    assert (out - ref).abs().max().item() < 2e-3 * max(ref.abs().max().item(), 0.05)


def test_ulip_pn_mlp_train_step_runs_and_only_prompt_trains():
    from ppt_amd.models import ULIP_models as M
    from ppt_amd.train import Trainer
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=0, evaluate_3d=False, ulip2=False, synthetic_weights=True)
    m = M.ULIP_PN_MLP(args)
    m.load_state_dict(W.ulip_pn_mlp_state_dict(seed=0), strict=False)
    m.cuda().train()
    tr = Trainer(m)
    pc_np, _ = W.synth_clouds(8, 1024, seed=81)
    pc = torch.from_numpy(pc_np).cuda()
    label = torch.arange(8, device="cuda") % 40
    before = {k: v.clone() for k, v in m.state_dict().items()}
    tr.inputs_ready = True            # pc is resident: FPS + kNN of a step run ahead on the grouping stream after warm-up
    losses = []
    for _ in range(6):
        loss, pred = tr.step(pc, label)
        losses.append(loss.item())
    tr.finish()
    assert sorted(str(k[0]) for k in m.point_encoder._graphs.entries) == ["pointmlp", "pointmlp_group", "pointmlp_group"]
    assert all(np.isfinite(losses)) and pred.shape == (8, 40)
    after = m.state_dict()
    changed = {k for k in before if before[k].dtype.is_floating_point and not torch.equal(before[k], after[k])
               and "running_" not in k}
    assert changed == {"prompt_learner.learnable_tokens"}, changed


def test_pointnet2_grouping_ahead_is_identical():
    """ULIP_PN_MSG training with FPS + ball queries of an iteration replayed on the grouping stream ahead of the step
    (Trainer.inputs_ready) gives exactly the losses and parameters of the in-order schedule."""
    from ppt_amd.models import ULIP_models as M
    from ppt_amd.train import Trainer
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=0, evaluate_3d=False, ulip2=False, synthetic_weights=True)
    B = 4
    pc_np, s0 = W.synth_clouds(B, 2048, seed=91)
    _, s1 = W.synth_clouds(B, 512, seed=92)
    rng = np.random.default_rng(3)
    dm = (torch.from_numpy((rng.random((B, 512)) > 0.4).astype(np.float32) / 0.6),
          torch.from_numpy((rng.random((B, 256)) > 0.5).astype(np.float32) / 0.5))
    label = torch.tensor([1, 5, 9, 30]).cuda()
    results = []
    for ahead in (False, True):
        m = M.ULIP_PN_MSG(args)
        m.load_state_dict(W.ulip_pn2_msg_state_dict(seed=0), strict=False)
        m.prompt_learner.embedding = W.synth_prompt_embedding(40, seed=0)
        m.cuda().train()
        m.point_encoder.fps_start = (torch.from_numpy(s0).cuda(), torch.from_numpy(s1).cuda())
        m.point_encoder.dropout_masks = dm
        tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
        tr.inputs_ready = ahead
        losses = []
        for it in range(6):
            loss, pred = tr.step(torch.from_numpy(np.roll(pc_np, it, 0)).cuda(), label)
            losses.append(loss)
        tr.finish()
        torch.cuda.synchronize()
        kinds = sorted(str(k[0]) for k in m.point_encoder._graphs.entries)
        assert kinds == (["pn2_group", "pn2_group", "pn2_msg"] if ahead else ["pn2_msg"]), kinds
        assert m.point_encoder.group_ahead is None         # the vouching ends with the step: later forwards run in order
        results.append(([l.item() for l in losses], pred.clone(), m.prompt_learner.learnable_tokens.detach().clone(),
                        m.point_encoder.bn2.running_var.clone()))
    (la, pa, ta, va), (lb, pb, tb, vb) = results
    assert la == lb and torch.equal(pa, pb) and torch.equal(ta, tb) and torch.equal(va, vb)


def test_ulip_pn_msg_train_step_runs_and_only_prompt_trains():
    from ppt_amd.models import ULIP_models as M
    from ppt_amd.train import Trainer
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=0, evaluate_3d=False, ulip2=False, synthetic_weights=True)
    m = M.ULIP_PN_MSG(args)
    m.load_state_dict(W.ulip_pn2_msg_state_dict(seed=0), strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(40, seed=0)
    m.cuda().train()
    assert [n for n, p in m.named_parameters() if p.requires_grad] == ["prompt_learner.learnable_tokens"]
    pc, _ = W.synth_clouds(4, 2048, seed=3)
    tr = Trainer(m, distributed=False)
    before = m.prompt_learner.learnable_tokens.detach().clone()
    loss, pred = tr.step(torch.from_numpy(pc).cuda(), torch.tensor([1, 2, 3, 4]).cuda())
    assert pred.shape == (4, 40) and torch.isfinite(pred).all() and np.isfinite(loss.item())
    assert (m.prompt_learner.learnable_tokens.detach() - before).abs().max().item() > 0


# ------------------------------------------------------------------ part segmentation (BASELINE config C5)
@pytest.mark.parametrize("precision", [torch.float32, "split16", torch.bfloat16])
def test_partseg_train_step_matches_golden(precision):
    """ULIP_PointBERT_partseg forward + CE + backward (main_partseg.py:204-215) on the golden case: B=2 x 2048 points
    with duplicate points, injected FPS starts / DropPath / Dropout.  fp32: logits 2e-2 abs (|logits| <= 47), loss 1e-3,
    gradients of the top layers 2e-3, of the deep decoder 6e-2 relative L2 (the reference itself differs from the
    oracle by up to 2e-2 there: max-pool arg-max flips amplify fp32 rounding -- see make_golden.gen_partseg)."""
    from ppt_amd.models import ULIP_models as M
    g = np.load(os.path.join(G, "g_partseg.npz"), allow_pickle=False)
    args = SimpleNamespace(classnames=M.dataset_classnames("shapenetpart"), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=0, task='partseg', head_type=0, evaluate_3d=False, ulip2=False, synthetic_weights=True)
    m = M.ULIP_PointBERT_partseg(args)
    assert sorted(n for n, p in m.named_parameters() if p.requires_grad) == sorted(g["trainable"].tolist())
    m.load_state_dict(W.ulip_partseg_state_dict(seed=0), strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(50, seed=0)
    m.cuda().set_precision(precision)
    m.overlap_text_tower = False
    m.train()
    pe = m.point_encoder
    pe.fps_start = tuple(torch.from_numpy(g[k]).cuda() for k in ("s0", "s1", "s2"))
    pe.drop_path_factors = torch.from_numpy(g["dp_masks"]).cuda()
    pe.dropout_mask = torch.from_numpy(np.unpackbits(g["drop"]).reshape(2, 2048, 128).astype(np.float32) * 2.0)
    pc_np, _ = W.synth_clouds(2, 2048, seed=55, duplicates=True)
    labels = torch.from_numpy(g["labels"].astype(np.int64)).cuda()
    pred = m(torch.from_numpy(pc_np).cuda(), torch.from_numpy(g["onehot"]).cuda())
    assert pred.shape == (2, 2048, 50)
    loss = torch.nn.CrossEntropyLoss(label_smoothing=0.2)(pred.reshape(-1, 50), labels.reshape(-1))
    loss.backward()
    f32 = precision in FP32_GRADE             # (split16 is held to the fp32 mode's bounds)
    err = np.abs(pred.detach().cpu().numpy()[:, ::16] - g["logits_sub"]).max()
    # performance mode: logits are logit_scale (14.3) x a cosine, |logits| <= 47 here.  With bf16 operands everywhere they moved
    # by 1.27; with fp16 operands in the text tower, tokenizer, blocks and decoder GEMMs and the fp16 per-point head: 0.254
    # (tools/f16_grad_range.py partseg), bound 0.3
    _bound(f"partseg {precision} logits abs err", err, 2e-2 if f32 else 0.3)
    _bound(f"partseg {precision} loss abs err", abs(loss.item() - float(g["loss"])), 1e-3 if f32 else 0.005)
    live = dict(m.named_parameters())
    top = ("point_encoder.conv1.weight", "point_encoder.bn1.weight", "point_encoder.bn1.bias", "prompt_learner.learnable_tokens")
    worst = 0.0
    for k in g["trainable"].tolist():
        if "gradnorm_" + k not in g:
            assert live[k].grad is None          # conv2: unused in forward (find_unused_parameters=True in the reference)
            continue
        ref_n = float(g["gradnorm_" + k])
        if ref_n < 1e-4:                          # biases in front of a BatchNorm: mathematically zero gradient
            continue
        sub = live[k].grad.detach().flatten()[::211].cpu().numpy()
        ref = g["gradsub_" + k]
        rel = np.linalg.norm(sub - ref) / np.linalg.norm(ref)
        if f32:
            worst = max(worst, rel if k not in top else 0.0)
            assert rel < (2e-3 if k in top else 6e-2), (k, rel)
            assert abs(live[k].grad.double().norm().item() / ref_n - 1) < 6e-2, k
        elif k in top:
            _bound(f"partseg bf16 grad {k} rel-L2", rel, 0.07)              # measured: conv1.weight 0.058, tokens 0.002 (bf16: 0.15)
        elif live[k].dim() >= 2:
            # bf16: logits = 100 x a cosine, so a feature error of a few 1e-3 moves a logit by ~1 and the per-point softmax by
            # tens of percent (and flips max-pool arg-maxima): measured rel-L2 0.18 ... 0.38 on the decoder's weight matrices,
            # while their NORMS agree within 1.2 % and the directions within 1 - cos <= 0.074.  All three are bounded.
            # (1-D norm parameters are sums with heavy cancellation and are only pinned in fp32 mode.)
            cos = float(np.dot(sub, ref) / (np.linalg.norm(sub) * np.linalg.norm(ref)))
            # measured with fp16 operands: <= 0.137 (bf16: 0.38), bound 0.15.  Not reachable below ~0.1 with ANY 16-bit operand format:
            # the decoder is ill-conditioned in its input features -- in fp32, 1e-4 relative noise on the backbone's feature taps
            # already moves these gradients by 5 % (tests/test_fullsize_gpu.py, C5 conditioning leg; tools/partseg_error.py)
            _bound(f"partseg bf16 grad {k} rel-L2", rel, 0.15)
            _bound(f"partseg bf16 grad {k} |norm ratio - 1|", abs(live[k].grad.double().norm().item() / ref_n - 1), 0.03)
            _bound(f"partseg bf16 grad {k} 1 - cos", 1 - cos, 0.1)
    if f32:
        print(f"PARITY partseg {precision} worst deep-decoder gradient rel-L2: {worst:.4g} (bound 0.06)")


def test_partseg_graphed_step_is_bit_identical_to_eager():
    """PointTransformer_partseg replays the frozen backbone's part of its forward (three FPS, kNN, tokenizer, 12 blocks), the
    decoder's forward and the decoder's hand-scheduled backward (ppt_amd.autograd._PartsegDecoder) from hipGraphs once a shape
    has been seen graphs.WARMUP_CALLS times.  Six training steps (Trainer: CE + backward + AdamW) with the
    RNG draws injected as static tensors: losses, trained parameters and BatchNorm running statistics are the same bits as with
    use_hip_graphs = False."""
    from ppt_amd.models import ULIP_models as M
    from ppt_amd.train import Trainer
    g = np.load(os.path.join(G, "g_partseg.npz"), allow_pickle=False)
    pc_np, _ = W.synth_clouds(2, 2048, seed=55, duplicates=True)
    pc = torch.from_numpy(pc_np).cuda()
    labels = torch.from_numpy(g["labels"].astype(np.int64)).cuda()
    onehot = torch.from_numpy(g["onehot"]).cuda()
    outs = {}
    for graphed in (False, True):
        args = SimpleNamespace(classnames=M.dataset_classnames("shapenetpart"), template_init='', class_name_position='middle',
                               num_learnable_prompt_tokens=32, gpu=0, task='partseg', head_type=0, evaluate_3d=False, ulip2=False,
                               synthetic_weights=True)
        m = M.ULIP_PointBERT_partseg(args)
        m.load_state_dict(W.ulip_partseg_state_dict(seed=0), strict=False)
        m.prompt_learner.embedding = W.synth_prompt_embedding(50, seed=0)
        m.cuda().set_precision(torch.bfloat16)
        m.train()
        pe = m.point_encoder
        pe.fps_start = tuple(torch.from_numpy(g[k]).cuda() for k in ("s0", "s1", "s2"))
        pe.drop_path_factors = torch.from_numpy(g["dp_masks"]).cuda()
        pe.dropout_mask = (torch.from_numpy(np.unpackbits(g["drop"]).reshape(2, 2048, 128).astype(np.float32)) * 2.0).cuda()
        pe.use_hip_graphs, pe._graph_injected = graphed, True
        tr = Trainer(m, lr=1e-3, label_smoothing=0.2, distributed=False)
        tr.extra_inputs = (onehot,)
        tr.inputs_ready = graphed                  # (the frozen backbone then runs ahead on its own stream)
        losses = [tr.step(pc, labels)[0] for _ in range(6)]
        tr.finish()
        torch.cuda.synchronize()
        # the ahead stage (two slots: the whole frozen backbone -- PointTransformer_partseg.backbone_ahead -- or the grouping stage +
        # one blocks graph) + decoder forward + decoder backward when graphed, nothing otherwise
        want = 4 if pe.backbone_ahead else 5
        assert (len(pe._graphs.entries) == want) == graphed and (graphed or not pe._graphs.entries), list(pe._graphs.entries)
        outs[graphed] = ([l.item() for l in losses], {n: q.detach().cpu().clone() for n, q in m.named_parameters() if q.requires_grad},
                         {n: b.detach().cpu().clone() for n, b in pe.named_buffers()})
    assert outs[False][0] == outs[True][0], (outs[False][0], outs[True][0])
    assert all(torch.equal(outs[False][1][k], outs[True][1][k]) for k in outs[False][1])
    assert all(torch.equal(outs[False][2][k], outs[True][2][k]) for k in outs[False][2])


def test_eval_text_cache_fast_path():
    """validate()-style inference: text features are computed once and reused until the prompt tokens change."""
    m, _ = build(0, torch.bfloat16)
    m.eval()
    pc, start = oracle_inputs()
    m.point_encoder.fps_start = torch.from_numpy(start).cuda()
    with torch.no_grad():
        a = m(pc.cuda())
        te = m._te_cache[1]
        b = m(pc.cuda())
        assert m._te_cache[1] is te and torch.equal(a, b)
        m.prompt_learner.learnable_tokens.add_(0.01)          # in-place update bumps the version -> cache refreshed
        c = m(pc.cuda())
        assert m._te_cache[1] is not te and not torch.equal(a, c)


@pytest.mark.parametrize("head_type", [0, 3])
def test_eval_with_the_tokenizer_ahead_is_identical(head_type):
    """validate() with ULIP_WITH_IMAGE.eval_inputs_ready: the grouping / tokenizer stage of each batch on its own stream, replayed
    from ping-pong graphs under the previous batch's blocks, gives bit-identical logits to the in-order forward -- over enough
    different batches to go through warm-up, capture and both buffer pairs."""
    pc, start = oracle_inputs()
    outs = []
    for ahead in (False, True):
        m, _ = build(head_type, torch.bfloat16)
        m.eval()
        m.eval_inputs_ready = ahead
        m.point_encoder.fps_start = torch.from_numpy(start).cuda()
        res = []
        with torch.no_grad():
            for it in range(7):
                batch = torch.roll(pc, it, 0).cuda()
                torch.cuda.synchronize()                       # the batch is complete on the device before forward() is called
                res.append(m(batch).clone())
        torch.cuda.synchronize()
        outs.append(res)
        kinds = {k[0] for k in m.point_encoder._graphs.entries}
        assert ("tokens" in kinds) == ahead
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("head_type", [0, 3])
def test_single_rank_rccl_step_is_identical(head_type):
    """SURVEY §8(e) on one GPU: Trainer(distributed=True) under a single-rank `nccl` (RCCL) process group -- the N-GPU code
    path: one all-reduce of the flat gradient buffer on the text stream, BatchNorm buffers re-bound for the broadcast --
    gives bit-identical losses, parameters and running statistics to the non-distributed run, at the same step time
    (no N > 1 run exists: the builder has one GPU at a time).  Child process: tests/dist_single_rank.py."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, MASTER_PORT=str(29611 + head_type))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_single_rank.py"), str(head_type), "12"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    print(res)
    assert res["finite"] and res["losses_equal"] and res["params_equal"] and res["bn_equal"], res
    assert res["n_params"] == (1 if head_type == 0 else 12)
    # the collective and the re-bound buffers must not cost step time (measured < 3 %; the bound is wide because this is a
    # 12-step wall-clock sample in a child process on a shared box: a tight one failed once in a run that took 1.6x as long overall)
    assert res["ms_dist"] < 1.35 * res["ms_plain"] + 0.5, res


def test_two_ranks_real_model_on_one_gpu(tmp_path):
    """SURVEY §8(e) with the REAL model on N = 2 ranks (reference main_cls.py:39,47-49,74-76): two processes on
    cuda:0 run ULIP_PointBERT head_type 3 under train.Trainer(distributed=True) on the halves of a B = 8 batch -- seeded `seed +
    rank`, with DIFFERENT initial prompt tokens / last block per rank.  Checked against single-process runs (child processes:
    tests/dist_two_ranks.py):
      * the DDP-constructor broadcast makes rank 1's trainable set equal rank 0's (it differed before);
      * the all-reduced gradient of step 1 is the mean of the two halves' single-process gradients;
      * after 3 steps every trained parameter is bit-identical on both ranks;
      * after finish() both ranks hold rank 0's BatchNorm running statistics = those of a single-process run on rank 0's half."""
    import subprocess
    import sys
    script = os.path.join(ROOT, "tests", "dist_two_ranks.py")
    out = str(tmp_path)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, script, "rank", str(r), out], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p_ in procs:
        try:
            o, _ = p_.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p_.returncode == 0 for p_ in procs), "\n".join(l[-3000:] for l in logs)
    r = subprocess.run([sys.executable, script, "ref", out], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r0, r1 = (torch.load(os.path.join(out, f"rank{k}.pt")) for k in range(2))
    ref = torch.load(os.path.join(out, "ref.pt"))
    names = sorted(r0["params"])
    assert len(names) == 12 and r0["init_broadcasts"] >= 1
    # the ranks started apart, the constructor's broadcast brought rank 1 to rank 0
    assert any(not torch.equal(r0["before"][n], r1["before"][n]) for n in names)
    for n in names:
        assert torch.equal(r1["after_bcast"][n], r0["before"][n]) and torch.equal(r0["after_bcast"][n], r0["before"][n]), n
    # step 1: the reduced gradient is the mean of the two single-process gradients
    worst, exact = 0.0, 0
    for n in names:
        mean = (ref["grads"][0][n] + ref["grads"][1][n]) / 2
        assert torch.equal(r0["grads_step1"][n], r1["grads_step1"][n]), n
        e = ((r0["grads_step1"][n] - mean).norm() / mean.norm()).item()
        worst, exact = max(worst, e), exact + int(torch.equal(r0["grads_step1"][n], mean))
        assert e < 1e-5, (n, e)
    print(f"two ranks ({r0['backend']}): reduced gradient vs mean of single-process gradients: worst rel-L2 {worst:.3g}, "
          f"{exact}/{len(names)} tensors bit-identical")
    # 3 steps: the same parameters on both ranks, bit for bit; they moved; nothing was skipped
    for n in names:
        assert torch.equal(r0["params"][n], r1["params"][n]), n
    assert any(not torch.equal(r0["params"][n], r0["before"][n]) for n in names)
    assert r0["skipped"] == 0 and r1["skipped"] == 0 and np.isfinite(r0["losses"]).all() and np.isfinite(r1["losses"]).all()
    assert r0["losses"] != r1["losses"]                               # different halves
    # BatchNorm running statistics: rank 0's, everywhere, = a single-process run over rank 0's inputs
    for k in r0["bn"]:
        assert torch.equal(r0["bn"][k], r1["bn"][k]), k
        assert torch.equal(r0["bn"][k], ref["bn_half0"][k]), k


def test_rowgemm_tower_matches_the_tile_gemm_tower():
    """The frozen ViT blocks on the weight-stationary linears with fused LayerNorms (csrc/rowgemm.hip; the engine switches to
    them from engine.ROWGEMM_MIN_ROWS token rows) against the LayerNorm kernel + tile-GEMM path, same model, bf16: features
    agree to bf16 rounding noise (the LayerNorm statistics are summed in another order: a rare 1-ulp operand flip), and
    against the fp32 golden features as well as the tile path does."""
    from ppt_amd import engine
    g = np.load(os.path.join(G, "g_eval.npz"))
    f0 = np.load(os.path.join(G, "g_step_h0.npz"))
    m, _ = build(0, torch.bfloat16)
    m.eval()
    m.point_encoder.fps_start = torch.from_numpy(f0["fps_start"]).cuda()
    m.point_encoder.use_hip_graphs = False
    pc, _ = oracle_inputs()
    saved = engine.ROWGEMM_MIN_ROWS
    feats = {}
    try:
        for name, thr in (("tile", 1 << 30), ("rowgemm", 0)):
            engine.ROWGEMM_MIN_ROWS = thr
            with torch.no_grad():
                feats[name] = m.point_encoder(pc.cuda()).float().cpu()
    finally:
        engine.ROWGEMM_MIN_ROWS = saved
    rel = ((feats["tile"] - feats["rowgemm"]).norm() / feats["tile"].norm()).item()
    assert rel < 5e-3, rel
    ref = torch.from_numpy(g["pc_feat"])
    e_tile, e_row = (feats["tile"] - ref).abs().max().item(), (feats["rowgemm"] - ref).abs().max().item()
    assert e_row < 0.15 and e_row < 1.5 * e_tile + 0.02, (e_tile, e_row)


def _token_structured_model(head_type, precision, position="middle"):
    from ppt_amd.models import ULIP_models as M
    args = SimpleNamespace(classnames=M.dataset_classnames("modelnet40"), template_init='', class_name_position=position,
                           num_learnable_prompt_tokens=32, gpu=0, task='cls', head_type=head_type, evaluate_3d=False, ulip2=False,
                           synthetic_weights=True)
    m = M.ULIP_PointBERT(args)
    m.load_state_dict(W.ulip_pointbert_state_dict(seed=0), strict=False)
    # token_embedding(tokenized_prompts) as the reference caches it (ULIP_models.py:102): same token id, same row
    m.prompt_learner.embedding = W.synth_prompt_embedding_from_tokens(m.tokenized_prompts, seed=0)
    m.cuda().set_precision(precision)
    m.overlap_text_tower = False
    return m


@pytest.mark.parametrize("position,want_p", [("middle", 17), ("end", 33), ("front", 0)])
def test_text_prefix_sharing_is_exact(position, want_p):
    """VERDICT r1 #1a: with the class name in the middle / at the end, the start token and the leading learnable context
    tokens are the same in every prompt and the mask is causal, so their activations are computed once (817 instead of
    1 480 rows for ModelNet40).  fp32 parity mode: text features BIT-identical to the unshared evaluation; the gradient of
    the learnable tokens equal up to the order in which the prompts' contributions are summed (1e-5 relative)."""
    m = _token_structured_model(0, torch.float32, position)
    assert m.prompt_learner.shared_prefix() == want_p
    res = {}
    for share in (False, True):
        m.share_text_prefix = share
        m.zero_grad()
        te = m.encode_text(m.prompt_learner(), m.tokenized_prompts)
        g = torch.Generator().manual_seed(1)
        (te * torch.randn(te.shape, generator=g).cuda()).sum().backward()
        res[share] = (te.detach().clone(), m.prompt_learner.learnable_tokens.grad.detach().clone())
    assert torch.equal(res[False][0], res[True][0])
    rel = ((res[False][1] - res[True][1]).norm() / res[False][1].norm()).item()
    assert rel < 1e-5, rel


def test_encode_text_with_per_class_prefixes_is_not_prefix_shared():
    """the public encode_text is general in `prompts`.  Prompts whose leading positions differ per class must
    not take the shared-prefix evaluation (which reads positions 0 .. P-1 from prompt 0 only): features and input gradient equal
    the evaluation with sharing switched off, bit for bit."""
    m = _token_structured_model(0, torch.float32, "middle")
    assert m.prompt_learner.shared_prefix() == 17
    g = torch.Generator().manual_seed(3)
    base = m.prompt_learner().detach()
    prompts = (base + 0.05 * torch.randn(base.shape, generator=g).cuda()).requires_grad_(True)     # per-class contexts
    res = {}
    for share in (False, True):
        m.share_text_prefix = share
        prompts.grad = None
        te = m.encode_text(prompts, m.tokenized_prompts)
        (te * torch.randn(te.shape, generator=torch.Generator().manual_seed(1)).cuda()).sum().backward()
        res[share] = (te.detach().clone(), prompts.grad.detach().clone())
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])


def test_text_prefix_sharing_bf16_and_training_step():
    """bf16 performance mode: shared-prefix tower == unshared tower (forward bit-identical, gradient to bf16 rounding), and a
    whole training step through the graph-replayed path gives the same loss and update as with sharing off."""
    from ppt_amd.train import Trainer
    m = _token_structured_model(0, torch.bfloat16)
    res = {}
    for share in (False, True):
        m.share_text_prefix = share
        m.zero_grad()
        te = m.encode_text(m.prompt_learner(), m.tokenized_prompts)
        g = torch.Generator().manual_seed(1)
        (te * torch.randn(te.shape, generator=g).cuda()).sum().backward()
        res[share] = (te.detach().clone(), m.prompt_learner.learnable_tokens.grad.detach().clone())
    assert torch.equal(res[False][0], res[True][0])
    assert ((res[False][1] - res[True][1]).norm() / res[False][1].norm()).item() < 2e-2
    pc, start = oracle_inputs()
    labels = torch.tensor([1, 7, 30, 12]).cuda()
    outs = {}
    for share in (False, True):
        mm = _token_structured_model(0, torch.bfloat16)
        mm.share_text_prefix = share
        mm.train()
        mm.point_encoder.fps_start = torch.from_numpy(start).cuda()
        mm.point_encoder.drop_path_factors = torch.ones(12, 2, 4)
        tr = Trainer(mm, lr=3e-3, distributed=False)
        first = None
        for i in range(4):                               # eager calls, then the hipGraph replays of both layouts
            loss, pred = tr.step(pc.cuda(), labels)
            if first is None:
                first = (loss.item(), pred.float().cpu().clone())
        tr.finish()
        torch.cuda.synchronize()
        outs[share] = (first, loss.item())
    # the first step's forward is the same bits (text features identical); later steps follow AdamW's sign-like first updates,
    # which amplify rounding-level gradient differences: only closeness is asked of them
    assert outs[False][0][0] == outs[True][0][0] and torch.equal(outs[False][0][1], outs[True][0][1])
    assert np.isfinite(outs[True][1]) and abs(outs[False][1] - outs[True][1]) < 0.1 * abs(outs[False][1])


def test_frozen_tower_on_its_own_stream_is_bit_identical():
    """Trainer.inputs_ready with a fully frozen point side (head_type 0): the point tower runs on a stream of its own that
    does not wait for the caller's stream (train.Trainer.tower_own_stream, ULIP_WITH_IMAGE.forward_loss).  Ten steps --
    eager calls, then hipGraph replays -- give the same losses, logits and learnable tokens, bit for bit, as the in-order
    schedule: the streams change when the work runs, not what it computes."""
    from ppt_amd.train import Trainer
    pc, start = oracle_inputs()
    labels = torch.tensor([1, 7, 30, 12]).cuda()
    outs = {}
    for own in (False, True):
        m = _token_structured_model(0, torch.bfloat16)
        m.overlap_text_tower = True
        m.train()
        m.point_encoder.fps_start = torch.from_numpy(start).cuda()
        m.point_encoder.drop_path_factors = torch.ones(12, 2, 4)
        tr = Trainer(m, lr=3e-3, distributed=False)
        tr.inputs_ready, tr.tower_own_stream = own, own
        pcs = [pc.cuda() + 0.001 * i for i in range(10)]          # a different cloud per step: a stale feature would show
        torch.cuda.synchronize()
        losses, preds = [], []
        for i in range(10):
            loss, pred = tr.step(pcs[i], labels)
            losses.append(loss)
            preds.append(pred)
        tr.finish()
        torch.cuda.synchronize()
        assert (tr._tower_used is not None) == own
        outs[own] = ([l.item() for l in losses], torch.stack([q.float() for q in preds]).cpu(),
                     m.prompt_learner.learnable_tokens.detach().cpu().clone())
    assert outs[False][0] == outs[True][0]
    assert torch.equal(outs[False][1], outs[True][1]) and torch.equal(outs[False][2], outs[True][2])


def test_text_tower_fused_paths_match_the_unfused_tower():
    """bf16: the prompt chain's fused nodes -- one-kernel prompt assembly (ppt_prompt_rows), LayerNorm inside the in_proj / c_fc
    linears (ppt_rowgemm_bf16), one-kernel AdamW -- against the same step with all of them off: text features within bf16
    rounding of the LayerNorm operands, token gradient within 8e-2 (cosine > 0.995)."""
    from ppt_amd import engine
    from ppt_amd.train import Trainer
    pc, start = oracle_inputs()
    labels = torch.tensor([1, 7, 30, 12]).cuda()
    res = {}
    saved = engine.TEXT_FUSE_LN
    try:
        for fused in (False, True):
            engine.TEXT_FUSE_LN = fused
            m = _token_structured_model(0, torch.bfloat16)
            m.fused_prompt_rows = fused
            m.train()
            m.point_encoder.fps_start = torch.from_numpy(start).cuda()
            m.point_encoder.drop_path_factors = torch.ones(12, 2, 4)
            tr = Trainer(m, lr=3e-3, distributed=False)
            tr.fused_adamw = fused
            te = m._text_raw().detach().float().cpu()
            loss, pred = tr.step(pc.cuda(), labels)
            tr.finish()
            torch.cuda.synchronize()
            res[fused] = (te, loss.item(), m.prompt_learner.learnable_tokens.grad.detach().cpu().clone(),
                          m.prompt_learner.learnable_tokens.detach().cpu().clone())
    finally:
        engine.TEXT_FUSE_LN = saved
    a, b = res[False], res[True]
    assert ((a[0] - b[0]).norm() / a[0].norm()).item() < 5e-3
    assert abs(a[1] - b[1]) < 2e-2 * abs(a[1])
    # (each variant is ~4.5e-2 away from the fp32 oracle's gradient, tools/bf16_error.py; the two differ by a 1-ulp flip of a
    # few bf16 LayerNorm outputs -- another summation order of the statistics -- that the 12 layers then carry along)
    assert ((a[2] - b[2]).norm() / a[2].norm()).item() < 8e-2
    cos = (a[2].flatten() @ b[2].flatten() / (a[2].norm() * b[2].norm())).item()
    assert cos > 0.995, cos


def _stressed_model(head_type=0):
    """LayerNorm gains of 10, four x30 outlier channels per LayerNorm, a x10 residual stream: the hazards of real checkpoints
    (tools/fp16_stress.py) -- on these the text tower's half BACKWARD overflows although features and loss stay finite."""
    sd = W.ulip_pointbert_state_dict(seed=0)
    gen = torch.Generator().manual_seed(7)
    for k in list(sd):
        if k.endswith(("norm1.weight", "norm2.weight", "ln_1.weight", "ln_2.weight", "ln_final.weight", "point_encoder.norm.weight")):
            v = sd[k] * 10.0
            v[torch.randperm(v.numel(), generator=gen)[:4]] *= 30.0
            sd[k] = v
        if k.endswith(("cls_token", "pos_embed.2.weight")) or k in ("positional_embedding", "token_embedding.weight"):
            sd[k] = sd[k] * 10.0
    m, _ = build(head_type, torch.bfloat16)
    m.load_state_dict(sd, strict=False)
    m.prompt_learner.embedding = W.synth_prompt_embedding(40, seed=0) * 10.0
    # (these tests are about the RUN-TIME monitor: the load-time self-check of the text tower -- calibrate_text_precision, which on
    # such weights moves the tower to fp32 operands before any step runs -- is switched off for them)
    m._text_calibrated = True
    m.train()
    return m


class _health_every:
    def __init__(self, n):
        self.n = str(n)

    def __enter__(self):
        # (PPT_TEXT_CALIBRATE=0: these tests are about the RUN-TIME monitor; the load-time self-check of the text tower --
        # calibrate_text_precision -- would move the stressed tower to fp32 operands before any step could overflow)
        self.old = {k: os.environ.get(k) for k in ("PPT_HEALTH_EVERY", "PPT_TEXT_CALIBRATE", "PPT_GRAD_CHECK")}
        os.environ["PPT_HEALTH_EVERY"] = self.n
        os.environ["PPT_TEXT_CALIBRATE"] = "0"
        os.environ["PPT_GRAD_CHECK"] = "off"            # (... and likewise the Trainer's first-batch gradient self-check)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_health_monitor_demotes_an_overflowing_half_stage():
    """VERDICT r3 #5(b, c): the mixed 16-bit mode watches itself (ppt_amd/health.py).  (1) ppt_health_check flags non-finite values
    and tracks max |x|.  (2) On weights with LayerNorm gains of 10, outlier channels and a x 10 residual stream -- the hazards of
    real checkpoints, tools/fp16_stress.py -- the text tower's half backward overflows although features and loss stay finite:
    the optimizer skips and counts those gradient elements, the monitor sees the counter move (BIT_GRAD), the Trainer demotes the
    16-bit backward stages to bf16 and training continues with finite gradients, nothing skipped any more, parameters finite.
    The demotion is the MODEL's (ULIP_WITH_IMAGE.demoted), not the process's: a second model built afterwards starts in half."""
    import warnings
    from ppt_amd import health, ops
    from ppt_amd.train import Trainer
    x = torch.randn(5000, device="cuda")
    flags, mx = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(1, device="cuda")
    ops.health_check(x, flags, 4, mx)
    assert flags.item() == 0 and mx.item() == x.abs().max().item()
    x[4321] = float("inf")
    ops.health_check(x.to(torch.float16), flags, 4, mx)
    assert flags.item() == 4
    m = _stressed_model()
    pc, _ = oracle_inputs()
    labels = torch.tensor([1, 2, 3, 4]).cuda()
    with _health_every(1):
        tr = Trainer(m, distributed=False)
        assert tr.health is not None and m.health is tr.health
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            skipped = []
            for _ in range(8):
                loss, _ = tr.step(pc.cuda(), labels)
                tr.finish()
                skipped.append(tr.nonfinite_grad_elements())
        assert skipped[0] > 0, "the stressed weights no longer overflow the half backward: make the stress harsher"
        assert tr.demotions and tr.demotions[0][1] & health.BIT_GRAD and m.text_f16 is False
        assert any(issubclass(c.category, RuntimeWarning) and "bf16" in str(c.message) for c in caught)
        assert skipped[-1] == skipped[-3], skipped                      # nothing skipped any more after the demotion
        assert np.isfinite(loss.item()) and all(bool(torch.isfinite(p).all()) for p in m.parameters())
        assert torch.isfinite(m.prompt_learner.learnable_tokens.grad).all()
        assert {"head", "last_block", "decoder"} <= m.demoted and m.point_encoder.demoted is m.demoted
    other, _ = build(0, torch.bfloat16)
    assert not other.demoted and other.text_f16, "a demotion must not leak into other models of the process"


def test_split16_mode_is_guarded_and_leaves_on_a_range_overflow():
    """split16 forms its products from IEEE-half pairs, so it has half's RANGE -- and the Trainer used to treat it
    as the fp32 mode (no monitor, no non-finite skip in AdamW).  Now: (1) Trainer installs the health monitor for a split16 model and
    the AdamW kernels keep their skip counter; (2) an activation beyond 65 504 (a LayerNorm gain of 1e6 in the text tower: 1-D, so
    the weight-range fit cannot see it) is SATURATED by the split -- loss, gradients and parameters stay finite -- and counted;
    (3) the monitor reports BIT_SPLIT and the model leaves the split16 products for the fp32 MFMA (`precision_name == "fp32"`), with
    a warning; (4) a split16 model WITHOUT such a value runs its steps without an event."""
    import warnings
    from ppt_amd import health
    from ppt_amd.train import Trainer
    pc, _ = oracle_inputs()
    labels = torch.tensor([1, 2, 3, 4]).cuda()
    with _health_every(1):
        m, _ = build(0, "split16")
        m.train()
        tr = Trainer(m, distributed=False)
        assert tr.health is not None and m.health is tr.health
        for _ in range(3):
            tr.step(pc.cuda(), labels)
        tr.finish()
        assert tr._skipped is not None and tr.nonfinite_grad_elements() == 0        # the guard exists, nothing tripped it
        assert not tr.demotions and m.precision_name == "split16" and not (tr.health.read_now() & health.BIT_SPLIT)

        m, _ = build(0, "split16")
        with torch.no_grad():
            m.transformer.resblocks[0].ln_1.weight[3] = 1.0e6
        m.reset_caches()
        m.train()
        tr = Trainer(m, distributed=False)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            for _ in range(4):
                loss, _ = tr.step(pc.cuda(), labels)
                tr.finish()
        assert tr.demotions and tr.demotions[0][1] & health.BIT_SPLIT, tr.demotions
        assert m.precision_name == "fp32" and not m.split16
        assert any(issubclass(c.category, RuntimeWarning) and "split16" in str(c.message) for c in caught)
        assert np.isfinite(loss.item()) and all(bool(torch.isfinite(p).all()) for p in m.parameters())


def test_health_monitor_stays_armed_after_an_event():
    """the flag word used to be sticky -- after the first BIT_POINT / BIT_LOSS event nothing new was ever
    reported, so a LATER half-backward overflow was skipped by AdamW every step, silently, for the rest of the run.  Now the word
    is cleared behind every read: (1) an injected non-finite feature event demotes the point tower; (2) the text tower's backward
    overflow that follows is still seen as BIT_GRAD and demotes the text side; (3) once nothing is left to demote, events that
    keep arriving raise FloatingPointError instead of making every step a silent no-op."""
    import warnings
    from ppt_amd import health
    from ppt_amd.train import Trainer
    m = _stressed_model()
    pc, _ = oracle_inputs()
    pc = pc.cuda()
    labels = torch.tensor([1, 2, 3, 4]).cuda()
    with _health_every(1), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tr = Trainer(m, distributed=False)
        # (1) a non-finite point-feature event in the first window (injected: the flag the tower's check would have set), together
        # with its echo BIT_LOSS
        tr.health.check(health.BIT_POINT, torch.tensor([float("inf")], device="cuda"))
        tr.health.check(health.BIT_LOSS, torch.tensor([float("nan")], device="cuda"))
        for _ in range(8):
            tr.step(pc, labels)
            tr.finish()
        bits = [d[1] for d in tr.demotions]
        assert bits[0] & health.BIT_POINT and {"tokenizer", "blocks"} <= m.demoted
        # (2) ... did not mask the gradient overflow of the text tower's half backward
        assert any(b & health.BIT_GRAD for b in bits[1:]), bits
        assert m.text_f16 is False
        n_skipped = tr.nonfinite_grad_elements()
        for _ in range(3):
            tr.step(pc, labels)
            tr.finish()
        assert tr.nonfinite_grad_elements() == n_skipped            # training goes on, nothing skipped any more
        # (3) every stage is on bf16 now: a loss that keeps coming back non-finite is not a half overflow
        with pytest.raises(FloatingPointError):
            for _ in range(3 * health.GIVE_UP_AFTER):
                tr.health.check(health.BIT_LOSS, torch.tensor([float("nan")], device="cuda"))
                tr.step(pc, labels)
                tr.finish()


def test_corrupt_label_is_reported_as_a_data_error_not_an_overflow():
    """a label outside [0, C) that is not ignore_index makes the loss NaN (ATen: device assert).  The monitor must not
    read that as a half overflow and demote the text tower for good: BIT_LABEL raises ValueError at the next poll, nothing demoted."""
    from ppt_amd.train import Trainer
    pc, _ = oracle_inputs()
    pc = pc.cuda()
    for head_type in (0, 3):                                        # fused head (ppt_head_ce_bwd) and ppt_cross_entropy_rows
        m, _ = build(head_type, torch.bfloat16)
        m.train()
        with _health_every(1):
            tr = Trainer(m, distributed=False)
            tr.step(pc, torch.tensor([1, 2, 3, 4]).cuda())
            with pytest.raises(ValueError, match="label"):
                for _ in range(3):
                    tr.step(pc, torch.tensor([1, 2, 40, 4]).cuda())
                    tr.finish()
        assert not m.demoted and m.text_f16 and not tr.demotions
        assert all(bool(torch.isfinite(p).all()) for p in m.parameters())      # the NaN step was skipped by the optimizer


def test_fp32_mode_keeps_the_reference_nan_behaviour():
    """the non-finite skip belongs to the 16-bit backward stages.  In the fp32 parity mode the AdamW kernels get no skip
    counter and do what torch.optim.AdamW does: a NaN gradient reaches the parameter (main_cls.py:205-207 then stops the run)."""
    from ppt_amd import ops
    p = torch.ones(8, device="cuda")
    g = torch.zeros(8, device="cuda")
    g[3] = float("nan")
    m_, v_ = torch.zeros(8, device="cuda"), torch.zeros(8, device="cuda")
    ops.adamw_step(p, g, m_, v_, 1e-3, 0.9, 0.98, 1e-8, 0.1, 1, skipped=None)
    assert torch.isnan(p[3]) and torch.isfinite(p[[0, 1, 2, 4, 5, 6, 7]]).all()
    p2 = torch.ones(8, device="cuda")
    sk = torch.zeros(1, dtype=torch.int64, device="cuda")
    ops.adamw_step(p2, g.clone(), torch.zeros(8, device="cuda"), torch.zeros(8, device="cuda"), 1e-3, 0.9, 0.98, 1e-8, 0.1, 1, skipped=sk)
    assert p2[3].item() == 1.0 and sk.item() == 1
    ref = torch.ones(8, device="cuda", requires_grad=True)
    opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.1, foreach=False)
    ref.grad = g.clone()
    opt.step()
    assert torch.equal(torch.isnan(ref.detach()), torch.isnan(p)) and torch.equal(ref.detach()[[0, 1, 2]], p[[0, 1, 2]])


def test_graph_capture_runs_without_the_cyclic_collector():
    """graphs.GraphedCall captures with Python's cyclic garbage collector switched off (a collection inside a capture finalises
    CUDA objects left in reference cycles -- not a capturable operation: the process aborted once in the full suite) and restores
    the collector's state afterwards, also when the captured function raises."""
    import gc
    from ppt_amd import graphs
    seen = []
    x = torch.ones(8, device="cuda")

    def fn(t):
        seen.append(gc.isenabled())
        return (t * 2,), None
    assert gc.isenabled()
    g = graphs.GraphedCall(fn, [x])
    assert seen == [False] and gc.isenabled()
    (y,), _ = g(torch.full((8,), 3.0, device="cuda"))
    torch.cuda.synchronize()
    assert float(y[0]) == 6.0

    def bad(t):
        raise ValueError("inside the capture")
    with pytest.raises(ValueError):
        graphs.GraphedCall(bad, [x])
    assert gc.isenabled()
    gc.disable()
    try:
        graphs.GraphedCall(fn, [x])
        assert not gc.isenabled()                # (a caller that had it off keeps it off)
    finally:
        gc.enable()
