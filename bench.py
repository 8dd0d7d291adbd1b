#!/usr/bin/env python3
"""bench.py -- headline benchmark of the PPT hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], "C2"): ModelNet40 class list, 1024-point clouds, PointBERT
(ULIP_PointBERT, head_type=0: frozen backbone + PromptLearner), batch 32 PER GPU, model.train()
(batch-statistics BatchNorm + DropPath), forward + CrossEntropy(label_smoothing 0.2) + backward +
AdamW, bf16 MFMA operands / fp32 accumulate.  One step == one iteration of main_cls.py:179-214.
Synthetic clouds (resident in HBM before the timed region) and deterministic random weights.

Timing: BURN_IN_STEPS (40) untimed steps (clock ramp-up, hipGraph capture; reported as "burn_in"), then --warmup W
untimed steps, then EXACTLY --steps K steps bracketed by barrier + torch.cuda.synchronize(): `value` = clouds / that
time; "ms_per_step_median" is the median of the K per-step times taken with HIP events on the caller's stream.

Prints ONE JSON line (rank 0): metric/value/unit..., plus
  "roofline":     the dominant kernel (the 16-bit -- fp16 / bf16 -- MFMA GEMM family) -- algorithmic FLOPs / its summed launch
                  time measured with HIP events on the launch stream during a second, instrumented pass
                  over the same K steps -- against the 2.5 PFLOP/s dense bf16 peak;
                  "roofline.kernels" lists EVERY hot kernel of that pass the same way: FPS / kNN / ball query against
                  8 TB/s with SURVEY §8(d)'s algorithmic bytes, attention and each GEMM kernel against the MFMA peak;
  "parity_mode":  clouds/s of the same step in the fp32 parity mode (N = 1 only);
  "split16_mode": ... and in the split16 mode (fp32 storage, products from hi + lo half pairs on the 16-bit matrix pipe: the fp32
                  mode's parity bounds at ~2x its rate);
  "cpu_baseline": the oracle (CPU restatement of the reference, `kind: "port"`) timed on this host's cores
                  on the reference's CPU-runnable case C1 (batch 8), rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # two compute streams + RCCL's: see ppt_amd/__init__.py

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PER_GPU_BATCH = 32
NPOINTS = 1024
HEAD_TYPE = 0
CONFIGS = {   # BASELINE.json configs[1] (headline) and configs[2] (secondary, --config C3)
    "C2": dict(dataset="modelnet40", batch=32, npoints=1024, head_type=0,
               name="C2: ModelNet40 1024-pt PointBERT (ULIP_PointBERT head_type=0, frozen backbone + PromptLearner)"),
    "C3": dict(dataset="scanobjectnn", batch=64, npoints=2048, head_type=3,
               name="C3: ScanObjectNN (PB_T50_RS shape) 2048-pt PointBERT + PointAdapter (head_type=3: last block un-frozen)"),
    "C4": dict(dataset="modelnet40", batch=32, npoints=8192, head_type=0, model="ULIP_PN_MSG",
               name="C4: ModelNet40 8192-pt PointNet2-MSG encoder (ULIP_PN_MSG, frozen) + PromptLearner, 32 clouds per GPU"),
    "MLP": dict(dataset="modelnet40", batch=32, npoints=1024, head_type=0, model="ULIP_PN_MLP",
                name="N4: ModelNet40 1024-pt PointMLP encoder (ULIP_PN_MLP, frozen) + PromptLearner, 32 clouds per GPU"),
    "C5": dict(dataset="shapenetpart", batch=16, npoints=2048, head_type=0, model="ULIP_PointBERT_partseg", task="partseg",
               name="C5: ShapeNetPart 2048-pt part segmentation (ULIP_PointBERT_partseg: frozen PointBERT + trainable decoder "
                    "+ PromptLearner), per-point logits, 16 clouds per GPU"),
}
METRICS = {"C2": "point-clouds/sec fwd+bwd, PointBERT 1024-pt ModelNet40",
           "C3": "point-clouds/sec fwd+bwd, PointBERT 2048-pt ScanObjectNN + PointAdapter",
           "C4": "point-clouds/sec fwd+bwd, PointNet2-MSG 8192-pt ModelNet40",
           "MLP": "point-clouds/sec fwd+bwd, PointMLP 1024-pt ModelNet40",
           "C5": "point-clouds/sec fwd+bwd, PointBERT part-seg 2048-pt ShapeNetPart"}
GROUP_AHEAD = os.environ.get("PPT_GROUP_AHEAD", "1") != "0"
BURN_IN_STEPS = 40               # untimed, before the --warmup steps (clock ramp, graph capture); reported as "burn_in"
PEAK_BF16_TFLOPS = 2500.0        # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3          # f32-input MFMA == f32 vector rate (same guide)
PEAK_HBM_GBS = 8000.0            # HBM3E spec (6.3 TB/s is what a copy kernel reaches)
HBM_KERNELS = ("fps", "knn_group", "ball_query")    # index work: accounted against algorithmic HBM bytes


def kernel_table(by_kernel, overhead_ms, steps):
    """roofline.kernels: one entry per hot kernel of the instrumented pass -- time per step, algorithmic work per step
    (bytes for the index kernels with SURVEY §8(d)'s formulas, FLOPs for the MFMA kernels), achieved rate and its
    fraction of the roofline that bounds it."""
    rows = []
    for k, d in sorted(by_kernel.items(), key=lambda kv: -kv[1]["ms"]):
        ms = max(d["ms"] - overhead_ms * d["launches"], 1e-6)
        hbm = d["family"] in HBM_KERNELS
        f32 = d["family"].endswith("f32")
        # achieved / frac are priced on the work the launch EXECUTES; the model's algorithmic figure (e.g. conv3
        # as the 512-wide conv whose broadcast half is evaluated once per group) is printed beside it
        rate = d["executed"] / (ms * 1e-3) / (1e9 if hbm else 1e12)
        peak = PEAK_HBM_GBS if hbm else (PEAK_F32_TFLOPS if f32 else PEAK_BF16_TFLOPS)
        row = {"kernel": k, "bound": "hbm" if hbm else "mfma", "launches_per_step": round(d["launches"] / steps, 2),
               "us_per_launch": round(1e3 * ms / d["launches"], 2), "ms_per_step": round(ms / steps, 4),
               ("algorithmic_MB_per_step" if hbm else "algorithmic_GFLOP_per_step"): round(d["work"] / steps / (1e6 if hbm else 1e9), 3)}
        if not hbm:
            row["executed_GFLOP_per_step"] = round(d["executed"] / steps / 1e9, 3)
        row.update({"achieved": round(rate, 2), "unit": "GB/s" if hbm else "TFLOP/s", "peak": peak, "frac": round(rate / peak, 5)})
        rows.append(row)
    return rows


def parity_mode_rate(cfg, pc, label, extra, steps, burn_in=6, mode="fp32", like=None):
    """clouds/s of the SAME step with set_precision("fp32") -- fp32 operands on the fp32 MFMA / fp32 VALU attention -- or
    set_precision("split16") -- the same fp32 storage with every GEMM and the attention forward formed from hi + lo half pairs on
    the 16-bit matrix pipe: the two modes whose results meet the fp32-level tolerances of tests/test_model_gpu.py."""
    from ppt_amd.train import Trainer
    m = build_model(cfg["dataset"], cfg["head_type"], precision=torch.float32, model=cfg.get("model", "ULIP_PointBERT"),
                    task=cfg.get("task", "cls"))
    if mode == "split16":
        m.set_precision("split16")
    m.train()
    tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
    tr.extra_inputs = extra
    if like is not None:               # the same promises the headline's trainer runs with (resident batch: the input stages run ahead)
        tr.inputs_ready, tr.group_ahead_when_frozen = like.inputs_ready, like.group_ahead_when_frozen
    for _ in range(burn_in):
        tr.step(pc, label)
    tr.finish()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = tr.step(pc, label)
    tr.finish()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert np.isfinite(loss.item())
    return {"dtype": "f32" if mode == "fp32" else "f32 as hi+lo f16 pairs", "value": round(pc.shape[0] * steps / dt, 2), "unit": "point-clouds/s",
            "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "burn_in": burn_in}


def operand_formats(model_name):
    """Which 16-bit operand format each part of the performance mode computes in (engine.BLOCKS_F16 / TOKENIZER_F16,
    ULIP_WITH_IMAGE.text_f16): IEEE half where the operands are bounded by a normalisation, bf16 elsewhere; fp32 accumulation,
    fp32 residual streams / statistics / heads everywhere."""
    from ppt_amd import engine
    text = "f16" if os.environ.get("PPT_TEXT_F16", "1") != "0" else "bf16"
    blocks = "f16" if engine.BLOCKS_F16 else "bf16"
    tok = "f16" if engine.TOKENIZER_F16 else "bf16"
    if model_name == "ULIP_PointBERT":
        return {"text_tower": text, "pointbert_tokenizer": tok, "transformer_blocks": blocks, "heads": "f32"}
    if model_name == "ULIP_PointBERT_partseg":
        return {"text_tower": text, "pointbert_tokenizer": tok, "transformer_blocks": blocks,
                "partseg_decoder": "f16" if engine.DECODER_F16 else "bf16", "per_point_head": "f16"}
    return {"text_tower": text, "point_encoder": "bf16", "heads": "f32"}


def formats_of(model, model_name):
    """operand_formats() corrected by what the model decided at run time: the text tower's half-vs-fp32 self-check
    (ULIP_WITH_IMAGE.calibrate_text_precision) and the health monitor's demotions."""
    f = operand_formats(model_name)
    if getattr(model, "text_precision", None) is torch.float32:
        f["text_tower"] = "f32 as hi+lo f16 pairs (split16)" if getattr(model, "text_split16", False) else "f32"
    cal = getattr(model, "text_calibration", None)
    if cal:
        f["text_tower_half_vs_fp32_rel_l2"] = round(cal["rel_l2"], 6)
    if getattr(model, "demoted", None):
        f["demoted_to_bf16"] = sorted(model.demoted)
    return f


def measured_parity():
    """The performance mode's error on the golden train step (tests/golden/g_step_h0.npz: B = 4 x 1024 points, head_type 0,
    logits / loss / token gradient captured from the upstream reference; fixtures are data, no oracle involved): what the
    headline number's arithmetic is worth, measured in the same process.  PPT_BENCH_WEIGHTS=ckpt_like: the same step on
    checkpoint-like weight magnitudes against g_step_h0_ckpt.npz (the reference on THOSE weights)."""
    import warnings
    from ppt_amd import weights as W
    from ppt_amd.train import Trainer
    name = "g_step_h0_ckpt.npz" if WEIGHTS == "ckpt_like" else "g_step_h0.npz"
    g = np.load(os.path.join(ROOT, "tests", "golden", name))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")             # (the self-check's verdict is reported in operand_formats, not on stderr)
        m = build_model("modelnet40", 0)
        m.prompt_learner.embedding = W.synth_prompt_embedding(40, seed=0).cuda()
        m.use_hip_graphs = m.point_encoder.use_hip_graphs = False
        m.overlap_text_tower = False
        m.train()
        pc, _ = W.synth_clouds(4, 1024, seed=77)
        m.point_encoder.fps_start = torch.from_numpy(g["fps_start"]).cuda()
        m.point_encoder.drop_path_factors = torch.from_numpy(g["dp_masks"]).cuda()
        tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
        loss, pred = tr.step(torch.from_numpy(pc).cuda(), torch.from_numpy(g["labels"]).cuda())
    torch.cuda.synchronize()
    gt = m.prompt_learner.learnable_tokens.grad.detach().cpu().numpy()
    gr = g["grad_prompt_learner.learnable_tokens"]
    return {"fixture": f"tests/golden/{name} (reference logits / loss / gradient, B = 4)",
            "logits_abs_err": round(float(np.abs(pred.detach().float().cpu().numpy() - g["logits"]).max()), 4),
            "logits_abs_max": round(float(np.abs(g["logits"]).max()), 1),
            "loss_abs_err": round(abs(float(loss.item()) - float(g["loss"])), 5),
            "token_grad_rel_l2": round(float(np.linalg.norm(gt - gr) / np.linalg.norm(gr)), 5)}


def build_model(dataset="modelnet40", head_type=HEAD_TYPE, precision=torch.bfloat16, model="ULIP_PointBERT", task="cls"):
    from ppt_amd import weights as W
    from ppt_amd.models import ULIP_models as M
    import contextlib
    import io
    args = SimpleNamespace(classnames=M.dataset_classnames(dataset), template_init='', class_name_position='middle',
                           num_learnable_prompt_tokens=32, gpu=torch.cuda.current_device(), task=task,
                           head_type=head_type, evaluate_3d=False, ulip2=False, synthetic_weights=True)
    with contextlib.redirect_stdout(io.StringIO()):
        m = getattr(M, model)(args)
    sd = {"ULIP_PointBERT": W.ulip_pointbert_state_dict, "ULIP_PN_MSG": W.ulip_pn2_msg_state_dict,
          "ULIP_PN_MLP": W.ulip_pn_mlp_state_dict, "ULIP_PointBERT_partseg": W.ulip_partseg_state_dict}[model](seed=0)
    if WEIGHTS == "ckpt_like":
        # checkpoint-LIKE magnitudes (ppt_amd.weights.checkpoint_like: LayerNorm gains log-normal around 1 with 5-10 x outlier
        # channels, 3 x larger weight matrices): what the mixed mode's load-time self-check decides on such weights -- text tower
        # moved to split16 products -- is part of what is timed (real ULIP / SLIP weights are not available)
        sd = W.checkpoint_like(sd, seed=0)
    m.load_state_dict(sd, strict=False)
    # the cached prompt embedding with the structure the reference's has (ULIP_models.py:102: a token_embedding lookup, so the
    # start token's row is the same in every prompt) -- which is what lets the text tower share the prompts' common prefix
    m.prompt_learner.embedding = W.synth_prompt_embedding_from_tokens(m.tokenized_prompts, seed=0)
    m.cuda()
    m.set_precision("fp32" if precision == torch.float32 else "mixed16")
    return m


def cpu_baseline(max_seconds=30.0):
    """Oracle train_step on the host cores, config C1 (B=8, N=1024, head_type 0)."""
    from oracle import oracle as O
    from ppt_amd import weights as W
    from ppt_amd.models import ULIP_models as M
    torch.set_num_threads(min(16, os.cpu_count() or 1))   # 16 was the fastest on the 256-core GPU-box host (tools/cpu_threads.py)
    names = M.dataset_classnames("modelnet40")
    ids, name_lengths = M.tokenize_prompts(names, 32)
    eot = ids.argmax(-1).numpy()
    sd = W.ulip_pointbert_state_dict(seed=0)
    emb = W.synth_prompt_embedding(len(names), seed=0)
    B = 8
    pc, start = W.synth_clouds(B, NPOINTS, seed=1234)
    labels = torch.from_numpy(np.random.default_rng(0).integers(0, len(names), size=(B,)))
    pc = torch.from_numpy(pc)
    O.train_step(sd, pc, labels, start, emb, name_lengths, eot, head_type=0)          # warm-up
    times = []
    t_all = time.time()
    while len(times) < 5 and (time.time() - t_all) < max_seconds:
        t0 = time.time()
        O.train_step(sd, pc, labels, start, emb, name_lengths, eot, head_type=0)
        times.append(time.time() - t0)
    med = float(np.median(times))
    return {"value": round(B / med, 3), "unit": "point-clouds/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle train_step (fwd+CE+bwd+AdamW), C1: batch 8 x 1024 pts, head_type 0, "
                      f"{len(times)} timed steps after 1 warm-up, median {med:.2f} s/step"}


def host_feed(tensors, steps):
    """PPT_BENCH_FEED=prefetch: the batch arrives from the HOST every step, as in main_cls.py:171-194 -- `steps` batches of pinned
    host tensors (what DataLoader(pin_memory=True) hands out; the same synthetic batch each time) through
    ppt_amd.data.DevicePrefetcher, whose copy-stream event lets the input-only stages run ahead WITHOUT the caller vouching
    for anything (Trainer.inputs_ready / eval_inputs_ready stay False)."""
    from ppt_amd.data import DevicePrefetcher
    host = tuple(t.cpu().pin_memory() for t in tensors)
    return iter(DevicePrefetcher((host for _ in range(steps)), depth=2))


FEED = os.environ.get("PPT_BENCH_FEED", "resident")
WEIGHTS = os.environ.get("PPT_BENCH_WEIGHTS", "synthetic")    # "ckpt_like": ppt_amd.weights.checkpoint_like magnitudes (secondary leg)
BENCH_MODE = os.environ.get("PPT_BENCH_MODE", "mixed16")      # "split16": the step in the split16 precision mode (secondary legs)


def secondary_runs():
    """Short runs of the other BASELINE configurations (and of validate()) as CHILD processes of this one, so that the driver's
    single `python bench.py` also times them (VERDICT r2 #4c): {name: {value, ms_per_step, ...}}.  Each child is this script with
    --config X --no-roofline --no-parity-mode --no-cpu-baseline; the parent's GPU work is finished and synchronised by now."""
    import subprocess
    out = {}
    # *_in_order: the same step WITHOUT the opt-in the headline runs with (Trainer.inputs_ready / eval_inputs_ready: the caller
    # vouches that the batch is resident, so the next step's FPS + kNN + tokenizer -- C5: the whole frozen backbone -- run under
    # the current one): what a caller gets who hands over batches that are merely queued on the stream
    in_order = {"PPT_GROUP_AHEAD": "0", "PPT_EVAL_AHEAD": "0"}
    # *_prefetch (round 5): batches copied from pinned host memory every step through ppt_amd.data.DevicePrefetcher, nothing
    # vouched for -- the unchanged caller's loop with its loader wrapped
    feed = {"PPT_BENCH_FEED": "prefetch"}
    # C5 (part segmentation) since round 6: the Trainer's first-batch gradient self-check (train.Trainer.calibrate_gradients) finds the
    # mixed mode's deep decoder gradients 0.11 rel-L2 away from the fp32-grade ones -- inherited from the frozen backbone's 16-bit
    # forward, no single stage fixes it -- and continues in split16: "C5" is what a caller gets by default,
    # "C5_mixed16" the same step with the check off (round 5's C5), and the feed variants of C5 compare like with like (check off)
    mixed = {"PPT_GRAD_CHECK": "off"}
    runs = [("C3", ["--config", "C3"], {}), ("C4", ["--config", "C4"], {}), ("C5", ["--config", "C5"], {}),
            ("C5_mixed16", ["--config", "C5"], mixed),
            ("C2_eval", ["--config", "C2", "--eval"], {}),
            ("C2_prefetch", ["--config", "C2"], feed), ("C3_prefetch", ["--config", "C3"], feed),
            ("C5_prefetch", ["--config", "C5"], dict(feed, **mixed)), ("C2_eval_prefetch", ["--config", "C2", "--eval"], feed),
            ("C2_in_order", ["--config", "C2"], in_order), ("C3_in_order", ["--config", "C3"], in_order),
            ("C5_in_order", ["--config", "C5"], dict(in_order, **mixed)), ("C2_eval_in_order", ["--config", "C2", "--eval"], in_order),
            # the split16 mode (fp32 storage, products from hi + lo half pairs: the fp32 mode's parity bounds) on the other configurations
            ("C3_split16", ["--config", "C3"], {"PPT_BENCH_MODE": "split16"}), ("C5_split16", ["--config", "C5"], {"PPT_BENCH_MODE": "split16"}),
            # the headline step on checkpoint-LIKE weight magnitudes (VERDICT r5 #2c): throughput of the mixed mode AFTER its load-time
            # self-check (text tower on split16 products there) + its parity triple against the reference's fixture on those weights
            ("C2_ckpt_like", ["--config", "C2", "--parity"], {"PPT_BENCH_WEIGHTS": "ckpt_like"})]
    for name, extra, env in runs:
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "30", "--warmup", "5", "--no-roofline", "--no-parity-mode",
               "--no-cpu-baseline", "--no-secondary"] + extra
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=ROOT, env=dict(os.environ, **env))
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            j = json.loads(line[-1])
            out[name] = {"metric": j["metric"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"],
                         "steps": j["steps"], "workload": j["config"]["workload"]}
            if "timing" in j["config"]:
                out[name]["ms_per_step_median"] = j["config"]["timing"]["ms_per_step_median"]
            if j["config"].get("gradient_self_check"):
                out[name]["gradient_self_check"] = j["config"]["gradient_self_check"]
                out[name]["dtype"] = j["dtype"]
            if "parity" in j:
                out[name]["parity"] = j["parity"]
                out[name]["operand_formats"] = j["config"]["operand_formats"]
        except Exception as e:            # a failed child must not cost the headline line
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def main_eval(a):
    """validate() (main_cls.py:237-299) throughput on one GPU: model.eval(), torch.no_grad(), logits = model(pc) for the
    same resident batch -- running-statistics BatchNorm (an affine prologue, no reduction kernels), no DropPath, the text
    features computed once and cached (ULIP_WITH_IMAGE._text_embed), both towers replayed from hipGraphs."""
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    from ppt_amd import graphs, weights as W
    graphs.shared_text_stream()
    graphs.shared_group_stream()
    cfg = CONFIGS[a.config]
    model = build_model(cfg["dataset"], cfg["head_type"], model=cfg.get("model", "ULIP_PointBERT"), task=cfg.get("task", "cls"))
    if BENCH_MODE == "split16":
        model.set_precision("split16")
    model.eval()
    # the synthetic batch is resident and complete before every call: the next batch's grouping / tokenizer stage may start
    # when forward() is called (ULIP_WITH_IMAGE.eval_inputs_ready; PPT_EVAL_AHEAD=0 for the in-order forward)
    model.eval_inputs_ready = os.environ.get("PPT_EVAL_AHEAD", "1") != "0" and FEED != "prefetch"
    B, N = cfg["batch"], cfg["npoints"]
    pc = torch.from_numpy(W.synth_clouds(B, N, seed=1234)[0]).cuda()
    extra = ()
    if cfg.get("task") == "partseg":
        extra = (torch.nn.functional.one_hot(torch.arange(B) % 16, 16).float().cuda(),)
    feed = host_feed((pc,), BURN_IN_STEPS + a.warmup + a.steps) if FEED == "prefetch" else None
    with torch.no_grad():
        for _ in range(BURN_IN_STEPS + a.warmup):
            logits = model(next(feed)[0] if feed else pc, *extra)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            logits = model(next(feed)[0] if feed else pc, *extra)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert torch.isfinite(logits).all()
    out = {"metric": METRICS[a.config].replace("fwd+bwd", "validate() forward"), "value": round(B * a.steps / dt, 2),
           "unit": "point-clouds/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "burn_in": BURN_IN_STEPS,
           "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f16" if set(operand_formats(cfg.get("model", "ULIP_PointBERT")).values()) <= {"f16", "f32"} else "f16/bf16",
           "data": "synthetic",
           "config": {"workload": cfg["name"] + ", eval-mode forward under no_grad (validate(), main_cls.py:237-299)", "feed": FEED,
                      "operand_formats": formats_of(model, cfg.get("model", "ULIP_PointBERT")),
                      "per_gpu_batch": B, "npoints": N, "parallelism": "dp1"}}
    print(json.dumps(out), flush=True)


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE in the environment): this parent -- which has made NO GPU call
    (importing torch does not initialise HIP) -- starts one child of this script per rank with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, as `python -m torch.distributed.run --nproc-per-node N` would (main_cls.py:47-49's launcher
    contract, utils/utils.py:104-143), relays rank 0's output (its JSON line stays the LAST line) and exits with the worst child's
    code.  Children are separate processes from the start: nothing here exec()s after touching the GPU."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, PPT_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    worst, deadline = 0, time.time() + float(os.environ.get("PPT_BENCH_LAUNCH_TIMEOUT", "1500"))
    live = set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                print(f"bench.py: rank {r} exited with code {rc}", file=sys.stderr)
                worst = worst or rc
        if live and (worst or time.time() > deadline):
            # a dead rank leaves the others waiting in the rendezvous / a collective: stop exactly the children started here
            for r in sorted(live):
                procs[r].kill()
            worst = worst or 124
        time.sleep(0.05)
    reader.join(timeout=10)
    out0 = (chunks[0] if chunks else "") or ""
    lines = out0.splitlines()
    js = [ln for ln in lines if ln.startswith("{")]
    for ln in lines:
        if not js or ln is not js[-1]:
            print(ln)
    if js:
        print(js[-1], flush=True)
    sys.exit(worst)


def main_dry(a):
    """PPT_BENCH_DRY=gloo: the launch / rendezvous / timing / JSON plumbing of a multi-rank run WITHOUT a GPU (tests/test_dp_cpu.py:
    the self-launcher under world size 2).  A "step" is one all-reduce of a head_type-0-sized gradient buffer (64 KiB) over gloo;
    the line says `"dry": true` and is not a measurement of anything."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if os.environ.get("PPT_BENCH_DRY_FAIL_RANK") == str(rank):     # (test hook: a rank that dies before the rendezvous)
        sys.exit(7)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=20))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    buf = torch.ones(16384)
    for _ in range(a.warmup):
        dist.all_reduce(buf.clone())
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        g = buf.clone()
        dist.all_reduce(g)
    dist.barrier()
    mine = time.perf_counter() - t0
    t = torch.tensor([mine], dtype=torch.float64)
    every = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(every, t)
    assert float(g[0]) == world
    dist.destroy_process_group()
    if rank == 0:
        per_rank = [1e3 * float(x) / a.steps for x in every]
        elapsed = max(float(x) for x in every)
        print(json.dumps({"metric": METRICS[a.config], "value": round(PER_GPU_BATCH * world * a.steps / elapsed, 2),
                          "unit": "point-clouds/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(1e3 * elapsed / a.steps, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "synthetic", "dry": True,
                          "config": {"workload": "DRY RUN (gloo, no GPU work): launcher / rendezvous / JSON plumbing only",
                                     "parallelism": f"dp{world}",
                                     "ms_per_step_per_rank": {"min": round(min(per_rank), 4), "max": round(max(per_rank), 4)}}}),
              flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)     # (the two-stream pipeline drains once per timed region: ~1.2 ms / K)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true")
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--eval", action="store_true", help="validate() throughput (main_cls.py:237-299): eval-mode forward under "
                    "no_grad, text features cached, no backward / optimizer")
    ap.add_argument("--parity", action="store_true", help="measure the parity triple against the golden fixture even when --no-parity-mode "
                    "skips the fp32 / split16 mode runs (the ckpt_like secondary leg)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short C3 / C4 / C5 / eval runs reported under `secondary`")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a.gpus, sys.argv[1:])        # (before ANY GPU call of this process)
    if os.environ.get("PPT_BENCH_DRY") == "gloo":
        return main_dry(a)
    if a.eval:
        return main_eval(a)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    from ppt_amd import graphs
    # before RCCL creates its streams: same hardware-queue position as at N = 1.  High priority where only the prompt trains
    # (the prompt side's backward -> AdamW -> forward chain is then what the head waits for; see graphs.shared_text_stream)
    c0_ = CONFIGS[a.config]
    graphs.shared_text_stream(priority=-1 if (c0_["head_type"] == 0 and c0_.get("model", "ULIP_PointBERT") == "ULIP_PointBERT") else 0)
    frozen_too = os.environ.get("PPT_GROUP_AHEAD_FROZEN", "1") != "0"     # also for a fully frozen PointBERT (C2): pays since round 3
    c_ = CONFIGS[a.config]
    group_ahead = GROUP_AHEAD and (frozen_too or (c_["head_type"] > 0 and c_.get("model", "ULIP_PointBERT") == "ULIP_PointBERT")
                                   or c_.get("model") in ("ULIP_PN_MSG", "ULIP_PN_MLP", "ULIP_PointBERT_partseg"))
    if group_ahead:                    # (only where Trainer uses it: an extra stream shifts the others' queue positions)
        graphs.shared_group_stream()
    force_dist = os.environ.get("PPT_FORCE_DIST") == "1"      # exercise the RCCL path with a single rank (dev aid)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from ppt_amd import ops, weights as W
    from ppt_amd.train import Trainer
    torch.manual_seed(1234 + rank)                                   # main_cls.py:39: seed + rank
    cfg = CONFIGS[a.config]
    global PER_GPU_BATCH, NPOINTS
    PER_GPU_BATCH, NPOINTS = cfg["batch"], cfg["npoints"]
    model = build_model(cfg["dataset"], cfg["head_type"], model=cfg.get("model", "ULIP_PointBERT"), task=cfg.get("task", "cls"))
    if BENCH_MODE == "split16":        # (secondary legs: the fp32-grade mode on the 16-bit matrix pipe)
        model.set_precision("split16")
    partseg = cfg.get("task") == "partseg"
    n_classes = len(model.prompt_learner.classnames)
    model.train()
    trainer = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=world > 1 or force_dist)
    # the synthetic batch is resident and complete before the first step: FPS + kNN of a step may start when it is called
    trainer.inputs_ready = group_ahead and FEED != "prefetch"
    trainer.group_ahead_when_frozen = frozen_too
    pc_np, _ = W.synth_clouds(PER_GPU_BATCH, NPOINTS, seed=1234 + rank)
    pc = torch.from_numpy(pc_np).cuda()
    lab_shape = (PER_GPU_BATCH, NPOINTS) if partseg else (PER_GPU_BATCH,)
    label = torch.from_numpy(np.random.default_rng(rank).integers(0, n_classes, size=lab_shape)).cuda()
    if partseg:                                    # main_partseg.py:210: model(pc, one-hot of the 16 shape categories)
        onehot = torch.nn.functional.one_hot(torch.arange(PER_GPU_BATCH) % 16, 16).float().cuda()
        trainer.extra_inputs = (onehot,)

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # burn-in before the W warm-up steps: on a fresh box the first ~100 ms of GPU work run at ramping clocks, and the first
    # two calls of every shape run eagerly and then capture their hipGraphs -- neither belongs in a W as small as 1
    # Round 6: the burn-in is at least BURN_IN_STEPS steps AND at least PPT_BENCH_BURN_IN_S seconds (default 1.0) of continuous GPU work:
    # 40 steps are 0.12 s, and the power state a fresh box's GPU reaches in that time is not the one it holds after a second
    # (profiles/r06_headline_repro.md); the count that ran is reported as "burn_in".
    burn_s = float(os.environ.get("PPT_BENCH_BURN_IN_S", "1.0"))
    max_burn = BURN_IN_STEPS + int(os.environ.get("PPT_BENCH_BURN_IN_MAX", "2000"))
    feed = host_feed((pc, label), max_burn + a.warmup + a.steps) if FEED == "prefetch" else None
    import gc
    gc_mode = os.environ.get("PPT_BENCH_GC", "freeze")
    # (under a process group every step carries a collective, so every rank must run the SAME number of steps: the burn-in is then a
    # fixed count -- what one second amounts to at this configuration's single-GPU rate would differ from rank to rank)
    fixed_burn = BURN_IN_STEPS + int(os.environ.get("PPT_BENCH_BURN_IN_DIST", "320")) if (world > 1 or force_dist) else None
    burned, tb = 0, time.perf_counter()
    while (burned < fixed_burn) if fixed_burn is not None else \
            (burned < BURN_IN_STEPS or (time.perf_counter() - tb < burn_s and burned < max_burn)):
        trainer.step(*(next(feed) if feed else (pc, label)))
        burned += 1
        if burned % 20 == 0:
            torch.cuda.current_stream().synchronize()        # (the host must not queue seconds of work ahead: the clock above is the GPU's)
        if burned == BURN_IN_STEPS and gc_mode != "raw":     # (every graph is captured, every cache filled: see below)
            gc.collect()
            gc.freeze()
            tb = time.perf_counter()                         # ... and the burn_s seconds of continuous work start AFTER that pause
    # The cyclic collector stays ON, as in any caller's loop; what is taken out of the timed region is the BACKLOG of the set-up:
    # ~10^6 container objects from building the model's modules, state dicts and graphs, which a generation-2 pass would walk in
    # the middle of a 60 ms region.  collect() + freeze() moves them to the permanent generation; the timed
    # steps' own garbage is collected as usual.  It runs ABOVE, after the first BURN_IN_STEPS steps -- not between the warm-up and
    # the timed region: the GPU idles while the collector walks the heap, and an idle of tens of milliseconds right before t0 costs
    # the first timed steps ~1 ms (measured: first step 4.7 instead of 4.25 ms, 9 960 instead of 10 210 clouds/s at K = 20).
    # PPT_BENCH_GC=0 also disables the collector for the region, PPT_BENCH_GC=raw leaves everything as it was in round 5.
    for _ in range(a.warmup):
        trainer.step(*(next(feed) if feed else (pc, label)))
    # K steps bracketed by barrier + synchronize (the contract's `value`); a HIP event on the caller's stream after every
    # step also gives the per-step times (hipEvents, median reported beside the mean)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    host_s = [0.0] * (a.steps + 1)
    if gc_mode == "0":
        gc.disable()
    gc_before = sum(st_["collections"] for st_ in gc.get_stats())
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    host_s[0] = t0
    for i in range(a.steps):
        loss, _ = trainer.step(*(next(feed) if feed else (pc, label)))
        marks[i + 1].record()
        host_s[i + 1] = time.perf_counter()
    trainer.finish()                         # (the deferred BatchNorm-buffer broadcast of a multi-rank run is timed too)
    barrier()
    elapsed = time.perf_counter() - t0
    gc_runs = sum(st_["collections"] for st_ in gc.get_stats()) - gc_before
    if gc_mode == "0":
        gc.enable()
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
    step_ms = sorted(per_step)
    host_ms = sorted(1e3 * (host_s[i + 1] - host_s[i]) for i in range(a.steps))
    if os.environ.get("PPT_BENCH_VERBOSE") == "1":
        print("per-step ms (sorted, top 5):", [round(x, 2) for x in step_ms[-5:]], file=sys.stderr)
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    # what the driver's record keeps is `config`: the distribution of the K timed steps goes there (GPU time between the events
    # the caller's stream records after each step; host time between the returns of step())
    timing = {"ms_per_step_median": round(median_ms, 3), "ms_per_step_min": round(step_ms[0], 3), "ms_per_step_max": round(step_ms[-1], 3),
              "slowest_steps": [[i, round(per_step[i], 3)] for i in sorted(range(a.steps), key=lambda i: -per_step[i])[:5]],
              "host_ms_per_step_median": round(host_ms[len(host_ms) // 2], 3), "host_ms_per_step_max": round(host_ms[-1], 3),
              "gc": gc_mode, "gc_collections_in_timed_region": gc_runs}
    if world > 1 or force_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank = [1e3 * float(x.item()) / a.steps for x in every]
        elapsed = max(float(x.item()) for x in every)
        timing["ms_per_step_per_rank"] = {"min": round(min(per_rank), 3), "max": round(max(per_rank), 3)}
        # the step's ONE collective, alone: the all-reduce of the flat gradient buffer (64 KiB at head_type 0) back to back on an
        # otherwise idle GPU, barrier-bracketed -- its latency over xGMI, which in the step overlaps the next point tower
        buf = trainer.sync.flat
        for _ in range(5):
            dist.all_reduce(buf.clone())
        barrier()
        ta = time.perf_counter()
        for _ in range(50):
            dist.all_reduce(buf.clone())
        barrier()
        timing["allreduce_us_alone"] = round(1e6 * (time.perf_counter() - ta) / 50, 1)
        timing["allreduce_bytes"] = buf.numel() * 4
    final_loss = loss.item()
    assert np.isfinite(final_loss), "loss is not finite"

    roof = None
    if rank == 0 and not a.no_roofline:
        # further passes over the same K steps with every launch of the hot kernels bracketed by HIP events.  These
        # passes run eagerly (no hipGraph replay: events cannot be recorded inside one) and the host, slowed by two
        # event records per launch, falls behind the GPU, so how much of the text stream happens to overlap the point
        # tower's GEMMs -- and with it their duration -- varies from pass to pass (3.0 .. 4.2 ms of GEMM time per step
        # seen on one box).  Three passes; the median one (by GEMM time) is reported.
        # (rank 0 only: the instrumented passes must not issue collectives the other ranks never join)
        trainer.distributed, trainer.bcast = False, None
        passes = []
        for _ in range(3):
            ops.profiler = ops.KernelProfiler()
            for _ in range(a.steps):
                trainer.step(pc, label)
            torch.cuda.synchronize()
            passes.append((ops.profiler.summary(), ops.profiler.by_kernel()))
        summ, detail = sorted(passes, key=lambda d: d[0]["gemm_bf16"]["ms"])[1]
        # an event pair around ANY launch also times the dispatch gaps on both sides of it; calibrate that on a
        # trivial kernel (1-element dtype conversion, ~1.5 us of execution) and take it off every bracket, so that
        # the per-launch figure is comparable with rocprofv3's kernel-only durations (profiles/)
        ops.profiler = ops.KernelProfiler()
        one = torch.zeros(1, device="cuda")
        blk_a = torch.zeros(65536, 512, device="cuda", dtype=torch.bfloat16)
        blk_w = torch.zeros(512, 512, device="cuda", dtype=torch.bfloat16)
        blk_o = torch.empty(65536, 512, device="cuda", dtype=torch.bfloat16)
        torch.cuda.synchronize()
        prof, ops.profiler = ops.profiler, None
        for _ in range(120):                # keep the GPU busy for the WHOLE calibration so that the host runs ahead
            ops.gemm(blk_a, blk_w, out=blk_o)
        ops.profiler = prof
        for _ in range(200):
            ops.profiler.begin("null", 0)
            ops.convert(one, torch.bfloat16)
            ops.profiler.end()
        torch.cuda.synchronize()
        brackets = sorted(st.elapsed_time(en) for _, st, en, _ in ops.profiler.records)
        # lower quartile, not the median: a bracket that the host issued late is longer than the dispatch gap, and
        # subtracting too much would overstate the rate (the per-launch figure is cross-checked against rocprofv3)
        overhead_ms = max(0.0, brackets[len(brackets) // 4] - 0.0015)
        ops.profiler = None
        g = summ["gemm_bf16"]
        g_ms = g["ms"] - overhead_ms * g["launches"]
        achieved = g["executed"] / (g_ms * 1e-3) / 1e12          # EXECUTED FLOPs; the model's figure beside it
        # HBM bytes per launch of the family: NOT measured in this run (PMC counters need rocprofv3 around the process) -- the
        # figure of the newest committed profile of this same command (two --pmc passes: FETCH_SIZE x2 + WRITE_SIZE), labelled
        # with its source; likewise the rocprofv3 kernel-trace fraction of that round beside the live event-bracket one
        traffic = traffic_source = rocprof_family = None
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
            tf = os.path.join(ROOT, "profiles", f"{rnd}_gemm_hbm_traffic.json")
            if a.config == "C2" and os.path.exists(tf):
                tj = json.load(open(tf))
                traffic = round(tj["hbm_bytes_per_launch"])
                traffic_source = f"profiles/{rnd}_gemm_hbm_traffic.json (rocprofv3 --pmc passes of this command, round {rnd[1:]}; not measured in this run)"
                rocprof_family = tj.get("rocprof_family")
                break
        roof = {"bound": "mfma", "kernel": "16-bit (fp16 / bf16) MFMA GEMM family (ppt_amd/csrc/gemm.hip, rowgemm.hip, lnlin.hip, mlp_fused3.hip, text_mlp.hip, mpn1/mpn3/mpn4.hip)",
                "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "frac_source": "live HIP-event brackets on the launch stream, dispatch gap subtracted (reads ~8 % above rocprofv3's kernel-only durations)",
                "rocprof_family": rocprof_family,
                "launches_per_step": g["launches"] // a.steps,
                "avg_launch_us": round(1e3 * g_ms / g["launches"], 2),
                "avg_bracket_us": round(1e3 * g["ms"] / g["launches"], 2), "event_overhead_us": round(1e3 * overhead_ms, 2),
                "algorithmic_gflop_per_launch": round(g["work"] / g["launches"] / 1e9, 3),
                "executed_gflop_per_launch": round(g["executed"] / g["launches"] / 1e9, 3),
                "achieved_model_flops": round(g["work"] / (g_ms * 1e-3) / 1e12, 2),
                "per_kernel_ms_per_step": {k: round((v["ms"] - overhead_ms * v["launches"]) / a.steps, 4)
                                           for k, v in summ.items()},
                "kernels": kernel_table(detail, overhead_ms, a.steps)}
    if world > 1 or force_dist:
        dist.barrier()

    if rank == 0:
        total = PER_GPU_BATCH * world * a.steps
        out = {"metric": METRICS[a.config],
               "value": round(total / elapsed, 2), "unit": "point-clouds/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "burn_in": burned, "ms_per_step": round(1e3 * elapsed / a.steps, 3),
               "ms_per_step_median": round(median_ms, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None,
               "dtype": "f32 as hi+lo f16 pairs" if (BENCH_MODE == "split16" or getattr(model, "precision_name", "") == "split16") else
                        ("f16" if set(operand_formats(cfg.get("model", "ULIP_PointBERT")).values()) <= {"f16", "f32"} else "f16/bf16"),
               "data": "synthetic",
               "config": {"workload": cfg["name"] + ", train-mode BN + DropPath, fwd + CE(ls 0.2) + bwd + AdamW"
                                      + (", checkpoint-LIKE weight magnitudes (weights.checkpoint_like)" if WEIGHTS == "ckpt_like" else ""), "feed": FEED,
                          "operand_formats": formats_of(model, cfg.get("model", "ULIP_PointBERT")),
                          "per_gpu_batch": PER_GPU_BATCH, "global_batch": PER_GPU_BATCH * world, "npoints": NPOINTS,
                          "classes": n_classes, "parallelism": f"dp{world}", "final_loss": round(final_loss, 4), "timing": timing,
                          "gradient_self_check": ({k: (round(v, 5) if isinstance(v, float) else v) for k, v in trainer.grad_calibration.items()}
                                                  if trainer.grad_calibration and trainer.grad_calibration.get("checked") else None)},
               "roofline": roof}
        if world == 1 and not force_dist and not a.no_parity_mode:
            out["parity_mode"] = parity_mode_rate(cfg, pc, label, trainer.extra_inputs, max(3, a.steps // 2), like=trainer)
            out["split16_mode"] = parity_mode_rate(cfg, pc, label, trainer.extra_inputs, max(3, a.steps // 2), mode="split16", like=trainer)
            if a.config == "C2":
                out["parity"] = measured_parity()
        elif world == 1 and a.parity and a.config == "C2":
            out["parity"] = measured_parity()
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if world == 1 and not force_dist and a.config == "C2" and not a.no_secondary:
            out["secondary"] = secondary_runs()
    if world > 1 or force_dist:
        dist.barrier()                       # rank 0 spends a few seconds more (roofline passes): tear down together
        dist.destroy_process_group()
    if rank == 0:
        # (after the teardown: RCCL writes its version banner / "Librccl path" to the C library's stdout, which is flushed at exit --
        # i.e. BEHIND anything Python printed.  Flush the C stream first, so that the JSON line stays the LAST line of the output.)
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
