"""Thin torch-tensor wrappers over the C ABI of libppt_hip.so (include/ppt_hip.h).

PyTorch is plumbing here: it owns device memory (caching allocator) and the stream; every
operation below is ONE call into a hand-written gfx950 kernel.  Tensors must be CUDA(HIP),
contiguous, of the stated dtype; wrappers allocate outputs and raise RuntimeError on any error
code.  There is no CPU path.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_NONE, ACT_QUICKGELU, ACT_RELU, A_AFFINE_RELU, A_CONV1, A_PLAIN,
                   PPT_BF16, PPT_F16, PPT_F32, GemmParams, RowGemmParams)
from . import _lib as _libmod  # noqa: F401

_DT = {torch.float32: PPT_F32, torch.bfloat16: PPT_BF16, torch.float16: PPT_F16}
_TORCH_DT = {PPT_F32: torch.float32, PPT_BF16: torch.bfloat16, PPT_F16: torch.float16}
HALF = (torch.bfloat16, torch.float16)        # the two 16-bit operand formats (same MFMA rate)


class KernelProfiler:
    """Optional per-launch timing with HIP events recorded on the launch stream (used by bench.py for
    the live roofline numbers).  Off by default: `ops.profiler = KernelProfiler()` turns it on."""

    def __init__(self):
        self.records = []          # (name, start_event, end_event, algorithmic work)
        self.kernels = []          # the kernel behind each record (a finer name than the family in `name`)
        self.executed = []         # FLOPs / bytes the launch really performs (differs from the model's where work is shared
                                   # -- the broadcast half of conv3 evaluated once per group -- or recomputed)

    def begin(self, name, work, kernel=None, executed=None):
        """work: the MODEL's algorithmic FLOPs (or bytes) of the layer(s) the launch stands for;
        executed: what the launch itself computes (default: the same)."""
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        self._cur = (name, ev, work, kernel or name, work if executed is None else executed)

    def end(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        name, st, work, kernel, executed = self._cur
        self.records.append((name, st, ev, work))
        self.kernels.append(kernel)
        self.executed.append(executed)

    def by_kernel(self):
        """-> {kernel: dict(family, launches, ms, work, executed)} (call after a device synchronise)."""
        out = {}
        for (name, st, en, work), kernel, ex in zip(self.records, self.kernels, self.executed):
            d = out.setdefault(kernel, dict(family=name, launches=0, ms=0.0, work=0.0, executed=0.0))
            d["launches"] += 1
            d["ms"] += st.elapsed_time(en)
            d["work"] += work
            d["executed"] += ex
        return out

    def summary(self):
        """-> {name: dict(launches, ms, work, executed)} (call after a device synchronise)."""
        out = {}
        for (name, st, en, work), ex in zip(self.records, self.executed):
            d = out.setdefault(name, dict(launches=0, ms=0.0, work=0.0, executed=0.0))
            d["launches"] += 1
            d["ms"] += st.elapsed_time(en)
            d["work"] += work
            d["executed"] += ex
        return out


profiler = None


def dtype_code(t):
    return _DT[t.dtype if isinstance(t, torch.Tensor) else t]


_raw_stream = torch._C._cuda_getCurrentRawStream      # current HIP stream handle without building a Stream object


def _stream():
    return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(t, dtype=None, name="tensor"):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA/HIP tensor (ppt_amd has no CPU path)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")


# ---------------------------------------------------------------------------------------------
def fps(xyz, M, start):
    """H1.  xyz [B,N,3] f32, start [B] i64 -> (idx [B,M] i64, center [B,M,3] f32)."""
    _chk(xyz, torch.float32, "xyz"); _chk(start, torch.int64, "start")
    B, N, _ = xyz.shape
    idx = torch.empty((B, M), dtype=torch.int64, device=xyz.device)
    ctr = torch.empty((B, M, 3), dtype=torch.float32, device=xyz.device)
    if profiler is not None:
        profiler.begin("fps", float(B) * (12 * N + 8 * M))          # algorithmic HBM bytes
    _lib.check(_lib.lib().ppt_fps_f32(_p(xyz), B, N, M, _p(start), _p(idx), _p(ctr), _stream()), "ppt_fps_f32")
    if profiler is not None:
        profiler.end()
    return idx, ctr


def knn_group(xyz, center, k, want_idx=True, want_nbhd=True, want_dist=False):
    """H2.  xyz [B,N,3], center [B,G,3] -> (nbr_idx [B,G,k] i64, neighborhood [B,G,k,3] f32[, dist [B,G,k] f32])."""
    _chk(xyz, torch.float32, "xyz"); _chk(center, torch.float32, "center")
    B, N, _ = xyz.shape
    G = center.shape[1]
    idx = torch.empty((B, G, k), dtype=torch.int64, device=xyz.device) if want_idx else None
    nb = torch.empty((B, G, k, 3), dtype=torch.float32, device=xyz.device) if want_nbhd else None
    if profiler is not None:
        profiler.begin("knn_group", float(B) * (12 * N + 12 * G + 8 * G * k + 12 * G * k))
    nd = torch.empty((B, G, k), dtype=torch.float32, device=xyz.device) if want_dist else None
    _lib.check(_lib.lib().ppt_knn_group_f32(_p(xyz), _p(center), B, N, G, k, _p(idx), _p(nb), _p(nd), _stream()),
               "ppt_knn_group_f32")
    if profiler is not None:
        profiler.end()
    return (idx, nb, nd) if want_dist else (idx, nb)


def ball_query_multi(xyz, center, queries, want_grouped=False):
    """queries = [(radius, K), ...] (1..3) around the same centres, one pass over the cloud (ppt_ball_query_multi_f32)
    -> [(idx [B,S,K] i64, grouped_xyz [B,S,K,3] | None), ...]; same results as ball_query per pair."""
    _chk(xyz, torch.float32, "xyz"); _chk(center, torch.float32, "center")
    import numpy as np
    B, N, _ = xyz.shape
    S = center.shape[1]
    q = _lib.BallMulti()
    q.n = len(queries)
    outs, bytes_ = [], 12 * N + 12 * S
    for j, (r, K) in enumerate(queries):
        idx = torch.empty((B, S, K), dtype=torch.int64, device=xyz.device)
        g = torch.empty((B, S, K, 3), dtype=torch.float32, device=xyz.device) if want_grouped else None
        q.r2[j], q.K[j] = float(np.float32(r * r)), K
        q.idx[j], q.gxyz[j] = idx.data_ptr(), (g.data_ptr() if g is not None else None)
        outs.append((idx, g))
        bytes_ += 8 * S * K + (12 * S * K if want_grouped else 0)
    if profiler is not None:
        profiler.begin("ball_query", float(B) * bytes_)
    _lib.check(_lib.lib().ppt_ball_query_multi_f32(_p(xyz), _p(center), B, N, S, ctypes.byref(q), _stream()), "ppt_ball_query_multi_f32")
    if profiler is not None:
        profiler.end()
    return outs


def square_distance(src, dst):
    """dvae.py:130-149: src [B,S,3], dst [B,N,3] f32 -> [B,S,N] f32 with the reference's rounding sequence."""
    _chk(src, torch.float32, "src"); _chk(dst, torch.float32, "dst")
    B, S, _ = src.shape
    N = dst.shape[1]
    out = torch.empty((B, S, N), dtype=torch.float32, device=src.device)
    _lib.check(_lib.lib().ppt_square_distance_f32(_p(src), _p(dst), B, S, N, _p(out), _stream()), "ppt_square_distance_f32")
    return out


def ball_query(xyz, center, radius, K, want_grouped=False):
    """H7.  -> idx [B,S,K] i64 (query_ball_point semantics) [, grouped_xyz [B,S,K,3] = xyz[idx] - center]."""
    _chk(xyz, torch.float32, "xyz"); _chk(center, torch.float32, "center")
    B, N, _ = xyz.shape
    S = center.shape[1]
    idx = torch.empty((B, S, K), dtype=torch.int64, device=xyz.device)
    import numpy as np
    r2 = float(np.float32(radius * radius))
    g = torch.empty((B, S, K, 3), dtype=torch.float32, device=xyz.device) if want_grouped else None
    if profiler is not None:                    # algorithmic HBM bytes per SA branch: cloud + centres + indices (+ grouped xyz)
        profiler.begin("ball_query", float(B) * (12 * N + 12 * S + 8 * S * K + (12 * S * K if want_grouped else 0)))
    _lib.check(_lib.lib().ppt_ball_query_f32(_p(xyz), _p(center), B, N, S, r2, K, _p(idx), _p(g), _stream()),
               "ppt_ball_query_f32")
    if profiler is not None:
        profiler.end()
    return (idx, g) if want_grouped else idx


# ---------------------------------------------------------------------------------------------
# split16 (ppt_gemm_params.split16): the powers of two the A / B operand values are multiplied by before they are split into
# hi + lo halves.  A: activations and S-scaled gradients (ppt_amd/gradscale.py) -- typical magnitude 1e-3 .. 1e2, left where they
# are; B: weights (typically 0.01 .. 0.1, |w| < 4 095 required) -- x 16 puts them where half keeps all of hi + lo's 22 bits.
SPLIT16_POW2 = (int(os.environ.get("PPT_SPLIT16_A_POW2", "0")), int(os.environ.get("PPT_SPLIT16_B_POW2", "4")))
# Process-wide on purpose (not thread-local): a node's backward runs on autograd's device thread.  The last model that fetched its
# WeightCache decides -- a model's backward follows its forward in every loop of the reference (main_cls.py:194-198).
_SPLIT16 = False
ATTN_SPLIT16 = os.environ.get("PPT_ATTN_SPLIT16", "1") != "0"      # 0: the split16 mode keeps the fp32 VALU attention forward


def set_split16(on, pow2=None):
    """fp32-operand ops.gemm calls that do not say `split=` themselves use the split16 products from now on (True) or the fp32
    MFMA (False).  pow2: the (A, B) pre-scales of the model that says so (ULIP_WITH_IMAGE.split_pow2: its weight range decides B's,
    _fit_split16_range) -- every model re-asserts its OWN pair with the flag whenever it fetches its WeightCache, so one model's
    fit no longer changes another's."""
    global _SPLIT16, SPLIT16_POW2
    _SPLIT16 = bool(on)
    if on and pow2 is not None:
        SPLIT16_POW2 = (int(pow2[0]), int(pow2[1]))


_SPLIT_OVERFLOW = {}


def split16_overflow_counter(device=None):
    """The device word (uint32, one per GPU, process-wide, never re-allocated: captured hipGraphs keep its address) to which every
    split16 GEMM wave that SATURATED a finite operand beyond IEEE half's range adds 1 (ppt_gemm_params.split_overflow).  A counter,
    not a flag: readers compare with the value they saw last (health.Monitor), nothing ever clears it, so a read on one stream
    cannot race with an add from another."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev not in _SPLIT_OVERFLOW:
        _SPLIT_OVERFLOW[dev] = torch.zeros((1,), dtype=torch.int32, device=torch.device("cuda", dev))
    return _SPLIT_OVERFLOW[dev]


def split16_enabled():
    return _SPLIT16


def gemm(A, B, *, out=None, out_dtype=None, M=None, bias=None, act=ACT_NONE, dact_pre=None,
         group_add=None, group_rows=0, row_scale=None, row_scale_rows=0, residual=None,
         residual2=None, out2=None, out2_pre=False, col_stats=None, pool_max=None, pool_min=None, pool_rows=0,
         batch=1, strideA=0, strideB=0, strideC=0,
         a_mode=A_PLAIN, a_scale=None, a_shift=None, pts=None, w1=None, b1=None, want_out=True, algo_k=None, core=None, split=None):
    """C[M,N] = epilogue(prologue(A)[M,K] @ B[N,K]^T) -- see struct ppt_gemm_params.
    A [M,K] (or None with a_mode=A_CONV1 and pts [M,3]); B [N,K]; 2-D, last-dim contiguous
    (row stride may exceed K).  Returns out (or None when want_out=False).
    split (fp32 operands only): True / (a_pow2, b_pow2) multiplies hi + lo half pairs on the 16-bit matrix pipe (split16,
    ppt_gemm_params.split16); False: the fp32 MFMA; None: what set_split16() last said (a model in the "split16" precision mode
    says it whenever it fetches its WeightCache, in forward and in backward)."""
    p = GemmParams()
    N, K = B.shape
    if a_mode == A_CONV1:
        M = pts.shape[0]
        dev = pts.device
    else:
        M = A.shape[0] if M is None else M
        dev = A.device
        assert A.shape[1] == K and A.stride(1) == 1 and A.dtype == B.dtype
        p.A, p.lda = _p(A), A.stride(0)
    assert B.stride(1) == 1
    p.B, p.ldb = _p(B), B.stride(0)
    p.M, p.N, p.K = M, N, K
    p.dtype = dtype_code(B)
    if out is None and want_out:
        out = torch.empty((M, N), dtype=out_dtype or B.dtype, device=dev)
    if out is not None:
        assert out.stride(1) == 1
        p.C, p.ldc, p.c_dtype = _p(out), out.stride(0), dtype_code(out)
    p.a_mode = a_mode
    p.a_scale, p.a_shift, p.pts, p.w1, p.b1 = _p(a_scale), _p(a_shift), _p(pts), _p(w1), _p(b1)
    p.bias = _p(bias)
    p.group_add, p.group_rows = _p(group_add), group_rows
    p.act = act
    if dact_pre is not None:
        assert dact_pre.dtype == B.dtype and dact_pre.stride(1) == 1
        p.dact_pre, p.ld_dact = _p(dact_pre), dact_pre.stride(0)
    p.row_scale, p.row_scale_rows = _p(row_scale), row_scale_rows
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.stride(-1) == 1
        p.residual, p.ld_res = _p(residual), residual.stride(-2)
    if residual2 is not None:
        assert residual2.dtype == torch.float32 and residual2.stride(-1) == 1
        p.residual2, p.ld_res2 = _p(residual2), residual2.stride(-2)
    if out2 is not None:
        p.C2, p.ldc2, p.c2_dtype, p.c2_pre = _p(out2), out2.stride(-2), dtype_code(out2), int(out2_pre)
    if col_stats is not None:
        p.col_sum, p.col_sqsum = _p(col_stats[0]), _p(col_stats[1])
    if pool_max is not None:
        p.pool_max, p.pool_dtype, p.pool_rows = _p(pool_max), dtype_code(pool_max), pool_rows
        p.pool_min = _p(pool_min)
    p.batch, p.strideA, p.strideB, p.strideC = batch, strideA, strideB, strideC
    if split is None:
        split = _SPLIT16
    if split and p.dtype == PPT_F32:
        p.split16 = 2 if core == "tiles" else 1          # ("tiles": keep the launch off the 256 x 128 split kernel -- A/B, tests)
        p.split_a_pow2, p.split_b_pow2 = split if isinstance(split, tuple) else SPLIT16_POW2
        if not torch.cuda.is_current_stream_capturing() or dev.index in _SPLIT_OVERFLOW:
            p.split_overflow = _p(split16_overflow_counter(dev))      # (first use outside a capture: the word is allocated there)
    if profiler is not None:
        kk = K if algo_k is None else algo_k
        # ("bf16" names the 16-bit MFMA family of the roofline table: bf16 and fp16 operands run at the same rate)
        profiler.begin("gemm_" + ("bf16" if p.dtype != PPT_F32 else "f32"), 2.0 * M * N * kk * max(1, batch),
                       "ppt_gemm " + ({PPT_BF16: "bf16", PPT_F16: "f16"}.get(p.dtype, "f32")) + (" (A-prologue)" if a_mode != A_PLAIN else ""),
                       executed=2.0 * M * N * K * max(1, batch))
    if core == "256":                  # the 256-row macro-tile core, explicitly (tests / tools; ppt_gemm picks it by itself)
        _lib.check(_lib.lib().ppt_gemm256(ctypes.byref(p), _stream()), "ppt_gemm256")
    else:
        _lib.check(_lib.lib().ppt_gemm(ctypes.byref(p), _stream()), "ppt_gemm")
    if profiler is not None:
        profiler.end()
    if probe is not None:
        _probe("gemm", out); _probe("gemm (second output)", out2)
    return out


class wave_priority:
    """with ops.wave_priority(1): every launch made inside raises its waves' issue priority (ppt_set_wave_priority; the
    kernels of the prompt chain take it as an argument, so a hipGraph captured inside keeps it)."""

    def __init__(self, prio):
        self.prio = int(bool(prio))

    def __enter__(self):
        self.old = _lib.lib().ppt_get_wave_priority()
        _lib.lib().ppt_set_wave_priority(self.prio)

    def __exit__(self, *exc):
        _lib.lib().ppt_set_wave_priority(self.old)


def cross_entropy_rows(logits, labels, smoothing, ignore_index=-100):
    """(loss 0-d, dlogits [R,C], scale 0-d) of nn.CrossEntropyLoss(label_smoothing=smoothing) with mean reduction
    (ppt_cross_entropy_rows).  Rows whose label == ignore_index (outside [0, C)) are ignored as ATen ignores them; any other label
    outside [0, C) makes the loss NaN (ATen: device assert).  dlogits is scaled by 1 / R and `scale` = R / counted rows (1.0 when
    none is ignored) completes it."""
    assert not (0 <= ignore_index < logits.shape[1]), "ignore_index inside [0, C) is not covered by ppt_cross_entropy_rows"
    _chk(logits, torch.float32, "logits"); _chk(labels, torch.int64, "labels")
    R, C = logits.shape
    out = torch.empty((2,), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    partial = torch.empty((2 * ((R + 127) // 128),), dtype=torch.float32, device=logits.device)
    _lib.check(_lib.lib().ppt_cross_entropy_rows(_p(logits), _p(labels), float(smoothing), R, C, int(ignore_index), _p(out), _p(dlogits),
                                                 _p(partial), _stream()), "ppt_cross_entropy_rows")
    return out[0], dlogits, out[1]


class persistent_occupancy:
    """with ops.persistent_occupancy(60): the persistent point-tower kernels launched inside size their grids for 60 % of the
    CUs (ppt_set_persistent_occupancy) -- room for the prompt chain on the other stream."""

    def __init__(self, percent):
        self.percent = int(percent)

    def __enter__(self):
        self.old = _lib.lib().ppt_get_persistent_occupancy()
        _lib.lib().ppt_set_persistent_occupancy(self.percent)

    def __exit__(self, *exc):
        _lib.lib().ppt_set_persistent_occupancy(self.old)


def get_persistent_occupancy():
    return _lib.lib().ppt_get_persistent_occupancy()


def rows_matmul(a, w_kn, alpha=1.0):
    """out [M,N] = alpha * a [M,K] @ w_kn [K,N], fp32, for a few rows (ppt_rows_matmul_f32) -- None when the shape is not covered."""
    _chk(a, torch.float32, "a"); _chk(w_kn, torch.float32, "w_kn")
    M, K = a.shape
    N = w_kn.shape[1]
    if K > 1536 or K % 32 or w_kn.shape[0] != K:
        return None
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    if profiler is not None:
        profiler.begin("gemm_f32", 2.0 * M * N * K, "ppt_rows_matmul_f32")
    _lib.check(_lib.lib().ppt_rows_matmul_f32(_p(a), _p(w_kn), M, K, N, float(alpha), _p(out), _stream()), "ppt_rows_matmul_f32")
    if profiler is not None:
        profiler.end()
    return out


# Which schedule of the fused frozen-block kernel runs (supported switch): 3 = csrc/mlp_fused3.hip (round 6: 256-unit
# slabs, GELU under fc2's MFMAs, residual kept in the accumulators), 2 = csrc/mlp_fused.hip (rounds 2-5).  The fragment-ordered
# weight copies differ, so vit_mlp_retile tags what it returns and vit_mlp follows the tag.
VIT_MLP_VARIANT = int(os.environ.get("PPT_VIT_MLP", "3"))


def vit_mlp_retile(w1, w2, variant=None):
    """(w1 [1536,384], w2 [384,1536]) 16-bit -> the fragment-ordered copies the fused kernel reads (ppt_vit_mlp_retile for
    variant 2, ppt_vit_mlp3_retile for variant 3; default VIT_MLP_VARIANT)."""
    assert w1.dtype in HALF and w2.dtype == w1.dtype
    _chk(w1, w1.dtype, "w1"); _chk(w2, w1.dtype, "w2")
    assert tuple(w1.shape) == (1536, 384) and tuple(w2.shape) == (384, 1536)
    variant = VIT_MLP_VARIANT if variant is None else int(variant)
    w1t, w2t = torch.empty_like(w1), torch.empty_like(w2)
    fn = _lib.lib().ppt_vit_mlp3_retile if variant == 3 else _lib.lib().ppt_vit_mlp_retile
    _lib.check(fn(_p(w1), _p(w2), _p(w1t), _p(w2t), _stream()), "ppt_vit_mlp_retile")
    w1t.ppt_variant = w2t.ppt_variant = variant          # (a plain attribute, like the copy event of data/prefetch.py)
    return w1t, w2t


def lnlin_retile(w):
    """w [N, 384] 16-bit (N a multiple of 384) -> the fragment-ordered copy ppt_lnlin reads."""
    assert w.dtype in HALF and w.dim() == 2 and w.shape[1] == 384 and w.shape[0] % 384 == 0
    _chk(w, w.dtype, "w")
    wt = torch.empty_like(w)
    _lib.check(_lib.lib().ppt_lnlin_retile(_p(w), _p(wt), w.shape[0], _stream()), "ppt_lnlin_retile")
    return wt


def lnlin(x, wt, ln, *, bias=None, ln_eps=1e-5, out=None):
    """LayerNorm(x [M, 384] f32; ln = (weight, bias)) @ W^T (+ bias) -> [M, N] in the weight's 16-bit dtype (csrc/lnlin.hip: rows
    stationary, weight streamed; wt = lnlin_retile(W))."""
    _chk(x, torch.float32, "x")
    M, K = x.shape
    N = wt.shape[0]
    assert K == 384 and wt.dtype in HALF
    out = torch.empty((M, N), dtype=wt.dtype, device=x.device) if out is None else out
    p = _lib.LnLinParams()
    p.x, p.W, p.C, p.ln_w, p.ln_b, p.ln_eps, p.bias = _p(x), _p(wt), _p(out), _p(ln[0]), _p(ln[1]), float(ln_eps), _p(bias)
    p.M, p.N, p.K, p.dtype = M, N, K, dtype_code(wt)
    if profiler is not None:
        profiler.begin("gemm_bf16", 2.0 * M * N * K, "ppt_lnlin (LayerNorm prologue)")
    _lib.check(_lib.lib().ppt_lnlin(ctypes.byref(p), _stream()), "ppt_lnlin")
    if profiler is not None:
        profiler.end()
    return out


def text_mlp_retile(w1, w2):
    """(w1 [2048, 512], w2 [512, 2048]) 16-bit row-major -> the fragment-ordered copies ppt_text_mlp_pair reads.  Forward:
    (c_fc.weight, c_proj.weight); backward: (c_proj.weight^T, c_fc.weight^T) -- the same shapes."""
    assert w1.dtype in HALF and w2.dtype == w1.dtype and tuple(w1.shape) == (2048, 512) and tuple(w2.shape) == (512, 2048)
    _chk(w1, w1.dtype, "w1"); _chk(w2, w1.dtype, "w2")
    w1t, w2t = torch.empty_like(w1), torch.empty_like(w2)
    _lib.check(_lib.lib().ppt_text_mlp_retile(_p(w1), _p(w2), _p(w1t), _p(w2t), _stream()), "ppt_text_mlp_retile")
    return w1t, w2t


def text_mlp_retile_split(w1, w2, b_pow2=None):
    """(w1 [2048, 512], w2 [512, 2048]) fp32 row-major -> the fragment-ordered hi + lo half copies (uint8 [4 MB] each) of the split16
    form of ppt_text_mlp_pair (csrc/text_mlp_split.hip), multiplied by 2^b_pow2 (default: the current SPLIT16_POW2's B scale).  The
    copies remember the scale they were made with (`.ppt_b_pow2`)."""
    assert w1.dtype == torch.float32 and w2.dtype == torch.float32 and tuple(w1.shape) == (2048, 512) and tuple(w2.shape) == (512, 2048)
    _chk(w1, torch.float32, "w1"); _chk(w2, torch.float32, "w2")
    b = SPLIT16_POW2[1] if b_pow2 is None else int(b_pow2)
    w1t = torch.empty((2048 * 512 * 4,), dtype=torch.uint8, device=w1.device)
    w2t = torch.empty((2048 * 512 * 4,), dtype=torch.uint8, device=w1.device)
    _lib.check(_lib.lib().ppt_text_mlp_retile_split(_p(w1), _p(w2), _p(w1t), _p(w2t), b, _stream()), "ppt_text_mlp_retile_split")
    w1t.ppt_b_pow2 = w2t.ppt_b_pow2 = b
    return w1t, w2t


def text_mlp_pair_split(a, w1t, w2t, *, bias=None, pre=None, backward=False, a_pow2=None, ln=None, ln_eps=1e-5, save_stats=False):
    """text_mlp_pair on fp32 operands multiplied as hi + lo half pairs (split16; csrc/text_mlp_split.hip): a [M, 512] f32, w1t / w2t
    from text_mlp_retile_split, pre [M, 2048] f32 -> the eight slices' partial products [8, M, 512] f32."""
    M = a.shape[0]
    assert a.dim() == 2 and a.shape[1] == 512 and a.stride(1) == 1 and a.dtype == torch.float32 and w1t.dtype == torch.uint8
    parts = torch.empty((8, M, 512), dtype=torch.float32, device=a.device)
    p = _lib.TextMlpParams()
    p.A, p.lda, p.W1, p.W2, p.b1, p.pre, p.parts = _p(a), a.stride(0), _p(w1t), _p(w2t), _p(bias), _p(pre), _p(parts)
    p.M, p.D, p.hidden, p.mode, p.dtype = M, 512, 2048, int(bool(backward)), PPT_F32
    p.split_a_pow2 = SPLIT16_POW2[0] if a_pow2 is None else int(a_pow2)
    p.split_b_pow2 = int(w1t.ppt_b_pow2)
    if not torch.cuda.is_current_stream_capturing() or a.device.index in _SPLIT_OVERFLOW:
        p.split_overflow = _p(split16_overflow_counter(a.device))
    if pre is not None:
        assert pre.dtype == torch.float32 and tuple(pre.shape) == (M, 2048) and pre.is_contiguous()
    mean = rstd = None
    if ln is not None:              # (forward: `a` is the residual stream, ln_2 applied while the rows are staged; -> (parts, mean, rstd))
        assert not backward
        _chk(ln[0], torch.float32, "ln weight"); _chk(ln[1], torch.float32, "ln bias")
        if save_stats:
            mean = torch.empty((M,), dtype=torch.float32, device=a.device)
            rstd = torch.empty((M,), dtype=torch.float32, device=a.device)
        p.ln_w, p.ln_b, p.ln_eps, p.ln_mean, p.ln_rstd = _p(ln[0]), _p(ln[1]), float(ln_eps), _p(mean), _p(rstd)
    if profiler is not None:
        profiler.begin("gemm_f32", 4.0 * M * 512 * 2048, "ppt_text_mlp_pair split16 (" + ("backward" if backward else "forward") + ")")
    _lib.check(_lib.lib().ppt_text_mlp_pair(ctypes.byref(p), _stream()), "ppt_text_mlp_pair (split16)")
    if profiler is not None:
        profiler.end()
    return (parts, mean, rstd) if ln is not None else parts


def text_lin_retile_split(w, b_pow2=None):
    """w [N, K] fp32 row-major (N a multiple of 256, K of 512) -> the fragment-ordered hi + lo half copy text_lin_split reads (uint8,
    N * K * 4 bytes), multiplied by 2^b_pow2 (default: the current SPLIT16_POW2's B scale); remembers (N, K, b_pow2)."""
    assert w.dtype == torch.float32 and w.dim() == 2
    _chk(w, torch.float32, "w")
    N, K = w.shape
    b = SPLIT16_POW2[1] if b_pow2 is None else int(b_pow2)
    wt = torch.empty((N * K * 4,), dtype=torch.uint8, device=w.device)
    _lib.check(_lib.lib().ppt_text_lin_retile_split(_p(w), _p(wt), N, K, b, _stream()), "ppt_text_lin_retile_split")
    wt.ppt_shape, wt.ppt_b_pow2 = (N, K), b
    return wt


def text_lin_split(a, wt, *, bias=None, residual=None, out=None, a_pow2=None):
    """a [M, K] fp32 @ W^T (+ bias) (+ residual) on split16 products, rows stationary (csrc/text_lin_split.hip; wt =
    text_lin_retile_split(W)) -> out [M, N] fp32; K > 512: the K / 512 partial products [K / 512, M, N] (no bias / residual)."""
    N, K = wt.ppt_shape
    M = a.shape[0]
    assert a.dtype == torch.float32 and a.dim() == 2 and a.shape[1] == K and a.stride(1) == 1
    kc = K // 512
    if kc > 1:
        assert bias is None and residual is None and out is None
        out = torch.empty((kc, M, N), dtype=torch.float32, device=a.device)
    elif out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    p = _lib.TextLinParams()
    p.A, p.lda, p.W, p.bias = _p(a), a.stride(0), _p(wt), _p(bias)
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.stride(-1) == 1
        p.residual, p.ld_res = _p(residual), residual.stride(-2)
    p.C, p.ldc = _p(out), (N if kc > 1 else out.stride(0))
    p.M, p.N, p.K = M, N, K
    p.split_a_pow2 = SPLIT16_POW2[0] if a_pow2 is None else int(a_pow2)
    p.split_b_pow2 = int(wt.ppt_b_pow2)
    if not torch.cuda.is_current_stream_capturing() or a.device.index in _SPLIT_OVERFLOW:
        p.split_overflow = _p(split16_overflow_counter(a.device))
    if profiler is not None:
        profiler.begin("gemm_f32", 2.0 * M * N * K, "ppt_text_lin_split")
    _lib.check(_lib.lib().ppt_text_lin_split(ctypes.byref(p), _stream()), "ppt_text_lin_split")
    if profiler is not None:
        profiler.end()
    return out


def text_mlp_pair(a, w1t, w2t, *, bias=None, pre=None, backward=False, ln=None, ln_eps=1e-5, save_stats=False):
    """The MLP half of a CLIP text layer in one launch (csrc/text_mlp.hip) -> the eight slices' partial products [8, M, 512] f32.
    forward: QuickGELU(a w1^T + bias) w2^T, `pre` (optional, [M, 2048] 16-bit) receives the pre-activation; backward=True:
    ((a w1^T) * QuickGELU'(pre)) w2^T with w1 / w2 the transposed weights' tiled copies.  The caller's LayerNorm sums the slices
    (layernorm_fwd_sum / layernorm_bwd_sum).
    ln = (weight, bias) (forward only): `a` is the FP32 residual stream and the LayerNorm in front of the branch (ln_2) is applied while
    the rows are staged; save_stats -> also returns (mean, rstd) [M] for the LayerNorm backward: (parts, mean, rstd)."""
    M = a.shape[0]
    assert a.dim() == 2 and a.shape[1] == 512 and a.stride(1) == 1 and w1t.dtype in HALF and w2t.dtype == w1t.dtype
    if ln is None:
        assert a.dtype == w1t.dtype
    else:
        assert a.dtype == torch.float32 and not backward
    parts = torch.empty((8, M, 512), dtype=torch.float32, device=a.device)
    p = _lib.TextMlpParams()
    p.A, p.lda, p.W1, p.W2, p.b1, p.pre, p.parts = _p(a), a.stride(0), _p(w1t), _p(w2t), _p(bias), _p(pre), _p(parts)
    p.M, p.D, p.hidden, p.mode, p.dtype = M, 512, 2048, int(bool(backward)), dtype_code(w1t)
    mean = rstd = None
    if ln is not None:
        _chk(ln[0], torch.float32, "ln weight"); _chk(ln[1], torch.float32, "ln bias")
        if save_stats:
            mean = torch.empty((M,), dtype=torch.float32, device=a.device)
            rstd = torch.empty((M,), dtype=torch.float32, device=a.device)
        p.ln_w, p.ln_b, p.ln_eps, p.ln_mean, p.ln_rstd = _p(ln[0]), _p(ln[1]), float(ln_eps), _p(mean), _p(rstd)
    if pre is not None:
        assert pre.dtype == w1t.dtype and tuple(pre.shape) == (M, 2048) and pre.is_contiguous()
    if profiler is not None:
        profiler.begin("gemm_bf16", 4.0 * M * 512 * 2048, "ppt_text_mlp_pair (" + ("backward" if backward else "forward") + ")")
    _lib.check(_lib.lib().ppt_text_mlp_pair(ctypes.byref(p), _stream()), "ppt_text_mlp_pair")
    if profiler is not None:
        profiler.end()
    return (parts, mean, rstd) if ln is not None else parts


def vit_proj_retile(wp):
    """attn.proj.weight [384,384] (16-bit) -> the fragment-ordered copy the fused proj prologue of ppt_vit_mlp_bf16 reads."""
    assert wp.dtype in HALF and tuple(wp.shape) == (384, 384)
    _chk(wp, wp.dtype, "wp")
    wpt = torch.empty_like(wp)
    _lib.check(_lib.lib().ppt_vit_proj_retile(_p(wp), _p(wpt), _stream()), "ppt_vit_proj_retile")
    return wpt


def vit_mlp(x, w1, b1, w2, b2, ln, *, out=None, ln_eps=1e-5, row_scale=None, row_scale_rows=0, residual2=None, workgroups=0, proj=None):
    """ppt_vit_mlp_bf16 (csrc/mlp_fused.hip): out = x + row_scale * (GELU(LN(x) w1^T + b1) w2^T + b2) (+ residual2), x [M,384]
    f32, w1 / w2: the fragment-ordered bf16 weights of vit_mlp_retile; out defaults to x (in place).
    proj = (a [M,384] 16-bit, wp_tiled (vit_proj_retile), bias | None, row_scale1 | None, rows): the attention branch's tail in
    front -- x <- x + row_scale1 * (a wp^T + bias) first (written to out), then the MLP branch on that."""
    assert w1.dtype in HALF and w2.dtype == w1.dtype
    _chk(x, torch.float32, "x"); _chk(w1, w1.dtype, "w1"); _chk(w2, w1.dtype, "w2")
    M, D = x.shape
    out = x if out is None else out
    p = _lib.VitMlpParams()
    p.dtype = dtype_code(w1)
    p.x, p.out, p.W1, p.W2, p.ln_w, p.ln_b, p.ln_eps = _p(x), _p(out), _p(w1), _p(w2), _p(ln[0]), _p(ln[1]), ln_eps
    p.b1, p.b2, p.row_scale, p.row_scale_rows, p.residual2 = _p(b1), _p(b2), _p(row_scale), row_scale_rows, _p(residual2)
    p.M, p.D, p.hidden, p.workgroups = M, D, w1.shape[0], workgroups
    flops = 4.0 * M * D * w1.shape[0]
    if proj is not None:
        a, wpt, pb, rs1, rs1_rows = proj
        assert a.dtype == w1.dtype and wpt.dtype == w1.dtype and tuple(a.shape) == (M, D)
        _chk(a, a.dtype, "proj a"); _chk(wpt, a.dtype, "proj w")
        p.proj_a, p.proj_W, p.proj_b, p.proj_row_scale, p.proj_row_scale_rows = _p(a), _p(wpt), _p(pb), _p(rs1), int(rs1_rows)
        flops += 2.0 * M * D * D
    if profiler is not None:
        profiler.begin("gemm_bf16", flops, "ppt_vit_mlp_bf16 (" + ("proj + residual + " if proj is not None else "") + "LN + fc1 + GELU + fc2 + residual)")
    variant = getattr(w1, "ppt_variant", 2)                  # (the order the weights were re-tiled in decides the kernel)
    assert getattr(w2, "ppt_variant", 2) == variant
    if variant == 3:
        _lib.check(_lib.lib().ppt_vit_mlp3_bf16(ctypes.byref(p), _stream()), "ppt_vit_mlp3_bf16")
    else:
        _lib.check(_lib.lib().ppt_vit_mlp_bf16(ctypes.byref(p), _stream()), "ppt_vit_mlp_bf16")
    if profiler is not None:
        profiler.end()
    return out


ROWGEMM_K = (384, 512)


def rowgemm(A, W, *, ln=None, ln_eps=1e-5, ln_stats=None, bias=None, act=ACT_NONE, out=None, out2=None, residual=None, residual2=None,
            row_scale=None, row_scale_rows=0, walkers=0):
    """ppt_rowgemm_bf16 (csrc/rowgemm.hip): C = epilogue(prologue(A) @ W^T) with W [N,K] bf16 held in registers.
    A: bf16 [M,K], or -- with ln = (gamma, beta) -- the f32 residual stream, LayerNorm applied while the rows are staged.
    residual given: the f32 form out = residual + row_scale[m // row_scale_rows] * (acc + bias) + residual2 (out may be
    residual itself); else out (bf16) = act(acc + bias) and out2 (bf16, optional) receives the pre-activation."""
    M, K = A.shape
    N = W.shape[0]
    T = W.dtype
    assert W.shape[1] == K and T in HALF and W.is_contiguous() and A.is_contiguous()
    assert A.dtype == (torch.float32 if ln is not None else T)
    p = RowGemmParams()
    p.A, p.W, p.M, p.N, p.K = _p(A), _p(W), M, N, K
    p.dtype = dtype_code(W)
    if ln is not None:
        p.a_ln, p.ln_w, p.ln_b, p.ln_eps = 1, _p(ln[0]), _p(ln[1]), ln_eps
        if ln_stats is not None:                     # (mean [M], rstd [M]) f32: what ppt_layernorm_bwd needs
            p.ln_mean, p.ln_rstd = _p(ln_stats[0]), _p(ln_stats[1])
    p.bias, p.act = _p(bias), act
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.is_contiguous() and residual.shape == (M, N)
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=A.device)
        assert out.dtype == torch.float32 and out.is_contiguous()
        p.residual_form, p.residual, p.residual2 = 1, _p(residual), _p(residual2)
        p.row_scale, p.row_scale_rows = _p(row_scale), row_scale_rows
    else:
        if out is None:
            out = torch.empty((M, N), dtype=T, device=A.device)
        assert out.dtype == T and out.is_contiguous()
        if out2 is not None:
            assert out2.dtype == T and out2.is_contiguous()
            p.C2 = _p(out2)
    p.C, p.walkers = _p(out), walkers
    if profiler is not None:
        profiler.begin("gemm_bf16", 2.0 * M * N * K, "ppt_rowgemm_bf16" + (" (LayerNorm prologue)" if ln is not None else
                                                                            " (residual form)" if residual is not None else ""))
    _lib.check(_lib.lib().ppt_rowgemm_bf16(ctypes.byref(p), _stream()), "ppt_rowgemm_bf16")
    if profiler is not None:
        profiler.end()
    if probe is not None:
        _probe("rowgemm", out); _probe("rowgemm (second output)", out2)
    return out


def layernorm_fwd(x, w, b, y_dtype, add=None, add_rows=0, write_xs=None, save_stats=False, eps=1e-5):
    """y = LN(x (+ add)); x f32 [..., D].  write_xs: f32 tensor receiving x+add (may be x itself).
    Returns (y, mean, rstd)."""
    _chk(x, torch.float32, "x")
    D = x.shape[-1]
    M = x.numel() // D
    y = torch.empty(x.shape, dtype=y_dtype, device=x.device)
    mean = torch.empty((M,), dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device) if save_stats else None
    _lib.check(_lib.lib().ppt_layernorm_fwd(_p(x), _p(add), add_rows, _p(write_xs), _p(w), _p(b), _p(y),
                                            dtype_code(y), _p(mean), _p(rstd), M, D, eps, _stream()),
               "ppt_layernorm_fwd")
    if probe is not None:
        _probe("layernorm", y)
    return y, mean, rstd


def layernorm_fwd_sum(x, bias, parts, w, b, y_dtype, write_xs, save_stats=False, eps=1e-5):
    """y = LN(x + bias + parts[0] + ... + parts[S-1]) -- the consumer of a split-K linear (ppt_layernorm_fwd_sum): x f32 [M, D],
    parts f32 [S, M, D] (the S partial products, ops.gemm_splitk), bias [D] or None; write_xs receives the summed rows.
    Returns (y, mean, rstd)."""
    _chk(x, torch.float32, "x"); _chk(parts, torch.float32, "parts"); _chk(write_xs, torch.float32, "write_xs")
    D = x.shape[-1]
    M = x.numel() // D
    S = parts.shape[0]
    assert parts.numel() == S * M * D
    y = torch.empty(x.shape, dtype=y_dtype, device=x.device)
    mean = torch.empty((M,), dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device) if save_stats else None
    _lib.check(_lib.lib().ppt_layernorm_fwd_sum(_p(x), _p(bias), _p(parts), S, _p(write_xs), _p(w), _p(b), _p(y), dtype_code(y),
                                                _p(mean), _p(rstd), M, D, eps, _stream()), "ppt_layernorm_fwd_sum")
    if probe is not None:
        _probe("layernorm", y)
    return y, mean, rstd


def layernorm_bwd_sum(dy_parts, xs, w, mean, rstd, dx, accumulate=True, copy_dtype=None):
    """layernorm_bwd's input-gradient form with dy = dy_parts[0] + ... + dy_parts[S-1] ([S, M, D] f32) -> (dx, dx_copy)."""
    _chk(dy_parts, torch.float32, "dy_parts"); _chk(xs, torch.float32, "xs")
    D = xs.shape[-1]
    M = xs.numel() // D
    S = dy_parts.shape[0]
    assert dy_parts.numel() == S * M * D
    cp = torch.empty(xs.shape, dtype=copy_dtype, device=xs.device) if copy_dtype is not None else None
    _lib.check(_lib.lib().ppt_layernorm_bwd_sum(_p(dy_parts), S, _p(xs), _p(w), _p(mean), _p(rstd), _p(dx), int(accumulate), _p(cp),
                                                dtype_code(cp) if cp is not None else 0, M, D, _stream()), "ppt_layernorm_bwd_sum")
    return dx, cp


def gemm_splitk(A, B, S):
    """The S partial products of A [M, K] @ B [N, K]^T over K slices of K / S as fp32 slices [S, M, N] (one batched ppt_gemm launch,
    plain epilogue): for skinny problems (the prompt chain: 817 rows) whose K loop is what takes the time.  K / S must be a multiple
    of 64 elements (the 64 x 64 LDS-DMA tile loop's slab)."""
    M, K = A.shape
    N = B.shape[0]
    Ks = K // S
    assert Ks * S == K and Ks % 64 == 0 and A.stride(1) == 1 and B.stride(1) == 1
    part = torch.empty((S, M, N), dtype=torch.float32, device=A.device)
    gemm(A[:, :Ks], B[:, :Ks], out=part.view(S * M, N), M=M, batch=S, strideA=Ks, strideB=Ks, strideC=M * N)
    return part


def layernorm_bwd(dy, xs, w, mean, rstd, dx=None, accumulate=False, want_wgrad=False, partial_rows=None,
                  copy_dtype=None):
    """-> (dx, dw, db[, dx_copy]).  dx f32; accumulate=True adds into the given dx; copy_dtype: also
    return the final dx converted to that dtype (operand of the next GEMM)."""
    _chk(dy, torch.float32, "dy"); _chk(xs, torch.float32, "xs")
    D = xs.shape[-1]
    M = xs.numel() // D
    if dx is None:
        dx = torch.empty_like(xs)
        accumulate = False
    dwp = dbp = None
    if partial_rows is None:
        # one wave per partial row walks M / partial_rows rows: enough waves to fill the chip (256 partial rows left the
        # 32 832-row LayerNorm backwards of C3 on 64 workgroups, 0.4 ms each), at most ~16 rows per wave
        partial_rows = max(256, min(4096, (M + 15) // 16))
    if want_wgrad:
        dwp = torch.empty((partial_rows, D), dtype=torch.float32, device=xs.device)
        dbp = torch.empty((partial_rows, D), dtype=torch.float32, device=xs.device)
    cp = torch.empty(xs.shape, dtype=copy_dtype, device=xs.device) if copy_dtype is not None else None
    _lib.check(_lib.lib().ppt_layernorm_bwd(_p(dy), _p(xs), _p(w), _p(mean), _p(rstd), _p(dx), int(accumulate),
                                            _p(cp), dtype_code(cp) if cp is not None else 0,
                                            _p(dwp), _p(dbp), partial_rows, M, D, _stream()), "ppt_layernorm_bwd")
    dw, db = (reduce_rows(dwp), reduce_rows(dbp)) if want_wgrad else (None, None)
    return (dx, dw, db, cp) if copy_dtype is not None else (dx, dw, db)


def attention_fwd(qkv, Bt, T, H, scale, causal, want_lse=True):
    """qkv [Bt*T, 3*H*64] -> out [Bt*T, H*64], lse [Bt,H,T] f32."""
    _chk(qkv, None, "qkv")
    out = torch.empty((Bt * T, H * 64), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((Bt, H, T), dtype=torch.float32, device=qkv.device) if want_lse else None
    if profiler is not None:
        profiler.begin("attention_fwd", 4.0 * Bt * H * T * T * 64 * (0.5 if causal else 1.0))
    if qkv.dtype == torch.float32 and _SPLIT16 and ATTN_SPLIT16:      # split16 mode: both products from hi + lo half pairs
        _lib.check(_lib.lib().ppt_attention_fwd_split16(_p(qkv), _p(out), _p(lse), Bt, T, 0, H, 64, scale, int(causal), _stream()),
                   "ppt_attention_fwd_split16")
    else:
        _lib.check(_lib.lib().ppt_attention_fwd(_p(qkv), _p(out), _p(lse), Bt, T, H, 64, scale, int(causal),
                                                dtype_code(qkv), _stream()), "ppt_attention_fwd")
    if profiler is not None:
        profiler.end()
    if probe is not None:
        _probe("attention", out)
    return out, lse


def attention_bwd(qkv, out, dout, lse, Bt, T, H, scale, causal):
    _chk(qkv, None, "qkv"); _chk(out, qkv.dtype, "out"); _chk(dout, qkv.dtype, "dout")
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((Bt, H, T), dtype=torch.float32, device=qkv.device)
    if profiler is not None:
        profiler.begin("attention_bwd", 10.0 * Bt * H * T * T * 64 * (0.5 if causal else 1.0))
    if qkv.dtype == torch.float32 and _SPLIT16 and ATTN_SPLIT16:      # split16 mode: every product from hi + lo half pairs
        _lib.check(_lib.lib().ppt_attention_bwd_split16(_p(qkv), _p(out), _p(dout), _p(lse), _p(delta), _p(dqkv), None, Bt, T, 0, H, 64,
                                                        scale, int(causal), _stream()), "ppt_attention_bwd_split16")
    else:
        _lib.check(_lib.lib().ppt_attention_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(delta), _p(dqkv), Bt, T, H, 64,
                                                scale, int(causal), dtype_code(qkv), _stream()), "ppt_attention_bwd")
    if profiler is not None:
        profiler.end()
    return dqkv


def prefix_rows(C, T, P):
    """rows of the prefix-shared layout (include/ppt_hip.h, ppt_attention_prefix_fwd): P shared + C * (T - P)."""
    return P + C * (T - P)


def attention_prefix_fwd(qkv, C, T, P, H, scale, want_lse=True):
    """causal attention over C prompts of T positions sharing their first P: qkv [P + C(T-P), 3*H*64] ->
    (out [rows, H*64], lse [rows, H] f32)."""
    _chk(qkv, None, "qkv")
    rows = prefix_rows(C, T, P)
    assert qkv.shape[0] == rows
    out = torch.empty((rows, H * 64), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((rows, H), dtype=torch.float32, device=qkv.device) if want_lse else None
    if profiler is not None:
        profiler.begin("attention_fwd", 2.0 * (C * (T * T - P * P) + P * P) * H * 64)
    if qkv.dtype == torch.float32 and _SPLIT16 and ATTN_SPLIT16:
        _lib.check(_lib.lib().ppt_attention_fwd_split16(_p(qkv), _p(out), _p(lse), C, T, P, H, 64, scale, 1, _stream()),
                   "ppt_attention_fwd_split16")
    else:
        _lib.check(_lib.lib().ppt_attention_prefix_fwd(_p(qkv), _p(out), _p(lse), C, T, P, H, 64, scale, dtype_code(qkv), _stream()),
                   "ppt_attention_prefix_fwd")
    if profiler is not None:
        profiler.end()
    if probe is not None:
        _probe("attention", out)
    return out, lse


def attention_prefix_bwd(qkv, out, dout, lse, C, T, P, H, scale):
    _chk(qkv, None, "qkv"); _chk(out, qkv.dtype, "out"); _chk(dout, qkv.dtype, "dout")
    rows = prefix_rows(C, T, P)
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((rows, H), dtype=torch.float32, device=qkv.device)
    ws = torch.empty((_lib.lib().ppt_attention_prefix_workspace_bytes(C, P, H, 64) // 4,), dtype=torch.float32, device=qkv.device)
    if profiler is not None:
        profiler.begin("attention_bwd", 5.0 * (C * (T * T - P * P) + P * P) * H * 64)
    if qkv.dtype == torch.float32 and _SPLIT16 and ATTN_SPLIT16:
        _lib.check(_lib.lib().ppt_attention_bwd_split16(_p(qkv), _p(out), _p(dout), _p(lse), _p(delta), _p(dqkv), _p(ws), C, T, P, H, 64,
                                                        scale, 1, _stream()), "ppt_attention_bwd_split16")
    else:
        _lib.check(_lib.lib().ppt_attention_prefix_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(delta), _p(dqkv), _p(ws), C, T, P, H, 64,
                                                       scale, dtype_code(qkv), _stream()), "ppt_attention_prefix_bwd")
    if profiler is not None:
        profiler.end()
    return dqkv


def conv1_stats(pts, w1, b1):
    """partial (sum, sumsq) of y = w1.p + b1 over points [M,3] -> ([P,C], [P,C])."""
    _chk(pts, torch.float32, "pts")
    M = pts.shape[0]
    C = w1.shape[0]
    P = _lib.lib().ppt_conv1_stats_max_partials(M)
    ps = torch.empty((P, C), dtype=torch.float32, device=pts.device)
    pq = torch.empty((P, C), dtype=torch.float32, device=pts.device)
    n = ctypes.c_int(0)
    _lib.check(_lib.lib().ppt_conv1_stats(_p(pts), M, _p(w1), _p(b1), C, _p(ps), _p(pq), ctypes.byref(n), _stream()),
               "ppt_conv1_stats")
    return ps, pq, _lib.lib().ppt_conv1_stats_rows_per_partial()


def bn_finalize(gamma, beta, train, partials=None, rows_per_partial=0, count=0, running_mean=None, running_var=None,
                num_batches_tracked=None, eps=1e-5, momentum=0.1, update_running=True, want_moments=False):
    """-> (scale, shift) f32 [C] (+ (mean, rstd) with want_moments); updates the running buffers in place when train."""
    C = gamma.shape[0]
    mean = torch.empty((C,), dtype=torch.float32, device=gamma.device) if want_moments else None
    rstd = torch.empty((C,), dtype=torch.float32, device=gamma.device) if want_moments else None
    scale = torch.empty((C,), dtype=torch.float32, device=gamma.device)
    shift = torch.empty((C,), dtype=torch.float32, device=gamma.device)
    ps, pq = partials if partials is not None else (None, None)
    upd = train and update_running
    P = 0 if ps is None else ps.shape[0]
    wbytes = _lib.lib().ppt_bn_finalize_workspace_bytes(P, C) if train else 0
    ws = torch.empty((wbytes // 8,), dtype=torch.float64, device=gamma.device) if wbytes else None
    _lib.check(_lib.lib().ppt_bn_finalize_ws(_p(ps), _p(pq), P, rows_per_partial, count, C, _p(gamma),
                                             _p(beta), eps, int(train), momentum,
                                             _p(running_mean) if (upd or not train) else None,
                                             _p(running_var) if (upd or not train) else None,
                                             _p(num_batches_tracked) if upd else None, _p(scale), _p(shift), _p(mean), _p(rstd),
                                             _p(ws), wbytes, _stream()),
               "ppt_bn_finalize_ws")
    return (scale, shift, mean, rstd) if want_moments else (scale, shift)


def rows_stats(x):
    """x [M,C] f32 -> ((sum, M2) partials [P,C], rows per partial) for bn_finalize."""
    _chk(x, torch.float32, "x")
    M, C = x.shape
    rpp = _lib.lib().ppt_rows_stats_rows_per_partial()
    P = (M + rpp - 1) // rpp
    ps = torch.empty((P, C), dtype=torch.float32, device=x.device)
    pq = torch.empty_like(ps)
    _lib.check(_lib.lib().ppt_rows_stats_f32(_p(x), M, C, _p(ps), _p(pq), _stream()), "ppt_rows_stats_f32")
    return (ps, pq), rpp


def bn_rows_backward(dy, x, scale, shift, mean, rstd, relu, batch_stats, want_dx=True, want_bf16=False, half_dtype=torch.bfloat16):
    """BatchNorm1d(+ReLU) backward over rows: -> (dx [M,C] f32 | None, d gamma [C], d beta [C][, dx in half_dtype (bf16 | f16)])."""
    _chk(dy, torch.float32, "dy"); _chk(x, torch.float32, "x")
    M, C = x.shape
    rpp = _lib.lib().ppt_rows_stats_rows_per_partial()
    P = (M + rpp - 1) // rpp
    pg = torch.empty((P, C), dtype=torch.float32, device=x.device)
    pgx = torch.empty_like(pg)
    _lib.check(_lib.lib().ppt_bn_rows_bwd_reduce(_p(dy), _p(x), _p(scale), _p(shift), _p(mean), _p(rstd), int(relu), M, C,
                                                 _p(pg), _p(pgx), _stream()), "ppt_bn_rows_bwd_reduce")
    sg, sgx = reduce_rows(pg), reduce_rows(pgx)
    dx = torch.empty_like(x) if want_dx else None
    dxb = torch.empty(x.shape, dtype=half_dtype, device=x.device) if want_bf16 else None
    _lib.check(_lib.lib().ppt_bn_rows_bwd_apply(_p(dy), _p(x), _p(scale), _p(shift), _p(mean), _p(rstd), _p(sg), _p(sgx),
                                                int(relu), int(batch_stats), M, C, _p(dx), _p(dxb), _DT[half_dtype], _stream()),
               "ppt_bn_rows_bwd_apply")
    return (dx, sgx, sg, dxb) if want_bf16 else (dx, sgx, sg)


def three_nn_interp(points1, points2, idx, dist, out_dtype, mult):
    """ppt_three_nn_interp_fwd: rows [B*N, ld] = [points1 | sum_j w_j points2[idx_j] | 0], ld = D1 + D2 rounded up to `mult`
    (the K alignment of the GEMM that reads them), in out_dtype; -> (rows, normalised weights [B,N,3] f32)."""
    _chk(points2, torch.float32, "points2"); _chk(idx, torch.int64, "idx"); _chk(dist, torch.float32, "dist")
    _chk(points1, torch.float32, "points1")
    B, N, _ = idx.shape
    S, D2 = points2.shape[1], points2.shape[2]
    D1 = 0 if points1 is None else points1.shape[2]
    ld = (D1 + D2 + mult - 1) // mult * mult
    rows = torch.empty((B * N, ld), dtype=out_dtype, device=points2.device)
    w = torch.empty((B, N, 3), dtype=torch.float32, device=points2.device)
    _lib.check(_lib.lib().ppt_three_nn_interp_fwd(_p(points1), D1, _p(points2), D2, _p(idx), _p(dist), B, N, S, _p(rows),
                                                  dtype_code(rows), ld, _p(w), _stream()), "ppt_three_nn_interp_fwd")
    return rows, w


def scatter_rows_bwd(idx, weight, d_rows, col_off, rows_div, S, C):
    """ppt_scatter_rows_bwd: idx [B,E] i64 (E entries per cloud), weight [B,E] f32 | None, d_rows [B * E / rows_div, ld] f32
    -> d_src [B,S,C] f32 = sum of the (weighted) gradient rows of the entries that gathered source s, ascending entry order."""
    _chk(idx, torch.int64, "idx"); _chk(weight, torch.float32, "weight"); _chk(d_rows, torch.float32, "d_rows")
    B = idx.shape[0]
    E = idx.numel() // B
    out = torch.empty((B, S, C), dtype=torch.float32, device=d_rows.device)
    _lib.check(_lib.lib().ppt_scatter_rows_bwd(_p(idx), _p(weight), _p(d_rows), d_rows.shape[1], col_off, B, E, rows_div, S, C,
                                               _p(out), _stream()), "ppt_scatter_rows_bwd")
    return out


def sum_groups(x, k):
    """x [G*k, C] f32 -> [G, C]: sums of each k consecutive rows."""
    _chk(x, torch.float32, "x")
    C = x.shape[-1]
    G = x.numel() // (k * C)
    out = torch.empty((G, C), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ppt_sum_groups(_p(x), G, k, C, _p(out), _stream()), "ppt_sum_groups")
    return out


def gather_add(P, Q, idx, Nsrc, y_dtype, want_stats=True):
    """y[b,s,k,:] = P[b*Nsrc + idx[b,s,k], :] + Q[b*S + s, :] -> (y [B*S*K, C], (psum, pm2) | None)."""
    _chk(P, torch.float32, "P"); _chk(Q, torch.float32, "Q"); _chk(idx, torch.int64, "idx")
    B, S, K = idx.shape
    C = P.shape[1]
    M = B * S * K
    y = torch.empty((M, C), dtype=y_dtype, device=P.device)
    ps = pm = None
    if want_stats:
        ps = torch.empty(((M + 31) // 32, C), dtype=torch.float32, device=P.device)
        pm = torch.empty_like(ps)
    _lib.check(_lib.lib().ppt_gather_add(_p(P), dtype_code(P), _p(Q), _p(idx), B, Nsrc, S, K, C, _p(y), dtype_code(y),
                                         _p(ps), _p(pm), _stream()), "ppt_gather_add")
    return y, ((ps, pm) if want_stats else None)


def pool_finish(pmax, pmin, fold, scale, shift, out):
    """out[g, :C] = relu(scale * (scale >= 0 ? max : min) + shift) folded over `fold` pooled rows; out is a 2-D
    view (row stride free) -- e.g. a column slice of the concatenated MSG feature matrix."""
    G = pmax.shape[0] // fold
    C = pmax.shape[1]
    assert out.shape[0] == G and out.shape[1] == C and out.stride(1) == 1
    _lib.check(_lib.lib().ppt_pool_finish(_p(pmax), _p(pmin), dtype_code(pmax), G, fold, C, _p(scale), _p(shift), _p(out),
                                          dtype_code(out), out.stride(0), _stream()), "ppt_pool_finish")
    return out


def bn_act_rows(x, scale, shift, y_dtype, mask=None):
    _chk(x, torch.float32, "x")
    M, C = x.shape
    y = torch.empty((M, C), dtype=y_dtype, device=x.device)
    _lib.check(_lib.lib().ppt_bn_act_rows(_p(x), M, C, _p(scale), _p(shift), _p(mask), _p(y), dtype_code(y), _stream()),
               "ppt_bn_act_rows")
    return y


def gn_lrelu_max_forward(y, gamma, beta, groups, eps, slope):
    """y [B,Q,K,C] f32 -> (out [B,Q,C], arg [B,Q,C] i32, mean [B,G], rstd [B,G]): GroupNorm(groups) + LeakyReLU(slope) +
    max over K (DGCNN_Propagation, pointbert/pointnet2_utils.py:371-467)."""
    _chk(y, torch.float32, "y")
    B, Q, K, C = y.shape
    L = _lib.lib()
    nch = L.ppt_gn_stats_chunks(Q * K)
    part = torch.empty((B, nch, groups, 2), dtype=torch.float64, device=y.device)
    _lib.check(L.ppt_gn_stats(_p(y), B, Q * K, C, groups, _p(part), _stream()), "ppt_gn_stats")
    n = float(Q * K * (C // groups))
    mean = torch.empty((B, groups), dtype=torch.float32, device=y.device)
    rstd = torch.empty((B, groups), dtype=torch.float32, device=y.device)
    _lib.check(L.ppt_gn_finish(_p(part), B, nch, groups, n, float(eps), 0, _p(mean), _p(rstd), _stream()), "ppt_gn_finish")   # biased, as nn.GroupNorm
    out = torch.empty((B, Q, C), dtype=torch.float32, device=y.device)
    arg = torch.empty((B, Q, C), dtype=torch.int32, device=y.device)
    _lib.check(L.ppt_gn_lrelu_max(_p(y), _p(mean), _p(rstd), _p(gamma), _p(beta), B, Q, K, C, groups, slope, _p(out), _p(arg),
                                  _stream()), "ppt_gn_lrelu_max")
    return out, arg, mean, rstd


def gn_lrelu_max_backward(y, dout, out, arg, mean, rstd, gamma, groups, slope):
    """-> (dy [B,Q,K,C], dgamma [C], dbeta [C])."""
    _chk(y, torch.float32, "y"); _chk(dout, torch.float32, "dout")
    B, Q, K, C = y.shape
    L = _lib.lib()
    nch = L.ppt_gn_bwd_chunks(Q)
    psum = torch.empty((B, nch, groups, 2), dtype=torch.float64, device=y.device)
    pgb = torch.empty((B, nch, C, 2), dtype=torch.float32, device=y.device)
    _lib.check(L.ppt_gn_bwd_sums(_p(y), _p(dout), _p(out), _p(arg), _p(mean), _p(rstd), _p(gamma), B, Q, K, C, groups, slope,
                                 _p(psum), _p(pgb), _stream()), "ppt_gn_bwd_sums")
    n = float(Q * K * (C // groups))
    s12n = torch.empty((B, groups, 2), dtype=torch.float32, device=y.device)
    _lib.check(L.ppt_gn_finish(_p(psum), B, nch, groups, n, 0.0, 1, _p(s12n), None, _stream()), "ppt_gn_finish")
    gb = reduce_rows(pgb.view(B * nch, 2 * C))                        # [2C] interleaved (dgamma, dbeta)
    dy = torch.empty_like(y)
    _lib.check(L.ppt_gn_bwd_apply(_p(y), _p(dout), _p(out), _p(arg), _p(mean), _p(rstd), _p(gamma), _p(s12n), B, Q, K, C, groups,
                                  slope, _p(dy), _stream()), "ppt_gn_bwd_apply")
    gb = gb.view(C, 2)
    return dy, gb[:, 0].contiguous(), gb[:, 1].contiguous()


def mini_pointnet_conv12(pts, w1, b1, a_scale, a_shift, w2, bias2):
    """pts [M,3] f32 -> (y2 [M,256] bf16, gmax [M/32,256] bf16): conv1 + folded BN + ReLU + conv2 + bias and the max over
    each group of 32 rows, one kernel (ppt_mini_pointnet_conv12_bf16).  w2 [256,128] bf16."""
    assert w2.dtype in HALF
    _chk(pts, torch.float32, "pts"); _chk(w2, w2.dtype, "w2")
    M = pts.shape[0]
    N, C1 = w2.shape
    y2 = torch.empty((M, N), dtype=w2.dtype, device=pts.device)
    gmax = torch.empty((M // 32, N), dtype=w2.dtype, device=pts.device)
    if profiler is not None:
        profiler.begin("gemm_bf16", 2.0 * M * N * C1, "ppt_mini_pointnet_conv12_bf16")
    _lib.check(_lib.lib().ppt_mini_pointnet_conv12_half(_p(pts), M, _p(w1), _p(b1), _p(a_scale), _p(a_shift), C1, _p(w2), _p(bias2),
                                                        N, _p(y2), _p(gmax), dtype_code(w2), _stream()), "ppt_mini_pointnet_conv12_half")
    if profiler is not None:
        profiler.end()
    if probe is not None:
        _probe("mini-PointNet conv2 output (pre-BN)", y2)
    return y2, gmax


def mini_pointnet_conv3(A, w, gterm, col_stats=None, store=True):
    """A [M,256] bf16, w [512,256] bf16, gterm [M/32,512] f32 -> y [M,512] bf16 = A @ w^T + gterm[group] (+ BatchNorm partials
    into col_stats) -- ppt_mini_pointnet_conv3_bf16.  The FLOPs are accounted at the 512-wide conv this is the local half of.
    store=False: the statistics pass alone (y is not written; returns None)."""
    assert w.dtype in HALF
    _chk(A, w.dtype, "A"); _chk(w, w.dtype, "w"); _chk(gterm, torch.float32, "gterm")
    M, K = A.shape
    N = w.shape[0]
    assert store or col_stats is not None
    y = torch.empty((M, N), dtype=w.dtype, device=A.device) if store else None
    ps, pm = col_stats if col_stats is not None else (None, None)
    if profiler is not None:
        # model: the 512-wide Conv1d on cat(global, local) (dvae.py:194); executed: the local half (the global half is one row
        # per group, a separate small GEMM booked with algo_k = 0 / its own executed FLOPs)
        profiler.begin("gemm_bf16", 2.0 * M * N * 2 * K, "ppt_mini_pointnet_conv3_bf16", executed=2.0 * M * N * K)
    _lib.check(_lib.lib().ppt_mini_pointnet_conv3_half(_p(A), M, K, _p(w), _p(gterm), N, _p(y), _p(ps), _p(pm), dtype_code(w), _stream()),
               "ppt_mini_pointnet_conv3_half")
    if profiler is not None:
        profiler.end()
    if probe is not None:
        _probe("mini-PointNet conv3 output (pre-BN)", y)
    return y


def mpn34_retile(w4):
    """W4 [256,512] 16-bit -> the fragment order csrc/mpn34.hip streams (ppt_mpn34_retile)."""
    assert w4.dtype in HALF and tuple(w4.shape) == (256, 512)
    _chk(w4, w4.dtype, "w4")
    out = torch.empty_like(w4)
    _lib.check(_lib.lib().ppt_mpn34_retile(_p(w4), _p(out), _stream()), "ppt_mpn34_retile")
    return out


def mini_pointnet_conv34(y2, w3s, gs, w4_tiled, bias4):
    """tok [M/32,256] = max over each group of 32 rows of relu(y2 @ w3s^T + gs[group]) @ w4^T + bias4, the [M,512] intermediate
    never leaving the chip (ppt_mini_pointnet_conv34_half).  y2 [M,256], w3s [512,256] (BatchNorm scale folded in), w4_tiled from
    mpn34_retile, all one 16-bit format; gs [M/32,512] f32."""
    T = w3s.dtype
    assert T in HALF and tuple(w3s.shape) == (512, 256) and w4_tiled.dtype == T and w4_tiled.numel() == 256 * 512
    _chk(y2, T, "y2"); _chk(w3s, T, "w3s"); _chk(w4_tiled, T, "w4_tiled"); _chk(gs, torch.float32, "gs")
    M = y2.shape[0]
    assert y2.shape[1] == 256 and tuple(gs.shape) == (M // 32, 512) and M % 32 == 0
    tok = torch.empty((M // 32, 256), dtype=T, device=y2.device)
    if profiler is not None:
        # model: the 512-wide conv3 on cat(global, local) + conv4; executed: the local half of conv3 (17 k-steps) + conv4
        profiler.begin("gemm_bf16", 2.0 * M * 512 * 512 + 2.0 * M * 256 * 512, "ppt_mini_pointnet_conv34_half",
                       executed=2.0 * M * 512 * 272 + 2.0 * M * 256 * 512)
    _lib.check(_lib.lib().ppt_mini_pointnet_conv34_half(_p(y2), M, _p(w3s), _p(gs), _p(w4_tiled), _p(bias4), _p(tok), dtype_code(T), _stream()),
               "ppt_mini_pointnet_conv34_half")
    if profiler is not None:
        profiler.end()
    if probe is not None:
        _probe("mini-PointNet tokens", tok)
    return tok


def mini_pointnet_conv4(A, a_scale, a_shift, w, bias):
    """A [M,512] bf16 -> tok [M/32,256] bf16 = max over each group of 32 rows of relu(a_scale*A + a_shift) @ w^T + bias
    (ppt_mini_pointnet_conv4_bf16)."""
    assert w.dtype in HALF
    _chk(A, w.dtype, "A"); _chk(w, w.dtype, "w")
    M, K = A.shape
    N = w.shape[0]
    tok = torch.empty((M // 32, N), dtype=w.dtype, device=A.device)
    if profiler is not None:
        profiler.begin("gemm_bf16", 2.0 * M * N * K, "ppt_mini_pointnet_conv4_bf16")
    _lib.check(_lib.lib().ppt_mini_pointnet_conv4_half(_p(A), M, K, _p(a_scale), _p(a_shift), _p(w), _p(bias), N, _p(tok), dtype_code(w),
                                                       _stream()), "ppt_mini_pointnet_conv4_half")
    if profiler is not None:
        profiler.end()
    if probe is not None:
        _probe("mini-PointNet tokens", tok)
    return tok


CONV12_STATS_SHAPES = {(32, 32), (64, 64), (64, 96), (64, 128), (128, 128)}      # (C1, N) ppt_conv12_stats_bf16 is built for


def conv12_stats(pts, w1, b1, a_scale, a_shift, w2, bias2):
    """pts [M,3] f32 -> (y2 [M,N] bf16, (part_sum, part_m2) [M/32, N]): conv1 + folded BN + ReLU + conv2 (+ bias) with the
    BatchNorm partials of the output, one barrier-free kernel (ppt_conv12_stats_bf16).  w2 [N,C1] bf16."""
    _chk(pts, torch.float32, "pts"); _chk(w2, torch.bfloat16, "w2")
    M = pts.shape[0]
    N, C1 = w2.shape
    y2 = torch.empty((M, N), dtype=torch.bfloat16, device=pts.device)
    ps = torch.empty((M // 32, N), dtype=torch.float32, device=pts.device)
    pm = torch.empty_like(ps)
    if profiler is not None:
        profiler.begin("gemm_bf16", 2.0 * M * N * C1, "ppt_conv12_stats_bf16")
    _lib.check(_lib.lib().ppt_conv12_stats_bf16(_p(pts), M, _p(w1), _p(b1), _p(a_scale), _p(a_shift), C1, _p(w2), _p(bias2), N,
                                                _p(y2), _p(ps), _p(pm), _stream()), "ppt_conv12_stats_bf16")
    if profiler is not None:
        profiler.end()
    return y2, (ps, pm)


AFFINE_CONV_POOL_SHAPES = {(32, 64, 16), (64, 128, 32), (96, 128, 64), (128, 256, 64)}     # (K, N, pool_rows)


def affine_conv_pool(A, a_scale, a_shift, w, bias, pool_rows, pmax, pmin, col_stats):
    """relu(a_scale * A + a_shift) @ w^T + bias, not written: BatchNorm partials into col_stats = (sum, M2) [M/32, N] and the
    max / min over every pool_rows rows into pmax / pmin (ppt_affine_conv_pool_bf16)."""
    _chk(A, torch.bfloat16, "A"); _chk(w, torch.bfloat16, "w")
    M, K = A.shape
    N = w.shape[0]
    if profiler is not None:
        profiler.begin("gemm_bf16", 2.0 * M * N * K, "ppt_affine_conv_pool_bf16")
    _lib.check(_lib.lib().ppt_affine_conv_pool_bf16(_p(A), A.stride(0), M, K, _p(a_scale), _p(a_shift), _p(w), _p(bias), N, pool_rows,
                                                    _p(pmax), _p(pmin), _p(col_stats[0]), _p(col_stats[1]), _stream()),
               "ppt_affine_conv_pool_bf16")
    if profiler is not None:
        profiler.end()


def group_anchor_stats(x, idx, anchor, Nsrc):
    """x [B*Nsrc, D] (f32 | bf16), idx [B,S,K], anchor [B,S] -> [B,S,2] f32: (sum, sum of squares) of x[idx] - x[anchor] per
    group (the statistic behind LocalGrouper's per-cloud std, pointMLP.py:170-175)."""
    _chk(x, None, "x"); _chk(idx, torch.int64, "idx"); _chk(anchor, torch.int64, "anchor")
    B, S, K = idx.shape
    out = torch.empty((B, S, 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ppt_group_anchor_stats(_p(x), dtype_code(x), _p(idx), _p(anchor), B, Nsrc, S, K, x.shape[1], _p(out),
                                                 _stream()), "ppt_group_anchor_stats")
    return out


def pointmlp_cloud_rstd(stats, n):
    """stats [B,S,2] f32 (group_anchor_stats) -> r [B] f32 = 1 / (unbiased std over the cloud's n values + 1e-5), fp64 inside
    (pointMLP.py:174-175)."""
    _chk(stats, torch.float32, "stats")
    B, S, _ = stats.shape
    r = torch.empty((B,), dtype=torch.float32, device=stats.device)
    _lib.check(_lib.lib().ppt_pointmlp_cloud_rstd(_p(stats), B, S, float(n), _p(r), _stream()), "ppt_pointmlp_cloud_rstd")
    return r


def pointmlp_pq(PQ, r, cidx, c0, B, N):
    """PQ [B*N, 2C] f32 -> (P [B*N, C] = PQ[:, :C] * r[b], Q [B*S, C] = (c0 + PQ[a, C:]) - P[a], a = b*N + cidx[b, s])."""
    _chk(PQ, torch.float32, "PQ"); _chk(r, torch.float32, "r"); _chk(cidx, torch.int64, "cidx"); _chk(c0, torch.float32, "c0")
    C = PQ.shape[1] // 2
    S = cidx.shape[1]
    P = torch.empty((B * N, C), dtype=torch.float32, device=PQ.device)
    Q = torch.empty((B * S, C), dtype=torch.float32, device=PQ.device)
    _lib.check(_lib.lib().ppt_pointmlp_pq(_p(PQ), _p(r), _p(cidx), _p(c0), _p(P), _p(Q), B, N, S, C, _stream()), "ppt_pointmlp_pq")
    return P, Q


def bn_res_act_rows(x, res, scale, shift, y_dtype, pool=1, res_affine=None):
    """relu(scale * x + shift + res') over rows [M,C]; pool > 1 returns the max over each `pool` consecutive rows.
    res' = res, or relu(rs * res + rh) with res_affine = (rs, rh)."""
    _chk(x, None, "x"); _chk(res, None, "res")
    M, C = x.shape
    assert res.shape == x.shape and M % pool == 0
    y = torch.empty((M // pool, C), dtype=y_dtype, device=x.device)
    rs, rh = res_affine if res_affine is not None else (None, None)
    _lib.check(_lib.lib().ppt_bn_res_act_rows(_p(x), dtype_code(x), _p(res), dtype_code(res), M, C, pool, _p(scale), _p(shift),
                                              _p(rs), _p(rh), _p(y), dtype_code(y), _stream()), "ppt_bn_res_act_rows")
    return y


def linear3_gelu(pts, w, b, y_dtype):
    _chk(pts, torch.float32, "pts")
    M, C = pts.shape[0], w.shape[0]
    y = torch.empty((M, C), dtype=y_dtype, device=pts.device)
    _lib.check(_lib.lib().ppt_linear3_gelu(_p(pts), M, _p(w), _p(b), C, _p(y), dtype_code(y), _stream()),
               "ppt_linear3_gelu")
    if probe is not None:
        _probe("pos_embed hidden", y)
    return y


def cls_max_pool(x, want_argmax=False):
    """x [B,T,D] -> feat [B,2D] f32 (+ argmax [B,D] i32)."""
    _chk(x, None, "x")
    B, T, D = x.shape
    out = torch.empty((B, 2 * D), dtype=torch.float32, device=x.device)
    am = torch.empty((B, D), dtype=torch.int32, device=x.device) if want_argmax else None
    _lib.check(_lib.lib().ppt_cls_max_pool(_p(x), dtype_code(x), B, T, D, _p(out), _p(am), _stream()), "ppt_cls_max_pool")
    return out, am


def convert(src, dst_dtype, scale=1.0):
    """src in another dtype; scale != 1: convert(src * scale) -- the operand copy of an fp32 activation gradient entering a 16-bit
    backward stage, multiplied by the stage's power-of-two gradient scale on the way (ppt_convert_scaled)."""
    _chk(src, None, "src")
    if src.dtype == dst_dtype and scale == 1.0:
        return src
    dst = torch.empty(src.shape, dtype=dst_dtype, device=src.device)
    if scale == 1.0:
        _lib.check(_lib.lib().ppt_convert(_p(src), dtype_code(src), _p(dst), dtype_code(dst), src.numel(), _stream()), "ppt_convert")
    else:
        _lib.check(_lib.lib().ppt_convert_scaled(_p(src), dtype_code(src), _p(dst), dtype_code(dst), src.numel(), float(scale), _stream()),
                   "ppt_convert_scaled")
    return dst


def weights_prep(items, dtype):
    """items: list of (w f32 [N, ld] contiguous 2-D view, col0, K, sub_col0 | None, Kp) -> list of (out [N, Kp], out_t [Kp, N]) in
    `dtype`: convert(w[:, col0:col0+K] (- w[:, sub_col0:sub_col0+K])), zero-padded to Kp columns, and its transpose -- ONE launch for
    all items (ppt_weights_prep)."""
    assert dtype in HALF or dtype == torch.float32
    arr = (_lib.WprepItem * len(items))()
    outs = []
    for a, (w, col0, K, sub, Kp) in zip(arr, items):
        _chk(w, torch.float32, "w")
        assert w.dim() == 2
        N = w.shape[0]
        out = torch.empty((N, Kp), dtype=dtype, device=w.device)
        out_t = torch.empty((Kp, N), dtype=dtype, device=w.device)
        a.w, a.ldw, a.N, a.col0, a.K, a.sub_col0, a.Kp, a.out, a.out_t = _p(w), w.shape[1], N, col0, K, (-1 if sub is None else sub), Kp, _p(out), _p(out_t)
        outs.append((out, out_t))
    _lib.check(_lib.lib().ppt_weights_prep(ctypes.cast(arr, ctypes.c_void_p), len(items), dtype_code(dtype), _stream()), "ppt_weights_prep")
    return outs


def labels_check(labels, n_classes, ignore_index, flags, bit):
    """flags[0] |= bit when a label lies outside [0, n_classes) and is not ignore_index (ppt_labels_check)."""
    _chk(labels, torch.int64, "labels"); _chk(flags, torch.int32, "flags")
    _lib.check(_lib.lib().ppt_labels_check(_p(labels), labels.numel(), int(n_classes), int(ignore_index), _p(flags), int(bit), _stream()),
               "ppt_labels_check")


def health_check(x, flags, bit, maxabs=None):
    """flags[0] |= bit if x (any dtype, contiguous) holds a non-finite value; maxabs[0] = max(maxabs[0], max |finite x|) when
    given (ppt_health_check).  flags: int32 [1]; maxabs: f32 [1], non-negative."""
    _chk(x, None, "x"); _chk(flags, torch.int32, "flags"); _chk(maxabs, torch.float32, "maxabs")
    _lib.check(_lib.lib().ppt_health_check(_p(x), dtype_code(x), x.numel(), _p(flags), int(bit), _p(maxabs), _stream()), "ppt_health_check")


# tools/fp16_stress.py: when set, every wrapper below that returns a 16-bit activation reports it -- probe(name, tensor)
probe = None


def _probe(name, t):
    if probe is not None and t is not None and t.dtype in HALF:
        probe(name, t)
    return t


def scale_rows_convert(W, scale, out_dtype, cols=None, bias=None, shift=None):
    """convert(scale[:, None] * W[:, cols[0]:cols[1]]) for a row-major f32 matrix W [N, Kfull] -> [N, K] contiguous in out_dtype
    (ppt_scale_rows_convert); with bias / shift also returns scale * bias + shift [N] f32."""
    _chk(W, torch.float32, "W"); _chk(scale, torch.float32, "scale")
    N, Kf = W.shape
    c0, c1 = cols if cols is not None else (0, Kf)
    K = c1 - c0
    out = torch.empty((N, K), dtype=out_dtype, device=W.device)
    want_bs = bias is not None or shift is not None
    bs = torch.empty((N,), dtype=torch.float32, device=W.device) if want_bs else None
    src = W if c0 == 0 else W[:, c0:]
    _lib.check(_lib.lib().ppt_scale_rows_convert(ctypes.c_void_p(src.data_ptr()), Kf, N, K, _p(scale), _p(out), dtype_code(out), _p(bias), _p(shift),
                                                 _p(bs), _stream()), "ppt_scale_rows_convert")
    return (out, bs) if want_bs else out


def transpose(src, dst_dtype=None, pad_to=1):
    """[R,C] -> [C,R] (optionally converting).  pad_to > 1: the result is [C, Rp] with Rp = R rounded up
    to a multiple of pad_to and a zero tail (so that it can be the K dimension of a GEMM operand)."""
    _chk(src, None, "src")
    R, C = src.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    dt_ = dst_dtype or src.dtype
    dst = torch.empty((C, Rp), dtype=dt_, device=src.device) if Rp == R else torch.zeros((C, Rp), dtype=dt_, device=src.device)
    _lib.check(_lib.lib().ppt_transpose(_p(src), dtype_code(src), _p(dst), dtype_code(dst), R, C, Rp, _stream()),
               "ppt_transpose")
    return dst


def col_sums(x):
    """sum over the rows of x [M,D] (any dtype, row stride free) -> [D] f32."""
    M, D = x.shape
    part = torch.empty(((M + 255) // 256, D), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().ppt_col_sums(_p(x), dtype_code(x), M, D, x.stride(0), _p(part), _stream()), "ppt_col_sums")
    return reduce_rows(part)


def reduce_rows(partial, out=None, accumulate=False):
    _chk(partial, torch.float32, "partial")
    P, D = partial.shape
    if out is None:
        out = torch.empty((D,), dtype=torch.float32, device=partial.device)
        accumulate = False
    _lib.check(_lib.lib().ppt_reduce_rows(_p(partial), P, D, _p(out), int(accumulate), _stream()), "ppt_reduce_rows")
    return out


def head_loss(feat, w, text_raw, logit_scale, labels, smoothing):
    """-> (loss 0-d, logits [B,C], d loss / d text_raw [C,E]): ppt_head_logits + ppt_head_ce_bwd (include/ppt_hip.h)."""
    for t, n in ((feat, "feat"), (w, "w [F,E]"), (text_raw, "text_raw")):
        _chk(t, torch.float32, n)
    _chk(labels, torch.int64, "labels")
    B, F = feat.shape
    C, E = text_raw.shape
    dev = feat.device
    spc = torch.empty((B, E), dtype=torch.float32, device=dev)
    logits = torch.empty((B, C), dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    d_text = torch.empty((C, E), dtype=torch.float32, device=dev)
    L = _lib.lib()
    _lib.check(L.ppt_head_logits(_p(feat), _p(w), _p(text_raw), _p(logit_scale), B, F, E, C, _p(spc), _p(logits), _stream()),
               "ppt_head_logits")
    _lib.check(L.ppt_head_ce_bwd(_p(logits), _p(labels), _p(spc), _p(text_raw), float(smoothing), B, E, C, _p(loss), _p(d_text),
                                 _stream()), "ppt_head_ce_bwd")
    return loss, logits, d_text


def gemm_tn_splitk(x_a, x_b, min_blocks=768, max_splits=16):
    """C[N1,N2] (f32) = x_a[M,N1]^T @ x_b[M,N2] -- a weight gradient: the reduction runs over the M rows, and the output
    has few tiles (proj: 6 x 6), so the rows are cut into S slices that run as one batched NT GEMM over the transposed
    operands (slice z = columns [z*Mc, (z+1)*Mc) of both) and the S fp32 partial products are folded by reduce_rows in a
    fixed order.  S is chosen to reach `min_blocks` 64x64 workgroups."""
    M, N1 = x_a.shape
    N2 = x_b.shape[1]
    tiles = ((N1 + 63) // 64) * ((N2 + 63) // 64)
    if (x_a.dtype in HALF and x_b.dtype == x_a.dtype and M % 32 == 0 and N1 % 8 == 0 and N2 % 8 == 0
            and x_a.stride(1) == 1 and x_b.stride(1) == 1 and x_a.stride(0) % 8 == 0 and x_b.stride(0) % 8 == 0):
        # operands as stored (ppt_gemm_tn_bf16: transposing LDS reads), no transposed copies; the number of slices must
        # divide the number of 32-row slabs
        slabs = M // 32
        big = ((N1 + 127) // 128) * ((N2 + 127) // 128)
        want = max(1, min(max_splits * 2, (min_blocks + big - 1) // big, slabs))
        S = max(d for d in range(1, want + 1) if slabs % d == 0)
        part = torch.empty((S, N1 * N2), dtype=torch.float32, device=x_a.device)
        if profiler is not None:
            profiler.begin("gemm_bf16", 2.0 * M * N1 * N2, "ppt_gemm_tn_bf16")
        _lib.check(_lib.lib().ppt_gemm_tn_half(_p(x_a), x_a.stride(0), _p(x_b), x_b.stride(0), M, N1, N2, S, _p(part),
                                               dtype_code(x_a), _stream()), "ppt_gemm_tn_half")
        if profiler is not None:
            profiler.end()
        return (reduce_rows(part) if S > 1 else part[0]).view(N1, N2)
    S = max(1, min(max_splits, (min_blocks + tiles - 1) // tiles, (M + 63) // 64))
    Mc = ((M + S - 1) // S + 63) // 64 * 64
    at = transpose(x_a, pad_to=S * Mc)                                  # [N1, S*Mc], zero tail
    bt = transpose(x_b, pad_to=S * Mc)
    # split16: the pre-scale goes by an operand's ROLE, not by its position -- B is a weight everywhere else (x 2^4), here both
    # operands are activations / S-scaled gradients, which stay where they are like every other A operand
    role = (SPLIT16_POW2[0], SPLIT16_POW2[0]) if (_SPLIT16 and at.dtype == torch.float32) else None
    if S == 1:
        return gemm(at, bt, out_dtype=torch.float32, split=role)
    part = torch.empty((S * N1, N2), dtype=torch.float32, device=x_a.device)
    gemm(at[:, :Mc], bt[:, :Mc], out=part, batch=S, strideA=Mc, strideB=Mc, strideC=N1 * N2, split=role)
    return reduce_rows(part.view(S, N1 * N2)).view(N1, N2)


def adamw_step(p, g, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, skipped=None):
    """torch.optim.AdamW's update of ONE tensor in one launch (ppt_adamw_step); p, exp_avg, exp_avg_sq updated in place.
    grad_scale != 1: g is first multiplied by it IN PLACE (the un-scaling of a caller's loss-scaled backward).  An element whose
    gradient is not finite is skipped and counted in `skipped` (1-element int64 device tensor, or None)."""
    for t, nm in ((p, "p"), (g, "g"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, torch.float32, nm)
    if skipped is not None:
        _chk(skipped, torch.int64, "skipped")
    _lib.check(_lib.lib().ppt_adamw_step(_p(p), _p(g), _p(exp_avg), _p(exp_avg_sq), p.numel(), lr, beta1, beta2, eps, weight_decay,
                                         int(step), float(grad_scale), _p(skipped), _stream()), "ppt_adamw_step")


def adamw_multi(items, lr, beta1, beta2, eps, weight_decay, grad_scale=1.0, skipped=None):
    """The same update for a list of (p, g, exp_avg, exp_avg_sq, step) in one launch per 64 tensors (ppt_adamw_multi)."""
    arr = (_lib.AdamwTensor * len(items))()
    for a, (p, g, m, v, step) in zip(arr, items):
        for t, nm in ((p, "p"), (g, "g"), (m, "exp_avg"), (v, "exp_avg_sq")):
            _chk(t, torch.float32, nm)
        assert g.numel() == p.numel() == m.numel() == v.numel()
        a.p, a.g, a.exp_avg, a.exp_avg_sq, a.n, a.step = _p(p), _p(g), _p(m), _p(v), p.numel(), int(step)
    if skipped is not None:
        _chk(skipped, torch.int64, "skipped")
    _lib.check(_lib.lib().ppt_adamw_multi(ctypes.cast(arr, ctypes.c_void_p), len(items), lr, beta1, beta2, eps, weight_decay,
                                          float(grad_scale), _p(skipped), _stream()), "ppt_adamw_multi")


def prompt_rows(base, slot, tokens, pos_rows):
    """out[i] = tokens[slot[i]] + pos_rows[i] where slot[i] >= 0, else base[i] (ppt_prompt_rows)."""
    _chk(base, torch.float32, "base"); _chk(slot, torch.int32, "slot"); _chk(tokens, torch.float32, "tokens")
    _chk(pos_rows, torch.float32, "pos_rows")
    rows, W = base.shape
    out = torch.empty_like(base)
    _lib.check(_lib.lib().ppt_prompt_rows(_p(base), _p(slot), _p(tokens), _p(pos_rows), rows, W, _p(out), _stream()), "ppt_prompt_rows")
    return out


def prompt_rows_bwd(g, rows_of, n_tok, scale=1.0):
    """d tokens [n_tok, W] = scale * sums of the rows of g listed per token in rows_of [n_tok, max_rows] i32 (-1 terminated)."""
    _chk(g, torch.float32, "g"); _chk(rows_of, torch.int32, "rows_of")
    W = g.shape[1]
    out = torch.empty((n_tok, W), dtype=torch.float32, device=g.device)
    _lib.check(_lib.lib().ppt_prompt_rows_bwd(_p(g), _p(rows_of), rows_of.shape[1], n_tok, W, float(scale), _p(out), _stream()),
               "ppt_prompt_rows_bwd")
    return out
