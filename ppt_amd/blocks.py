"""Callable transformer sub-modules on the HIP kernels.

The hot path never calls a sub-module: each tower is ONE autograd node over a hand-scheduled kernel pipeline
(ppt_amd/engine.py).  The reference's module classes are nevertheless callable on their own
(models/pointbert/point_encoder.py:24-30 Mlp, :46-58 Attention, :76-79 Block, :99-110 TransformerEncoder;
models/ULIP_models.py:49-56 ResidualAttentionBlock, :66-67 Transformer), and a maintainer who writes
`model.point_encoder.blocks(x, pos)` or `model.transformer(x)` must get the reference's result and gradients.  This file
holds the autograd nodes those `forward`s are made of: every GEMM, attention and LayerNorm -- forward, input gradient AND
weight gradient -- is a ppt_amd.ops call (libppt_hip.so); torch only adds residuals and applies DropPath factors.

Operand precision: `precision` = torch.bfloat16 (bf16 MFMA operands, fp32 accumulation, fp32 outputs) or torch.float32 (parity).
"""
import torch

from . import gradscale, ops
from .ops import ACT_GELU, ACT_NONE, ACT_QUICKGELU

DEFAULT_PRECISION = torch.bfloat16


def _rows(x):
    """[..., D] -> contiguous fp32 [M, D]"""
    return x.reshape(-1, x.shape[-1]).float().contiguous()


def _opnd(w, T, kind):
    """operand copy of a weight: 'w' = [N,K] in T, 'wt' = [K,N] in T"""
    w2 = w.detach().reshape(w.shape[0], -1).contiguous()
    return ops.convert(w2, T) if kind == "w" else ops.transpose(w2, T)


class _LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension (fp32 statistics): ppt_layernorm_fwd / ppt_layernorm_bwd."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x2 = _rows(x)
        y, mean, rstd = ops.layernorm_fwd(x2, w.detach().float().contiguous(), b.detach().float().contiguous(), torch.float32,
                                          save_stats=True, eps=eps)
        ctx.save_for_backward(x2, w, mean, rstd)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w, mean, rstd = ctx.saved_tensors
        dx, dw, db = ops.layernorm_bwd(_rows(dy), x2, w.detach().float().contiguous(), mean, rstd, want_wgrad=True)
        return dx.view(ctx.shape), dw, db, None


def layer_norm(x, w, b, eps=1e-5):
    return _LayerNormFn.apply(x, w, b, eps).to(x.dtype)


class _MlpFn(torch.autograd.Function):
    """y = act(x W1^T + b1) W2^T + b2 (point_encoder.py:24-30 with GELU, ULIP_models.py:41-42 with QuickGELU; the Dropouts
    of the reference have p = 0 in every PPT configuration).  The activation and its derivative ride in GEMM epilogues."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act, T):
        x2 = _rows(x)
        xt = ops.convert(x2, T)
        pre = torch.empty((x2.shape[0], w1.shape[0]), dtype=T, device=x2.device)
        f = ops.gemm(xt, _opnd(w1, T, "w"), out_dtype=T, bias=b1.detach().float() if b1 is not None else None, act=act,
                     out2=pre, out2_pre=True)
        y = ops.gemm(f, _opnd(w2, T, "w"), out_dtype=torch.float32, bias=b2.detach().float() if b2 is not None else None)
        ctx.save_for_backward(xt, pre, f, w1, w2)
        ctx.act, ctx.T, ctx.shape, ctx.has_b = act, T, x.shape, (b1 is not None, b2 is not None)
        # a 16-bit backward stage (ppt_amd/gradscale.py).  Default when no criterion announced its row count: the BATCH size, not
        # the token rows -- a bare block's caller averages over samples, and B x T would over-scale by ~2^9
        ctx.grad_scale = gradscale.current(T, default_rows=x.shape[0] if x.dim() >= 3 else x2.shape[0])
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    @gradscale.scaled_backward
    def backward(ctx, dy):
        xt, pre, f, w1, w2 = ctx.saved_tensors
        T = ctx.T
        g = _rows(dy)
        gt = ops.convert(g, T)
        dw2 = ops.gemm_tn_splitk(gt, f)
        db2 = ops.col_sums(g) if ctx.has_b[1] else None
        d_pre = ops.gemm(gt, _opnd(w2, T, "wt"), out_dtype=T, act=ctx.act, dact_pre=pre)
        dw1 = ops.gemm_tn_splitk(d_pre, xt)
        db1 = ops.col_sums(d_pre) if ctx.has_b[0] else None
        dx = ops.gemm(d_pre, _opnd(w1, T, "wt"), out_dtype=torch.float32)
        return dx.view(ctx.shape), dw1, db1, dw2, db2, None, None


def mlp(x, w1, b1, w2, b2, act=ACT_GELU, precision=None):
    return _MlpFn.apply(x, w1, b1, w2, b2, act, precision or DEFAULT_PRECISION).to(x.dtype)


class _SelfAttentionFn(torch.autograd.Function):
    """proj(softmax(q k^T * scale [causal]) v) with q, k, v = x Wqkv^T (+ bqkv) split [3, heads, 64] along the output
    (point_encoder.py:46-58; nn.MultiheadAttention's packed in_proj at ULIP_models.py:38,49-51 has the same layout and the
    same result: scaling q before the product instead of the scores after it differs only in rounding).  x [B, T, D]."""

    @staticmethod
    def forward(ctx, x, wqkv, bqkv, wproj, bproj, heads, scale, causal, T):
        B, Tn, D = x.shape
        assert D == heads * 64, "the attention kernels are built for 64-wide heads"
        x2 = _rows(x)
        xt = ops.convert(x2, T)
        qkv = ops.gemm(xt, _opnd(wqkv, T, "w"), out_dtype=T, bias=bqkv.detach().float() if bqkv is not None else None)
        a, lse = ops.attention_fwd(qkv, B, Tn, heads, scale, causal, want_lse=True)
        y = ops.gemm(a, _opnd(wproj, T, "w"), out_dtype=torch.float32, bias=bproj.detach().float() if bproj is not None else None)
        ctx.save_for_backward(xt, qkv, a, lse, wqkv, wproj)
        ctx.cfg = (B, Tn, D, heads, scale, causal, T, bqkv is not None, bproj is not None)
        ctx.grad_scale = gradscale.current(T, default_rows=B)       # a 16-bit backward stage; default: the batch size (see _MlpFn)
        return y.view(B, Tn, D)

    @staticmethod
    @gradscale.scaled_backward
    def backward(ctx, dy):
        xt, qkv, a, lse, wqkv, wproj = ctx.saved_tensors
        B, Tn, D, heads, scale, causal, T, has_bq, has_bp = ctx.cfg
        g = _rows(dy)
        gt = ops.convert(g, T)
        dwp = ops.gemm_tn_splitk(gt, a)
        dbp = ops.col_sums(g) if has_bp else None
        d_a = ops.gemm(gt, _opnd(wproj, T, "wt"), out_dtype=T)
        d_qkv = ops.attention_bwd(qkv, a, d_a, lse, B, Tn, heads, scale, causal)
        dwq = ops.gemm_tn_splitk(d_qkv, xt)
        dbq = ops.col_sums(d_qkv) if has_bq else None
        dx = ops.gemm(d_qkv, _opnd(wqkv, T, "wt"), out_dtype=torch.float32)
        return dx.view(B, Tn, D), dwq, dbq, dwp, dbp, None, None, None, None


def self_attention(x, wqkv, bqkv, wproj, bproj, heads, scale, causal=False, precision=None):
    return _SelfAttentionFn.apply(x, wqkv, bqkv, wproj, bproj, heads, scale, causal, precision or DEFAULT_PRECISION).to(x.dtype)


def drop_path(x, drop_prob, training):
    """timm 0.4.12 DropPath (point_encoder.py:4,68): per-sample factor floor(keep + U[0,1)) / keep in train(), identity otherwise."""
    if not drop_prob or not training:
        return x
    keep = 1.0 - drop_prob
    shape = (x.shape[0],) + (1,) * (x.dim() - 1)
    mask = (keep + torch.rand(shape, dtype=x.dtype, device=x.device)).floor_()
    return x.div(keep) * mask
