"""torch.autograd bridges onto the C-ABI GEMM for layers whose WEIGHTS train (the part-segmentation decoder,
models/pointbert/pointnet2_utils.py:297-467).  The frozen towers do not use these: their forward/backward are
the hand-scheduled pipelines of ppt_amd.engine."""
import torch

from . import ops


def _pad_k(t, mult, dtype):
    """[R, K] -> operand copy in `dtype` with K zero-padded to a multiple of `mult`."""
    R, K = t.shape
    Kp = (K + mult - 1) // mult * mult
    if Kp == K:
        return ops.convert(t.contiguous(), dtype)
    out = torch.zeros((R, Kp), dtype=dtype, device=t.device)
    out[:, :K] = t
    return out


class _Linear(torch.autograd.Function):
    """y[M,N] = x[M,K] @ w[N,K]^T (+ b): forward, dX, dW and db all on ppt_gemm (operand dtype `prec`)."""

    @staticmethod
    def forward(ctx, x, w, b, prec):
        mult = 8 if prec == torch.bfloat16 else 4
        x2 = x.detach().float()
        w2 = w.detach().reshape(w.shape[0], -1).float()
        xt, wt = _pad_k(x2, mult, prec), _pad_k(w2, mult, prec)
        y = ops.gemm(xt, wt, out_dtype=torch.float32, bias=b.detach().float().contiguous() if b is not None else None)
        ctx.save_for_backward(xt, wt)
        ctx.k, ctx.wshape, ctx.has_bias, ctx.prec = x2.shape[1], w.shape, b is not None, prec
        return y

    @staticmethod
    def backward(ctx, dy):
        xt, wt = ctx.saved_tensors
        prec = ctx.prec
        dy = dy.contiguous().float()
        dyt = ops.convert(dy, prec)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dyt, ops.transpose(wt), out_dtype=torch.float32)[:, :ctx.k]
        if ctx.needs_input_grad[1]:
            dw = ops.gemm_tn_splitk(dyt, xt)                          # rows split into slices: few output tiles, long K
            dw = dw[:, :ctx.k].reshape(ctx.wshape)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.col_sums(dy)
        return dx, dw, db, None


def linear(x, w, b=None, prec=torch.bfloat16):
    """x [..., K] -> [..., N] through the MFMA GEMM with gradients for x, w and b."""
    lead = x.shape[:-1]
    y = _Linear.apply(x.reshape(-1, x.shape[-1]), w, b, prec)
    return y.view(*lead, -1)


class _BatchNormReLURows(torch.autograd.Function):
    """relu(batch_norm(x)) over the rows of x [M,C] with trainable gamma / beta (pointnet2_utils.py:362-366): statistics,
    normalisation and the whole backward on the HIP kernels (ppt_rows_stats_f32, ppt_bn_finalize_ws, ppt_bn_act_rows,
    ppt_bn_rows_bwd_*); running statistics updated in place as nn.BatchNorm1d does."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps):
        x = x.detach().float().contiguous()
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        if training:
            parts, rpp = ops.rows_stats(x)
            sc, sh, mean, rstd = ops.bn_finalize(g, b, True, partials=parts, rows_per_partial=rpp, count=x.shape[0],
                                                 running_mean=running_mean, running_var=running_var,
                                                 num_batches_tracked=num_batches_tracked, eps=eps, momentum=momentum,
                                                 want_moments=True)
        else:
            sc, sh, mean, rstd = ops.bn_finalize(g, b, False, running_mean=running_mean, running_var=running_var, eps=eps,
                                                 want_moments=True)
        ctx.save_for_backward(x, sc, sh, mean, rstd)
        ctx.training = training
        return ops.bn_act_rows(x, sc, sh, torch.float32)

    @staticmethod
    def backward(ctx, dy):
        x, sc, sh, mean, rstd = ctx.saved_tensors
        dx, dgamma, dbeta = ops.bn_rows_backward(dy.contiguous().float(), x, sc, sh, mean, rstd, True, ctx.training)
        return dx, dgamma, dbeta, None, None, None, None, None, None


class _GroupNormLReLUMax(torch.autograd.Function):
    """max over k of LeakyReLU(GroupNorm(y)) for the channels-last conv output y [B,Q,K,C] (DGCNN_Propagation,
    pointbert/pointnet2_utils.py:371-467 -- there: permute to [B,C,Q,K], nn.GroupNorm, nn.LeakyReLU, max(dim=-1)):
    forward and backward on the HIP kernels of csrc/groupnorm.hip; the normalised tensor is never materialised."""

    @staticmethod
    def forward(ctx, y, gamma, beta, groups, eps, slope):
        y = y.contiguous().float()
        g, b = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        out, arg, mean, rstd = ops.gn_lrelu_max_forward(y, g, b, groups, eps, slope)
        ctx.save_for_backward(y, out, arg, mean, rstd, g)
        ctx.groups, ctx.slope = groups, slope
        return out

    @staticmethod
    def backward(ctx, dout):
        y, out, arg, mean, rstd, g = ctx.saved_tensors
        dy, dgamma, dbeta = ops.gn_lrelu_max_backward(y, dout.contiguous().float(), out, arg, mean, rstd, g, ctx.groups, ctx.slope)
        return dy, dgamma, dbeta, None, None, None


def group_norm_lrelu_max(y, gn, slope):
    """y [B,Q,K,C] -> [B,Q,C] for an nn.GroupNorm `gn` over C."""
    return _GroupNormLReLUMax.apply(y, gn.weight, gn.bias, gn.num_groups, gn.eps, slope)


class _GatherAddRows(torch.autograd.Function):
    """y[b,q,j,:] = P[b, idx[b,q,j], :] + Q[b,q,:] (ppt_gather_add) with gradients for P and Q.  The gradient of P is a
    scatter-add over idx: index_put_(accumulate=True), the sort-based deterministic kernel advanced indexing also uses."""

    @staticmethod
    def forward(ctx, P, Q, idx):
        B, S, C = P.shape
        y, _ = ops.gather_add(P.reshape(B * S, C).float().contiguous(), Q.reshape(-1, C).float().contiguous(), idx.contiguous(), S,
                              torch.float32, want_stats=False)
        ctx.save_for_backward(idx)
        ctx.S = S
        return y.view(B, idx.shape[1], idx.shape[2], C)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, Nq, K, C = dy.shape
        dQ = dy.sum(2) if ctx.needs_input_grad[1] else None
        dP = None
        if ctx.needs_input_grad[0]:
            flat = (idx + torch.arange(B, device=idx.device).view(B, 1, 1) * ctx.S).reshape(-1)
            dP = torch.zeros((B * ctx.S, C), dtype=dy.dtype, device=dy.device)
            dP.index_put_((flat,), dy.reshape(-1, C), accumulate=True)
            dP = dP.view(B, ctx.S, C)
        return dP, dQ, None


def gather_add_rows(P, Q, idx):
    """P [B,S,C], Q [B,Nq,C], idx [B,Nq,k] -> [B,Nq,k,C]."""
    return _GatherAddRows.apply(P, Q, idx)


def batch_norm_relu_rows(x, bn, training):
    """F.relu(bn(x)) for x [M,C] and an nn.BatchNorm1d `bn` (momentum must be a number)."""
    return _BatchNormReLURows.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                    bn.num_batches_tracked if training else None, training, bn.momentum, bn.eps)
