"""torch.autograd bridges onto the C-ABI GEMM for layers whose WEIGHTS train (the part-segmentation decoder,
models/pointbert/pointnet2_utils.py:297-467).  The frozen towers do not use these: their forward/backward are
the hand-scheduled pipelines of ppt_amd.engine."""
import os

import torch

from . import gradscale, ops


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
        mult = 8 if prec in ops.HALF else 4
        x2 = x.detach().float()
        w2 = w.detach().reshape(w.shape[0], -1).float()
        xt, wt = _pad_k(x2, mult, prec), _pad_k(w2, mult, prec)
        y = ops.gemm(xt, wt, out_dtype=torch.float32, bias=b.detach().float().contiguous() if b is not None else None)
        ctx.save_for_backward(xt, wt)
        ctx.k, ctx.wshape, ctx.has_bias, ctx.prec = x2.shape[1], w.shape, b is not None, prec
        ctx.grad_scale = gradscale.current(prec, default_rows=x2.shape[0])      # a 16-bit backward stage (ppt_amd/gradscale.py)
        return y

    @staticmethod
    @gradscale.scaled_backward
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


# =================================================================================================
# Fused modules of the part-segmentation decoder: ONE autograd node per reference module, so the tensors between its
# layers exist only in the form their consumer wants -- the operand dtype for the next GEMM, never an fp32 tensor that is
# converted (ppt_convert), padded, concatenated or gathered by ATen kernels on the way.
# =================================================================================================
def _mult(prec):
    return 8 if prec in ops.HALF else 4


def _w_operand(w, prec):
    """Conv / Linear weight [N, K, ...] -> operand copy [N, Kp] in `prec`, K zero-padded to the GEMM's alignment."""
    return _pad_k(w.detach().reshape(w.shape[0], -1).float(), _mult(prec), prec)


def _conv_bn_relu_fwd(xT, w, b, bn, training, prec, out_dtype, wprep=None):
    """relu(bn(conv(x))) over rows (pointnet2_utils.py:362-366): x [M, Kp] in the operand dtype -> (h [M, N] in out_dtype,
    what the backward needs).  The conv output is kept in fp32 (the BatchNorm statistics and its backward read it); its
    chunk statistics come out of the GEMM epilogue.  wprep = (operand copy [N, Kp], its transpose [Kp, N]) when the caller made
    them already (decoder_weight_prep: all of the decoder's weights in one launch)."""
    M = xT.shape[0]
    wT, wTT = wprep if wprep is not None else (_w_operand(w, prec), None)
    N = wT.shape[0]
    st = None
    if training and M % 32 == 0:
        st = (torch.empty((M // 32, N), dtype=torch.float32, device=xT.device), torch.empty((M // 32, N), dtype=torch.float32, device=xT.device))
    y = ops.gemm(xT, wT, out_dtype=torch.float32, bias=b.detach().float().contiguous() if b is not None else None, col_stats=st)
    g, be = bn.weight.detach().float().contiguous(), bn.bias.detach().float().contiguous()
    if training:
        parts, rpp = (st, 32) if st is not None else ops.rows_stats(y)
        sc, sh, mean, rstd = ops.bn_finalize(g, be, True, partials=parts, rows_per_partial=rpp, count=M, running_mean=bn.running_mean,
                                             running_var=bn.running_var, num_batches_tracked=bn.num_batches_tracked, eps=bn.eps,
                                             momentum=bn.momentum, want_moments=True)
    else:
        sc, sh, mean, rstd = ops.bn_finalize(g, be, False, running_mean=bn.running_mean, running_var=bn.running_var, eps=bn.eps,
                                             want_moments=True)
    h = ops.bn_act_rows(y, sc, sh, out_dtype)
    return h, (xT, wT, y, sc, sh, mean, rstd, wTT)


# (Tried: the weight-gradient work of every layer -- dW, db: nothing downstream waits for them -- forked onto a second stream
# inside the capture, so that the hipGraph carries it as a parallel branch beside the dX chain.  The replay of the branched
# graph took 16.5 ms instead of 7.5 ms per part-seg step, and tensors freed on the capturing stream were reused under the
# branch's pending reads.  The backward stays one chain.)
def _conv_bn_relu_bwd(dh, saved, training, prec, w, need_dx):
    """-> (dx [M, Kp] f32 | None, dW like w, db [N], dgamma, dbeta) for _conv_bn_relu_fwd; dh [M, N] f32."""
    xT, wT, y, sc, sh, mean, rstd, wTT = saved
    bf = prec in ops.HALF
    res = ops.bn_rows_backward(dh.contiguous().float(), y, sc, sh, mean, rstd, True, training, want_dx=not bf, want_bf16=bf,
                               half_dtype=prec if bf else torch.bfloat16)
    dgamma, dbeta = res[1], res[2]
    dyT = res[3] if bf else res[0]
    K = w[0].numel()
    dW = ops.gemm_tn_splitk(dyT, xT)
    dW = (dW if dW.shape[1] == K else dW[:, :K]).reshape(w.shape)
    db = ops.col_sums(dyT)
    dx = ops.gemm(dyT, wTT if wTT is not None else ops.transpose(wT), out_dtype=torch.float32) if need_dx else None
    return dx, dW, db, dgamma, dbeta


class _FeaturePropagation(torch.autograd.Function):
    """PointNetFeaturePropagation.forward (pointbert/pointnet2_utils.py:310-368) after the 3-NN search: interpolation +
    concatenation (ppt_three_nn_interp_fwd writes the first conv's operand), two conv + BatchNorm + ReLU layers, and the whole
    backward (weights, biases, BatchNorm affine, and the interpolated features through ppt_scatter_rows_bwd)."""

    @staticmethod
    def forward(ctx, points2, w0, b0, g0, be0, w1, b1, g1, be1, mod, idx, dist, points1, prec):
        training = mod.training
        B, N, _ = idx.shape
        p2 = points2.detach().float().contiguous()
        p1 = points1.detach().float().contiguous() if points1 is not None else None
        rows, wgt = ops.three_nn_interp(p1, p2, idx, dist, prec, _mult(prec))
        wp = getattr(ctx, "wprep", None) or (None, None)
        h0, s0 = _conv_bn_relu_fwd(rows, w0, b0, mod.mlp_bns[0], training, prec, prec, wprep=wp[0])
        h1, s1 = _conv_bn_relu_fwd(h0, w1, b1, mod.mlp_bns[1], training, prec, torch.float32, wprep=wp[1])
        ctx.s0, ctx.s1, ctx.idx, ctx.wgt = s0, s1, idx, wgt
        ctx.training, ctx.prec, ctx.w0, ctx.w1 = training, prec, w0, w1
        ctx.D1, ctx.S, ctx.D2 = (0 if p1 is None else p1.shape[2]), p2.shape[1], p2.shape[2]
        ctx.grad_scale = gradscale.current(prec, default_rows=B * N)           # a 16-bit backward stage (ppt_amd/gradscale.py)
        return h1.view(B, N, -1)

    @staticmethod
    @gradscale.scaled_backward
    def backward(ctx, dout):
        B, N, C = dout.shape
        dh0, dW1, db1, dg1, dbe1 = _conv_bn_relu_bwd(dout.reshape(B * N, C), ctx.s1, ctx.training, ctx.prec, ctx.w1, True)
        d_rows, dW0, db0, dg0, dbe0 = _conv_bn_relu_bwd(dh0, ctx.s0, ctx.training, ctx.prec, ctx.w0, ctx.needs_input_grad[0])
        dp2 = None
        if ctx.needs_input_grad[0]:
            dp2 = ops.scatter_rows_bwd(ctx.idx.view(B, N * 3), ctx.wgt.view(B, N * 3), d_rows, ctx.D1, 3, ctx.S, ctx.D2)
        return dp2, dW0, db0, dg0, dbe0, dW1, db1, dg1, dbe1, None, None, None, None, None


def feature_propagation(mod, points1, points2, idx, dist, prec):
    """mod: a PointNetFeaturePropagation with two conv + bn layers; idx / dist [B,N,3] from the 3-NN search."""
    c0, c1, n0, n1 = mod.mlp_convs[0], mod.mlp_convs[1], mod.mlp_bns[0], mod.mlp_bns[1]
    return _FeaturePropagation.apply(points2, c0.weight, c0.bias, n0.weight, n0.bias, c1.weight, c1.bias, n1.weight, n1.bias,
                                     mod, idx, dist, points1, prec)


class _ConvBNReLURows(torch.autograd.Function):
    """relu(bn(conv(x))) for fp32 rows x [M,K] (the decoder's conv1 + bn1, point_encoder.py:414-416) as one node."""

    @staticmethod
    def forward(ctx, x, w, b, g, be, conv_bn, prec):
        bn, training = conv_bn
        xT = _pad_k(x.detach().float().contiguous(), _mult(prec), prec)
        h, s = _conv_bn_relu_fwd(xT, w, b, bn, training, prec, torch.float32, wprep=getattr(ctx, "wprep", None))
        ctx.s, ctx.training, ctx.prec, ctx.w, ctx.K = s, training, prec, w, x.shape[1]
        ctx.grad_scale = gradscale.current(prec, default_rows=x.shape[0])      # a 16-bit backward stage (ppt_amd/gradscale.py)
        return h

    @staticmethod
    @gradscale.scaled_backward
    def backward(ctx, dh):
        dx, dW, db, dg, dbe = _conv_bn_relu_bwd(dh, ctx.s, ctx.training, ctx.prec, ctx.w, ctx.needs_input_grad[0])
        if dx is not None and dx.shape[1] != ctx.K:
            dx = dx[:, :ctx.K]
        return dx, dW, db, dg, dbe, None, None


def conv_bn_relu_rows(x, conv, bn, training, prec):
    return _ConvBNReLURows.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, (bn, training), prec)


class _DGCNNLayer(torch.autograd.Function):
    """One layer of DGCNN_Propagation (pointbert/pointnet2_utils.py:404-440): conv(cat(x_k[nn] - x_q, x_q)) + GroupNorm +
    LeakyReLU + max over the k neighbours.  By linearity of the 1x1 conv, W = [Wa | Wb]:
        Wa.(x_j - x_q) + Wb.x_q = Wa.x_j + (Wb - Wa).x_q
    so the conv runs once per source point (P) and once per query (Q) and ppt_gather_add forms the [B,Nq,k,Cout] rows.
    Backward: GroupNorm kernels -> dy; dP by the owner-computes scatter (ppt_scatter_rows_bwd), dQ = sum over k
    (ppt_sum_groups); dWa = dP^T x_k, d(Wb - Wa) = dQ^T x_q on the NT weight-gradient GEMM; input gradients by two GEMMs."""

    @staticmethod
    def forward(ctx, x_k, x_q, w, gn_w, gn_b, idx, gn, slope, prec):
        B, S, C = x_k.shape
        Nq = x_q.shape[1]
        K = idx.shape[2]
        w2 = w.detach().reshape(w.shape[0], -1).float()
        Cout = w2.shape[0]
        wp = getattr(ctx, "wprep", None)
        if wp is not None:
            (waT, waTT), (wdT, wdTT) = wp
        else:
            waT = _pad_k(w2[:, :C], _mult(prec), prec)
            wdT = _pad_k(w2[:, C:] - w2[:, :C], _mult(prec), prec)
            waTT = wdTT = None
        xkT = _pad_k(x_k.detach().reshape(B * S, C).float(), _mult(prec), prec)
        xqT = xkT if x_q is x_k else _pad_k(x_q.detach().reshape(B * Nq, C).float(), _mult(prec), prec)
        P = ops.gemm(xkT, waT, out_dtype=torch.float32)
        Q = ops.gemm(xqT, wdT, out_dtype=torch.float32)
        y, _ = ops.gather_add(P, Q, idx.contiguous(), S, torch.float32, want_stats=False)
        y = y.view(B, Nq, K, Cout)
        g, b = gn_w.detach().float().contiguous(), gn_b.detach().float().contiguous()
        out, arg, mean, rstd = ops.gn_lrelu_max_forward(y, g, b, gn.num_groups, gn.eps, slope)
        ctx.saved = (xkT, xqT, waT, wdT, y, out, arg, mean, rstd, g, idx, waTT, wdTT)
        ctx.groups, ctx.slope, ctx.prec, ctx.w, ctx.C, ctx.S, ctx.same = gn.num_groups, slope, prec, w, C, S, x_q is x_k
        ctx.grad_scale = gradscale.current(prec, default_rows=B * Nq)          # a 16-bit backward stage (ppt_amd/gradscale.py)
        return out

    @staticmethod
    @gradscale.scaled_backward
    def backward(ctx, dout):
        xkT, xqT, waT, wdT, y, out, arg, mean, rstd, g, idx, waTT, wdTT = ctx.saved
        B, Nq, K, Cout = y.shape
        C, prec = ctx.C, ctx.prec
        dy, dgamma, dbeta = ops.gn_lrelu_max_backward(y, dout.contiguous().float(), out, arg, mean, rstd, g, ctx.groups, ctx.slope)
        dy2 = dy.view(B * Nq * K, Cout)
        dP = ops.scatter_rows_bwd(idx.view(B, Nq * K), None, dy2, 0, 1, ctx.S, Cout).view(B * ctx.S, Cout)
        dQ = ops.sum_groups(dy2, K)
        dPT, dQT = ops.convert(dP, prec), ops.convert(dQ, prec)
        dWa = ops.gemm_tn_splitk(dPT, xkT)[:, :C]
        dWd = ops.gemm_tn_splitk(dQT, xqT)[:, :C]
        dW = torch.cat([dWa - dWd, dWd], dim=1).reshape(ctx.w.shape)
        dxk = dxq = None
        if ctx.needs_input_grad[0]:
            dxk = ops.gemm(dPT, waTT if waTT is not None else ops.transpose(waT), out_dtype=torch.float32)[:, :C].reshape(B, ctx.S, C)
        if ctx.needs_input_grad[1]:
            dxq = ops.gemm(dQT, wdTT if wdTT is not None else ops.transpose(wdT), out_dtype=torch.float32)[:, :C].reshape(B, Nq, C)
        return dxk, dxq, dW, dgamma, dbeta, None, None, None, None


def dgcnn_layer(x_k, x_q, conv, gn, idx, slope, prec):
    """x_k [B,S,C] sources, x_q [B,Nq,C] queries, idx [B,Nq,k] neighbours of each query among the sources -> [B,Nq,Cout]."""
    return _DGCNNLayer.apply(x_k, x_q, conv.weight, gn.weight, gn.bias, idx, gn, slope, prec)


# ---------------------------------------------------------------------------------------------------------------------
# The whole part-segmentation decoder as ONE autograd node whose forward and backward are hipGraph replays.
#
# The decoder's step is ~450 launches of small kernels issued from Python autograd nodes, and enqueueing them is what bounds the
# part-seg step (tools/host_time.py).  torch.cuda.make_graphed_callables cannot be used (capturing an autograd backward
# segfaults in hipStreamEndCapture in this build), so the backward is hand-scheduled here: the per-module nodes above already
# ARE explicit forward / backward pairs, and this node calls their static methods with stand-in contexts in topological /
# reverse order and routes the gradients between them itself -- plain functions of tensors that graphs.GraphedCall captures.
# Same kernels in the same order as the per-module path: bit-identical results (tests/test_model_gpu.py).
class _Ctx:
    """Stand-in for the autograd context of the per-module nodes: attributes + needs_input_grad."""

    def __init__(self, needs):
        self.needs_input_grad = needs


def partseg_decoder_params(pe):
    """The trainable tensors the decoder node returns gradients for, in its fixed order."""
    ps = []
    for fp in (pe.propagation_0, pe.propagation_1, pe.propagation_2):
        for conv, bn in zip(fp.mlp_convs, fp.mlp_bns):
            ps += [conv.weight, conv.bias, bn.weight, bn.bias]
    for dg in (pe.dgcnn_pro_1, pe.dgcnn_pro_2):
        for layer in (dg.layer1, dg.layer2):
            ps += [layer[0].weight, layer[1].weight, layer[1].bias]
    return ps + [pe.conv1.weight, pe.conv1.bias, pe.bn1.weight, pe.bn1.bias]


DECODER_WPREP = os.environ.get("PPT_DECODER_WPREP", "1") != "0"          # 0: per-weight conversions (A/B runs)
DECODER_WPREP_F32 = os.environ.get("PPT_DECODER_WPREP_F32", "1") != "0"  # the same single launch in the fp32 / split16 modes (copies: same bits)


def decoder_weight_prep(pe):
    """The 16-bit operand copy AND its transpose of every conv weight of the decoder -- 6 feature-propagation convs, the Wa /
    (Wb - Wa) halves of the 4 DGCNN convs, conv1: 15 pairs -- in ONE launch (ppt_weights_prep) instead of ~41: per weight and step
    the per-module path pays a zero-fill + padded copy or a slice copy + convert (+ a subtraction for Wb - Wa) in the forward and a
    transpose in the backward, and these weights all train, so nothing can be cached across steps.  Same values bit for bit.
    -> {key: (wT [N, Kp], wTT [Kp, N])}, keys (module, layer index) / (module, layer, 'a' | 'd'); None in the fp32 mode."""
    prec = pe._dec_precision
    if not DECODER_WPREP or not (prec in ops.HALF or (prec == torch.float32 and DECODER_WPREP_F32)):
        return None
    mult = _mult(prec)
    keys, items = [], []

    def add(key, w, col0, K, sub):
        w2 = w.detach().reshape(w.shape[0], -1)
        if w2.dtype != torch.float32 or not w2.is_contiguous():
            w2 = w2.float().contiguous()
        keys.append(key)
        items.append((w2, col0, K, sub, (K + mult - 1) // mult * mult))
    for fp in (pe.propagation_0, pe.propagation_1, pe.propagation_2):
        for li, conv in enumerate(fp.mlp_convs):
            add((fp, li), conv.weight, 0, conv.weight.shape[1], None)
    for dg in (pe.dgcnn_pro_1, pe.dgcnn_pro_2):
        for li, layer in enumerate((dg.layer1, dg.layer2)):
            C = layer[0].weight.shape[1] // 2
            add((dg, li, 'a'), layer[0].weight, 0, C, None)             # Wa
            add((dg, li, 'd'), layer[0].weight, C, C, 0)                # Wb - Wa
    add((pe, 'conv1'), pe.conv1.weight, 0, pe.conv1.weight.shape[1], None)
    return dict(zip(keys, ops.weights_prep(items, prec)))


def _fp_forward(mod, xyz1, xyz2, points1, points2, need_dp2, prep=None):
    idx, _, d = ops.knn_group(xyz2.contiguous().float(), xyz1.contiguous().float(), 3, want_nbhd=False, want_dist=True)
    c = _Ctx((need_dp2,))
    if prep is not None:
        c.wprep = (prep[(mod, 0)], prep[(mod, 1)])
    c0, c1, n0, n1 = mod.mlp_convs[0], mod.mlp_convs[1], mod.mlp_bns[0], mod.mlp_bns[1]
    out = _FeaturePropagation.forward(c, points2, c0.weight, c0.bias, n0.weight, n0.bias, c1.weight, c1.bias, n1.weight, n1.bias,
                                      mod, idx, d, points1, mod.precision)
    return out, c


def _dg_forward(mod, seq, coor_q, x_q, coor_k, x_k, need_k, need_q, prep=None):
    idx, _ = ops.knn_group(coor_k.contiguous(), coor_q.contiguous(), mod.k, want_nbhd=False)
    c = _Ctx((need_k, need_q))
    if prep is not None:
        li = 0 if seq is mod.layer1 else 1
        c.wprep = (prep[(mod, li, 'a')], prep[(mod, li, 'd')])
    out = _DGCNNLayer.forward(c, x_k, x_q, seq[0].weight, seq[1].weight, seq[1].bias, idx, seq[1], 0.2, mod.precision)
    return out, c


def partseg_decoder_forward(pe, f_a, f_b, f_c, center, c1, c2, pts, cls_label, drop):
    """point_encoder.py:396-416 on rows -> (y [B,N,128], contexts for partseg_decoder_backward).  drop: multiplicative Dropout
    factors [B,N,128] or None."""
    B, N, _ = pts.shape
    f0 = torch.cat([cls_label.float().view(B, 1, 16).expand(-1, N, -1), pts], dim=-1)
    prep = decoder_weight_prep(pe)               # every weight's operand copy + transpose: one launch
    F2, k1 = _fp_forward(pe.propagation_2, c2, center, c2, f_b, False, prep)
    F1, k2 = _fp_forward(pe.propagation_1, c1, center, c1, f_a, False, prep)
    L1, k3a = _dg_forward(pe.dgcnn_pro_2, pe.dgcnn_pro_2.layer1, c2, F2, center, f_c, False, True, prep)
    L2, k3b = _dg_forward(pe.dgcnn_pro_2, pe.dgcnn_pro_2.layer2, c2, L1, c2, L1, True, True, prep)
    L3, k4a = _dg_forward(pe.dgcnn_pro_1, pe.dgcnn_pro_1.layer1, c1, F1, c2, L2, True, True, prep)
    L4, k4b = _dg_forward(pe.dgcnn_pro_1, pe.dgcnn_pro_1.layer2, c1, L3, c1, L3, True, True, prep)
    F0, k5 = _fp_forward(pe.propagation_0, pts, c1, f0, L4, True, prep)
    k6 = _Ctx((True,))
    if prep is not None:
        k6.wprep = prep[(pe, 'conv1')]
    y = _ConvBNReLURows.forward(k6, F0.reshape(B * N, -1), pe.conv1.weight, pe.conv1.bias, pe.bn1.weight, pe.bn1.bias,
                                (pe.bn1, pe.training), pe._dec_precision).view(B, N, -1)
    if drop is not None:
        y = y * drop
    return y, (k1, k2, k3a, k3b, k4a, k4b, k5, k6, drop, (B, N))


def partseg_decoder_backward(ctxs, dy, grad_scale=1.0):
    """dy [B,N,128] -> the gradients of partseg_decoder_params(pe), in that order.
    grad_scale = S: the whole decoder is ONE gradient-scaled stage (ppt_amd/gradscale.py) -- dy x S on the way in, the 47
    parameter gradients x 1 / S in one multi-tensor launch on the way out; the per-module backwards run un-wrapped (`.raw`)."""
    k1, k2, k3a, k3b, k4a, k4b, k5, k6, drop, (B, N) = ctxs
    S = float(grad_scale)
    dy = dy.float()
    if drop is not None:
        dy = dy * (drop if S == 1.0 else drop * S)
    elif S != 1.0:
        dy = dy * S
    fp_bwd, dg_bwd, cv_bwd = _FeaturePropagation.backward.raw, _DGCNNLayer.backward.raw, _ConvBNReLURows.backward.raw
    g6 = cv_bwd(k6, dy.reshape(B * N, -1).contiguous())
    g5 = fp_bwd(k5, g6[0].reshape(B, N, -1))                                             # -> d L4 (points2 of propagation_0)
    g4b = dg_bwd(k4b, g5[0])
    g4a = dg_bwd(k4a, g4b[0] + g4b[1])                                                    # L3 is both operands of layer 2
    g2 = fp_bwd(k2, g4a[1])                                                               # F1 = propagation_1's output
    g3b = dg_bwd(k3b, g4a[0])                                                             # L2 = dgcnn_pro_2's output
    g3a = dg_bwd(k3a, g3b[0] + g3b[1])
    g1 = fp_bwd(k1, g3a[1])                                                               # F2 = propagation_2's output
    out = list(g5[1:9]) + list(g2[1:9]) + list(g1[1:9])
    for gd in (g4a, g4b, g3a, g3b):
        out += [gd[2], gd[3], gd[4]]
    out = out + [g6[1], g6[2], g6[3], g6[4]]
    if S != 1.0:
        out = [o.contiguous() for o in out]            # (a weight gradient may be a column slice of its padded GEMM result)
        gradscale.unscale_(out, S)
    return out


STATIC_GRADS_OK = False          # set by train.Trainer.step around loss.backward(): see _PartsegDecoder.backward


class _PartsegDecoder(torch.autograd.Function):
    """forward(pe, f_a, f_b, f_c, center, c1, c2, pts, cls_label, drop, *partseg_decoder_params(pe)) -> y [B,N,128]; forward and
    backward are replayed from hipGraphs (captured at the first call per shape; activations live in the forward graph's pool)."""

    @staticmethod
    def forward(ctx, pe, f_a, f_b, f_c, center, c1, c2, pts, cls_label, drop, *params):
        from . import graphs
        ins = [f_a, f_b, f_c, center, c1, c2, pts.contiguous().float(), cls_label.contiguous().float()]
        key = ("partseg_decoder", tuple(pts.shape), pe.training, pe._precision, drop is not None)

        def build():
            def fn(*t):
                d = drop if drop is not None else ((torch.rand((t[6].shape[0], t[6].shape[1], 128), device=t[6].device) >= 0.5).float() * 2.0
                                                   if pe.training else None)                 # nn.Dropout(0.5), point_encoder.py:417
                y, ctxs = partseg_decoder_forward(pe, *t, d)
                return (y,), ctxs
            return graphs.GraphedCall(fn, ins)
        g = pe._graphs.get(key, build)
        (y,), ctxs = g(*ins)
        g.generation = getattr(g, "generation", 0) + 1
        ctx.pe, ctx.g, ctx.key, ctx.generation, ctx.ctxs = pe, g, key, g.generation, ctxs
        ctx.grad_scale = gradscale.current(pe._dec_precision, default_rows=pts.shape[0] * pts.shape[1])
        return y.clone()

    @staticmethod
    def backward(ctx, dy):
        from . import graphs
        if ctx.g.generation != ctx.generation:
            raise RuntimeError("the part-seg decoder's captured activations were overwritten by a later forward; set "
                               "point_encoder.use_hip_graphs = False to keep several forwards alive before backward")
        ctxs, fwd = ctx.ctxs, ctx.g
        S = ctx.grad_scale

        def build():
            def fn(d):
                return tuple(partseg_decoder_backward(ctxs, d, grad_scale=S)), None
            return graphs.GraphedCall(fn, [dy.contiguous()], pool=fwd.pool())
        grads, _ = ctx.pe._graphs.get(("partseg_decoder_bwd",) + ctx.key[1:] + (S,), build)(dy.contiguous())
        # Inside train.Trainer.step (STATIC_GRADS_OK): fresh tensor objects over the graph's static buffers, which autograd adopts
        # as .grad without a copy -- the step drops every .grad before each backward and its optimizer reads them before the next
        # replay overwrites them.  Any other caller (gradient accumulation, a custom loop holding p.grad) gets copies: an adopted
        # alias would be overwritten by the next replay and then added to itself.
        if STATIC_GRADS_OK:
            return (None,) * 10 + tuple(gr.detach() for gr in grads)
        return (None,) * 10 + tuple(gr.clone() for gr in grads)
