"""Hand-scheduled kernel pipelines for the two towers of ULIP/PointBERT.

Instead of an autograd graph of hundreds of tiny ops (the reference) or a tracing compiler, the
forward AND backward of each tower are explicit Python sequences of C-ABI kernel launches
(ppt_amd.ops) over caller-visible tensors; torch.autograd sees ONE node per tower
(ppt_amd.models.*).  All launches go to the current stream, nothing synchronises, so a whole
training step can be captured in a hipGraph.

dtype T is the operand dtype of the MFMA kernels: torch.bfloat16 (performance mode) or
torch.float32 (parity mode).  The residual streams, LayerNorm statistics, BatchNorm statistics and
all gradients that are accumulated stay fp32 in both modes.

Reference citations are relative to the upstream repo.
"""
import os

import torch

from . import ops
from .ops import ACT_GELU, ACT_NONE, ACT_QUICKGELU, A_AFFINE_RELU, A_CONV1

ATTN_SCALE = 64 ** -0.5


class WeightCache:
    """Operand-dtype / pre-transposed copies of parameters.  Frozen parameters are converted once;
    a trainable parameter is re-converted when the optimiser has bumped its version counter."""

    def __init__(self, dtype, demoted=None):
        self.dtype = dtype
        self._c = {}
        # stages of the OWNING model that ppt_amd/health.py moved from IEEE half back to bf16 at run time (the owner's set,
        # shared by reference: per model, not per process -- ADVICE r4)
        self.demoted = demoted if demoted is not None else set()

    def get(self, t, kind="w", cols=None, pad_to=None):
        """kind 'w': [N,K] operand copy (optionally a column slice; K zero-padded to a multiple of pad_to) in
        self.dtype; 'wt': transposed [K,N] copy (the B operand of dX = dY @ W); 'f32': fp32 2-D view."""
        key = (id(t), kind, cols, pad_to)
        ent = self._c.get(key)
        if ent is not None and ent[0] is t and ent[1] == t._version:
            return ent[2]
        with torch.no_grad():
            w = t.detach()
            w = w.reshape(w.shape[0], -1)                   # Conv1d [out,in,1] -> [out,in]
            if cols is not None:
                w = w[:, cols[0]:cols[1]].contiguous()
            if pad_to is not None and w.shape[1] % pad_to:
                wp = torch.zeros((w.shape[0], (w.shape[1] + pad_to - 1) // pad_to * pad_to), dtype=w.dtype, device=w.device)
                wp[:, :w.shape[1]] = w
                w = wp
            if kind == "w":
                out = ops.convert(w.contiguous(), self.dtype)
            elif kind == "wt":
                out = ops.transpose(w.contiguous(), self.dtype)
            elif kind == "f32":
                out = w.contiguous()
            else:
                raise ValueError(kind)
        self._c[key] = (t, t._version, out)
        return out

    def derived(self, key, tensors, fn):
        """A value computed from several parameters (folded weights ...): rebuilt when any of them changed."""
        stamp = tuple((id(t), t._version) for t in tensors)
        ent = self._c.get(("derived", key))
        if ent is not None and ent[0] == stamp:
            return ent[2]
        with torch.no_grad():
            out = fn()
        self._c[("derived", key)] = (stamp, tensors, out)
        return out


# Diagnostic (tools/bf16_error.py): operand precision of ONE stage of the point tower overridden -- {"tokenizer" | "blocks" (0 ..
# depth-2) | "last_block": torch.float32 / torch.bfloat16} -- to attribute the bf16 mode's error to stages.  Eager execution only
# (the hipGraph keys do not carry it); empty in production.
STAGE_DTYPE = {}


# The transformer blocks' operand format in the performance mode is IEEE half (PPT_BLOCKS_F16=0: bf16, as the tokenizer keeps):
# the same MFMA rate with 11 significand bits instead of 8.  Their activations are LayerNorm outputs, projections of them,
# softmax weights and GELU outputs (the residual stream is fp32) -- far inside fp16's range -- and tools/bf16_error.py puts what
# is left of the logits / gradient error after the text tower moved to fp16 (ULIP_WITH_IMAGE._cache) in blocks 0-10.
BLOCKS_F16 = os.environ.get("PPT_BLOCKS_F16", "1") != "0"
# ... and the PointBERT tokenizer (mini-PointNet + reduce_dim + pos_embed; PPT_TOKENIZER_F16=0: bf16): with text tower and blocks
# on fp16 it is what remains (logits 0.10 -> 0.04 max, last-block weight gradients 2 % -> 0.4 % with it in fp32).  Its stored
# activations are raw Conv1d outputs in front of a BatchNorm (y2, y3) and group maxima: |values| of order 1-100, fp16's range is
# 65 504; the conv1 output is rebuilt in fp32 from the coordinates and never stored.
TOKENIZER_F16 = os.environ.get("PPT_TOKENIZER_F16", "1") != "0"
# ... and the GEMM operands of the part-segmentation decoder (feature propagation, DGCNN propagation, conv1; PPT_DECODER_F16=0:
# bf16).  Its activations stay fp32 between layers either way; what changes is the rounding of each GEMM's two operands
# (8 -> 11 significand bits) in the forward and of dY in the backward.
DECODER_F16 = os.environ.get("PPT_DECODER_F16", "1") != "0"


def _stage_wc(wc, stage):
    """The WeightCache a stage runs with: `wc`, or a sibling of another operand precision (kept on `wc`).  A stage that the
    owning model's health monitor demoted (WeightCache.demoted; ppt_amd/health.py) stays on bf16."""
    dt = STAGE_DTYPE.get(stage)
    if dt is None and wc.dtype == torch.bfloat16 and stage not in wc.demoted:
        if (BLOCKS_F16 and stage in ("blocks", "last_block")) or (TOKENIZER_F16 and stage == "tokenizer"):
            dt = torch.float16
    if dt is None or dt == wc.dtype:
        return wc
    alts = wc.__dict__.setdefault("_alts", {})
    if dt not in alts:
        alts[dt] = WeightCache(dt, wc.demoted)
    return alts[dt]


FUSED_CONV12 = os.environ.get("PPT_FUSED_CONV12", "1") != "0"      # 0: the generic PPT_A_CONV1 GEMM (A/B comparisons)
# conv3 + BN + ReLU + conv4 + max as ONE kernel (csrc/mpn34.hip).  Alone on the chip at C2's size (524 288 points,
# tools/mpn34_bench.py): 283 us = 1.0 PFLOP/s executed, against 466 us for conv3 (y3 written) + conv4 (y3 read back).
#   eval (BatchNorm is a constant affine): ON -- validate() 2.425 -> 2.363 ms per batch of 32 (the tokenizer runs beside the
#     previous batch's blocks, so most of the 183 us were hidden already);
#   train: OFF -- the batch statistics need a conv3 pass of their own in front (mpn3 without its store: 178 us), 461 us
#     against 466 alone, and in the C2 step 3.305 against 3.220 ms (tower 2.764 / 2.679, same-box A/B tools/ab_env.py): the
#     137 GFLOP computed twice cost more beside the prompt chain than the 1.1 GB of HBM traffic they save.  PPT_FUSED_CONV34_TRAIN=1.
FUSED_CONV34 = os.environ.get("PPT_FUSED_CONV34", "1") != "0"
FUSED_CONV34_TRAIN = os.environ.get("PPT_FUSED_CONV34_TRAIN", "0") != "0"
# csrc/rowgemm.hip (weight-stationary K = 384 linears with the LayerNorm applied while the rows are staged) from this many
# token rows on.  Measured (same box, graph-replayed steps): C3 (32 832 rows) 7.60 -> 7.18 ms per step, the three linears
# 11-27 % faster each; C2 (16 416 rows) 3.99 -> 4.17 ms although two of the three are ~10 % faster in isolation -- its
# workgroups take a whole CU each (8 waves x 256 VGPRs), and with only 7 row tiles per workgroup to amortise the weight
# preload the small win is paid back by the text tower's kernels, which can no longer share those CUs.  PPT_ROWGEMM=0 / 1
# forces it off / on.  (Round 3, after the LDS-DMA fix of ppt_common.h made the prompt chain's GEMMs 6-23 % faster: C2 3.422 ->
# 3.345 ms WITH it, same box, tools/ab_env.py -- the threshold went from 24 000 to 16 000 rows.)
FUSED_MLP = os.environ.get("PPT_FUSED_MLP", "1") != "0"             # LayerNorm + fc1 + GELU + fc2 + residual of a frozen block: one kernel
FUSED_PROJ = os.environ.get("PPT_FUSED_PROJ", "1") != "0"           # ... with attn.proj + DropPath + residual in front of it (rowgemm path)
# which kernel each linear of a FROZEN ViT block runs on from ROWGEMM_MIN_ROWS token rows on (vit_block_forward):
# PPT_BLOCK_QKV = rowgemm | v2, PPT_BLOCK_PROJ = fused | rowgemm | v2, PPT_BLOCK_MLP = fused | rowgemm | v2
# (round 6: qkv "auto" = csrc/lnlin.hip -- rows stationary, weight streamed, two workgroups per CU -- up to LNLIN_MAX_ROWS token rows,
# csrc/rowgemm.hip above: alone 33.6 -> 28.0 us at 16 416 rows and a tie (51 us) at 32 832; in the step C2 2.688 -> 2.639 ms with the
# prompt chain critical (its workgroups do not take whole CUs), C3 5.631 -> 5.678 ms)
LNLIN_MAX_ROWS = int(os.environ.get("PPT_LNLIN_MAX_ROWS", "24576"))
BLOCK_PATH = {"qkv": os.environ.get("PPT_BLOCK_QKV", "auto"), "proj": os.environ.get("PPT_BLOCK_PROJ", "fused"),
              "mlp": os.environ.get("PPT_BLOCK_MLP", "fused")}
_RG = os.environ.get("PPT_ROWGEMM", "")
ROWGEMM_MIN_ROWS = 1 << 30 if _RG == "0" else (0 if _RG == "1" else 16000)


def _bn_params(sd, p):
    return (sd[p + "weight"], sd[p + "bias"], sd[p + "running_mean"], sd[p + "running_var"],
            sd[p + "num_batches_tracked"])


# =================================================================================================
# point branch
# =================================================================================================
def group_points(pc, num_group, group_size, fps_start):
    """Group.forward (dvae.py:159-181): FPS centres + kNN neighbourhoods (centre-subtracted)."""
    _, center = ops.fps(pc, num_group, fps_start)
    _, nbhd = ops.knn_group(pc, center, group_size, want_idx=False)
    return nbhd, center


def mini_pointnet(sd, p, wc, nbhd, bn_train, update_running=True):
    """Encoder.forward (dvae.py:201-215): [B,G,n,3] -> [B*G,256] in wc.dtype.  n must be 32
    (one MFMA row-tile == one group, pooled in the GEMM epilogue)."""
    B, G, n, _ = nbhd.shape
    assert n == 32, "the fused max-pool epilogue assumes group_size == 32"
    T = wc.dtype
    M = B * G * n
    pts = nbhd.view(M, 3)
    dev = pts.device
    w1 = wc.get(sd[p + "first_conv.0.weight"], "f32")
    b1 = sd[p + "first_conv.0.bias"]
    g1, be1, rm1, rv1, nb1 = _bn_params(sd, p + "first_conv.1.")
    if bn_train:
        ps, pq, rpp = ops.conv1_stats(pts, w1, b1)
        sc1, sh1 = ops.bn_finalize(g1, be1, True, partials=(ps, pq), rows_per_partial=rpp, count=M,
                                   running_mean=rm1, running_var=rv1, num_batches_tracked=nb1,
                                   update_running=update_running)
    else:
        sc1, sh1 = ops.bn_finalize(g1, be1, False, running_mean=rm1, running_var=rv1)
    # conv1 + BN1 + ReLU live in the A-prologue of the conv2 GEMM; epilogue: +bias, group max
    w2 = wc.get(sd[p + "first_conv.3.weight"])
    if T in ops.HALF and FUSED_CONV12 and tuple(w2.shape) == (256, 128) and w2.stride(0) == 128:
        # the dedicated kernel (csrc/mpn1.hip): same arithmetic, no tile staging (326 -> ~100 us for 524 288 points)
        y2, gmax = ops.mini_pointnet_conv12(pts, w1, b1, sc1, sh1, w2, sd[p + "first_conv.3.bias"])
    else:
        gmax = torch.empty((M // 32, 256), dtype=T, device=dev)
        y2 = ops.gemm(None, w2, out_dtype=T, a_mode=A_CONV1, pts=pts, w1=w1, b1=b1,
                      a_scale=sc1, a_shift=sh1, bias=sd[p + "first_conv.3.bias"], pool_max=gmax)
    # cat([global, local]) @ W3^T  ==  local @ W3[:,256:]^T + (global @ W3[:,:256]^T + b3) per group
    w3 = sd[p + "second_conv.0.weight"]
    g2, be2, rm2, rv2, nb2 = _bn_params(sd, p + "second_conv.1.")
    w4p = sd[p + "second_conv.3.weight"]
    fused34 = (FUSED_CONV34 and T in ops.HALF and FUSED_CONV12 and tuple(w3.shape[:2]) == (512, 512) and tuple(w4p.shape[:2]) == (256, 512)
               and y2.is_contiguous() and (not bn_train or FUSED_CONV34_TRAIN))
    if fused34 and not bn_train:
        # eval (validate(), main_cls.py:237-299 -- SURVEY 8(f) N1): BatchNorm 2 is a constant affine and is folded INTO conv3:
        # W3s = scale o W3, b3s = scale * b3 + shift (two small launches per forward -- not cached: the running statistics are
        # written through raw pointers and by hipGraph replays, which no version counter sees).  The group term then comes out of
        # its GEMM as the per-group bias gs, and conv3 + BN + ReLU + conv4 + max is one kernel (csrc/mpn34.hip): y3 never exists.
        sc2, sh2 = ops.bn_finalize(g2, be2, False, running_mean=rm2, running_var=rv2)
        w3f = wc.get(w3, "f32")
        w3s_a, b3s = ops.scale_rows_convert(w3f, sc2, T, cols=(0, 256), bias=sd[p + "second_conv.0.bias"], shift=sh2)
        w3s_b = ops.scale_rows_convert(w3f, sc2, T, cols=(256, 512))
        gs = ops.gemm(gmax, w3s_a, out_dtype=torch.float32, bias=b3s, algo_k=0)
        w4t = wc.derived(("mpn34_w4t", p), (w4p,), lambda: ops.mpn34_retile(wc.get(w4p)))
        return ops.mini_pointnet_conv34(y2, w3s_b, gs, w4t, sd[p + "second_conv.3.bias"])
    gterm = ops.gemm(gmax, wc.get(w3, cols=(0, 256)), out_dtype=torch.float32, bias=sd[p + "second_conv.0.bias"],
                     algo_k=0)      # its FLOPs are accounted to the 512-wide conv3 (algo_k=512 below)
    w3b = wc.get(w3, cols=(256, 512))
    fused3 = (T in ops.HALF and FUSED_CONV12 and tuple(w3b.shape) == (512, 256) and w3b.stride(0) == 256
              and y2.is_contiguous() and gterm.is_contiguous())          # csrc/mpn3.hip: W3b in registers, rows read once
    if fused34 and fused3:
        # train: the batch statistics of y3 need every row before any row can be normalised -- a statistics pass that computes
        # conv3 WITHOUT writing it (mpn3.hip, store = False), then the fused kernel on the folded weights of THIS batch:
        # 137 GFLOP recomputed (C2) against 1.1 GB of HBM traffic saved
        cs = torch.empty((M // 32, 512), dtype=torch.float32, device=dev)
        cq = torch.empty_like(cs)
        ops.mini_pointnet_conv3(y2, w3b, gterm, (cs, cq), store=False)
        sc2, sh2 = ops.bn_finalize(g2, be2, True, partials=(cs, cq), rows_per_partial=32, count=M,
                                   running_mean=rm2, running_var=rv2, num_batches_tracked=nb2,
                                   update_running=update_running)
        w3s_b = ops.scale_rows_convert(wc.get(w3, "f32"), sc2, T, cols=(256, 512))
        gs = torch.addcmul(sh2, gterm, sc2)
        w4t = wc.derived(("mpn34_w4t", p), (w4p,), lambda: ops.mpn34_retile(wc.get(w4p)))
        return ops.mini_pointnet_conv34(y2, w3s_b, gs, w4t, sd[p + "second_conv.3.bias"])
    if bn_train:
        cs = torch.empty((M // 32, 512), dtype=torch.float32, device=dev)
        cq = torch.empty_like(cs)
        if fused3:
            y3 = ops.mini_pointnet_conv3(y2, w3b, gterm, (cs, cq))
        else:
            y3 = ops.gemm(y2, w3b, out_dtype=T, group_add=gterm, group_rows=32, col_stats=(cs, cq), algo_k=512)
        sc2, sh2 = ops.bn_finalize(g2, be2, True, partials=(cs, cq), rows_per_partial=32, count=M,
                                   running_mean=rm2, running_var=rv2, num_batches_tracked=nb2,
                                   update_running=update_running)
    else:
        y3 = ops.mini_pointnet_conv3(y2, w3b, gterm) if fused3 else \
            ops.gemm(y2, w3b, out_dtype=T, group_add=gterm, group_rows=32, algo_k=512)
        sc2, sh2 = ops.bn_finalize(g2, be2, False, running_mean=rm2, running_var=rv2)
    # BN2 + ReLU in the A-prologue of conv4; only the pooled maximum is written
    w4 = wc.get(sd[p + "second_conv.3.weight"])
    if (T in ops.HALF and FUSED_CONV12 and tuple(w4.shape) == (256, 512) and w4.stride(0) == 512 and y3.is_contiguous()
            and y3.shape[1] == 512):
        # csrc/mpn4.hip: W4 stays in registers, every group's rows are read once (305 -> ~150 us for 524 288 points)
        return ops.mini_pointnet_conv4(y3, sc2, sh2, w4, sd[p + "second_conv.3.bias"])
    tok = torch.empty((M // 32, w4.shape[0]), dtype=T, device=dev)
    ops.gemm(y3, w4, a_mode=A_AFFINE_RELU, a_scale=sc2, a_shift=sh2, bias=sd[p + "second_conv.3.bias"], pool_max=tok, want_out=False)
    return tok


def _mlp_weights(sd, p, wc):
    """fc1 / fc2 of block `p` in the fragment order of csrc/mlp_fused.hip (re-made when a weight's version changes)."""
    w1, w2 = sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc2.weight"]
    return wc.derived(("vit_mlp_tiled", p, ops.VIT_MLP_VARIANT), (w1, w2), lambda: ops.vit_mlp_retile(wc.get(w1), wc.get(w2)))


def vit_block_forward(sd, p, wc, x, pos, B, Tn, heads, dp1, dp2, save=None, pos_in_x=False, add_pos_out=False):
    """Block.forward on block(x + pos) (point_encoder.py:76-79,103).  x [B*Tn, D] fp32 is updated in
    place unless `save` (a dict) is given: then every intermediate the backward needs is kept.
    pos_in_x: x already holds x + pos (the previous block's fc2 epilogue added it: add_pos_out there), so norm1
    neither reads pos nor rewrites the residual stream; the sums are formed in the same order either way."""
    T = wc.dtype
    keep = save is not None
    if x.shape[0] >= ROWGEMM_MIN_ROWS and T in ops.HALF and not keep and x.shape[1] in ops.ROWGEMM_K:
        # frozen block, nothing kept: every linear picks its kernel (BLOCK_PATH; round 5) --
        #   "rowgemm": csrc/rowgemm.hip, the K = 384 weight stationary in registers, the LayerNorm applied while the rows are staged;
        #   "v2": a LayerNorm launch + ppt_gemm, which routes these row counts to the 256-row macro-tile core (csrc/gemm256.hip);
        #   "fused" (proj / mlp): attn.proj in front of / the whole MLP branch inside csrc/mlp_fused.hip.
        # The residual stream is updated in place; the sums are formed in the same order on every path.
        path = BLOCK_PATH
        if path["qkv"] == "auto":
            path = dict(path, qkv="lnlin" if x.shape[0] <= LNLIN_MAX_ROWS else "rowgemm")
        fused_ok = FUSED_MLP and sd[p + "mlp.fc1.weight"].shape[0] == 1536
        mlp = path["mlp"] if (fused_ok or path["mlp"] != "fused") else "rowgemm"
        proj = path["proj"] if not (path["proj"] == "fused" and not (mlp == "fused" and FUSED_PROJ)) else "rowgemm"
        if pos_in_x and path["qkv"] == "lnlin" and sd[p + "attn.qkv.weight"].shape[0] % 384 == 0:
            # csrc/lnlin.hip (round 6): rows stationary, weight streamed in fragment order, two workgroups per CU
            wq = sd[p + "attn.qkv.weight"]
            wqt = wc.derived(("lnlin_tiled", p), (wq,), lambda: ops.lnlin_retile(wc.get(wq)))
            qkv = ops.lnlin(x, wqt, (sd[p + "norm1.weight"], sd[p + "norm1.bias"]))
        elif pos_in_x and path["qkv"] in ("rowgemm", "lnlin"):
            qkv = ops.rowgemm(x, wc.get(sd[p + "attn.qkv.weight"]), ln=(sd[p + "norm1.weight"], sd[p + "norm1.bias"]))
        else:           # (the first block: x + pos is formed -- and written back -- by the LayerNorm kernel)
            if pos_in_x:
                h, _, _ = ops.layernorm_fwd(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], T)
            else:
                h, _, _ = ops.layernorm_fwd(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], T, add=pos, write_xs=x)
            if path["qkv"] in ("rowgemm", "lnlin"):
                qkv = ops.rowgemm(h, wc.get(sd[p + "attn.qkv.weight"]))
            else:
                qkv = ops.gemm(h, wc.get(sd[p + "attn.qkv.weight"]), out_dtype=T)
        a, _ = ops.attention_fwd(qkv, B, Tn, heads, ATTN_SCALE, False, want_lse=False)
        if proj == "fused":
            # attn.proj + DropPath + residual ride in front of the MLP kernel (csrc/mlp_fused.hip, round 3)
            w1t, w2t = _mlp_weights(sd, p, wc)
            wp = sd[p + "attn.proj.weight"]
            wpt = wc.derived(("vit_proj_tiled", p), (wp,), lambda: ops.vit_proj_retile(wc.get(wp)))
            ops.vit_mlp(x, w1t, sd[p + "mlp.fc1.bias"], w2t, sd[p + "mlp.fc2.bias"],
                        (sd[p + "norm2.weight"], sd[p + "norm2.bias"]), row_scale=dp2, row_scale_rows=Tn,
                        residual2=pos if add_pos_out else None, proj=(a, wpt, sd[p + "attn.proj.bias"], dp1, Tn))
            return x
        if proj == "v2":
            ops.gemm(a, wc.get(sd[p + "attn.proj.weight"]), out=x, bias=sd[p + "attn.proj.bias"], row_scale=dp1, row_scale_rows=Tn,
                     residual=x)
        else:
            ops.rowgemm(a, wc.get(sd[p + "attn.proj.weight"]), bias=sd[p + "attn.proj.bias"], residual=x, out=x, row_scale=dp1,
                        row_scale_rows=Tn)
        if mlp == "fused":
            w1t, w2t = _mlp_weights(sd, p, wc)
            ops.vit_mlp(x, w1t, sd[p + "mlp.fc1.bias"], w2t, sd[p + "mlp.fc2.bias"],
                        (sd[p + "norm2.weight"], sd[p + "norm2.bias"]), row_scale=dp2, row_scale_rows=Tn,
                        residual2=pos if add_pos_out else None)
            return x
        if mlp == "v2":
            h2, _, _ = ops.layernorm_fwd(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], T)
            f = ops.gemm(h2, wc.get(sd[p + "mlp.fc1.weight"]), out_dtype=T, bias=sd[p + "mlp.fc1.bias"], act=ACT_GELU)
        else:
            f = ops.rowgemm(x, wc.get(sd[p + "mlp.fc1.weight"]), ln=(sd[p + "norm2.weight"], sd[p + "norm2.bias"]),
                            bias=sd[p + "mlp.fc1.bias"], act=ACT_GELU)
        ops.gemm(f, wc.get(sd[p + "mlp.fc2.weight"]), out=x, bias=sd[p + "mlp.fc2.bias"], row_scale=dp2, row_scale_rows=Tn,
                 residual=x, residual2=pos if add_pos_out else None)
        return x
    T = wc.dtype
    keep = save is not None
    if pos_in_x:
        xs = x
        h, mean1, rstd1 = ops.layernorm_fwd(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], T, save_stats=keep)
    else:
        xs = torch.empty_like(x) if keep else x
        h, mean1, rstd1 = ops.layernorm_fwd(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], T, add=pos, write_xs=xs,
                                            save_stats=keep)
    qkv = ops.gemm(h, wc.get(sd[p + "attn.qkv.weight"]), out_dtype=T)
    a, lse = ops.attention_fwd(qkv, B, Tn, heads, ATTN_SCALE, False, want_lse=keep)
    x_mid = torch.empty_like(x) if keep else xs
    fused_mlp = (FUSED_MLP and not keep and T in ops.HALF and x.shape[1] == 384 and sd[p + "mlp.fc1.weight"].shape[0] == 1536
                 and x_mid.is_contiguous())
    if fused_mlp and FUSED_PROJ and a.is_contiguous() and x.shape[0] >= 8192:
        # frozen block: attn.proj + DropPath + residual in front of the fused MLP kernel (csrc/mlp_fused.hip, round 3).  From
        # ~100 chunks of 80 rows on: with few rows the MLP kernel runs on a handful of CUs and the proj GEMM in front of it is
        # better off as a launch of its own over all of them (C5, 2 064 rows: 6.39 ms separate, 6.44 fused)
        w1t, w2t = _mlp_weights(sd, p, wc)
        wp = sd[p + "attn.proj.weight"]
        wpt = wc.derived(("vit_proj_tiled", p), (wp,), lambda: ops.vit_proj_retile(wc.get(wp)))
        return ops.vit_mlp(xs, w1t, sd[p + "mlp.fc1.bias"], w2t, sd[p + "mlp.fc2.bias"],
                           (sd[p + "norm2.weight"], sd[p + "norm2.bias"]), row_scale=dp2, row_scale_rows=Tn,
                           residual2=pos if add_pos_out else None, proj=(a, wpt, sd[p + "attn.proj.bias"], dp1, Tn))
    ops.gemm(a, wc.get(sd[p + "attn.proj.weight"]), out=x_mid, bias=sd[p + "attn.proj.bias"], row_scale=dp1,
             row_scale_rows=Tn, residual=xs)
    if fused_mlp:
        # frozen block: LayerNorm + fc1 + GELU + fc2 + DropPath + residual (+ the next block's "+ pos") in one kernel
        w1t, w2t = _mlp_weights(sd, p, wc)
        return ops.vit_mlp(x_mid, w1t, sd[p + "mlp.fc1.bias"], w2t, sd[p + "mlp.fc2.bias"],
                           (sd[p + "norm2.weight"], sd[p + "norm2.bias"]), row_scale=dp2, row_scale_rows=Tn,
                           residual2=pos if add_pos_out else None)
    h2, mean2, rstd2 = ops.layernorm_fwd(x_mid, sd[p + "norm2.weight"], sd[p + "norm2.bias"], T, save_stats=keep)
    pre = torch.empty((x.shape[0], sd[p + "mlp.fc1.weight"].shape[0]), dtype=T, device=x.device) if keep else None
    f = ops.gemm(h2, wc.get(sd[p + "mlp.fc1.weight"]), out_dtype=T, bias=sd[p + "mlp.fc1.bias"], act=ACT_GELU,
                 out2=pre, out2_pre=True)
    x_out = torch.empty_like(x) if keep else x_mid
    ops.gemm(f, wc.get(sd[p + "mlp.fc2.weight"]), out=x_out, bias=sd[p + "mlp.fc2.bias"], row_scale=dp2,
             row_scale_rows=Tn, residual=x_mid, residual2=pos if add_pos_out else None)
    if keep:
        save.update(xs=xs, mean1=mean1, rstd1=rstd1, h=h, qkv=qkv, a=a, lse=lse, x_mid=x_mid, mean2=mean2,
                    rstd2=rstd2, h2=h2, pre=pre, f=f, dp1=dp1, dp2=dp2)
    return x_out


def tokenize_points(sd, p, wc, pc, fps_start, bn_train, cfg, update_running=True, grouped=None):
    """The part of PointTransformer.forward in front of the blocks (point_encoder.py:234-249): Group (FPS + kNN), the
    mini-PointNet encoder, reduce_dim, cls token / cls_pos and pos_embed -> (x2, pos2), both [B*Tn, D] fp32 (Tn = G + 1).
    A function of the input cloud and frozen weights alone: train.Trainer can run it ahead of the step (PointTransformer._group_ahead)."""
    G, D = cfg["num_group"], cfg["trans_dim"]
    Tn = G + 1
    # grouped = (nbhd, center): Group.forward was already run (ahead of the step, on its own stream)
    nbhd, center = grouped if grouped is not None else group_points(pc, G, cfg["group_size"], fps_start)
    B = center.shape[0]
    dev = center.device
    wct = _stage_wc(wc, "tokenizer")
    tok = mini_pointnet(sd, p + "encoder.", wct, nbhd, bn_train, update_running)
    x = torch.empty((B, Tn, D), dtype=torch.float32, device=dev)
    pos = torch.empty((B, Tn, D), dtype=torch.float32, device=dev)
    x[:, 0] = sd[p + "cls_token"].view(D)
    pos[:, 0] = sd[p + "cls_pos"].view(D)
    x2, pos2 = x.view(B * Tn, D), pos.view(B * Tn, D)
    # reduce_dim and pos_embed write straight into rows 1.. of every sample (batched GEMM)
    ops.gemm(tok, wct.get(sd[p + "reduce_dim.weight"]), out=x2[1:], M=G, bias=sd[p + "reduce_dim.bias"], batch=B,
             strideA=G * tok.shape[1], strideC=Tn * D)
    pe = ops.linear3_gelu(center.view(B * G, 3), sd[p + "pos_embed.0.weight"], sd[p + "pos_embed.0.bias"], wct.dtype)
    ops.gemm(pe, wct.get(sd[p + "pos_embed.2.weight"]), out=pos2[1:], M=G, bias=sd[p + "pos_embed.2.bias"], batch=B,
             strideA=G * pe.shape[1], strideC=Tn * D)
    return x2, pos2, center


def point_encoder_forward(sd, p, wc, pc, fps_start, dp, bn_train, save_tier, cfg, update_running=True, fetch=None,
                          last_block=None, resume=None, grouped=None, tokens=None):
    """PointTransformer.forward (point_encoder.py:234-257) -> (feat [B,2*D] fp32, saved | None).
    dp: DropPath factors [depth,2,B] fp32 or None; save_tier > 0 keeps block-(depth-1) activations.

    The forward can be cut in front of the last block (the only one that may train, ULIP_models.py:461-470):
    last_block=False runs the tokenizer and blocks 0 .. depth-2 -- everything that is frozen whatever the head_type --
    and returns (x2, pos2), both [B*Tn, D] fp32, x2 with the last block's "+ pos" already added;
    resume=(x2, pos2) runs the last block, the final norm and the pooling on them.
    grouped=(neighbourhoods [B,G,n,3], centres [B,G,3]) replaces pc / fps_start; tokens=(x2, pos2) = tokenize_points' result
    (x2 is updated in place) replaces the whole tokenizer."""
    T = wc.dtype
    G, D, depth, heads = cfg["num_group"], cfg["trans_dim"], cfg["depth"], cfg["num_heads"]
    Tn = G + 1
    center = None
    if resume is None:
        if tokens is not None:
            x2, pos2 = tokens
        else:
            x2, pos2, center = tokenize_points(sd, p, wc, pc, fps_start, bn_train, cfg, update_running, grouped)
        B = x2.shape[0] // Tn
        first, pos_in_x = 0, False
    else:
        x2, pos2 = resume
        B = x2.shape[0] // Tn
        first, pos_in_x = depth - 1, depth > 1
    saved = None
    fetched = []
    for l in range(first, depth - 1 if last_block is False else depth):
        bp = f"{p}blocks.blocks.{l}."
        d1 = dp[l, 0] if dp is not None else None
        d2 = dp[l, 1] if dp is not None else None
        # the next block's "+ pos" rides in this block's fc2 epilogue (unless the block output itself is wanted)
        add_pos_out = l + 1 < depth and not (fetch is not None and l in fetch)
        wcl = _stage_wc(wc, "last_block" if l == depth - 1 else "blocks")
        if save_tier > 0 and l == depth - 1:
            saved = {"wc": wcl}
            x2 = vit_block_forward(sd, bp, wcl, x2, pos2, B, Tn, heads, d1, d2, save=saved, pos_in_x=pos_in_x,
                                   add_pos_out=add_pos_out)
        else:
            x2 = vit_block_forward(sd, bp, wcl, x2, pos2, B, Tn, heads, d1, d2, pos_in_x=pos_in_x, add_pos_out=add_pos_out)
        pos_in_x = add_pos_out
        if fetch is not None and l in fetch:        # part-seg: norm(x)[:, 1:] after blocks 3, 7, 11 (point_encoder.py:100-108,377)
            fn, _, _ = ops.layernorm_fwd(x2, sd[p + "norm.weight"], sd[p + "norm.bias"], torch.float32)
            fetched.append(fn.view(B, Tn, D)[:, 1:])
    if last_block is False:
        return x2, pos2
    if fetch is not None:
        return fetched, center
    keep = saved is not None
    xn, meanf, rstdf = ops.layernorm_fwd(x2, sd[p + "norm.weight"], sd[p + "norm.bias"], torch.float32, save_stats=keep)
    feat, argmax = ops.cls_max_pool(xn.view(B, Tn, D), want_argmax=keep)
    if keep:
        saved.update(x_out=x2, meanf=meanf, rstdf=rstdf, argmax=argmax, B=B, Tn=Tn, D=D, heads=heads,
                     prefix=f"{p}blocks.blocks.{depth - 1}.", norm_w=sd[p + "norm.weight"])
    return feat, saved


def _wgrad(dy_t, x_t):
    """dW[N,K] = dY[M,N]^T @ X[M,K]: both operands are M-major, so both are transposed and the NT GEMM runs split over
    the M rows (ops.gemm_tn_splitk: a 384 x 384 gradient over 32 832 rows is 36 tiles with 513 K slabs each otherwise)."""
    return ops.gemm_tn_splitk(dy_t, x_t)


def point_encoder_backward(sd, wc, s, dfeat, tier, grad_scale=1.0):
    """Backward of the un-frozen part (ULIP_models.py:461-470): final LN -> block depth-1.
    Returns {param name: grad} for the tier's parameters.
    grad_scale = S (ppt_amd/gradscale.py): the block's activation gradients travel in its 16-bit operand format; dfeat is
    multiplied by the power of two S on the way in ([B, 2D]: one tiny launch) and every returned gradient by 1 / S on the way
    out (one multi-tensor launch) -- the caller sees the true fp32 gradients."""
    wc = s.get("wc", wc)                       # (the cache block depth-1 ran its forward with: STAGE_DTYPE diagnostics)
    T = wc.dtype
    if grad_scale != 1.0:
        from . import gradscale
        grads = point_encoder_backward(sd, wc, s, dfeat * float(grad_scale), tier)
        gradscale.unscale_(list(grads.values()), float(grad_scale))
        return grads
    B, Tn, D, heads, p = s["B"], s["Tn"], s["D"], s["heads"], s["prefix"]
    M = B * Tn
    grads = {}
    # cat(cls, max) backward: scatter into the final-LN output gradient
    dxn = torch.zeros((B, Tn, D), dtype=torch.float32, device=dfeat.device)
    dxn[:, 0] = dfeat[:, :D]
    dxn.scatter_(1, s["argmax"].view(B, 1, D).long(), dfeat[:, D:].reshape(B, 1, D))
    g, _, _ = ops.layernorm_bwd(dxn.view(M, D), s["x_out"], s["norm_w"], s["meanf"], s["rstdf"])
    dp1, dp2 = s["dp1"], s["dp2"]
    # ---- MLP branch: x_out = x_mid + dp2 * (gelu(h2 W1^T + b1) W2^T + b2)
    gs = g if dp2 is None else (g.view(B, Tn, D) * dp2.view(B, 1, 1)).view(M, D)
    gs_t = ops.convert(gs, T)
    grads[p + "mlp.fc2.weight"] = _wgrad(gs_t, s["f"])
    grads[p + "mlp.fc2.bias"] = ops.col_sums(gs)
    d_pre = ops.gemm(gs_t, wc.get(sd[p + "mlp.fc2.weight"], "wt"), out_dtype=T, act=ACT_GELU, dact_pre=s["pre"])
    d_h2 = ops.gemm(d_pre, wc.get(sd[p + "mlp.fc1.weight"], "wt"), out_dtype=torch.float32)
    if tier >= 2:
        grads[p + "mlp.fc1.weight"] = _wgrad(d_pre, s["h2"])
        grads[p + "mlp.fc1.bias"] = ops.col_sums(d_pre)
    _, dw, db = ops.layernorm_bwd(d_h2, s["x_mid"], sd[p + "norm2.weight"], s["mean2"], s["rstd2"], dx=g,
                                  accumulate=True, want_wgrad=True)
    grads[p + "norm2.weight"], grads[p + "norm2.bias"] = dw, db
    if tier < 2:
        return grads
    # ---- attention branch: x_mid = xs + dp1 * (attn(LN1(xs)) Wp^T + bp)
    gs = g if dp1 is None else (g.view(B, Tn, D) * dp1.view(B, 1, 1)).view(M, D)
    gs_t = ops.convert(gs, T)
    if tier >= 3:
        grads[p + "attn.proj.weight"] = _wgrad(gs_t, s["a"])
        grads[p + "attn.proj.bias"] = ops.col_sums(gs)
    d_a = ops.gemm(gs_t, wc.get(sd[p + "attn.proj.weight"], "wt"), out_dtype=T)
    d_qkv = ops.attention_bwd(s["qkv"], s["a"], d_a, s["lse"], B, Tn, heads, ATTN_SCALE, False)
    if tier >= 3:
        grads[p + "attn.qkv.weight"] = _wgrad(d_qkv, s["h"])
    d_h = ops.gemm(d_qkv, wc.get(sd[p + "attn.qkv.weight"], "wt"), out_dtype=torch.float32)
    _, dw, db = ops.layernorm_bwd(d_h, s["xs"], sd[p + "norm1.weight"], s["mean1"], s["rstd1"], want_wgrad=True)
    grads[p + "norm1.weight"], grads[p + "norm1.bias"] = dw, db
    return grads


# =================================================================================================
# PointNet2-MSG encoder (models/pointnet2/pointnet2.py:40-73, pointnet2_utils.py:161-266)
# =================================================================================================
PN2_MSG = dict(   # pointnet2.py:44-46
    sa1=dict(npoint=512, radii=[0.1, 0.2, 0.4], nsample=[16, 32, 128]),
    sa2=dict(npoint=128, radii=[0.2, 0.4, 0.8], nsample=[32, 64, 128]))


def _bn_affine(sd, bnp, train, partials, rpp, count, update_running):
    g, be, rm, rv, nb = _bn_params(sd, bnp)
    if train:
        return ops.bn_finalize(g, be, True, partials=partials, rows_per_partial=rpp, count=count, running_mean=rm,
                               running_var=rv, num_batches_tracked=nb, update_running=update_running)
    return ops.bn_finalize(g, be, False, running_mean=rm, running_var=rv)


def _pad_cols(x, mult, dtype):
    """[R, C] -> [R, Cp] in `dtype`, zero-padded so that Cp % mult == 0 (GEMM K alignment)."""
    R, C = x.shape
    Cp = (C + mult - 1) // mult * mult
    out = torch.zeros((R, Cp), dtype=dtype, device=x.device)
    out[:, :C] = x
    return out


def _stats_bufs(M, C, dev, train):
    if not train:
        return None
    return (torch.empty(((M + 31) // 32, C), dtype=torch.float32, device=dev),
            torch.empty(((M + 31) // 32, C), dtype=torch.float32, device=dev))


def _sa_msg_level(sd, p, wc, cfg, xyz, feats, start, train, upd, pre=None):
    """PointNetSetAbstractionMsg.forward (pointnet2_utils.py:228-266): one (radius, nsample, MLP) branch per scale,
    channels = [features | centred xyz] (:250)."""
    branches = [(r, K, f"{p}conv_blocks.{i}.", f"{p}bn_blocks.{i}.") for i, (r, K) in enumerate(zip(cfg["radii"], cfg["nsample"]))]
    return _sa_level(sd, branches, wc, cfg["npoint"], xyz, feats, start, train, upd, xyz_first=False, pre=pre)


def _sa_group(xyz, npoint, branches, start, first):
    """The part of a set-abstraction level that depends on coordinates only: FPS + one ball query per branch.
    -> [new_xyz [B,S,3], per branch: grouped centred xyz [B,S,K,3] (first level) or neighbour indices [B,S,K]]."""
    _, new_xyz = ops.fps(xyz, npoint, start)
    out = [new_xyz]
    if 1 < len(branches) <= 3:          # the scales of an MSG level share their centres: one pass over the cloud for all radii
        for idx, gxyz in ops.ball_query_multi(xyz, new_xyz, list(branches), want_grouped=first):
            out.append(gxyz if first else idx)
        return out
    for r, K in branches:
        idx, gxyz = ops.ball_query(xyz, new_xyz, r, K, want_grouped=True)
        out.append(gxyz if first else idx)
    return out


def pointnet2_group(pc, fps_starts, levels):
    """Grouping stage of Pointnet2_Msg / Pointnet2_Ssg (both levels): levels = [(npoint, [(radius, K), ...]), ...].
    A function of the input cloud and the FPS starts alone, so a trainer may run it ahead of the step."""
    out, xyz = [], pc
    for li, ((npoint, branches), start) in enumerate(zip(levels, fps_starts)):
        g = _sa_group(xyz.contiguous(), npoint, branches, start, first=li == 0)
        out += g
        xyz = g[0]
    return out


def _sa_level(sd, branches, wc, npoint, xyz, feats, start, train, upd, xyz_first, pre=None):
    """FPS + ball query + shared three-layer MLP + max over the group, for every (radius, nsample, conv prefix, bn prefix)
    branch.  xyz [B,N,3] fp32, feats [B*N, D] (T) or None -> (new_xyz [B,S,3], new_feats [B*S, sum C] (T)).
    xyz_first: channel order of the first conv's input, [centred xyz | features] (sample_and_group, :132) or
    [features | centred xyz] (the MSG module, :250)."""
    T = wc.dtype
    S = npoint
    # pre = _sa_group's list for this level (grouping already done, ahead of the step; xyz may then be None at level 1)
    new_xyz = pre[0] if pre is not None else ops.fps(xyz, S, start)[1]
    B, dev = new_xyz.shape[0], new_xyz.device
    N = xyz.shape[1] if xyz is not None else 0
    c_out = [sd[cb + "2.weight"].shape[0] for _, _, cb, _ in branches]
    out = torch.empty((B * S, sum(c_out)), dtype=T, device=dev)
    col = 0
    mult = 8 if T in ops.HALF else 4
    src = cx = None                  # (what every branch of a level with input features shares: built once, not per branch)
    for i, (r, K, cb, bb) in enumerate(branches):
        if pre is not None:
            idx = gxyz = pre[1 + i]
        else:
            idx, gxyz = ops.ball_query(xyz, new_xyz, r, K, want_grouped=True)
        M = B * S * K
        w0, b0 = sd[cb + "0.weight"], sd[cb + "0.bias"]
        C1 = w0.shape[0]
        st1 = _stats_bufs(M, sd[cb + "1.weight"].shape[0], dev, train)
        if feats is None:
            # layer 0 is a K=3 conv on the centred coordinates: A-prologue (conv + BN + ReLU) of the layer-1 GEMM
            pts = gxyz.view(M, 3)
            w03 = wc.get(w0, "f32")
            part0, rpp0 = None, 0
            if train:
                ps, pq, rpp0 = ops.conv1_stats(pts, w03, b0)
                part0 = (ps, pq)
            sc0, sh0 = _bn_affine(sd, bb + "0.", train, part0, rpp0, M, upd)
            w1c = wc.get(sd[cb + "1.weight"])
            if (T == torch.bfloat16 and FUSED_CONV12 and train and M % 32 == 0 and w1c.stride(0) == w1c.shape[1]
                    and (w1c.shape[1], w1c.shape[0]) in ops.CONV12_STATS_SHAPES):
                y1, st1 = ops.conv12_stats(pts, w03, b0, sc0, sh0, w1c, sd[cb + "1.bias"])     # csrc/mpn1.hip: no tile staging
            else:
                y1 = ops.gemm(None, w1c, out_dtype=T, a_mode=A_CONV1, pts=pts, w1=w03, b1=b0,
                              a_scale=sc0, a_shift=sh0, bias=sd[cb + "1.bias"], col_stats=st1)
        else:
            # layer 0 by linearity of the 1x1 conv: per source point P = W.[feat|xyz], per centre Q = b - W_xyz.c
            D = feats.shape[1]
            fo, xo = (3, 0) if xyz_first else (0, D)              # column offsets of the features / the coordinates
            if src is None:
                src = torch.zeros((B * N, (D + 3 + mult - 1) // mult * mult), dtype=T, device=dev)
                src[:, fo:fo + D] = feats
                src[:, xo:xo + 3] = xyz.view(B * N, 3)
                cx = _pad_cols(new_xyz.view(B * S, 3), 4, torch.float32)
            w0p = wc.get(w0, "w", pad_to=mult)
            P = ops.gemm(src, w0p, out_dtype=torch.float32)
            # Q = b - W_xyz . c as ONE GEMM: the negated, padded coordinate columns of the (frozen) weight are a cached derived
            # operand and the bias rides in the epilogue (same products, b + (-(W c)) = b - W c: the same bits)
            wxn = wc.derived(("sa_wx_neg", cb, xo), (w0,), lambda w0=w0, xo=xo, C1=C1: _pad_cols(-w0.detach().reshape(C1, -1)[:, xo:xo + 3].float(), 4, torch.float32))
            Q = ops.gemm(cx, wxn, out_dtype=torch.float32, bias=b0)
            y0, part0 = ops.gather_add(P, Q, idx, N, T, want_stats=train)
            sc0, sh0 = _bn_affine(sd, bb + "0.", train, part0, 32, M, upd)
            y1 = ops.gemm(y0, wc.get(sd[cb + "1.weight"]), out_dtype=T, a_mode=A_AFFINE_RELU, a_scale=sc0, a_shift=sh0,
                          bias=sd[cb + "1.bias"], col_stats=st1)
        sc1, sh1 = _bn_affine(sd, bb + "1.", train, st1, 32, M, upd)
        C3 = c_out[i]
        pr = min(K, 64)
        pmax = torch.empty((M // pr, C3), dtype=torch.float32, device=dev)
        pmin = torch.empty_like(pmax)
        st2 = _stats_bufs(M, C3, dev, train)
        w2c = wc.get(sd[cb + "2.weight"])
        if (T == torch.bfloat16 and FUSED_CONV12 and train and M % 64 == 0 and y1.is_contiguous() and w2c.stride(0) == w2c.shape[1]
                and (w2c.shape[1], C3, pr) in ops.AFFINE_CONV_POOL_SHAPES):
            ops.affine_conv_pool(y1, sc1, sh1, w2c, sd[cb + "2.bias"], pr, pmax, pmin, st2)      # csrc/affpool.hip: no tile staging
        else:
            ops.gemm(y1, w2c, a_mode=A_AFFINE_RELU, a_scale=sc1, a_shift=sh1, bias=sd[cb + "2.bias"],
                     want_out=False, pool_max=pmax, pool_min=pmin, pool_rows=pr, col_stats=st2)
        sc2, sh2 = _bn_affine(sd, bb + "2.", train, st2, 32, M, upd)
        ops.pool_finish(pmax, pmin, K // pr, sc2, sh2, out[:, col:col + C3])
        col += C3
    return new_xyz, out


PN2_MSG_LEVELS = [(c["npoint"], list(zip(c["radii"], c["nsample"]))) for c in (PN2_MSG["sa1"], PN2_MSG["sa2"])]


def pointnet2_msg_forward(sd, p, wc, pc, fps_starts, train, drop_masks, update_running=True, grouped=None):
    """Pointnet2_Msg.forward (pointnet2.py:56-73): pc [B,N,3] -> [B,256] fp32.
    fps_starts = (start level 1 [B], start level 2 [B]); drop_masks = (m1 [B,512], m2 [B,256]) or None.
    grouped = pointnet2_group(pc, fps_starts, PN2_MSG_LEVELS) when the grouping stage already ran."""
    n1 = 1 + len(PN2_MSG_LEVELS[0][1])
    pre1, pre2 = (grouped[:n1], grouped[n1:]) if grouped is not None else (None, None)
    s0, s1 = fps_starts if fps_starts is not None else (None, None)
    l1_xyz, l1 = _sa_msg_level(sd, p + "sa1.", wc, PN2_MSG["sa1"], pc, None, s0, train, update_running, pre=pre1)
    l2_xyz, l2 = _sa_msg_level(sd, p + "sa2.", wc, PN2_MSG["sa2"], l1_xyz.contiguous(), l1, s1, train,
                               update_running, pre=pre2)
    return _pn2_tail(sd, p, wc, l2_xyz, l2, train, drop_masks, update_running)


def _pn2_tail(sd, p, wc, l2_xyz, l2, train, drop_masks, update_running):
    """sa3 (group_all) + the FC head shared by Pointnet2_Msg and Pointnet2_Ssg (pointnet2.py:33-36, 67-70)."""
    T = wc.dtype
    B = l2_xyz.shape[0]
    dev = l2_xyz.device
    mult = 8 if T in ops.HALF else 4
    # sa3: group_all over the 128 remaining points, channels = [xyz | features] (pointnet2_utils.py:152-157)
    S2 = l2_xyz.shape[1]
    M = B * S2
    D = l2.shape[1]
    a = torch.zeros((M, (D + 3 + mult - 1) // mult * mult), dtype=T, device=dev)
    a[:, :3] = l2_xyz.reshape(M, 3)
    a[:, 3:3 + D] = l2
    cb, bb = p + "sa3.mlp_convs.", p + "sa3.mlp_bns."
    st = _stats_bufs(M, 256, dev, train)
    y0 = ops.gemm(a, wc.get(sd[cb + "0.weight"], "w", pad_to=mult), out_dtype=T, bias=sd[cb + "0.bias"], col_stats=st)
    sc, sh = _bn_affine(sd, bb + "0.", train, st, 32, M, update_running)
    st = _stats_bufs(M, 512, dev, train)
    y1 = ops.gemm(y0, wc.get(sd[cb + "1.weight"]), out_dtype=T, a_mode=A_AFFINE_RELU, a_scale=sc, a_shift=sh,
                  bias=sd[cb + "1.bias"], col_stats=st)
    sc, sh = _bn_affine(sd, bb + "1.", train, st, 32, M, update_running)
    st = _stats_bufs(M, 1024, dev, train)
    pr = min(S2, 64)
    pmax = torch.empty((M // pr, 1024), dtype=torch.float32, device=dev)
    pmin = torch.empty_like(pmax)
    ops.gemm(y1, wc.get(sd[cb + "2.weight"]), a_mode=A_AFFINE_RELU, a_scale=sc, a_shift=sh, bias=sd[cb + "2.bias"],
             want_out=False, pool_max=pmax, pool_min=pmin, pool_rows=pr, col_stats=st)
    sc, sh = _bn_affine(sd, bb + "2.", train, st, 32, M, update_running)
    x = torch.empty((B, 1024), dtype=T, device=dev)
    ops.pool_finish(pmax, pmin, S2 // pr, sc, sh, x)
    # FC head: Linear -> BatchNorm1d (over the batch) -> ReLU -> Dropout, twice (pointnet2.py:69-70)
    for fc, bn, width, mask in (("fc1.", "bn1.", 512, 0), ("fc2.", "bn2.", 256, 1)):
        st = _stats_bufs(B, width, dev, train)
        h = ops.gemm(x, wc.get(sd[p + fc + "weight"]), out_dtype=torch.float32, bias=sd[p + fc + "bias"], col_stats=st)
        sc, sh = _bn_affine(sd, p + bn, train, st, 32, B, update_running)
        last = fc == "fc2."
        x = ops.bn_act_rows(h, sc, sh, torch.float32 if last else T, mask=drop_masks[mask] if drop_masks is not None else None)
    return x


PN2_SSG = dict(   # pointnet2.py:11-12
    sa1=dict(npoint=512, radius=0.2, nsample=32), sa2=dict(npoint=128, radius=0.4, nsample=64))


PN2_SSG_LEVELS = [(c["npoint"], [(c["radius"], c["nsample"])]) for c in (PN2_SSG["sa1"], PN2_SSG["sa2"])]


def pointnet2_ssg_forward(sd, p, wc, pc, fps_starts, train, drop_masks, update_running=True, grouped=None):
    """Pointnet2_Ssg.forward (pointnet2.py:22-38): pc [B,N,3] -> [B,256] fp32; same kernels as the MSG encoder, one
    scale per level and [centred xyz | features] channel order."""
    levels, xyz, feats = (("sa1.", PN2_SSG["sa1"]), ("sa2.", PN2_SSG["sa2"])), pc, None
    for li, (name, cfg) in enumerate(levels):
        br = [(cfg["radius"], cfg["nsample"], p + name + "mlp_convs.", p + name + "mlp_bns.")]
        pre = grouped[2 * li:2 * li + 2] if grouped is not None else None
        start = fps_starts[li] if fps_starts is not None else None
        xyz, feats = _sa_level(sd, br, wc, cfg["npoint"], xyz.contiguous() if xyz is not None else None, feats, start, train,
                               update_running, xyz_first=True, pre=pre)
    return _pn2_tail(sd, p, wc, xyz, feats, train, drop_masks, update_running)


POINTMLP = dict(points=1024, k_neighbors=[24] * 4, reducers=[2] * 4, pre_blocks=[2] * 4, pos_blocks=[2] * 4)   # pointMLP.py:359-363


def _conv_bn(sd, wc, x, wkey, bnp, train, upd, out_dtype, a_affine=None):
    """Conv1d(k=1, bias=False) over rows + the folded BatchNorm1d affine of its output: (y [M,C], scale, shift).
    a_affine = (scale, shift): the input is the raw output of the previous conv, BN + ReLU applied while it is loaded."""
    M = x.shape[0]
    w = wc.get(sd[wkey])
    st = _stats_bufs(M, w.shape[0], x.device, train)
    kw = dict(a_mode=A_AFFINE_RELU, a_scale=a_affine[0], a_shift=a_affine[1]) if a_affine is not None else {}
    y = ops.gemm(x, w, out_dtype=out_dtype, col_stats=st, **kw)
    sc, sh = _bn_affine(sd, bnp, train, st, 32, M, upd)
    return y, sc, sh


def _mlp_res_block(sd, wc, p, x, train, upd, pool=1, x_affine=None):
    """ConvBNReLURes1D.forward (pointMLP.py:188-221; groups=1): relu(BN(conv2(relu(BN(conv1(x))))) + x) on rows [M,C];
    pool > 1 also takes the max over each `pool` consecutive rows (:251, :332).  x_affine = (scale, shift): x is the raw
    output of the conv in front of the block and relu(scale * x + shift) is the block input, applied where x is read."""
    T = wc.dtype
    c1, sc1, sh1 = _conv_bn(sd, wc, x, p + "net1.0.weight", p + "net1.1.", train, upd, T, a_affine=x_affine)
    c2, sc2, sh2 = _conv_bn(sd, wc, c1, p + "net2.0.weight", p + "net2.1.", train, upd, T, a_affine=(sc1, sh1))
    return ops.bn_res_act_rows(c2, x, sc2, sh2, T, pool, res_affine=x_affine)


def pointmlp_group(pc, fps_starts, cfg=None):
    """The grouping of all stages (LocalGrouper's FPS + kNN, pointMLP.py:157-162): a function of the coordinates alone.
    -> flat list, per stage (anchor indices [B,S], anchor coordinates [B,S,3], neighbour indices [B,S,k])."""
    cfg = cfg or POINTMLP
    out, xyz, S = [], pc, cfg["points"]
    for i in range(len(cfg["reducers"])):
        S //= cfg["reducers"][i]
        cidx, new_xyz = ops.fps(xyz, S, fps_starts[i])
        nidx, _ = ops.knn_group(xyz, new_xyz, cfg["k_neighbors"][i], want_idx=True, want_nbhd=False)
        out += [cidx, new_xyz, nidx]
        xyz = new_xyz
    return out


def pointmlp_forward(sd, p, wc, pc, fps_starts, train, drop_masks, update_running=True, cfg=None, grouped=None):
    """Model.forward of pointMLP() / pointMLPElite() (pointMLP.py:320-334): pc [B,N,3] -> [B,256] fp32.
    fps_starts = one start vector [B] per stage (furthest_point_sample's randint, :77); drop_masks = (m1 [B,512],
    m2 [B,256]) or None.  Rows are (cloud, point) / (cloud, group, neighbour) throughout, channels last.

    LocalGrouper (:152-181, normalize="anchor", use_xyz=False) followed by the transfer conv (:243-247) is linear in the
    gathered features: with r_b = 1 / (std_b + 1e-5) and W = [Wa | Wb] the transfer weight,
        W.[alpha*(x_j - a)*r_b + beta | a] = r_b * (Wa*alpha).x_j + (Wa.beta + Wb.a - r_b * (Wa*alpha).a)
    so the conv runs once per SOURCE point (N rows instead of S*k = 12 N) and ppt_gather_add forms the rows."""
    cfg = cfg or POINTMLP
    T = wc.dtype
    B, N, _ = pc.shape
    dev = pc.device
    upd = update_running
    # embedding (:324): Conv1d(3, E, bias=False) + BN + ReLU, evaluated in the A-prologue of a GEMM against the identity
    w_e = wc.get(sd[p + "embedding.net.0.weight"], "f32")
    E = w_e.shape[0]
    b_e = torch.zeros((E,), dtype=torch.float32, device=dev)
    pts = pc.reshape(B * N, 3)
    part, rpp = None, 0
    if train:
        ps, pq, rpp = ops.conv1_stats(pts, w_e, b_e)
        part = (ps, pq)
    sc, sh = _bn_affine(sd, p + "embedding.net.1.", train, part, rpp, B * N, upd)
    x = ops.gemm(None, wc.derived(("eye", E), (), lambda: torch.eye(E, dtype=T, device=dev)), out_dtype=T, a_mode=A_CONV1,
                 pts=pts, w1=w_e, b1=b_e, a_scale=sc, a_shift=sh)
    xyz = pc
    S = cfg["points"]
    n_stage = len(cfg["reducers"])
    for i in range(n_stage):
        S //= cfg["reducers"][i]
        k = cfg["k_neighbors"][i]
        d = x.shape[1]
        if grouped is not None:                                                  # pointmlp_group ran ahead of the step
            cidx, new_xyz, nidx = grouped[3 * i:3 * i + 3]
        else:
            cidx, new_xyz = ops.fps(xyz, S, fps_starts[i])
            nidx, _ = ops.knn_group(xyz, new_xyz, k, want_idx=True, want_nbhd=False)
        # per-cloud std of the anchor-centred neighbour features (:174), unbiased, from per-group (sum, sumsq) in fp64
        st = ops.group_anchor_stats(x, nidx, cidx, N)
        n = float(S * k * d)
        if POINTMLP_FUSED_NORM:
            r = ops.pointmlp_cloud_rstd(st, n)                                   # (one launch for the ~12 of the expression below)
        else:
            st = st.double().sum(1)
            var = ((st[:, 1] - st[:, 0] * st[:, 0] / n) / (n - 1.0)).clamp_min(0.0)
            r = (1.0 / (var.sqrt() + 1e-5)).float()
        pp = f"{p}pre_blocks_list.{i}."
        g = f"{p}local_grouper_list.{i}."
        wt, alpha, beta = sd[pp + "transfer.net.0.weight"], sd[g + "affine_alpha"], sd[g + "affine_beta"]
        C = wt.shape[0]

        def fold(wt=wt, alpha=alpha, beta=beta, d=d):
            w2 = wt.detach().reshape(wt.shape[0], -1).float()
            wa = w2[:, :d] * alpha.detach().reshape(1, d)
            return ops.convert(torch.cat([wa, w2[:, d:]], 0).contiguous(), T), (w2[:, :d] @ beta.detach().reshape(d)).contiguous()
        wcat, c0 = wc.derived(("pointmlp_transfer", i), (wt, alpha, beta), fold)
        PQ = ops.gemm(x, wcat, out_dtype=torch.float32)                          # [B*N, 2C]: (Wa*alpha).x | Wb.x
        if POINTMLP_FUSED_NORM and C % 4 == 0:
            P, Q = ops.pointmlp_pq(PQ, r, cidx, c0, B, N)                        # (one launch, the same operations in the same order)
        else:
            P = (PQ[:, :C].reshape(B, N, C) * r.view(B, 1, 1)).reshape(B * N, C)
            a = (torch.arange(B, device=dev).view(B, 1) * N + cidx).view(-1)
            Q = (c0.view(1, C) + PQ[:, C:][a] - P[a]).contiguous()
        M = B * S * k
        # the transfer conv's output stays raw: its BN + ReLU (:246) is applied by the two readers of the first block
        y, part0 = ops.gather_add(P, Q, nidx, N, T, want_stats=train)
        aff = _bn_affine(sd, pp + "transfer.net.1.", train, part0, 32, M, upd)
        nb = cfg["pre_blocks"][i]
        for j in range(nb):                                                      # PreExtraction (:248-252)
            y = _mlp_res_block(sd, wc, f"{pp}operation.{j}.", y, train, upd, pool=k if j == nb - 1 else 1,
                               x_affine=aff if j == 0 else None)
        nb = cfg["pos_blocks"][i]
        for j in range(nb):                                                      # PosExtraction (:272-273), max of :332
            y = _mlp_res_block(sd, wc, f"{p}pos_blocks_list.{i}.operation.{j}.", y, train, upd,
                               pool=S if (i == n_stage - 1 and j == nb - 1) else 1)
        x, xyz, N = y, new_xyz, S
    # classifier (:307-316): Linear -> BatchNorm1d over the batch -> ReLU -> Dropout(0.5), twice
    c = p + "classifier."
    for fc, bn, mask in (("0.", "1.", 0), ("4.", "5.", 1)):
        w = wc.get(sd[c + fc + "weight"])
        st = _stats_bufs(B, w.shape[0], dev, train)
        h = ops.gemm(x, w, out_dtype=torch.float32, bias=sd[c + fc + "bias"], col_stats=st)
        sc, sh = _bn_affine(sd, c + bn, train, st, 32, B, upd)
        last = fc == "4."
        x = ops.bn_act_rows(h, sc, sh, torch.float32 if last else T, mask=drop_masks[mask] if drop_masks is not None else None)
    return x


# =================================================================================================
# text branch (CLIP text transformer, ULIP_models.py:35-67, 203-222)
# =================================================================================================
# LayerNorm -> in_proj / c_fc as ONE kernel (csrc/rowgemm.hip): two nodes fewer per layer, but OFF by default -- the
# weight-stationary kernel's 512-thread / 256-VGPR workgroups take whole CUs for ~15 us where the 817 rows give the 64 x 64 tile
# GEMM ~8 us of workgroups other kernels share a CU with, and the prompt chain runs BESIDE the point tower: interleaved same-box
# A/B of the C2 step 3.87 (on) vs 3.79 ms (off) (tools/ab_env.py)
TEXT_FUSE_LN = os.environ.get("PPT_TEXT_FUSE_LN", "0") != "0"
# round 5: the text tower's K = 2048 / 1536 linears (c_proj forward; the dX products of c_fc and in_proj) as split-K launches whose
# partial products are summed by the LayerNorm kernel that consumes them (ops.gemm_splitk + ops.layernorm_fwd_sum / _bwd_sum).
# 16-bit operand modes only: the fp32 parity mode keeps the single-launch summation order.
TEXT_SPLITK = os.environ.get("PPT_TEXT_SPLITK", "1") != "0"
# PointMLP's LocalGrouper normalisation as two launches per stage (csrc/pointmlp.hip: cloud_rstd_kernel, pointmlp_pq_kernel) instead of
# ~25 ATen ones; PPT_POINTMLP_FUSED_NORM=0: the ATen expressions (A/B, bit-identity test).
POINTMLP_FUSED_NORM = os.environ.get("PPT_POINTMLP_FUSED_NORM", "1") != "0"


# The MLP half of a text layer as ONE launch per direction (csrc/text_mlp.hip: 32-row blocks x 256-unit hidden slices, the hidden
# activation never in memory, eight partial products summed by the LayerNorm that follows).  Supported switch:
# PPT_TEXT_MLP_PAIR=0 restores c_fc + split-K c_proj as two launches.
TEXT_MLP_PAIR = os.environ.get("PPT_TEXT_MLP_PAIR", "1") != "0"
TEXT_MLP_PAIR_LN = os.environ.get("PPT_TEXT_MLP_PAIR_LN", "1") != "0"      # ln_2 inside that launch (forward); 0: its own launch in front
TEXT_MLP_PAIR_LAST = os.environ.get("PPT_TEXT_MLP_PAIR_LAST", "1") != "0"  # the last layer's forward too (its slices summed by an extra LayerNorm launch)
# ... and its split16 form (csrc/text_mlp_split.hip) where the text tower runs on fp32 operands as hi + lo half pairs: the whole-model
# split16 mode, and the mixed mode on weights whose text tower failed its self-check (checkpoint-like magnitudes).  0: the tile GEMMs.
TEXT_MLP_PAIR_SPLIT = os.environ.get("PPT_TEXT_MLP_PAIR_SPLIT", "1") != "0"


# The attention half's four linears of a text layer in that mode on csrc/text_lin_split.hip (rows stationary, weight halves streamed):
# in_proj, out_proj (+ bias + residual), and their input-gradient products.  0: the split16 tile GEMMs.
TEXT_LIN_SPLIT = os.environ.get("PPT_TEXT_LIN_SPLIT", "1") != "0"


def _text_lin_ok(Ta, width):
    return Ta == torch.float32 and TEXT_LIN_SPLIT and ops.split16_enabled() and width == 512


def _text_lin_tiles(sd, name, wca, transposed=False):
    """The hi + lo half copy of weight `name` ([N, K]; transposed: of its transpose, the dX operand) for ops.text_lin_split, made with
    the B pre-scale in force (a new fit re-makes it)."""
    w = sd[name]
    b = ops.SPLIT16_POW2[1]
    return wca.derived(("text_lin_split", name, transposed, b), (w,),
                       lambda: ops.text_lin_retile_split(wca.get(w, "wt") if transposed else wca.get(w), b))


def _text_mlp_pair_ok(Tm):
    return Tm in ops.HALF or (Tm == torch.float32 and TEXT_MLP_PAIR_SPLIT and ops.split16_enabled())


def _text_mlp_tiles(sd, p, wcm, backward):
    """The fragment-ordered weight copies of layer `p` for ops.text_mlp_pair: forward (c_fc.weight, c_proj.weight); backward the
    transposed pair (c_proj.weight^T [2048, 512], c_fc.weight^T [512, 2048]) -- the dX operands the WeightCache already keeps."""
    wfc, wpr = sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_proj.weight"]
    if wcm.dtype == torch.float32:          # split16: hi + lo half copies, made with the B pre-scale in force (a new fit re-makes them)
        b = ops.SPLIT16_POW2[1]
        if backward:
            return wcm.derived(("text_mlp_split_bwd", p, b), (wfc, wpr), lambda: ops.text_mlp_retile_split(wcm.get(wpr, "wt"), wcm.get(wfc, "wt"), b))
        return wcm.derived(("text_mlp_split_fwd", p, b), (wfc, wpr), lambda: ops.text_mlp_retile_split(wcm.get(wfc), wcm.get(wpr), b))
    if backward:
        return wcm.derived(("text_mlp_tiles_bwd", p), (wfc, wpr), lambda: ops.text_mlp_retile(wcm.get(wpr, "wt"), wcm.get(wfc, "wt")))
    return wcm.derived(("text_mlp_tiles_fwd", p), (wfc, wpr), lambda: ops.text_mlp_retile(wcm.get(wfc), wcm.get(wpr)))


def text_tower_forward(sd, wc, prompts, eot_pos, heads, layers, save, eff_len=None, prefix=0, rows_in=None):
    """encode_text: prompts [C,L,W] fp32 -> text features [C,E] fp32 (before L2 normalisation).
    save=True keeps what the input-gradient backward needs.

    eff_len: the attention is causal (ULIP_models.py:224-230) and only the EOT token is pooled
    (:222), so positions after the last EOT of any class can influence neither the output nor any
    gradient; the tower is evaluated on the first eff_len = max(eot)+1 positions only (37 of 77 for
    the ModelNet40 prompts).  Outputs and gradients are identical to the full-length evaluation.

    prefix = P > 0: the caller vouches that positions 0 .. P-1 are the same in every prompt (the start token and the
    leading learnable context tokens of PromptLearner's "middle" / "end" layouts, ULIP_models.py:112-148).  Causal attention
    then makes their activations identical in every prompt at every layer, so they are stored and computed ONCE: the tower
    runs on P + C (L - P) rows instead of C L (817 instead of 1 480 for ModelNet40: every LayerNorm / linear of the tower does
    45 % less work), with ppt_attention_prefix_fwd reading the shared rows as every prompt's first P keys.  The forward is
    bit-identical to the unshared evaluation; in the backward the prompts' contributions to the shared rows are summed layer by
    layer instead of at the very end (same sum, another order).

    rows_in = (x0 [M, W] f32, C, Lfull): the first layer's input ALREADY in the tower's row layout with the positional
    embedding added (ops.prompt_rows builds it from the learnable tokens in one kernel); `prompts` is then None and the
    backward hands back the gradient of x0 as it is."""
    T = wc.dtype
    if rows_in is not None:
        x0, C, Lfull = rows_in
        Wd = x0.shape[1]
    else:
        C, Lfull, Wd = prompts.shape
    L = Lfull if eff_len is None else min(Lfull, int(eff_len))
    P = int(prefix) if (prefix and 0 < int(prefix) < L and C > 1) else 0
    dev = x0.device if rows_in is not None else prompts.device
    if rows_in is not None:
        M = ops.prefix_rows(C, L, P) if P else C * L
        assert x0.shape[0] == M
        xin, add, add_rows = x0, None, 0
        # rows of the EOT tokens (ULIP_models.py:222): a function of the token ids alone -- cached, not four tiny kernels per forward
        rows = wc.derived(("text_eot_rows", C, L, P), (eot_pos,), lambda: ((P + torch.arange(C, device=dev) * (L - P) + (eot_pos - P)) if P
                                                                           else (torch.arange(C, device=dev) * L + eot_pos)))
    elif P:
        M = ops.prefix_rows(C, L, P)
        xin = torch.cat([prompts[0, :P], prompts[:, P:L].reshape(C * (L - P), Wd)], dim=0)
        add = wc.derived(("text_pos_prefix", C, L, P), (sd["positional_embedding"],),
                         lambda: torch.cat([sd["positional_embedding"][:P], sd["positional_embedding"][P:L].repeat(C, 1)], dim=0).contiguous())
        add_rows = 0
        rows = P + torch.arange(C, device=dev) * (L - P) + (eot_pos - P)             # EOT pooling (ULIP_models.py:222); eot >= P
    else:
        if L != Lfull:
            prompts = prompts[:, :L].contiguous()
        M = C * L
        add, add_rows = sd["positional_embedding"], L             # x = prompts + pos[:L] (ULIP_models.py:210)
        xin = prompts.reshape(M, Wd)
        rows = torch.arange(C, device=dev) * L + eot_pos            # EOT pooling (ULIP_models.py:222)
    saved = {"layers": []} if save else None
    x = xin if add is None else torch.empty((M, Wd), dtype=torch.float32, device=dev)
    fuse = TEXT_FUSE_LN and T in ops.HALF and Wd in ops.ROWGEMM_K
    # (diagnostics, tools/bf16_error.py: the attention half and the MLP half of every layer may run at another operand precision;
    # both start from and end in the fp32 residual stream)
    wca, wcm = _stage_wc(wc, "text_attn"), _stage_wc(wc, "text_mlp")
    Ta, Tm = wca.dtype, wcm.dtype
    if Ta != T or Tm != T:
        fuse = False

    def stats():
        return (torch.empty((M,), dtype=torch.float32, device=dev), torch.empty((M,), dtype=torch.float32, device=dev)) if save else None
    # split-K for the K = 2048 linear (c_proj): with 817 rows it is 104 workgroups walking 32 K slabs each at ONE CU's L2 -> LDS rate
    # (~9 of its 12.6 us); as four K slices it is 416 workgroups of 8 slabs, and the four fp32 partial products are added up -- with
    # the residual and the bias -- by the LayerNorm that reads the result anyway (ops.layernorm_fwd_sum): no reduction launch.
    # (16-bit modes and split16; the fp32 parity mode keeps the single-launch summation order)
    splitk = TEXT_SPLITK and (T in ops.HALF or ops.split16_enabled()) and Tm == T and not fuse and M <= 4096 and Wd == 512
    lin_split = _text_lin_ok(Ta, Wd) and not fuse
    pending = None                  # (x_mid, c_proj bias, partial products) of the previous layer: its output is formed by this layer's LN1
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        if pending is not None:
            x = torch.empty((M, Wd), dtype=torch.float32, device=dev)
            h, mean1, rstd1 = ops.layernorm_fwd_sum(pending[0], pending[1], pending[2], sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], Ta,
                                                    write_xs=x, save_stats=save)
            pending = None
            if lin_split:
                qkv = ops.text_lin_split(h, _text_lin_tiles(sd, p + "attn.in_proj_weight", wca), bias=sd[p + "attn.in_proj_bias"])
            else:
                qkv = ops.gemm(h, wca.get(sd[p + "attn.in_proj_weight"]), out_dtype=Ta, bias=sd[p + "attn.in_proj_bias"])
        elif fuse and add is None:
            # LayerNorm applied while the rows are staged (one node of the prompt chain instead of two); its statistics are
            # kept for the backward
            st1 = stats()
            mean1, rstd1 = st1 if save else (None, None)
            qkv = ops.rowgemm(xin, wc.get(sd[p + "attn.in_proj_weight"]), ln=(sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]), ln_stats=st1,
                              bias=sd[p + "attn.in_proj_bias"])
        else:
            h, mean1, rstd1 = ops.layernorm_fwd(xin, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], Ta, add=add,
                                                add_rows=add_rows, write_xs=x if add is not None else None,
                                                save_stats=save)
            if lin_split:
                qkv = ops.text_lin_split(h, _text_lin_tiles(sd, p + "attn.in_proj_weight", wca), bias=sd[p + "attn.in_proj_bias"])
            else:
                qkv = ops.gemm(h, wca.get(sd[p + "attn.in_proj_weight"]), out_dtype=Ta, bias=sd[p + "attn.in_proj_bias"])
        add, add_rows = None, 0
        if P:
            a, lse = ops.attention_prefix_fwd(qkv, C, L, P, heads, ATTN_SCALE, want_lse=save)
        else:
            a, lse = ops.attention_fwd(qkv, C, L, heads, ATTN_SCALE, True, want_lse=save)
        x_mid = torch.empty_like(x)
        if lin_split:
            ops.text_lin_split(a, _text_lin_tiles(sd, p + "attn.out_proj.weight", wca), bias=sd[p + "attn.out_proj.bias"], residual=x, out=x_mid)
        else:
            ops.gemm(a, wca.get(sd[p + "attn.out_proj.weight"]), out=x_mid, bias=sd[p + "attn.out_proj.bias"], residual=x)
        pre = torch.empty((M, sd[p + "mlp.c_fc.weight"].shape[0]), dtype=Tm, device=dev) if save else None
        # (the LAST layer has no LayerNorm of a next layer to add its partial products up: ln_final's launch over all rows does it --
        # its normalised output is not used, the summed rows are; split16: 2 x 41 us of tile GEMMs -> 23 + 6 us)
        pair = (TEXT_MLP_PAIR and splitk and (i + 1 < layers or TEXT_MLP_PAIR_LAST) and _text_mlp_pair_ok(Tm)
                and tuple(sd[p + "mlp.c_fc.weight"].shape) == (2048, 512))
        if pair:
            # c_fc + QuickGELU + c_proj in one launch; its eight partial products are this layer's output once the next layer's
            # LayerNorm has added them to x_mid and the bias (the split-K hand-over below, with 8 slices instead of 4)
            w1t, w2t = _text_mlp_tiles(sd, p, wcm, False)
            if Tm == torch.float32 and TEXT_MLP_PAIR_LN:
                parts, mean2, rstd2 = ops.text_mlp_pair_split(x_mid, w1t, w2t, bias=sd[p + "mlp.c_fc.bias"], pre=pre,
                                                              ln=(sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]), save_stats=save)
            elif Tm == torch.float32:
                h2, mean2, rstd2 = ops.layernorm_fwd(x_mid, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], Tm, save_stats=save)
                parts = ops.text_mlp_pair_split(h2, w1t, w2t, bias=sd[p + "mlp.c_fc.bias"], pre=pre)
            elif TEXT_MLP_PAIR_LN:
                # ... and ln_2 is applied while the kernel stages its rows: the LayerNorm launch in front of it goes as well
                parts, mean2, rstd2 = ops.text_mlp_pair(x_mid, w1t, w2t, bias=sd[p + "mlp.c_fc.bias"], pre=pre,
                                                        ln=(sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]), save_stats=save)
            else:
                h2, mean2, rstd2 = ops.layernorm_fwd(x_mid, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], Tm, save_stats=save)
                parts = ops.text_mlp_pair(h2, w1t, w2t, bias=sd[p + "mlp.c_fc.bias"], pre=pre)
            if save:
                saved["layers"].append(dict(x=x, mean1=mean1, rstd1=rstd1, qkv=qkv, a=a, lse=lse, x_mid=x_mid, mean2=mean2,
                                            rstd2=rstd2, pre=pre))
            if i + 1 < layers:
                pending = (x_mid, sd[p + "mlp.c_proj.bias"], parts)
            else:
                x = torch.empty_like(x_mid)
                ops.layernorm_fwd_sum(x_mid, sd[p + "mlp.c_proj.bias"], parts, sd["ln_final.weight"], sd["ln_final.bias"], torch.float32,
                                      write_xs=x)
                xin = x
            continue
        if fuse:
            st2 = stats()
            mean2, rstd2 = st2 if save else (None, None)
            f = ops.rowgemm(x_mid, wc.get(sd[p + "mlp.c_fc.weight"]), ln=(sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]), ln_stats=st2,
                            bias=sd[p + "mlp.c_fc.bias"], act=ACT_QUICKGELU, out2=pre)
        else:
            h2, mean2, rstd2 = ops.layernorm_fwd(x_mid, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], Tm, save_stats=save)
            f = ops.gemm(h2, wcm.get(sd[p + "mlp.c_fc.weight"]), out_dtype=Tm, bias=sd[p + "mlp.c_fc.bias"],
                         act=ACT_QUICKGELU, out2=pre, out2_pre=True)
        if save:
            saved["layers"].append(dict(x=x, mean1=mean1, rstd1=rstd1, qkv=qkv, a=a, lse=lse, x_mid=x_mid, mean2=mean2,
                                        rstd2=rstd2, pre=pre))
        if splitk and i + 1 < layers:
            pending = (x_mid, sd[p + "mlp.c_proj.bias"], ops.gemm_splitk(f, wcm.get(sd[p + "mlp.c_proj.weight"]), 4))
            continue
        x_next = torch.empty_like(x)
        ops.gemm(f, wcm.get(sd[p + "mlp.c_proj.weight"]), out=x_next, bias=sd[p + "mlp.c_proj.bias"], residual=x_mid)
        x = x_next
        xin = x
    x_eot = x.index_select(0, rows)
    hn, meanf, rstdf = ops.layernorm_fwd(x_eot, sd["ln_final.weight"], sd["ln_final.bias"], torch.float32,
                                         save_stats=save)
    wc32 = _f32_cache(wc)
    out = ops.rows_matmul(hn, wc32.get(sd["text_projection"], "f32")) if hn.shape[0] <= 256 else None      # x @ text_projection (ULIP_models.py:222)
    if out is None:
        out = ops.gemm(hn, wc32.get(sd["text_projection"], "wt"), out_dtype=torch.float32)
    if save:
        saved.update(x_eot=x_eot, meanf=meanf, rstdf=rstdf, rows=rows, C=C, L=L, Lfull=Lfull, W=Wd, heads=heads, P=P, M=M,
                     rows_mode=rows_in is not None, wca=wca, wcm=wcm)
    return out, saved


_F32_CACHES = {}


def _f32_cache(wc):
    """fp32 operand cache riding along a WeightCache (head GEMMs always run in fp32)."""
    c = _F32_CACHES.get(id(wc))
    if c is None or c[0] is not wc:
        c = (wc, WeightCache(torch.float32))
        _F32_CACHES[id(wc)] = c
    return c[1]


def text_tower_backward(sd, wc, s, dout, grad_scale=1.0):
    """Input gradient of encode_text: dout [C,E] -> d prompts [C,L,W] fp32.  The tower is frozen
    (ULIP_models.py:487-507), so no weight gradient is ever formed: 4 dX GEMMs, 2 LayerNorm
    backwards and one attention backward per layer.  With a shared prefix (text_tower_forward) the gradient of the shared
    rows is handed to prompt 0; the caller (PromptLearner's index_put) sums over the prompts anyway.

    grad_scale = S (ppt_amd/gradscale.py): the layers' activation gradients travel in the tower's 16-bit operand format, so the
    gradient enters them multiplied by the power of two S -- folded into the very first product, d_hn = S * dout @ P^T -- and
    leaves multiplied by 1 / S: in the row-layout mode by the CALLER (ops.prompt_rows_bwd(..., scale = 1 / S) folds it into the
    kernel that sums the rows onto the tokens; the returned g is still scaled), otherwise here."""
    T = wc.dtype
    S = float(grad_scale)
    C, L, Wd, heads, P, M = s["C"], s["L"], s["W"], s["heads"], s["P"], s["M"]
    wc32 = _f32_cache(wc)
    d_hn = ops.rows_matmul(dout.contiguous(), wc32.get(sd["text_projection"], "wt"), alpha=S) if dout.shape[0] <= 256 else None
    if d_hn is None:
        d_hn = ops.gemm(dout.contiguous() if S == 1.0 else dout * S, wc32.get(sd["text_projection"], "w"), out_dtype=torch.float32)
    d_eot, _, _ = ops.layernorm_bwd(d_hn, s["x_eot"], sd["ln_final.weight"], s["meanf"], s["rstdf"])
    g = torch.zeros((M, Wd), dtype=torch.float32, device=dout.device)
    g.index_copy_(0, s["rows"], d_eot)
    wca, wcm = s.get("wca", wc), s.get("wcm", wc)              # (per-half operand precision: diagnostics, see the forward)
    Ta, Tm = wca.dtype, wcm.dtype
    splitk = TEXT_SPLITK and (T in ops.HALF or ops.split16_enabled()) and Ta == T and Tm == T and M <= 4096 and Wd == 512
    lin_split = _text_lin_ok(Ta, Wd) and splitk and Tm == Ta
    g_t = ops.convert(g, Tm)
    for i in reversed(range(len(s["layers"]))):
        p = f"transformer.resblocks.{i}."
        ly = s["layers"][i]
        if TEXT_MLP_PAIR and splitk and _text_mlp_pair_ok(Tm) and tuple(sd[p + "mlp.c_fc.weight"].shape) == (2048, 512):
            # the branch's input gradient ((g W_proj) * QuickGELU'(pre)) W_fc in one launch: eight partial products, added up by the
            # LayerNorm backward that reads them
            w1t, w2t = _text_mlp_tiles(sd, p, wcm, True)
            pair_fn = ops.text_mlp_pair_split if Tm == torch.float32 else ops.text_mlp_pair
            _, g_t = ops.layernorm_bwd_sum(pair_fn(g_t, w1t, w2t, pre=ly["pre"], backward=True), ly["x_mid"],
                                           sd[p + "ln_2.weight"], ly["mean2"], ly["rstd2"], g, accumulate=True, copy_dtype=Ta)
            d_pre = None
        else:
            d_pre = ops.gemm(g_t, wcm.get(sd[p + "mlp.c_proj.weight"], "wt"), out_dtype=Tm, act=ACT_QUICKGELU,
                             dact_pre=ly["pre"])
        if d_pre is None:
            pass
        elif splitk:
            # K = 2048 over 817 rows: four K slices, added up by the LayerNorm backward that reads the product (see the forward)
            _, g_t = ops.layernorm_bwd_sum(ops.gemm_splitk(d_pre, wcm.get(sd[p + "mlp.c_fc.weight"], "wt"), 4), ly["x_mid"],
                                           sd[p + "ln_2.weight"], ly["mean2"], ly["rstd2"], g, accumulate=True, copy_dtype=Ta)
        else:
            d_h2 = ops.gemm(d_pre, wcm.get(sd[p + "mlp.c_fc.weight"], "wt"), out_dtype=torch.float32)
            _, _, _, g_t = ops.layernorm_bwd(d_h2, ly["x_mid"], sd[p + "ln_2.weight"], ly["mean2"], ly["rstd2"], dx=g,
                                             accumulate=True, copy_dtype=Ta)
        if lin_split:
            d_a = ops.text_lin_split(g_t, _text_lin_tiles(sd, p + "attn.out_proj.weight", wca, True))
        else:
            d_a = ops.gemm(g_t, wca.get(sd[p + "attn.out_proj.weight"], "wt"), out_dtype=Ta)
        if P:
            d_qkv = ops.attention_prefix_bwd(ly["qkv"], ly["a"], d_a, ly["lse"], C, L, P, heads, ATTN_SCALE)
        else:
            d_qkv = ops.attention_bwd(ly["qkv"], ly["a"], d_a, ly["lse"], C, L, heads, ATTN_SCALE, True)
        if splitk:                  # K = 1536: three slices
            dparts = (ops.text_lin_split(d_qkv, _text_lin_tiles(sd, p + "attn.in_proj_weight", wca, True)) if lin_split
                      else ops.gemm_splitk(d_qkv, wca.get(sd[p + "attn.in_proj_weight"], "wt"), 3))
            _, g_t = ops.layernorm_bwd_sum(dparts, ly["x"],
                                           sd[p + "ln_1.weight"], ly["mean1"], ly["rstd1"], g, accumulate=True, copy_dtype=Tm)
        else:
            d_h = ops.gemm(d_qkv, wca.get(sd[p + "attn.in_proj_weight"], "wt"), out_dtype=torch.float32)
            _, _, _, g_t = ops.layernorm_bwd(d_h, ly["x"], sd[p + "ln_1.weight"], ly["mean1"], ly["rstd1"], dx=g,
                                             accumulate=True, copy_dtype=Tm)
    if s["rows_mode"]:
        return g                                            # gradient of the row-layout input, STILL scaled by S (ops.prompt_rows_bwd folds it and 1 / S)
    if S != 1.0:
        g.mul_(1.0 / S)
    if not P and L == s["Lfull"]:
        return g.view(C, L, Wd)
    full = torch.zeros((C, s["Lfull"], Wd), dtype=torch.float32, device=dout.device)   # positions past the last EOT: zero gradient
    if P:
        full[0, :P] = g[:P]
        full[:, P:L] = g[P:].view(C, L - P, Wd)
    else:
        full[:, :L] = g.view(C, L, Wd)
    return full


def head_loss_forward_backward(feat, wt, text_raw, logit_scale, labels, smoothing, w=None):
    """The step between the towers when only the prompt trains (head_type 0): pc_embed = feat @ pc_projection
    (ULIP_models.py:257), text features L2-normalised (:279), logits = exp(logit_scale) * pc_embed @ text^T (:281),
    label-smoothed cross entropy with mean reduction (main_cls.py:52,196) -- and, in the same pass, the gradient of the
    loss w.r.t. the un-normalised text features, which is all the backward needs.  ~20 tiny launches with no autograd
    bookkeeping in between: shape-static, so the caller replays them from a hipGraph.
    feat [B,F] f32, wt [E,F] f32 (pc_projection^T) or w [F,E] (pc_projection), text_raw [C,E] f32, logit_scale 0-d, labels [B] i64
    -> (loss 0-d, logits [B,C], d loss / d text_raw [C,E])."""
    B, C = feat.shape[0], text_raw.shape[0]
    if w is not None and w.dtype == torch.float32 and feat.shape[1] <= 8192 and B <= 4096 and logit_scale.dtype == torch.float32:
        # three small kernels for the whole step (ppt_head_logits, ppt_head_ce_bwd) instead of ~30 tiny launches;
        # w = pc_projection [F,E] as stored
        return ops.head_loss(feat.contiguous(), w.contiguous(), text_raw.contiguous(), logit_scale.contiguous(),
                             labels.contiguous(), smoothing)
    pc = ops.gemm(feat.contiguous(), wt, out_dtype=torch.float32, split=False)          # [B,E]  (the head: B rows, un-scaled gradients -> fp32 MFMA)
    nrm = text_raw.norm(dim=-1, keepdim=True)
    tn = text_raw / nrm
    spc = (logit_scale.exp() * pc).contiguous()
    logits = ops.gemm(spc, tn.contiguous(), out_dtype=torch.float32, split=False)       # [B,C]
    logp = torch.log_softmax(logits, dim=1)
    nll = -logp.gather(1, labels.view(B, 1)).squeeze(1)
    loss = ((1.0 - smoothing) * nll + smoothing * (-logp.mean(dim=1))).mean()
    target = torch.full_like(logp, smoothing / C)
    target.scatter_add_(1, labels.view(B, 1), torch.full((B, 1), 1.0 - smoothing, dtype=logp.dtype, device=logp.device))
    dlogits = (logp.exp() - target) / B
    # d tn [C,E] = dlogits^T [C,B] @ spc [B,E]   (NT GEMM on the transposed operands, K = B padded to the chunk size)
    d_tn = ops.gemm(ops.transpose(dlogits, pad_to=4), ops.transpose(spc, pad_to=4), out_dtype=torch.float32, split=False)
    d_raw = (d_tn - tn * (d_tn * tn).sum(dim=-1, keepdim=True)) / nrm
    return loss, logits, d_raw
