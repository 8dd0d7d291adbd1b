"""Gradient scaling of the 16-bit backward stages, local to each autograd node.

The performance mode carries activation gradients in IEEE half inside four stages -- the CLIP text tower's backward, the
un-frozen last PointBERT block, the part-segmentation decoder and the per-point head (engine.*_F16) -- and half keeps its 11 bits
only down to 6.1e-5.  The criterion of the reference's callers is a MEAN over the rows of the batch (main_cls.py:52,
main_partseg.py:213), so d loss / d logits shrinks with the row count (1 / (B x 2048) per logit in part segmentation) and the
stages' gradients would slide into the subnormals.  Rounds 2-3 answered that in `train.Trainer` (seed `backward()` with S,
un-scale inside the optimizer), which an UNCHANGED caller -- `loss.backward(); optimizer.step()` as in main_cls.py:194-198 and
main_partseg.py:204-215 -- never gets.  Now every node that owns a 16-bit backward stage does it itself:

  * entry: the fp32 gradient it receives is multiplied by a power of two S where it is first converted / projected
    (ppt_convert_scaled, the alpha of ppt_rows_matmul_f32, one tiny ATen multiply for the [B, 768] feature gradient),
  * exit: every gradient it hands out -- input gradients and parameter .grads -- is multiplied by 1 / S (ppt_gemm's row_scale with
    one row group, ppt_prompt_rows_bwd's scale, one multi-tensor multiply for the parameter list),

so that what crosses a node boundary is always the true fp32 gradient: exact (powers of two), invisible to the caller, to ATen ops
between the nodes, to the all-reduce and to any optimizer.

S is fixed when the node's FORWARD runs, on the host, from the number of rows the caller's criterion averages over -- known to the
model's forward (B logits rows in recognition, B x N in part segmentation) and announced to the nodes through `rows(n)`:
S = 2^floor(log2(rows)), i.e. the stages see the gradient of ~the SUM over rows: batch-size invariant and in the middle of the
measured plateau (tools/f16_grad_range.py: the golden step's gradient error is flat while the per-row seed is >= 1/512 and there
is no overflow up to 32 768x above it).  A node whose forward runs outside any `rows` context (a bare PointTransformer, a bare
encode_text) uses its own default.  PPT_LOSS_SCALE = "auto" (default) | a number | "off".
"""
import os
import threading

import torch

_tls = threading.local()

POLICY = os.environ.get("PPT_LOSS_SCALE", "auto")


def pow2_floor(n):
    return float(2 ** max(0, int(n).bit_length() - 1))


class rows:
    """`with gradscale.rows(n):` -- the forward running inside feeds a criterion that averages over n rows."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.prev = getattr(_tls, "rows", None)
        _tls.rows = self.n
        return self

    def __exit__(self, *exc):
        _tls.rows = self.prev
        return False


def current(stage_dtype, default_rows=1):
    """The scale S a node whose forward runs NOW will use in its backward: 1.0 unless the stage carries gradients in a 16-bit
    format; else a power of two from the announced row count (or the node's own default)."""
    if stage_dtype not in (torch.float16, torch.bfloat16):
        # split16 (fp32 storage, hi + lo half products: ops.set_split16) scales like a 16-bit stage: the halves keep their 22 bits
        # only above half's subnormals, and a mean-reduced loss puts un-scaled gradients far below them
        from . import ops
        if not (stage_dtype == torch.float32 and ops.split16_enabled()):
            return 1.0
    pol = POLICY
    if pol in (None, "", "0", "1", "none", "off"):
        return 1.0
    if pol != "auto":
        return float(pol)
    n = getattr(_tls, "rows", None)
    return pow2_floor(n if n else max(1, int(default_rows)))


_INV = {}


def inv_tensor(S, device):
    """[1] fp32 device tensor holding 1 / S: the row_scale of a dX GEMM whose result leaves a scaled stage (row_scale_rows = M:
    one row group).  Created on first use per (S, device); never inside a stream capture (a tensor born there belongs to the
    graph's pool and is written only when the graph replays)."""
    key = (float(S), str(device))
    t = _INV.get(key)
    if t is None:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("gradscale.inv_tensor: first use inside a hipGraph capture (warm the shape up eagerly first)")
        t = torch.full((1,), 1.0 / float(S), dtype=torch.float32, device=device)
        _INV[key] = t
    return t


def unscale_(tensors, S):
    """tensors (a list, None entries skipped) *= 1 / S in one multi-tensor launch."""
    if S == 1.0:
        return
    ts = [t for t in tensors if t is not None]
    if ts:
        torch._foreach_mul_(ts, 1.0 / S)


def scaled_backward(raw):
    """Wraps the `backward` of an autograd node whose stage is 16-bit and that is NOT on a hot path (the callable sub-modules of
    ppt_amd/blocks.py, the per-module nodes of the part-seg decoder when it runs un-graphed): incoming gradients x S, outgoing
    gradients x 1 / S (out of place: an output may be a view of a buffer the node keeps), S = ctx.grad_scale set by the forward.
    The raw function stays reachable as `.raw` for callers that scale a whole chain of such nodes once (autograd._PartsegDecoder)."""
    def backward(ctx, *douts):
        S = getattr(ctx, "grad_scale", 1.0)
        if S == 1.0:
            return raw(ctx, *douts)
        outs = raw(ctx, *[d * S if d is not None else None for d in douts])
        single = not isinstance(outs, tuple)
        outs = [outs] if single else list(outs)
        idx = [i for i, o in enumerate(outs) if isinstance(o, torch.Tensor)]
        if idx:
            for i, o in zip(idx, torch._foreach_mul([outs[i] for i in idx], 1.0 / S)):
                outs[i] = o
        return outs[0] if single else tuple(outs)
    backward.raw = raw
    return backward
