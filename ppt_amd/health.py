"""fp16 defence of the mixed 16-bit mode.

IEEE half carries 11 significand bits but stops at 65 504.  The stages that run on it (engine.*_F16) are bounded by a normalisation
on the synthetic weights every parity number of this repository was measured with -- but real ULIP / SLIP checkpoints
(ULIP_models.py:472-507) are not available offline, and LayerNorm gains of 10 or a few outlier channels are what real checkpoints
have (tools/fp16_stress.py builds such weights and prints the per-stage ranges).  So the mode defends itself at run time:

  * every conversion to half rounds to +-inf on overflow (never saturates), and an inf in a stage's 16-bit activations reaches the
    stage's fp32 output as inf / NaN (GEMM accumulations, softmax, LayerNorm statistics) -- or is harmless (an -inf in front of a
    ReLU / max-pool is what the true large negative value would have become);
  * `Monitor` ORs a bit into a device word when the point tower's features (BIT_POINT) or the loss (BIT_LOSS) are not finite:
    two one-workgroup launches per step, neither on the prompt chain; the word is copied to pinned host memory every `every` steps
    without a stall and looked at one poll later;
  * the optimizer kernels skip -- and count -- every gradient element that is not finite (csrc/optim.hip), so the parameters and
    Adam moments of the steps in between stay clean (a NaN loss makes every gradient NaN: the whole step is a no-op, as
    torch.cuda.amp.GradScaler would make it);
  * on a set bit the Trainer DEMOTES the offending side to bf16 (same MFMA rate, fp32's exponent range, 8 significand bits) for
    the rest of the run and says so: BIT_POINT -> tokenizer + blocks (+ part-seg decoder); BIT_LOSS alone -> text tower + head;
    BIT_GRAD (the optimizer's skip counter moved although features and loss were finite: a BACKWARD stage overflowed -- with
    gains of 10 the text tower's half gradients do, tools/fp16_stress.py) -> every stage with a 16-bit backward: text tower,
    un-frozen last block, part-seg decoder and head.
"""
import warnings

import torch

BIT_POINT, BIT_LOSS, BIT_GRAD = 1, 2, 4


class Monitor:
    def __init__(self, device, every=50):
        self.flags = torch.zeros((1,), dtype=torch.int32, device=device)
        self.host = torch.zeros((1,), dtype=torch.int32).pin_memory() if torch.cuda.is_available() else torch.zeros((1,), dtype=torch.int32)
        self.host_skipped = torch.zeros((1,), dtype=torch.int64).pin_memory() if torch.cuda.is_available() else torch.zeros((1,), dtype=torch.int64)
        self.every = int(every)
        self._pending = None
        self.seen = 0
        self.polls = 0
        self.skipped_seen = 0
        self.skipped = None                 # the optimizer's device counter of skipped (non-finite) gradient elements (train.Trainer)

    def check(self, bit, t):
        """queue the non-finite check of tensor t on the current stream (ppt_health_check)"""
        from . import ops
        if t is not None and t.is_cuda and t.numel():
            ops.health_check(t.detach().contiguous(), self.flags, bit)

    def poll(self, step):
        """-> bits newly seen (0 almost always).  Every `every` steps the flag word is copied to pinned memory behind the work
        queued so far; the copy is looked at when it has completed -- normally one poll later -- so nothing ever waits."""
        new = 0
        if self._pending is not None and self._pending.query():
            self._pending = None
            bits = int(self.host.item())
            sk = int(self.host_skipped.item())
            if sk > self.skipped_seen and not (bits & (BIT_POINT | BIT_LOSS)) and not (self.seen & BIT_GRAD):
                bits |= BIT_GRAD
            self.skipped_seen = sk
            new = bits & ~self.seen
            self.seen |= new
        if self.every > 0 and step % self.every == 0 and self._pending is None and self.flags.is_cuda:
            self.host.copy_(self.flags, non_blocking=True)
            if self.skipped is not None:
                # (the counter is written on the text stream: order the copy behind that stream's work queued so far)
                self.host_skipped.copy_(self.skipped, non_blocking=True)
            self._pending = torch.cuda.Event()
            self._pending.record()
            self.polls += 1
        return new

    def read_now(self):
        """blocking read (tests, end of an epoch)"""
        bits = int(self.flags.item())
        new = bits & ~self.seen
        self.seen |= bits
        return new


def demote(model, bits):
    """Move the side named by `bits` from IEEE half to bf16 operands for the rest of the run; returns what was done."""
    from . import engine
    done = []
    pe = getattr(model, "point_encoder", None)
    if bits & BIT_POINT:
        engine.DEMOTED.update(("tokenizer", "blocks", "last_block", "decoder"))
        if pe is not None and hasattr(pe, "_dec_precision"):
            pe.precision = pe.precision                           # (the setter re-derives the decoder's operand format)
        done.append("point tower (tokenizer, transformer blocks" + (", part-seg decoder" if hasattr(pe, "_dec_precision") else "") + ")")
    elif bits & (BIT_LOSS | BIT_GRAD):
        # the features were finite and the loss was not: the text tower or the head products; or both were finite and gradient
        # elements were not: a 16-bit BACKWARD stage -- the text tower, the un-frozen last block, the decoder, the per-point head
        if getattr(model, "text_f16", False):
            model.text_f16 = False
            done.append("CLIP text tower")
        engine.DEMOTED.add("head")
        done.append("per-point head")
        if bits & BIT_GRAD:
            engine.DEMOTED.update(("last_block", "decoder"))
            if pe is not None and hasattr(pe, "_dec_precision"):
                pe.precision = pe.precision
            done.append("last block / part-seg decoder (the stages with a 16-bit backward)")
    if hasattr(model, "reset_caches"):
        model.reset_caches()                                      # operand copies and captured graphs were made for the old format
    elif pe is not None and hasattr(pe, "_graphs"):
        pe._graphs.clear()
    if done:
        warnings.warn("ppt_amd: a 16-bit stage overflowed IEEE half (non-finite " +
                      ("point features" if bits & BIT_POINT else ("loss" if bits & BIT_LOSS else "gradients")) +
                      "); its steps were skipped by the optimizer and these stages now run on bf16 operands: " + "; ".join(done),
                      RuntimeWarning, stacklevel=3)
    return done
