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
  * the word is CLEARED behind every read (stream-ordered, on the stream the checks are queued on) and the skip counter is compared
    with its value at the previous poll, so every poll reports what happened SINCE the last one and a later, different overflow is
    seen as well (a sticky word let BIT_POINT / BIT_LOSS mask every later BIT_GRAD);
  * on a set bit the Trainer DEMOTES the offending side to bf16 (same MFMA rate, fp32's exponent range, 8 significand bits) for
    the rest of the run and says so: BIT_POINT -> tokenizer + blocks (+ part-seg decoder); BIT_LOSS alone -> text tower + head;
    BIT_GRAD (the optimizer's skip counter moved although features and loss were finite: a BACKWARD stage overflowed -- with
    gains of 10 the text tower's half gradients do, tools/fp16_stress.py) -> every stage with a 16-bit backward: text tower,
    un-frozen last block, part-seg decoder and head;
  * split16 (fp32 operands as hi + lo half pairs) has half's RANGE too: its GEMMs saturate a finite operand value beyond 65 504
    instead of turning it into inf / NaN and count the waves that did (ppt_gemm_params.split_overflow -> BIT_SPLIT here): the
    model then leaves the split16 products for the fp32 MFMA (`set_precision("fp32")`, or `text_split16 = False` for a mixed-mode
    text tower that ran on them), with a warning -- the stored values are fp32 either way, so nothing else changes;
  * when events keep arriving and nothing is left to demote (bf16 has fp32's range: then it is not a half overflow -- diverging
    training or bad data), or a label is outside [0, C) (BIT_LABEL: ATen raises a device assert there), the Trainer raises
    FloatingPointError / ValueError instead of skipping steps silently.
"""
import contextlib
import warnings

import torch

BIT_POINT, BIT_LOSS, BIT_GRAD, BIT_LABEL, BIT_SPLIT = 1, 2, 4, 8, 16
NBITS = 5
GIVE_UP_AFTER = 3            # consecutive polls with an event that no demotion can answer -> FloatingPointError


class Monitor:
    def __init__(self, device, every=50):
        self.flags = torch.zeros((1,), dtype=torch.int32, device=device)
        pin = torch.cuda.is_available()
        self.host = torch.zeros((1,), dtype=torch.int32).pin_memory() if pin else torch.zeros((1,), dtype=torch.int32)
        self.host_skipped = torch.zeros((1,), dtype=torch.int64).pin_memory() if pin else torch.zeros((1,), dtype=torch.int64)
        self.every = int(every)
        self._pending = None
        self.seen = 0                       # every bit ever reported (a log, NOT a mask: detection stays armed)
        self.polls = 0
        self.skipped_seen = 0
        self.unanswered = 0                 # consecutive polls whose event found nothing left to demote (train.Trainer)
        self.skipped = None                 # the optimizer's device counter of skipped (non-finite) gradient elements (train.Trainer)
        # split16: the process-wide counter of GEMM waves that saturated an operand beyond half's range (ops.split16_overflow_counter)
        self.split_counter = None
        self.host_split = torch.zeros((1,), dtype=torch.int32).pin_memory() if pin else torch.zeros((1,), dtype=torch.int32)
        self.split_seen = 0
        if torch.device(device).type == "cuda":
            from . import ops
            self.split_counter = ops.split16_overflow_counter(device)
            self.split_seen = int(self.split_counter.item())          # (events of earlier models / runs in this process are not ours)
            self.host_split[0] = self.split_seen

    def check(self, bit, t):
        """queue the non-finite check of tensor t on the current stream (ppt_health_check).  Every check of a run is queued on
        the stream poll() is called on (the caller's), which is what makes the clear in poll() race-free."""
        from . import ops
        if t is not None and t.is_cuda and t.numel():
            ops.health_check(t.detach().contiguous(), self.flags, bit)

    def check_labels(self, labels, n_classes, ignore_index=-100):
        """queue the corrupt-label check (ppt_labels_check): BIT_LABEL is a data bug, reported as such, never a demotion"""
        from . import ops
        if labels is not None and labels.is_cuda and labels.dtype == torch.int64 and labels.numel():
            ops.labels_check(labels.contiguous(), n_classes, ignore_index, self.flags, BIT_LABEL)

    def _resolve(self):
        bits = int(self.host.item())
        sk = int(self.host_skipped.item())
        # gradient elements were skipped since the last poll although features and loss were finite in the same window: a 16-bit
        # BACKWARD stage overflowed.  (A NaN loss makes every gradient NaN: those skips are the BIT_LOSS event, not a new one.)
        if sk > self.skipped_seen and not (bits & (BIT_POINT | BIT_LOSS | BIT_LABEL)):
            bits |= BIT_GRAD
        self.skipped_seen = sk
        sp = int(self.host_split.item())
        if sp != self.split_seen:           # (a wrapping 32-bit counter: any change is an event)
            bits |= BIT_SPLIT
        self.split_seen = sp
        self.seen |= bits
        if not bits:
            self.unanswered = 0             # (a clean window: "consecutive" starts over)
        return bits

    def poll(self, step, side=None):
        """-> the bits set since the previous poll (0 almost always).  Every `every` steps the flag word is copied to pinned
        memory behind the work queued so far and cleared behind the copy; the optimizer's skip counter is copied on `side`, the
        stream the optimizer runs on (train.Trainer: the text stream), i.e. behind this step's AdamW (a copy on the
        caller's stream raced with it).  The copies are looked at when they have completed -- normally one poll later -- so
        nothing ever waits.  Call at the END of a step, after the optimizer has been queued."""
        new = 0
        if self._pending is not None and all(e.query() for e in self._pending):
            self._pending = None
            new = self._resolve()
        if self.every > 0 and step % self.every == 0 and self._pending is None and self.flags.is_cuda:
            self.host.copy_(self.flags, non_blocking=True)
            self.flags.zero_()                                     # (same stream: behind the copy, in front of the next check)
            if self.split_counter is not None:
                self.host_split.copy_(self.split_counter, non_blocking=True)
            evs = [torch.cuda.Event()]
            evs[0].record()
            if self.skipped is not None:
                with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                    self.host_skipped.copy_(self.skipped, non_blocking=True)
                    if side is not None:
                        evs.append(torch.cuda.Event())
                        evs[1].record()
                    else:
                        evs[0].record()
            self._pending = evs
            self.polls += 1
        return new

    def poll_all_ranks(self, step, side=None, group=None):
        """The data-parallel form of poll(): every rank must take the SAME demotion decision at the SAME step (a rank that re-captures
        its graphs alone stalls the others' collective, and `event.query()` timing differs per rank -- ADVICE r4).  Every `every` steps:
        a blocking read of this rank's word and skip counter, then ONE all-reduce (MAX over the ranks of [bit 0, bit 1, ..., counter])
        so that each rank sees the union of the events; ~one device synchronisation per `every` steps."""
        import torch.distributed as dist
        if not (self.every > 0 and step % self.every == 0 and self.flags.is_cuda):
            return 0
        if side is not None:
            side.synchronize()
        bits = int(self.flags.item())
        self.flags.zero_()
        sk = int(self.skipped.item()) if self.skipped is not None else 0
        sp = int(self.split_counter.item()) if self.split_counter is not None else self.split_seen
        if sp != self.split_seen:
            bits |= BIT_SPLIT
        self.split_seen = sp
        self.host_split[0] = sp
        v = torch.tensor([float((bits >> b) & 1) for b in range(NBITS)] + [float(sk)], dtype=torch.float64, device=self.flags.device)
        dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
        v = v.cpu()
        self.host[0] = sum(int(v[b].item()) << b for b in range(NBITS))
        self.host_skipped[0] = int(v[NBITS].item())
        self.polls += 1
        return self._resolve()

    def read_now(self, side=None):
        """blocking read of everything since the last poll (tests, end of an epoch, after a FloatingPointError)"""
        if self._pending is not None:
            for e in self._pending:
                e.synchronize()
            self._pending = None
            new = self._resolve()
        else:
            new = 0
        if side is not None:
            side.synchronize()
        self.host.copy_(self.flags)
        self.flags.zero_()
        if self.skipped is not None:
            self.host_skipped.copy_(self.skipped)
        if self.split_counter is not None:
            torch.cuda.synchronize(self.split_counter.device)     # (the GEMMs that add to it run on any stream)
            self.host_split.copy_(self.split_counter)
        return new | self._resolve()


def demote(model, bits):
    """Move the side named by `bits` from IEEE half to bf16 operands for the rest of the run; returns what was done -- an empty
    list when every stage the bits point at already runs on bf16 (the caller counts those: Monitor.unanswered)."""
    done = []
    pe = getattr(model, "point_encoder", None)
    demoted = model.__dict__.setdefault("demoted", set())
    if bits & BIT_SPLIT:
        # a split16 GEMM saturated an operand: the values are fp32 in memory, only the PRODUCTS were formed from half pairs --
        # form them on the fp32 MFMA from here on (slower, fp32's range)
        if getattr(model, "split16", False):
            model.set_precision("fp32")
            done.append("every split16 product (the model now runs in the fp32 mode)")
        elif getattr(model, "text_split16", False) and getattr(model, "text_precision", None) is torch.float32:
            model.text_split16 = False
            done.append("the text tower's split16 products (fp32 MFMA from here on)")
        if done:
            if hasattr(model, "reset_caches") and not getattr(model, "split16", False):
                keep = (model.text_precision, model.text_calibration, model._text_calibrated) if hasattr(model, "text_calibration") else None
                model.reset_caches()
                if keep is not None:        # (the calibration's verdict stands: the weights did not change)
                    model.text_precision, model.text_calibration, model._text_calibrated = keep
            warnings.warn("ppt_amd: a split16 GEMM operand exceeded IEEE half's range (saturated to +-65 504 for that launch): "
                          + "; ".join(done), RuntimeWarning, stacklevel=3)
        bits &= ~BIT_SPLIT
        if not bits:
            return done
    point_left = not {"tokenizer", "blocks", "last_block", "decoder"} <= demoted
    if bits & BIT_POINT and point_left:
        # (non-finite features make the loss non-finite too: BIT_LOSS in the same window is this event's echo, not a second one)
        demoted.update(("tokenizer", "blocks", "last_block", "decoder"))
        if pe is not None and hasattr(pe, "_dec_precision"):
            pe.precision = pe.precision                           # (the setter re-derives the decoder's operand format)
        done.append("point tower (tokenizer, transformer blocks" + (", part-seg decoder" if hasattr(pe, "_dec_precision") else "") + ")")
    elif bits & (BIT_POINT | BIT_LOSS | BIT_GRAD):
        # the features were finite (or the point tower is on bf16 already) and the loss was not: the text tower or the head
        # products; or both were finite and gradient elements were not: a 16-bit BACKWARD stage -- the text tower, the un-frozen
        # last block, the decoder, the per-point head
        if getattr(model, "text_f16", False):
            model.text_f16 = False
            done.append("CLIP text tower")
        if "head" not in demoted:
            demoted.add("head")
            done.append("per-point head")
        if bits & BIT_GRAD and not {"last_block", "decoder"} <= demoted:
            demoted.update(("last_block", "decoder"))
            if pe is not None and hasattr(pe, "_dec_precision"):
                pe.precision = pe.precision
            done.append("last block / part-seg decoder (the stages with a 16-bit backward)")
    if not done:
        return done
    if hasattr(model, "reset_caches"):
        model.reset_caches()                                      # operand copies and captured graphs were made for the old format
    elif pe is not None and hasattr(pe, "_graphs"):
        pe._graphs.clear()
    warnings.warn("ppt_amd: a 16-bit stage overflowed IEEE half (non-finite " +
                  ("point features" if bits & BIT_POINT else ("loss" if bits & BIT_LOSS else "gradients")) +
                  "); its steps were skipped by the optimizer and these stages now run on bf16 operands: " + "; ".join(done),
                  RuntimeWarning, stacklevel=3)
    return done
