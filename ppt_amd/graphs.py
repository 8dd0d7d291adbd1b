"""HIP-graph replay of the launch-bound, shape-static sections of the step.

The text tower runs ~110 kernels forward and ~150 backward on 40x37 rows: each
takes 5-15 us on the GPU and about as long to enqueue from Python, so with the point tower running
ahead (train.Trainer.step) the host became the limiter.  A section whose launches depend only on
tensor shapes -- no host read of device data, no data-dependent control flow -- is captured once into
a hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture on the stream our C-ABI launches go to)
and replayed with its inputs copied into the captured buffers.  Eager execution stays the reference
behaviour: the first `warmup` calls of every key run eagerly (they also populate the operand caches,
which must not be allocated inside a capture), and `enabled = False` turns replay off globally.
"""
import os

import torch

enabled = os.environ.get("PPT_HIP_GRAPHS", "1") != "0"      # PPT_HIP_GRAPHS=0: eager launches only (per-dispatch counters)
WARMUP_CALLS = 2


class GraphedCall:
    """fn(*tensors) -> (outputs tuple, keepalive) captured with static inputs.

    `outputs` are the tensors handed back on every replay (same storage each time: a later replay
    overwrites them, callers that keep a result across replays clone it); `keepalive` is anything else
    the capture allocated that a later graph reads (saved activations)."""

    def __init__(self, fn, inputs, pool=None, alias_inputs=False):
        # alias_inputs: the inputs ARE static buffers (another graph's outputs, handed over in place): captured as they are, and
        # __call__ copies nothing for an input that is the captured tensor itself
        self.static_in = list(inputs) if alias_inputs else [t.clone() for t in inputs]
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: only THIS thread's calls are checked against the capture.  Under a process group RCCL's watchdog thread
        # polls the events of earlier collectives (the DDP-constructor broadcast, the previous step's all-reduce); in the default
        # "global" mode such a query during a capture aborts the process with hipErrorStreamCaptureUnsupported (seen once the
        # constructor broadcast existed: tests/dist_single_rank.py, head_type 3)
        # ... and no cyclic garbage collection inside the capture: a collection that happens to run there finalises whatever CUDA
        # objects earlier code left in reference cycles (graphs, events, streams of models that are gone), and a destroy / query of
        # those is not a capturable operation -- the process aborts ("Fatal Python error: Aborted ... Garbage-collecting" under
        # GraphedCall.__init__, seen in the full GPU suite once an allocation count shifted).  torch.cuda.graph collects right
        # before it starts capturing; reference-counted frees inside the capture are the caching allocator's business as before.
        import gc
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(self.graph, pool=pool, capture_error_mode="thread_local"):
                self.outputs, self.keepalive = fn(*self.static_in)
        finally:
            if gc_was_on:
                gc.enable()

    def pool(self):
        return self.graph.pool()

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if dst is not src:
                dst.copy_(src)
        self.graph.replay()
        return self.outputs, self.keepalive


class GraphCache:
    """Per-model table key -> GraphedCall with a call counter per key (eager until warmed up)."""

    def __init__(self):
        self.entries = {}
        self.calls = {}

    def clear(self):
        self.entries.clear()
        self.calls.clear()

    def ready(self, key):
        """True when `key` should be served by a graph (captured already, or warmed up and capturable now)."""
        if not enabled:
            return False
        if key in self.entries:
            return True
        n = self.calls.get(key, 0)
        self.calls[key] = n + 1
        return n >= WARMUP_CALLS

    def get(self, key, build):
        g = self.entries.get(key)
        if g is None:
            g = self.entries[key] = build()
        return g


_TEXT_STREAMS = {}


def shared_text_stream(device=None, priority=None):
    """The process-wide side stream of `device` (one per GPU, shared by every model): created on first call.
    Which hardware queue a HIP stream gets depends on how many streams the process created before it, and some
    positions execute in order with the caller's stream (4.8 -> 6.2 ... 7.5 ms per C2 step, tools/prio_probe.py) -- so the
    stream is created ONCE, and callers that know better create it early: bench.py calls this right after
    set_device, before init_process_group lets RCCL create its own streams, which puts it in the same position as in
    the single-GPU run."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev not in _TEXT_STREAMS:
        with torch.cuda.device(dev):
            # priority (first call only; PPT_TEXT_STREAM_PRIORITY overrides): -1 lets the prompt side's small kernels ahead of
            # the point tower's.  Every iteration's text forward waits for the previous iteration's text backward + AdamW,
            # and once the tower is fast enough that chain -- stretched by contention -- is what the head waits for
            # (C2: +1 ... 1.6 % at -1 in same-box A/B runs; C3, whose caller stream waits for the optimizer anyway: -1.6 %; PointMLP -0.7 %)
            prio = os.environ.get("PPT_TEXT_STREAM_PRIORITY")
            prio = int(prio) if prio is not None else (0 if priority is None else int(priority))
            _TEXT_STREAMS[dev] = torch.cuda.Stream(priority=prio)
    return _TEXT_STREAMS[dev]


def ready_event(t):
    """The event that marks tensor `t` complete in device memory, if its producer attached one (ppt_amd.data.DevicePrefetcher:
    the end of the host-to-device copy on the copy stream), else None.  An input-only stage may then start on its own stream
    behind this event alone -- it need not wait for the caller's stream, and the caller need not vouch for anything."""
    return getattr(t, "_ppt_ready", None)


def wait_inputs(stream, tensors, main=None, vouched=False):
    """Order `stream` (an ahead stage's) behind the producers of `tensors`: the copy event a tensor carries (DevicePrefetcher), or
    -- for a tensor WITHOUT one that the caller has not vouched for either -- the caller's stream `main`.  The second case is the
    derived tensor: the batch object carried the event, but what the stage reads is `pc.contiguous().float()` / a slice / a `.to()`
    of it, a NEW tensor produced on the caller's stream that no event covers (the stage, which deliberately does not
    wait for the caller's stream, could read it before it was written).  Falling back to stream order costs that step its overlap
    and nothing else.  vouched: Trainer.inputs_ready / eval_inputs_ready -- the caller's promise covers every tensor of the call."""
    in_order = False
    for t in tensors:
        ev = ready_event(t)
        if ev is not None:
            stream.wait_event(ev)
        elif not vouched:
            in_order = True
    if in_order and main is not None:
        stream.wait_stream(main)


class AheadStage:
    """A shape-static stage that depends on the step's INPUTS only (FPS, ball queries, kNN), replayed on its own stream
    as soon as the step is called -- under the previous iteration's compute instead of at the head of this one's.  Two
    captured copies with their own output buffers alternate, because the consumer of the previous outputs may still be
    running; a copy is reused only after the consumer that read it has finished.  The stage's stream waits for nothing
    else -- in particular not for the caller's stream: the caller vouches that the inputs are complete in memory."""

    def __init__(self):
        self.slot = 0
        self.free = [None, None]

    def run(self, cache, key, fn, ins, side, vouched=False):
        """fn(*ins) -> (outputs, keepalive) as for GraphedCall.  -> (outputs of this call, slot).  vouched: the caller promised that
        `ins` are complete in device memory (Trainer.inputs_ready); otherwise every tensor of `ins` must carry its copy event, or
        the stage falls back to waiting for the caller's stream (wait_inputs)."""
        main = torch.cuda.current_stream()
        slot = self.slot = 1 - self.slot
        with torch.cuda.stream(side):
            if self.free[slot] is not None:
                side.wait_event(self.free[slot])
            wait_inputs(side, ins, main, vouched)     # (a DevicePrefetcher batch: behind its copy; a vouched-for tensor: nothing)
            outs, _ = cache.get(tuple(key) + (slot,), lambda: GraphedCall(fn, ins))(*ins)
            done = side.record_event()
        for t in ins:
            t.record_stream(side)
        main.wait_event(done)
        return outs, slot

    def consumed(self, slot):
        """Call after the consumer of `slot`'s outputs has been queued on the current stream."""
        self.free[slot] = torch.cuda.current_stream().record_event()


_GROUP_STREAMS = {}


def shared_group_stream(device=None):
    """The process-wide stream the grouping stage of the NEXT iteration runs on (FPS + kNN depend on the input cloud
    only: point_encoder.PointTransformer._group_ahead).  Created once per GPU, like the text stream and for the same
    reason; callers that care about hardware-queue placement create it right after shared_text_stream()."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev not in _GROUP_STREAMS:
        with torch.cuda.device(dev):
            _GROUP_STREAMS[dev] = torch.cuda.Stream()
    return _GROUP_STREAMS[dev]


_TOWER_STREAMS = {}


def shared_tower_stream(device=None):
    """The process-wide stream a fully frozen point tower runs on when the caller vouches for its inputs
    (train.Trainer.inputs_ready + tower_own_stream): created once per GPU, like the text stream."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev not in _TOWER_STREAMS:
        with torch.cuda.device(dev):
            _TOWER_STREAMS[dev] = torch.cuda.Stream()
    return _TOWER_STREAMS[dev]


def _lead_over(ref, cand):
    """Fraction of `ref`'s busy time by which a tiny kernel queued on `cand` AFTER ref's work finishes BEFORE it:
    ~1 when the streams sit on different hardware queues, <= 0 when they share one (in-order execution)."""
    big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    small = torch.empty(1024, dtype=torch.float32, device="cuda")
    e0, e_ref, e_cand = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    with torch.cuda.stream(ref):
        e0.record(ref)
        for _ in range(6):
            big.fill_(1.0)
        e_ref.record(ref)
    with torch.cuda.stream(cand):
        small.fill_(1.0)
        e_cand.record(cand)
    torch.cuda.synchronize()
    return e_cand.elapsed_time(e_ref) / max(e0.elapsed_time(e_ref), 1e-6)


def concurrent_stream(ref=None, tries=8):
    """A new stream that really runs concurrently with `ref` (default: the current stream).  The HIP runtime deals
    streams round-robin onto GPU_MAX_HW_QUEUES hardware queues; a stream that lands on ref's queue executes in order
    with it (seen: the text stream serialised with the point tower after RCCL, or a few unrelated torch.cuda.Stream()
    objects, had been created -- 4.8 -> 6.2 / 8.0 ms per C2 step).  Each candidate is probed (~1 ms); the first with a
    clear lead is taken, rejected ones stay alive until then so that the next candidate gets the next queue."""
    ref = torch.cuda.current_stream() if ref is None else ref
    rejected, best, best_lead = [], None, -1e9
    for _ in range(tries):
        cand = torch.cuda.Stream()
        lead = _lead_over(ref, cand)
        if lead > 0.8:
            return cand
        if lead > best_lead:
            best, best_lead = cand, lead
        rejected.append(cand)
    return best
