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

    def __init__(self, fn, inputs, pool=None):
        self.static_in = [t.clone() for t in inputs]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=pool):
            self.outputs, self.keepalive = fn(*self.static_in)

    def pool(self):
        return self.graph.pool()

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
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
