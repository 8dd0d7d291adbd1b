"""Host-to-device feeding of the hot path with the copies off the compute stream.

main_cls.py:171-194 takes a batch from the DataLoader and copies it inside the loop (`pc = pc.cuda(args.gpu, non_blocking=True)`,
:188-191): the copy is queued on the compute stream, so the step's first kernels wait for it, and nothing about the batch is
known before `model(pc)` is called.  `DevicePrefetcher` wraps the loader: the copies of batch i + 1 run on a copy stream while
batch i computes, and every device tensor it yields carries the event that marks its copy complete.  The stages of the model that
depend on the inputs alone -- FPS + kNN + the frozen tokenizer (C2 / C3), the whole frozen backbone (C5), both levels' FPS + ball
queries (C4) -- wait for THAT event on their own stream instead of for the compute stream (ppt_amd/graphs.py: ready_event), i.e. they
run under the previous step's transformer blocks without the caller vouching for anything (`Trainer.inputs_ready` stays available
for tensors that simply are resident).

    for pc, target, target_name, rgb in DevicePrefetcher(train_loader):     # the loop body of main_cls.py:179-214 unchanged:
        pc = pc.cuda(args.gpu, non_blocking=True)                           #   a no-op on a device tensor (same object, same event)
        ...
"""
import collections

import torch


def _map(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(o, fn) for o in obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    return obj


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors(o)
    elif isinstance(obj, dict):
        for o in obj.values():
            yield from _tensors(o)


class DevicePrefetcher:
    """Iterate `loader` (any iterable of tensors / nested lists, tuples, dicts of tensors; non-tensor leaves pass through) with
    the host-to-device copies `depth` batches ahead on a copy stream.  Pinned host tensors (DataLoader(pin_memory=True)) make
    the copies asynchronous; pageable ones are copied synchronously by the runtime -- still correct, no overlap."""

    def __init__(self, loader, device=None, depth=2, stream=None):
        self.loader = loader
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.depth = max(1, int(depth))
        self.stream = stream

    def __len__(self):
        return len(self.loader)

    def _stream(self):
        """The stream the copies are queued on: the process-wide GROUPING stream (graphs.shared_group_stream) unless the caller gives
        one.  Not a stream of its own: which hardware queue a new HIP stream lands on depends on how many streams the process has
        created, and a copy stream that shares the text stream's queue executes in order with the prompt chain -- measured: every
        step's tower then waits for its batch behind the previous step's text backward (C2 3.11 -> 4.74 ms, C5 6.2 -> 11.3 ms per
        step).  The grouping stream is where the first consumer of the batch runs anyway (the ahead stage), its queue placement is
        the one the two-stream schedule was tuned with, and the copy of batch i + 1 is queued on it before step i is even called."""
        if self.stream is not None:
            return self.stream
        from .. import graphs
        return graphs.shared_group_stream(self.device)

    def __iter__(self):
        it = iter(self.loader)
        copy = self._stream()
        queue = collections.deque()

        def put():
            try:
                batch = next(it)
            except StopIteration:
                return False
            with torch.cuda.stream(copy):
                dev = _map(batch, lambda t: t.to(self.device, non_blocking=True))
                ev = torch.cuda.Event()
                ev.record(copy)
            for t in _tensors(dev):
                t._ppt_ready = ev              # (graphs.ready_event: the input-only stages of the model wait for this, not for the caller)
            queue.append((dev, ev))
            return True
        for _ in range(self.depth):
            if not put():
                break
        while queue:
            dev, ev = queue.popleft()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)
            for t in _tensors(dev):
                t.record_stream(cur)
            put()
            yield dev
