"""CPU, world_size 2 over gloo: the data-parallel machinery of ppt_amd.train (flat gradient
all-reduce + BatchNorm buffer broadcast) reproduces DistributedDataParallel semantics:
  averaged gradient over ranks == gradient of the mean loss over the concatenated batch
(the reference wraps the model in DDP, main_cls.py:47-49).  The per-rank "model" here is the oracle's
train-step math on the CPU (tests may use the oracle); the HIP model goes through exactly the same
Trainer / FlatGradSync / BufferBroadcast code on the GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class TinyModel(torch.nn.Module):
    """A stand-in with the same trainable/frozen/buffer structure as ULIP_PointBERT (a trainable
    prompt, a frozen weight, a BatchNorm buffer) -- small enough for a 2-process CPU test."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.prompt = torch.nn.Parameter(torch.randn(4, 8, generator=g))
        self.frozen = torch.nn.Parameter(torch.randn(8, 5, generator=g), requires_grad=False)
        self.logit_scale = torch.nn.Parameter(torch.tensor(2.0), requires_grad=False)
        self.bn = torch.nn.BatchNorm1d(8)
        self.bn.weight.requires_grad = False      # frozen, yet in train mode
        self.bn.bias.requires_grad = False

    def forward(self, x):
        h = self.bn(x @ self.prompt)
        return h @ self.frozen


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ppt_amd.train import Trainer
    torch.manual_seed(0)
    m = TinyModel()
    if rank == 1:                       # rank 1 starts with different BN buffers: the broadcast must fix that
        m.bn.running_mean.fill_(3.0)
    tr = Trainer(m, lr=1e-2, distributed=True)
    g = torch.Generator().manual_seed(123)
    x = torch.randn(8, 4, generator=g)
    y = torch.randint(0, 5, (8,), generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    p0 = m.prompt.detach().clone()
    rm_before = None
    loss, _ = tr.step(xs, ys)
    rm_mid = m.bn.running_mean.numpy().copy()
    tr.finish()                         # the buffer broadcast is deferred to here (train.Trainer.broadcast_buffers_every_step)
    ret[rank] = dict(grad=tr.sync.flat.clone().numpy(), prompt=m.prompt.detach().numpy().copy(), p0=p0.numpy(),
                     rm=m.bn.running_mean.numpy().copy(), rm_mid=rm_mid, loss=float(loss.detach()))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_grad_allreduce_matches_ddp_semantics():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    r0, r1 = ret[0], ret[1]
    # every rank ends with the same averaged gradient and the same parameters
    assert np.array_equal(r0["grad"], r1["grad"])
    assert np.array_equal(r0["prompt"], r1["prompt"])
    # reference: per-rank BN statistics (plain BN, not SyncBN), losses averaged over ranks
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(123)
    x = torch.randn(8, 4, generator=g)
    y = torch.randint(0, 5, (8,), generator=g)
    grads = []
    for rank in range(world):
        m = TinyModel()
        out = m(x[rank * 4:(rank + 1) * 4])
        loss = torch.nn.functional.cross_entropy(out, y[rank * 4:(rank + 1) * 4], label_smoothing=0.2)
        (gr,) = torch.autograd.grad(loss, [m.prompt])
        grads.append(gr.flatten().numpy())
    want = (grads[0] + grads[1]) / 2
    assert np.allclose(r0["grad"], want, atol=1e-6)
    # rank 1's divergent running_mean is overwritten by rank 0's (DDP broadcast_buffers) -- at finish(): a train-mode
    # forward never reads it, and rank 0's own statistics never depend on another rank's
    assert abs(r1["rm_mid"]).max() > 1.0
    assert np.array_equal(r1["rm"], r0["rm"]) and abs(r1["rm"]).max() < 1.0


def test_flat_grad_views_and_zero():
    from ppt_amd.train import FlatGradSync
    a = torch.nn.Parameter(torch.ones(3, 2))
    b = torch.nn.Parameter(torch.ones(5))
    c = torch.nn.Parameter(torch.ones(2), requires_grad=False)
    s = FlatGradSync([a, b, c])
    assert s.flat.numel() == 11 and c.grad is None
    (a.sum() * 2 + b.sum() * 3).backward()
    assert torch.equal(s.flat, torch.cat([torch.full((6,), 2.0), torch.full((5,), 3.0)]))
    a.grad = None
    s.zero()
    assert a.grad is not None and a.grad.data_ptr() == s.flat.data_ptr() and s.flat.abs().sum() == 0


def test_cosine_scheduler_matches_oracle():
    from oracle import oracle as O
    from ppt_amd.train import cosine_scheduler
    a = cosine_scheduler(3e-3, 1e-5, 7, 13, warmup_epochs=1, start_warmup_value=1e-6)
    b = O.cosine_scheduler(3e-3, 1e-5, 7, 13, warmup_epochs=1, start_warmup_value=1e-6)
    assert np.array_equal(a, b)
