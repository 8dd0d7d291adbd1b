#!/usr/bin/env python3
"""VERDICT r3 #4, measured: a phase boundary as a dependent launch vs as a grid barrier inside one persistent launch, for the
prompt chain's geometry (G = 104 workgroups, P = 85 phases, ~128 KiB streamed per workgroup and phase), alone and beside the tower's
fused-MLP kernel looping on another stream.  See tools/chain_resident_probe.hip.   python tools/chain_resident_probe.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ppt_amd import ops

so = os.path.join(ROOT, "tools", "_build", "libchain_probe.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", os.path.join(ROOT, "tools", "chain_resident_probe.hip"), "-o", so])
L = ctypes.CDLL(so)
P_ = lambda t: ctypes.c_void_p(t.data_ptr())
raw = lambda s: ctypes.c_void_p(s.cuda_stream)

G, P = 104, 85
src = torch.randint(0, 2 ** 31 - 1, (G * 8 * 128 * 64 * 4 + 4096,), dtype=torch.int32, device="cuda")          # G * 8 windows of 128 KiB
dst = torch.empty((G * 2 * 256 * 4,), dtype=torch.int32, device="cuda")
counter = torch.zeros(1, dtype=torch.int32, device="cuda")
err = torch.zeros(1, dtype=torch.int32, device="cuda")

# the tower-like load: the fused MLP kernel of a frozen block at C2's size, looping on its own stream
g = torch.Generator().manual_seed(0)
M = 32 * 513
x = torch.randn(M, 384, generator=g).cuda()
w1 = (torch.randn(1536, 384, generator=g) * 0.05).cuda().to(torch.float16)
w2 = (torch.randn(384, 1536, generator=g) * 0.02).cuda().to(torch.float16)
w1t, w2t = ops.vit_mlp_retile(w1, w2)
b1, b2 = torch.zeros(1536).cuda(), torch.zeros(384).cuda()
ln = (torch.ones(384).cuda(), torch.zeros(384).cuda())
load_stream, chain_stream = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps, beside):
    torch.cuda.synchronize()
    if beside:
        with torch.cuda.stream(load_stream), ops.persistent_occupancy(70):
            for _ in range(beside):
                ops.vit_mlp(x, w1t, b1, w2t, b2, ln)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(chain_stream):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for kb in (4, 64, 128):
    def launches():
        assert L.probe_launches(P_(src), P_(dst), kb, P, G, raw(chain_stream)) == 0

    def resident():
        counter.zero_()
        assert L.probe_resident(P_(src), P_(dst), kb, P, G, P_(counter), P_(err), raw(chain_stream)) == 0
    with torch.cuda.stream(chain_stream):
        launches(); resident()
    rows = []
    for beside in (0, 40):
        tl = timed(launches, 5, beside)
        with torch.cuda.stream(chain_stream):
            tr = timed(resident, 5, beside)
        rows.append((beside, tl, tr))
    assert err.item() == 0, "a barrier timed out (a workgroup was not resident)"
    for beside, tl, tr in rows:
        print(f"{kb:4d} KiB / workgroup / phase, {'beside the fused-MLP loop' if beside else 'alone on the chip         '}: "
              f"{P} dependent launches {tl:8.1f} us = {tl / P:5.2f} us per phase | one resident launch with {P} grid barriers {tr:8.1f} us = {tr / P:5.2f} us per phase",
              flush=True)
