#!/usr/bin/env python3
"""split16 attention forward (csrc/attention_split.hip) against the fp32 VALU kernel and an fp64 softmax(QK^T)V: error and time.
    python tools/attn_split_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppt_amd import ops

dev = torch.device("cuda:0")


def ref(qkv, Bt, T, H, scale, causal):
    q, k, v = qkv.double().view(Bt, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * scale
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool, device=qkv.device), 1), float("-inf"))
    return (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(Bt * T, H * 64), torch.logsumexp(s, -1)


for name, Bt, T, H, causal, gain in (("vit", 32, 513, 6, False, 1.0), ("vit x3", 32, 513, 6, False, 3.0), ("text", 40, 77, 8, True, 1.0),
                                     ("ragged", 3, 200, 2, False, 1.0), ("causal 300", 5, 300, 4, True, 2.0)):
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(Bt * T, 3 * H * 64, generator=g) * gain).to(dev)
    want, want_lse = ref(qkv, Bt, T, H, 0.125, causal)
    for tag, split in (("fp32 VALU", False), ("split16", True)):
        ops.set_split16(split)
        out, lse = ops.attention_fwd(qkv, Bt, T, H, 0.125, causal)
        err = ((out.double() - want).abs().max() / want.abs().max()).item()
        lerr = (lse.double() - want_lse).abs().max().item()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            st.record()
            for _ in range(10):
                ops.attention_fwd(qkv, Bt, T, H, 0.125, causal)
            en.record(); en.synchronize()
            ts.append(st.elapsed_time(en) * 100)
        us = sorted(ts)[2]
        fl = 4.0 * Bt * H * T * T * 64 * (0.5 if causal else 1.0)
        print(f"{name:11s} Bt {Bt:3d} T {T:4d} H {H} {tag:10s} {us:8.1f} us {fl / us / 1e6:7.1f} TFLOP/s  out max-err/max {err:.2e}  lse abs err {lerr:.2e}", flush=True)
ops.set_split16(False)
