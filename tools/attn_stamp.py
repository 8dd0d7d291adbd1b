#!/usr/bin/env python3
"""In-kernel phase timing of attn_fwd_resident (csrc/attention_mfma.hip, ATTN_STAMP): s_memtime ticks (shader cycles) per phase.

    bash tools/build_variant.sh attnstamp -DPPT_ATTN_STAMP && python tools/attn_stamp.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

L = ctypes.CDLL(os.path.join(ROOT, "tools", "_build", "libppt_attnstamp.so"))
f = L.ppt_attention_fwd
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
              ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
B, T, H = 32, 513, 6
qkv = torch.randn(B * T, 3 * H * 64, device="cuda").half()
out = torch.empty(B * T, H * 64, device="cuda", dtype=torch.float16)
st = torch.zeros(B * H * 8 * 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    st.zero_()
    torch.cuda.synchronize()
    assert f(qkv.data_ptr(), out.data_ptr(), st.data_ptr(), B, T, H, 64, 0.125, 0, 2, None) == 0
    torch.cuda.synchronize()
s = st.cpu().numpy().reshape(B * H, 8, 16).astype(np.int64)
t0 = s[:, :, 0].min()
med = lambda v: float(np.median(v))
names = ["entry", "fill issued", "-", "tile0", "tile1", "tile2", "tile3", "tile4", "tile5", "tile6", "tile7", "loop done", "rows out", "cls partial", "end"]
prev = 0
for i, n in enumerate(names):
    if n == "-":
        continue
    print(f"{n:12s} median {med(s[:, :, i] - s[:, :, 0]) :8.0f} cycles after entry (+{(med(s[:, :, i] - s[:, :, 0]) - prev):6.0f})   max {(s[:, :, i] - s[:, :, 0]).max()} cycles")
    prev = med(s[:, :, i] - s[:, :, 0])
print("entry spread (last workgroup entry - first):", (s[:, :, 0].max() - t0), "cycles; last exit:", (s[:, :, 14].max() - t0), "cycles")
