#!/usr/bin/env python3
"""In-kernel phase timing of lnlin_kernel (csrc/lnlin.hip, LNLIN_STAMP): s_memtime ticks per phase, per workgroup round.
    bash tools/build_variant.sh lnlinstamp -DPPT_LNLIN_STAMP && python3 tools/lnlin_stamp.py [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["PPT_HIP_LIB"] = os.path.join(ROOT, "tools", "_build", "libppt_lnlinstamp.so")
sys.path.insert(0, ROOT)
import ctypes
import numpy as np
import torch
from ppt_amd import ops, _lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16416
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, 384, generator=g) * 2).cuda()
gam, bet = torch.ones(384).cuda(), torch.zeros(384).cuda()
w = (torch.randn(1152, 384, generator=g) * 384 ** -0.5).cuda().half()
wt = ops.lnlin_retile(w)
chunks = (M + 63) // 64
grid = ((chunks + 7) // 8) * 8 * 3
st = torch.zeros(grid * 8 * 8, dtype=torch.int64, device="cuda")
C = torch.empty(M, 1152, dtype=torch.float16, device="cuda")
P = _lib.LnLinParams(x=x.data_ptr(), W=wt.data_ptr(), C=C.data_ptr(), ln_w=gam.data_ptr(), ln_b=bet.data_ptr(), ln_eps=1e-5, bias=st.data_ptr(),
                     M=M, N=1152, K=384, dtype=ops.dtype_code(w), slices=3)
for _ in range(3):
    st.zero_()
    torch.cuda.synchronize()
    assert _lib.lib().ppt_lnlin(ctypes.byref(P), None) == 0
    torch.cuda.synchronize()
s = st.cpu().numpy().reshape(grid, 8, 8)
live = s[:, 0, 0] > 0
s = s[live]
t0 = s[:, :, 0].min()
print(f"rows {M}: {live.sum()} workgroups; kernel span {s[:, :, 7].max() - t0} ticks")
entry = s[:, 0, 0] - t0
first = entry < np.percentile(entry, 60)             # the first round (entered at once) / the later ones
names = ["entry", "loads issued", "LayerNorm done", "barrier", "k loop done", "barrier", "C tile in LDS + barrier", "stores issued"]
for label, sel in (("first round", first), ("later rounds", ~first)):
    if sel.sum() == 0:
        continue
    print(f"-- {label}: {sel.sum()} workgroups, entry median {np.median(entry[sel]):.0f}")
    for i in range(1, 8):
        d = (s[sel][:, :, i] - s[sel][:, :, i - 1])
        print(f"   {names[i]:28s} +{np.median(d):8.0f} (p90 {np.percentile(d, 90):8.0f})")
    print(f"   workgroup lifetime median {np.median(s[sel][:, :, 7].max(1) - s[sel][:, :, 0].min(1)):.0f}")
