#!/usr/bin/env python3
"""In-kernel phase timing of csrc/mpn34.hip (diagnostic build, see M34_STAMP there): builds the stamped variant on the box.
    python tools/mpn34_stamp.py [extra -D flags ...]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ppt_amd import ops

so = os.path.join(ROOT, "tools", "_build", "libmpn34_stamp.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fno-slp-vectorize", "-DPPT_M34_STAMP",
                       *sys.argv[1:], "-shared", os.path.join(ROOT, "ppt_amd", "csrc", "mpn34.hip"), os.path.join(ROOT, "ppt_amd", "csrc", "api.hip"), "-o", so])
L = ctypes.CDLL(so)
tiles = 16384
T = torch.float16
M = 32 * tiles
g = torch.Generator().manual_seed(0)
y2 = torch.randn(M, 256, generator=g).cuda().to(T)
w3b = (torch.randn(512, 256, generator=g) * 0.06).cuda().to(T)
w4 = (torch.randn(256, 512, generator=g) * 0.04).cuda().to(T)
gs = torch.randn(tiles, 512, generator=g).cuda()
w4t = ops.mpn34_retile(w4)
tok = torch.empty(tiles, 256, dtype=T, device="cuda")
stamps = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())
fn = L.ppt_mini_pointnet_conv34_half
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
for _ in range(3):
    stamps.zero_()
    assert fn(P(y2), M, P(w3b), P(gs), P(w4t), P(stamps), P(tok), 2, None) == 0
    torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(256, 8, 8).astype(np.int64)
names = ["wait y2 (a) + barrier", "request (b) + phase 1 (a) + y3 rows 0-63", "wait y2 (b) + barrier", "phase 1 (b) + ring prime + barrier + y3 rows 64-127",
         "barrier", "phase 2 MFMA (W4 stream), 128 rows", "max + store"]
med = lambda a: int(np.median(a))
tot = s[:, :, 7] - s[:, :, 0]
print(f"second chunk of every workgroup, median over 256 x 8 waves (s_memtime ticks); total {med(tot)}")
for i, n in enumerate(names):
    d = s[:, :, i + 1] - s[:, :, i]
    print(f"  {n:52s} {med(d):7d}   (p10 {int(np.percentile(d, 10))}, p90 {int(np.percentile(d, 90))})")
