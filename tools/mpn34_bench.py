"""Time the second half of the mini-PointNet at C2's size (524 288 points = 16 384 groups), alone on the chip:
  unfused:  ppt_mini_pointnet_conv3 (y3 written) + ppt_mini_pointnet_conv4 (y3 read back)
  stats:    ppt_mini_pointnet_conv3 with store = False (the training step's statistics pass)
  fused:    ppt_mini_pointnet_conv34 (csrc/mpn34.hip)
Usage: python tools/mpn34_bench.py [groups] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppt_amd import ops  # noqa: E402

tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = torch.float16
M = 32 * tiles
g = torch.Generator().manual_seed(0)
y2 = torch.randn(M, 256, generator=g).cuda().to(T)
w3b = (torch.randn(512, 256, generator=g) * 0.06).cuda().to(T)
w4 = (torch.randn(256, 512, generator=g) * 0.04).cuda().to(T)
gterm = torch.randn(tiles, 512, generator=g).cuda()
sc, sh = (0.5 + torch.rand(512, generator=g)).cuda(), (0.1 * torch.randn(512, generator=g)).cuda()
b4 = torch.zeros(256).cuda()
w4t = ops.mpn34_retile(w4)
st = (torch.empty(tiles, 512, device="cuda"), torch.empty(tiles, 512, device="cuda"))


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def unfused():
    y3 = ops.mini_pointnet_conv3(y2, w3b, gterm, st)
    return ops.mini_pointnet_conv4(y3, sc, sh, w4, b4)


t_u = timed(unfused)
t_3 = timed(lambda: ops.mini_pointnet_conv3(y2, w3b, gterm, st))
t_s = timed(lambda: ops.mini_pointnet_conv3(y2, w3b, gterm, st, store=False))
t_f = timed(lambda: ops.mini_pointnet_conv34(y2, w3b, gterm, w4t, b4))
fl = 2.0 * M * 512 * 272 + 2.0 * M * 256 * 512
print(f"groups {tiles}: unfused conv3 + conv4 {t_u:.1f} us (conv3 alone {t_3:.1f}) | statistics pass {t_s:.1f} us | fused {t_f:.1f} us "
      f"= {fl / t_f * 1e-6:.0f} TFLOP/s executed | train: stats + fused {t_s + t_f:.1f} vs {t_u:.1f} us; eval: fused {t_f:.1f} vs {t_u:.1f} us")
