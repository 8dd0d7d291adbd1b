"""Dev tool: FPS kernel time at the shapes of the four configs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppt_amd import ops, weights as W
for B, N, M in ((32, 1024, 512), (64, 2048, 512), (32, 8192, 512), (32, 512, 128), (16, 2048, 512)):
    pc, start = W.synth_clouds(B, N, seed=3)
    pc = torch.from_numpy(pc).cuda(); start = torch.from_numpy(start).cuda()
    for _ in range(3): ops.fps(pc, M, start)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): ops.fps(pc, M, start)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(f"fps B={B} N={N} M={M}: {us:8.1f} us  ({us / M * 1e3:.0f} ns per pick)")
