"""Dev tool: attention forward time at the point-tower (B=32, T=513, H=6) and text-tower shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppt_amd import ops
for B, T, H, causal in ((32, 513, 6, False), (64, 513, 6, False), (40, 37, 8, True)):
    qkv = torch.randn(B * T, 3 * H * 64, device="cuda").half()
    f = lambda: ops.attention_fwd(qkv, B, T, H, 0.125, causal, want_lse=False)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 50
    print(f"attention fwd B={B} T={T} H={H} causal={causal}: {us:7.1f} us  {4 * T * T * 64 * H * B / us / 1e6:6.1f} TF")
