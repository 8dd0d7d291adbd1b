"""Dev tool: GPU-side cost of a dependent chain of tiny kernels, stream launches vs hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppt_amd import ops
x = torch.zeros(4096, device="cuda")
blk_a = torch.zeros(65536, 512, device="cuda", dtype=torch.bfloat16)
blk_w = torch.zeros(512, 512, device="cuda", dtype=torch.bfloat16)
blk_o = torch.empty(65536, 512, device="cuda", dtype=torch.bfloat16)
N = 300
def chain():
    for _ in range(N):
        ops.convert(x, torch.bfloat16)
def timed(fn, busy=True):
    torch.cuda.synchronize()
    if busy:
        for _ in range(30): ops.gemm(blk_a, blk_w, out=blk_o)      # let the host run ahead
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / N
chain(); torch.cuda.synchronize()
print("stream launches, host ahead : %.2f us per kernel" % timed(chain))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    chain()
g.replay(); torch.cuda.synchronize()
print("hipGraph replay             : %.2f us per kernel" % timed(g.replay))
