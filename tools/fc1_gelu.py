import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppt_amd import ops
M, N, K = 16416, 1536, 384
A = torch.randn(M, K, device="cuda").bfloat16(); W = torch.randn(N, K, device="cuda").bfloat16()
b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
def bench(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
print("fc1 plain        %.1f us" % bench(lambda: ops.gemm(A, W, out=out)))
print("fc1 +bias        %.1f us" % bench(lambda: ops.gemm(A, W, out=out, bias=b)))
print("fc1 +bias+relu   %.1f us" % bench(lambda: ops.gemm(A, W, out=out, bias=b, act=ops.ACT_RELU)))
print("fc1 +bias+gelu   %.1f us" % bench(lambda: ops.gemm(A, W, out=out, bias=b, act=ops.ACT_GELU)))
print("fc1 +bias+qgelu  %.1f us" % bench(lambda: ops.gemm(A, W, out=out, bias=b, act=ops.ACT_QUICKGELU)))
