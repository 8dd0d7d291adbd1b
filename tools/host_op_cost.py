"""Dev tool: host cost (enqueue time, no synchronisation inside the loop) of single ops.* calls and of the torch ops around them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppt_amd import ops
a = torch.randn(256, 384, device="cuda").to(torch.bfloat16); w = torch.randn(384, 384, device="cuda").to(torch.bfloat16)
out = torch.empty(256, 384, device="cuda"); b = torch.randn(384, device="cuda"); x = torch.randn(256, 384, device="cuda")
def t(name, fn, n=3000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{name:34s} {1e6 * (t1 - t0) / n:7.2f} us/call", flush=True)
t("torch.empty", lambda: torch.empty(256, 384, device="cuda"))
t("torch add (out=)", lambda: torch.add(x, x, out=out))
t("ops.gemm (out given, bias)", lambda: ops.gemm(a, w, out=out, bias=b))
t("ops.gemm (allocating)", lambda: ops.gemm(a, w, out_dtype=torch.float32))
t("ops.convert", lambda: ops.convert(x, torch.bfloat16))
t("ops.layernorm_fwd", lambda: ops.layernorm_fwd(x, b, b, torch.bfloat16))
t("x.contiguous().float() (no-op)", lambda: x.contiguous().float())
t("torch.cuda.current_stream()", lambda: torch.cuda.current_stream().cuda_stream)
