"""Weight-gradient GEMM: ppt_gemm_tn_bf16 (operands as stored) against transposed copies + batched NT GEMM."""
import sys
import time

import torch

sys.path.insert(0, ".")
from ppt_amd import ops  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def old(a, b, S=16):
    M, N1 = a.shape
    N2 = b.shape[1]
    Mc = ((M + S - 1) // S + 63) // 64 * 64
    at = ops.transpose(a, pad_to=S * Mc)
    bt = ops.transpose(b, pad_to=S * Mc)
    part = torch.empty((S * N1, N2), dtype=torch.float32, device=a.device)
    ops.gemm(at[:, :Mc], bt[:, :Mc], out=part, batch=S, strideA=Mc, strideB=Mc, strideC=N1 * N2)
    return ops.reduce_rows(part.view(S, N1 * N2)).view(N1, N2)


for M, N1, N2 in [(32832, 384, 384), (32832, 1152, 384), (32832, 1536, 384), (32832, 384, 1536), (32768, 1536, 384),
                  (65536, 256, 128), (65536, 128, 320), (32768, 512, 256)]:
    a = torch.randn(M, N1, device="cuda").to(torch.bfloat16)
    b = torch.randn(M, N2, device="cuda").to(torch.bfloat16)
    t_new = timed(lambda: ops.gemm_tn_splitk(a, b))
    t_old = timed(lambda: old(a, b))
    fl = 2.0 * M * N1 * N2
    print(f"M={M} N1={N1} N2={N2}: tn {t_new:7.1f} us ({fl / t_new / 1e6:6.1f} TF)   transposes+nt {t_old:7.1f} us", flush=True)
