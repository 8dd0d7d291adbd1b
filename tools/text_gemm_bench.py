"""Dev tool: the GEMM shapes of the prompt chain (M = 817 rows of the prefix-shared text tower) one by one, graph-timed.
    python tools/text_gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppt_amd import ops
from rowgemm_bench import timeit

g = torch.Generator().manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 817
for nm, K, N, resid in (("out_proj fwd/bwd", 512, 512, True), ("c_proj fwd", 2048, 512, True), ("d c_fc", 2048, 512, False),
                        ("d c_proj", 512, 2048, False), ("d in_proj", 1536, 512, False), ("in_proj (plain)", 512, 1536, False),
                        ("c_fc (plain)", 512, 2048, False)):
    a = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).cuda().to(torch.bfloat16)
    b = torch.randn(N, generator=g).cuda()
    x = torch.randn(M, N, generator=g).cuda()
    if resid:
        fn = lambda: ops.gemm(a, w, out=x, bias=b, residual=x)
    else:
        fn = lambda: ops.gemm(a, w, out_dtype=torch.float32)
    t = timeit(fn)
    fl = 2.0 * M * N * K
    print(f"{nm:18s} M={M} K={K:4d} N={N:4d}: {t:6.1f} us ({fl / t / 1e6:5.0f} TF), min traffic {(M * K + N * K) * 2 / 1e6:.1f} MB", flush=True)
