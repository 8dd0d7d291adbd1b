#!/usr/bin/env python3
"""Per-shape timing of every GEMM launch of one C2 training step (GPU).  Development tool:
prints TFLOP/s per (mode, M, N, K) so the slow shapes of ppt_gemm are visible."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppt_amd import ops

def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

def main():
    dev = "cuda"
    B = 32
    Mp, Mt, Mm = B * 513, 40 * 77, B * 512 * 32
    shapes = [("plain qkv", Mp, 1152, 384), ("plain proj+res", Mp, 384, 384), ("plain fc1 gelu", Mp, 1536, 384),
              ("plain fc2+res", Mp, 384, 1536), ("txt in_proj", Mt, 1536, 512), ("txt out_proj", Mt, 512, 512),
              ("txt c_fc", Mt, 2048, 512), ("txt c_proj", Mt, 512, 2048), ("conv3 local", Mm, 512, 256),
              ("gterm", Mm // 32, 512, 256), ("reduce_dim", B * 512, 384, 256)]
    for name, M, N, K in shapes:
        A = torch.randn(M, K, device=dev).bfloat16(); W = torch.randn(N, K, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ms = bench(lambda: ops.gemm(A, W, out=out))
        print(f"{name:16s} M={M:7d} N={N:5d} K={K:5d}  {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF")
        if "res" in name:
            res = torch.randn(M, N, device=dev); o32 = torch.empty(M, N, device=dev)
            bias = torch.randn(N, device=dev)
            ms = bench(lambda: ops.gemm(A, W, out=o32, residual=o32, bias=bias))
            print(f"{'  +bias+res f32':16s} {'':30s}  {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF")
    # mini-PointNet specials
    pts = torch.randn(Mm, 3, device=dev) * 0.1
    w1 = torch.randn(128, 3, device=dev); b1 = torch.randn(128, device=dev)
    sc = torch.rand(128, device=dev) + 0.5; sh = torch.randn(128, device=dev)
    W2 = torch.randn(256, 128, device=dev).bfloat16(); y2 = torch.empty(Mm, 256, device=dev, dtype=torch.bfloat16)
    gmax = torch.empty(Mm // 32, 256, device=dev, dtype=torch.bfloat16)
    ms = bench(lambda: ops.gemm(None, W2, out=y2, a_mode=ops.A_CONV1, pts=pts, w1=w1, b1=b1, a_scale=sc, a_shift=sh, pool_max=gmax))
    print(f"{'conv2 (CONV1 A)':16s} M={Mm:7d} N={256:5d} K={128:5d}  {ms*1e3:9.1f} us  {2*Mm*256*128/ms/1e9:8.1f} TF")
    y3 = torch.randn(Mm, 512, device=dev).bfloat16(); W4 = torch.randn(256, 512, device=dev).bfloat16()
    sc2 = torch.rand(512, device=dev) + 0.5; sh2 = torch.randn(512, device=dev)
    tok = torch.empty(Mm // 32, 256, device=dev, dtype=torch.bfloat16)
    ms = bench(lambda: ops.gemm(y3, W4, a_mode=ops.A_AFFINE_RELU, a_scale=sc2, a_shift=sh2, pool_max=tok, want_out=False))
    print(f"{'conv4 (AFFINE A)':16s} M={Mm:7d} N={256:5d} K={512:5d}  {ms*1e3:9.1f} us  {2*Mm*256*512/ms/1e9:8.1f} TF")
    W3 = torch.randn(512, 256, device=dev).bfloat16(); gt = torch.randn(Mm // 32, 512, device=dev)
    cs = torch.empty(Mm // 32, 512, device=dev); cq = torch.empty_like(cs); o3 = torch.empty(Mm, 512, device=dev, dtype=torch.bfloat16)
    ms = bench(lambda: ops.gemm(y2, W3, out=o3, group_add=gt, group_rows=32, col_stats=(cs, cq)))
    print(f"{'conv3 +stats':16s} M={Mm:7d} N={512:5d} K={256:5d}  {ms*1e3:9.1f} us  {2*Mm*512*256/ms/1e9:8.1f} TF")

if __name__ == "__main__":
    main()
