"""Dev tool: run each GEMM flavour several times on identical inputs and compare the results bitwise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppt_amd import ops

torch.manual_seed(0)
dev = "cuda"

def check(name, fn, n=6):
    ref = [t.clone() for t in fn()]
    bad = 0
    for _ in range(n):
        out = fn()
        torch.cuda.synchronize()
        for a, b in zip(ref, out):
            if not torch.equal(a, b):
                bad += 1
                d = (a.float() - b.float()).abs()
                print(f"   {name}: mismatch max {d.max().item():.3e} at {int((d > 0).sum())} elements of {d.numel()}; first idx {torch.nonzero(d > 0)[0].tolist()}")
                break
    print(f"{name:34s} {'NONDETERMINISTIC' if bad else 'ok'}", flush=True)

for M in (2052, 16416):
    A = torch.randn(M, 384, device=dev).bfloat16()
    Wq = torch.randn(1152, 384, device=dev).bfloat16(); W1 = torch.randn(1536, 384, device=dev).bfloat16()
    b1 = torch.randn(1536, device=dev)
    check(f"qkv M={M}", lambda: (ops.gemm(A, Wq),))
    check(f"fc1 gelu M={M}", lambda: (ops.gemm(A, W1, bias=b1, act=ops.ACT_GELU),))
    res = torch.randn(M, 384, device=dev); Wp = torch.randn(384, 384, device=dev).bfloat16()
    check(f"proj+res M={M}", lambda: (ops.gemm(A, Wp, out_dtype=torch.float32, residual=res, bias=b1[:384]),))
for Mm in (65536, 524288):
    pts = torch.randn(Mm, 3, device=dev) * 0.1
    w1 = torch.randn(128, 3, device=dev); bb = torch.randn(128, device=dev)
    sc = torch.rand(128, device=dev) + 0.5; sh = torch.randn(128, device=dev)
    W2 = torch.randn(256, 128, device=dev).bfloat16()
    def conv2():
        y2 = torch.empty(Mm, 256, device=dev, dtype=torch.bfloat16); gmax = torch.empty(Mm // 32, 256, device=dev, dtype=torch.bfloat16)
        ops.gemm(None, W2, out=y2, a_mode=ops.A_CONV1, pts=pts, w1=w1, b1=bb, a_scale=sc, a_shift=sh, pool_max=gmax, pool_rows=32)
        return y2, gmax
    check(f"conv2 M={Mm}", conv2)
    y2 = torch.randn(Mm, 256, device=dev).bfloat16()
    W3 = torch.randn(512, 256, device=dev).bfloat16(); gt = torch.randn(Mm // 32, 512, device=dev)
    def conv3():
        cs = torch.empty(Mm // 32, 512, device=dev); cq = torch.empty_like(cs); o3 = torch.empty(Mm, 512, device=dev, dtype=torch.bfloat16)
        ops.gemm(y2, W3, out=o3, group_add=gt, group_rows=32, col_stats=(cs, cq))
        return o3, cs, cq
    check(f"conv3 M={Mm}", conv3)
    y3 = torch.randn(Mm, 512, device=dev).bfloat16(); W4 = torch.randn(256, 512, device=dev).bfloat16()
    sc2 = torch.rand(512, device=dev) + 0.5; sh2 = torch.randn(512, device=dev)
    def conv4():
        tok = torch.empty(Mm // 32, 256, device=dev, dtype=torch.bfloat16)
        ops.gemm(y3, W4, a_mode=ops.A_AFFINE_RELU, a_scale=sc2, a_shift=sh2, pool_max=tok, pool_rows=32, want_out=False)
        return (tok,)
    check(f"conv4 M={Mm}", conv4)
