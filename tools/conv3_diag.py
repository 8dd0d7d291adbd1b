import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppt_amd import ops
torch.manual_seed(0)
dev = "cuda"
Mm = 65536
y2 = torch.randn(Mm, 256, device=dev).bfloat16()
W3 = torch.randn(512, 256, device=dev).bfloat16(); gt = torch.randn(Mm // 32, 512, device=dev)
def conv3():
    cs = torch.full((Mm // 32, 512), float("nan"), device=dev); cq = torch.full_like(cs, float("nan")); o3 = torch.empty(Mm, 512, device=dev, dtype=torch.bfloat16)
    ops.gemm(y2, W3, out=o3, group_add=gt, group_rows=32, col_stats=(cs, cq))
    torch.cuda.synchronize()
    return o3, cs, cq
full = (y2.float() @ W3.float().t()) + gt.repeat_interleave(32, 0)
ch = full.view(-1, 32, 512)
ts = ch.sum(1); tq = ((ch - ch.mean(1, keepdim=True)) ** 2).sum(1)
for it in range(4):
    o3, cs, cq = conv3()
    for nm, got, true in (("sum", cs, ts), ("m2", cq, tq)):
        err = (got - true).abs() / (true.abs() + 1.0)
        bad = torch.nonzero(~(err < 1e-2))
        print(it, nm, "bad", bad.shape[0], "nan", int(torch.isnan(got).sum()))
        for b in bad[:6].tolist():
            print("    chunk", b[0], "col", b[1], "got", got[b[0], b[1]].item(), "true", true[b[0], b[1]].item())
        if bad.shape[0]:
            cols = bad[:, 1]
            print("    cols%64 hist:", torch.bincount(cols % 64, minlength=64).tolist())
            print("    chunk%4 hist:", torch.bincount(bad[:, 0] % 4, minlength=4).tolist())

# which wrong value is it?  half-sums over the rows of lane half h=0 / h=1
rows = torch.arange(32)
h0 = ((rows % 8) < 4)
a = ch[:, h0].sum(1); b = ch[:, ~h0].sum(1)
o3, cs, cq = conv3()
err = (cs - ts).abs() / (ts.abs() + 1.0)
bad = torch.nonzero(~(err < 1e-2))
for name, cand in (("2a", 2 * a), ("2b", 2 * b), ("a", a), ("b", b)):
    c = cand[bad[:, 0], bad[:, 1]]; g = cs[bad[:, 0], bad[:, 1]]
    print(name, "matches", int(((c - g).abs() < 1e-2 * (g.abs() + 1)).sum()), "of", bad.shape[0])
# partial sums in the order of the code: s0 = v0+v2+..., s1 = v1+v3+...
vr = [ (r & 3) + 8 * (r >> 2) for r in range(16)]
for name, sel in (("s0 (even r) both halves", [vr[r] + 4 * hh for r in range(0, 16, 2) for hh in (0, 1)]),
                  ("s1 (odd r) both halves", [vr[r] + 4 * hh for r in range(1, 16, 2) for hh in (0, 1)])):
    c = ch[:, sel].sum(1)[bad[:, 0], bad[:, 1]]; g = cs[bad[:, 0], bad[:, 1]]
    print(name, "matches", int(((c - g).abs() < 1e-2 * (g.abs() + 1)).sum()), "of", bad.shape[0])
    c2 = 2 * c
    print("  2x:", int(((c2 - g).abs() < 1e-2 * (g.abs() + 1)).sum()))
print("sample got/true/a/b:", [(round(cs[i, j].item(), 2), round(ts[i, j].item(), 2), round(a[i, j].item(), 2), round(b[i, j].item(), 2)) for i, j in bad[:5].tolist()])
