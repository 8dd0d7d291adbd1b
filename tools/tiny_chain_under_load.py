"""Dev tool: is the stretch of the prompt chain under load a per-kernel-BOUNDARY cost?  A captured chain of 200 dependent
one-workgroup kernels (no real work) is timed alone and beside write-heavy / read-only / compute-only background loads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from ppt_amd import ops

torch.cuda.set_device(0)
prio = int(sys.argv[1]) if len(sys.argv) > 1 else -1
side = torch.cuda.Stream(priority=prio)
bg = torch.cuda.Stream()
small = torch.zeros(256, device="cuda")
big = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
big.fill_(1.0)
g = torch.Generator().manual_seed(0)
B = 32
M = B * 513
x = torch.randn(M, 384, generator=g).cuda()
gam, bet = torch.ones(384).cuda(), torch.zeros(384).cuda()
w1 = (torch.randn(1536, 384, generator=g) * 0.05).cuda().to(torch.bfloat16)
w2 = (torch.randn(384, 1536, generator=g) * 0.02).cuda().to(torch.bfloat16)
b1, b2, dp = torch.randn(1536, generator=g).cuda(), torch.randn(384, generator=g).cuda(), torch.ones(B).cuda()
w1t, w2t = ops.vit_mlp_retile(w1, w2)
xo = x.clone()
qkv = torch.randn(M, 1152, generator=g).cuda().to(torch.bfloat16)
acc = torch.zeros((), device="cuda")

LOADS = {
    "none": None,
    "hbm_fill (writes)": lambda: big.fill_(1.0),
    "hbm_sum (reads)": lambda: torch.sum(big, dim=0, keepdim=True, out=acc.view(1)),
    "mlp_fused": lambda: ops.vit_mlp(xo, w1t, b1, w2t, b2, (gam, bet), row_scale=dp, row_scale_rows=513),
    "attention": lambda: ops.attention_fwd(qkv, B, 513, 6, 0.125, False, want_lse=False),
}

with torch.cuda.stream(side):
    for _ in range(3):
        small.add_(1.0)
    torch.cuda.synchronize()
    chain = torch.cuda.CUDAGraph()
    with torch.cuda.graph(chain, stream=side):
        for _ in range(200):
            small.add_(1.0)
torch.cuda.synchronize()

for name, fn in LOADS.items():
    graph, t_graph = None, 1.0
    if fn is not None:
        with torch.cuda.stream(bg):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=bg):
                for _ in range(20):
                    fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            graph.replay()
            e.record()
            torch.cuda.synchronize()
            t_graph = s.elapsed_time(e)
    reps = 0 if graph is None else int(60 / t_graph) + 2
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(bg):
        for _ in range(reps):
            graph.replay()
    with torch.cuda.stream(side):
        e0.record()
        for _ in range(10):
            chain.replay()
        e1.record()
    torch.cuda.synchronize()
    print(f"{name:20s}: {e0.elapsed_time(e1) / 2000 * 1e3:6.2f} us per tiny kernel of the chain (priority {prio})", flush=True)
