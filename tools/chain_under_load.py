"""Dev tool: how much does ONE kind of point-tower kernel, looping on another stream, stretch the prompt chain (text forward
-> head -> text backward -> AdamW with a cached point feature; ~2.1 ms alone)?   python tools/chain_under_load.py [loads ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from ppt_amd import graphs, ops, weights as W
from ppt_amd.train import Trainer

torch.cuda.set_device(0)
cfg = bench.CONFIGS["C2"]
graphs.shared_text_stream(priority=-1)
model = bench.build_model(cfg["dataset"], cfg["head_type"], torch.bfloat16, "ULIP_PointBERT", "cls")
model.train()
tr = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=False)
B, N = cfg["batch"], cfg["npoints"]
pc = torch.from_numpy(W.synth_clouds(B, N, seed=1)[0]).cuda()
label = torch.randint(0, 40, (B,), device="cuda")
with torch.no_grad():
    feat = model.point_encoder(pc).detach()
model.point_encoder.forward = lambda x: feat

g = torch.Generator().manual_seed(0)
M = B * 513
x = torch.randn(M, 384, generator=g).cuda()
gam, bet = torch.ones(384).cuda(), torch.zeros(384).cuda()
w1 = (torch.randn(1536, 384, generator=g) * 0.05).cuda().to(torch.bfloat16)
w2 = (torch.randn(384, 1536, generator=g) * 0.02).cuda().to(torch.bfloat16)
wq = (torch.randn(1152, 384, generator=g) * 0.05).cuda().to(torch.bfloat16)
b1, b2, bq, dp = torch.randn(1536, generator=g).cuda(), torch.randn(384, generator=g).cuda(), torch.randn(1152, generator=g).cuda(), torch.ones(B).cuda()
w1t, w2t = ops.vit_mlp_retile(w1, w2)
xo = x.clone()
qkv = torch.randn(M, 1152, generator=g).cuda().to(torch.bfloat16)
enc_a = torch.randn(B * 512 * 32, 128, generator=g).cuda().to(torch.bfloat16)
enc_w = (torch.randn(256, 128, generator=g) * 0.05).cuda().to(torch.bfloat16)
big = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
start = torch.zeros(B, dtype=torch.long, device="cuda")


def mlp_unfused():
    h, _, _ = ops.layernorm_fwd(xo, gam, bet, torch.bfloat16)
    f = ops.gemm(h, w1, out_dtype=torch.bfloat16, bias=b1, act=ops.ACT_GELU)
    ops.gemm(f, w2, out=xo, bias=b2, row_scale=dp, row_scale_rows=513, residual=xo)


def mlp_rowgemm():
    f = ops.rowgemm(xo, w1, ln=(gam, bet), bias=b1, act=ops.ACT_GELU)
    ops.rowgemm(f, w2, bias=b2, residual=xo, out=xo, row_scale=dp, row_scale_rows=513)


LOADS = {
    "none": None,
    "mlp_fused": lambda: ops.vit_mlp(xo, w1t, b1, w2t, b2, (gam, bet), row_scale=dp, row_scale_rows=513),
    "mlp_tiles": mlp_unfused,
    "rowgemm_qkv": lambda: ops.rowgemm(x, wq, ln=(gam, bet), bias=bq),
    "attention": lambda: ops.attention_fwd(qkv, B, 513, 6, 0.125, False, want_lse=False),
    "enc_gemm": lambda: ops.gemm(enc_a, enc_w, out_dtype=torch.bfloat16),
    "fps": lambda: ops.fps(pc, 512, start),
    "knn": lambda: ops.knn_group(pc, pc[:, :512].contiguous(), 32),
    "hbm_fill": lambda: big.fill_(1.0),
}


def chain(n):
    for _ in range(n):
        tr.step(pc, label)
    tr.finish()


chain(30)
torch.cuda.synchronize()
bg = torch.cuda.Stream()
for name in (sys.argv[1:] or list(LOADS)):
    fn = LOADS[name]
    graph = None
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
    n_steps = 30
    reps = 0 if graph is None else int(n_steps * 5.5 / t_graph) + 2          # enough background work to cover the chain's run
    torch.cuda.synchronize()
    e0, e1, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(bg):
        for _ in range(reps):
            graph.replay()
        eb.record()
    e0.record()
    chain(n_steps)
    e1.record()
    torch.cuda.synchronize()
    t_chain = e0.elapsed_time(e1)
    msg = f"{name:12s}: chain {t_chain / n_steps:6.3f} ms/step"
    if graph is not None:
        msg += f" | load alone {t_graph / 20 * 1e3:7.1f} us/launch, {reps} x 20 launches took {e0.elapsed_time(eb):7.1f} ms (alone {reps * t_graph:7.1f})"
    print(msg, flush=True)
