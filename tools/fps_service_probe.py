import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from ppt_amd import data as PD
from ppt_amd import ops
# 1. the kernel alone, batch 8
x = torch.randn(8, 8192, 3, device="cuda"); st = torch.zeros(8, dtype=torch.int64, device="cuda")
for _ in range(3): ops.fps(x, 1024, st)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): ops.fps(x, 1024, st)
torch.cuda.synchronize(); print("fps kernel B=8 N=8192 M=1024: %.3f ms" % (1e3 * (time.perf_counter() - t0) / 20))
svc = PD.start_fps_service()
c = np.random.default_rng(0).standard_normal((8192, 3)).astype(np.float32)
import multiprocessing as mp
def work(wid, n, q):
    t0 = time.perf_counter()
    for i in range(n):
        svc.request(wid, c, 1024, i % 8192)
    q.put((wid, time.perf_counter() - t0))
ctx = mp.get_context("fork")
for W in (1, 8, 16):
    q = ctx.Queue(); n = 100
    s0, l0, ls0 = svc.served, svc.launches, svc.launch_s
    ps = [ctx.Process(target=work, args=(w, n, q)) for w in range(W)]
    t0 = time.perf_counter()
    for p in ps: p.start()
    for p in ps: p.join()
    dt = time.perf_counter() - t0
    print(f"{W} workers x {n}: {W * n / dt:.0f} clouds/s; {svc.launches - l0} launches, {(svc.served - s0) / max(svc.launches - l0, 1):.1f} per launch, {1e3 * (svc.launch_s - ls0) / max(svc.launches - l0, 1):.2f} ms per launch")
PD.stop_fps_service()
