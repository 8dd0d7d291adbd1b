"""Dev tool: host enqueue time vs GPU time of the C2 step (is the step launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    torch.cuda.set_device(0)
    cfg = bench.CONFIGS["C2"]
    import argparse
    from ppt_amd.train import Trainer
    from ppt_amd import weights as W
    model = bench.build_model(cfg["dataset"], cfg["head_type"], torch.bfloat16, cfg.get("model", "ULIP_PointBERT"), cfg.get("task", "cls"))
    model.train()
    tr = Trainer(model, lr=3e-3, label_smoothing=0.2, distributed=False)
    pc_np, _ = W.synth_clouds(cfg["batch"], cfg["npoints"], seed=1)
    pc = torch.from_numpy(pc_np).cuda()
    label = torch.randint(0, 40, (cfg["batch"],), device="cuda")
    from ppt_amd import graphs
    graphs.shared_text_stream(); graphs.shared_group_stream()
    for ra, ga in ((True, False), (True, True), (True, False), (True, True)):
        tr.run_ahead = ra
        tr.inputs_ready = ga
        for _ in range(10):
            tr.step(pc, label)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            tr.step(pc, label)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"run_ahead={ra} group_ahead={ga}: host enqueue {1e3*(t1-t0)/50:.3f} ms/step, total {1e3*(t2-t0)/50:.3f} ms/step", flush=True)
    for burst in (2, 4, 8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(burst):
            tr.step(pc, label)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"burst {burst}: host {1e3*(t1-t0)/burst:.3f} ms/step, total {1e3*(t2-t0)/burst:.3f} ms/step", flush=True)
    # host-only cost: same loop under a profiler of python time
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        tr.step(pc, label)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)



main()
