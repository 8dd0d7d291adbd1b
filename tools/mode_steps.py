"""Dev tool: N training steps of a bench configuration in a given precision mode (fp32 | split16 | mixed16) -- the thing to put
under rocprofv3 --kernel-trace, or to time.     python3 tools/mode_steps.py C2 fp32 12"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ppt_amd.train import Trainer

cfg_name, mode, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
torch.cuda.set_device(0)
cfg = bench.CONFIGS[cfg_name]
m = bench.build_model(cfg["dataset"], cfg["head_type"], precision=torch.float32 if mode != "mixed16" else torch.bfloat16,
                      model=cfg.get("model", "ULIP_PointBERT"), task=cfg.get("task", "cls"))
if mode == "split16":
    m.set_precision("split16")
m.train()
import numpy as np
from ppt_amd import weights as W
B, N = cfg["batch"], cfg["npoints"]
partseg = cfg.get("task") == "partseg"
pc = torch.from_numpy(W.synth_clouds(B, N, seed=1234)[0]).cuda()
label = torch.from_numpy(np.random.default_rng(0).integers(0, len(m.prompt_learner.classnames), size=(B, N) if partseg else (B,))).cuda()
extra = (torch.nn.functional.one_hot(torch.arange(B) % 16, 16).float().cuda(),) if partseg else None
tr = Trainer(m, lr=3e-3, label_smoothing=0.2, distributed=False)
if os.environ.get('PPT_MODE_STEPS_VOUCH', '1') != '0':      # the resident batch is complete: input-only stages run ahead (as bench.py)
    from ppt_amd import graphs
    graphs.shared_group_stream()
    tr.inputs_ready = True
    tr.group_ahead_when_frozen = True
if extra is not None:
    tr.extra_inputs = extra
for _ in range(6):
    tr.step(pc, label)
tr.finish(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss, _ = tr.step(pc, label)
tr.finish(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{cfg_name} {mode}: {1e3 * dt / steps:.3f} ms/step, {pc.shape[0] * steps / dt:.1f} clouds/s, loss {loss.item():.5f}")
