"""bf16 vs fp32 PointMLP in train mode: output and BatchNorm running statistics (diagnostic)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from ppt_amd import weights as W
from ppt_amd.models.pointmlp.pointMLP import pointMLP
sd = W.synth_state_dict(W.pointmlp_spec(prefix=""), seed=0)
m = pointMLP(); m.load_state_dict(sd); m.cuda().train()
B = 8
pc, s0 = W.synth_clouds(B, 1024, seed=63)
st = [s0] + [W.synth_clouds(B, n, seed=64 + i)[1] for i, n in enumerate((512, 256, 128))]
outs, stats = [], []
for prec in (torch.float32, torch.bfloat16):
    m.load_state_dict(sd); m.precision, m._wc = prec, None
    m.fps_start = tuple(torch.from_numpy(s).cuda() for s in st)
    m.dropout_masks = (torch.ones(B, 512), torch.ones(B, 256))
    outs.append(m(torch.from_numpy(pc).cuda()).cpu())
    stats.append({k: v.cpu().clone() for k, v in m.state_dict().items() if "running_" in k})
print("out rel", ((outs[0] - outs[1]).norm() / outs[0].norm()).item())
worst = sorted(((( stats[0][k] - stats[1][k]).norm() / (stats[0][k] - sd[k]).norm().clamp_min(1e-12)).item(), k) for k in stats[0])
for r, k in worst[-8:]:
    print(f"{r:.4f} {k}")
