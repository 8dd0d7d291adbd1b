"""Dev tool: the text tower's attention backward alone (prefix-shared layout: C prompts x T positions, P shared) -- mean launch time of
every kernel it runs, from HIP events around `n` back-to-back calls (PPT_HIP_LIB selects a variant build).
    python3 tools/attn_short_bench.py [C T P H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppt_amd import ops

C, T, P, H = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (40, 37, 17, 8)
rows = ops.prefix_rows(C, T, P)
torch.manual_seed(0)
qkv = (torch.randn(rows, 3 * H * 64, device="cuda") * 0.5).half()
out, lse = ops.attention_prefix_fwd(qkv, C, T, P, H, 0.125)
dout = torch.randn_like(out)
for what, fn in (("forward", lambda: ops.attention_prefix_fwd(qkv, C, T, P, H, 0.125)),
                 ("backward (+ prefix reduce)", lambda: ops.attention_prefix_bwd(qkv, out, dout, lse, C, T, P, H, 0.125))):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 300
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{what}: {a.elapsed_time(b) / n * 1e3:.2f} us per call (rows {rows})")
