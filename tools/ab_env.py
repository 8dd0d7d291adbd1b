"""Dev tool: interleaved A/B of environment switches on one box.  Every variant runs tools/step_parts.py (or bench.py) in a
child process, round-robin over `rounds`, so that box-to-box and minute-to-minute drift hits all variants alike.
    python tools/ab_env.py C2 3 "PPT_FUSED_MLP=0" "PPT_FUSED_MLP=1" "PPT_FUSED_MLP=1 PPT_TEXT_FUSE_LN=0" ..."""
import os, re, subprocess, sys

name, rounds, variants = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        env = dict(os.environ)
        for kv in v.split():
            k, val = kv.split("=")
            env[k] = val
        if name.startswith("bench:"):                 # bench.py's own loop (any configuration): ms per step, then two zeros
            out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", name[6:], "--steps", "120", "--warmup", "20",
                                  "--no-cpu-baseline", "--no-roofline", "--no-parity-mode", "--no-secondary"], env=env, capture_output=True, text=True)
            m = re.search(r'"ms_per_step": ([\d.]+)(), "()', out.stdout)
            m = m or re.search(r'"ms_per_step": ([\d.]+)()()', out.stdout)
        else:
            out = subprocess.run([sys.executable, os.path.join(root, "tools", "step_parts.py"), name], env=env, capture_output=True, text=True)
            m = re.search(r"full step ([\d.]+) ms \| point tower alone ([\d.]+) ms \| prompt side alone ([\d.]+)", out.stdout)
        if not m:
            print(v, "FAILED", out.stdout[-400:], out.stderr[-800:])
            continue
        res[v].append(tuple(float(x or 0) for x in m.groups()))
        print(r, v, res[v][-1], flush=True)
print("\nvariant | full (min / median) | tower | prompt")
for v, rows in res.items():
    if not rows:
        continue
    cols = list(zip(*rows))
    med = lambda c: sorted(c)[len(c) // 2]
    print(f"{v:60s} | {min(cols[0]):.3f} / {med(cols[0]):.3f} | {med(cols[1]):.3f} | {med(cols[2]):.3f}")
