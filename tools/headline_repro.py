#!/usr/bin/env python3
"""Dev tool: the driver's headline command, `python3 bench.py --gpus 1 --steps 20 --warmup 5`, repeated on ONE box in
child processes (the expensive legs that run AFTER the timed region are skipped: they cannot change it), with the distribution of
`value` and of the K per-step times of each run.  Variants are environment settings, interleaved round-robin.
    python3 tools/headline_repro.py 10 "PPT_BENCH_GC=raw" "PPT_BENCH_GC=freeze" ..."""
import json, os, subprocess, sys, time

rounds, variants = int(sys.argv[1]), sys.argv[2:] or [""]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        env = dict(os.environ)
        for kv in v.split():
            k, val = kv.split("=")
            env[k] = val
        t0 = time.time()
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                              "--no-cpu-baseline", "--no-roofline", "--no-parity-mode", "--no-secondary"], env=env, capture_output=True, text=True)
        js = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if not js:
            print(r, v, "FAILED", out.stderr[-600:], flush=True)
            continue
        j = json.loads(js[-1])
        t = j["config"]["timing"]
        rows[v].append((j["value"], j["ms_per_step"], t["ms_per_step_median"], t["ms_per_step_max"], t["host_ms_per_step_median"],
                        t["host_ms_per_step_max"], t["gc_collections_in_timed_region"]))
        print(f"{r} [{v}] value {j['value']:.0f} clouds/s | ms/step {j['ms_per_step']:.3f} | per-step GPU median {t['ms_per_step_median']:.3f} "
              f"max {t['ms_per_step_max']:.3f} slowest {t['slowest_steps'][:3]} | host median {t['host_ms_per_step_median']:.3f} max "
              f"{t['host_ms_per_step_max']:.3f} | gc runs {t['gc_collections_in_timed_region']} | burn-in {j.get('burn_in')} | wall {time.time() - t0:.1f} s", flush=True)
print("\n| variant | runs | value min / median / max (clouds/s) | spread | ms/step median | per-step median | worst step | host median |")
print("|---|---|---|---|---|---|---|---|")
for v, rs in rows.items():
    if not rs:
        continue
    vals = sorted(x[0] for x in rs)
    med = lambda c: sorted(c)[len(c) // 2]
    print(f"| `{v or 'default'}` | {len(rs)} | {vals[0]:.0f} / {med(vals):.0f} / {vals[-1]:.0f} | ±{50 * (vals[-1] - vals[0]) / med(vals):.1f} % | "
          f"{med([x[1] for x in rs]):.3f} | {med([x[2] for x in rs]):.3f} | {max(x[3] for x in rs):.3f} | {med([x[4] for x in rs]):.3f} |")
