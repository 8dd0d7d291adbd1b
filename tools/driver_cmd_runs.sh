#!/bin/bash
# Dev tool: the driver's own bench command, n times on this box; one summary line per run (gpurun_out/r6_final_driver_cmd_<i>.json).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
for i in $(seq 1 ${1:-3}); do
    t0=$(date +%s.%N)
    python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_final_driver_cmd_$i.json 2> gpurun_out/r6_final_driver_cmd_$i.err
    t1=$(date +%s.%N)
    python3 - $i $t0 $t1 <<'PY'
import json, sys
i, t0, t1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
d = json.loads(open(f"gpurun_out/r6_final_driver_cmd_{i}.json").read().strip().splitlines()[-1])
t = d["config"]["timing"]
sec = {k: (v.get("value") if isinstance(v, dict) else v) for k, v in d.get("secondary", {}).items()}
print(f"run {i}: wall {t1 - t0:.0f} s | {d['value']} clouds/s {d['ms_per_step']} ms (median {t.get('ms_per_step_median')}, max {t.get('ms_per_step_max')}) roofline {d['roofline']['frac']} | {sec}")
PY
done
