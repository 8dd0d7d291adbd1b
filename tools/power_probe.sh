# gpurun -- 'bash tools/power_probe.sh [bench args]': socket power and shader clock sampled while bench.py runs (rocm-smi, 10 samples)
cd $GRAFT_REPO_ROOT
rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | head -8
echo "--- under load: $@"
python bench.py "$@" --steps 20000 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-mode --no-secondary > /tmp/pp.json 2>/dev/null &
PID=$!
sleep 35
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk" | tr '\n' ' '; echo; sleep 0.7; done
wait $PID
python -c "import json; d=json.loads([l for l in open('/tmp/pp.json').read().splitlines() if l.startswith('{')][-1]); print('value', d['value'], 'ms', d['ms_per_step'])"
rocm-smi --showmaxpower 2>&1 | grep -i "max" | head -3
