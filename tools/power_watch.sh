#!/bin/bash
# Samples socket power / clocks (rocm-smi) while bench.py replays the training step: is the fp32 step power-limited?  bash tools/power_watch.sh [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
rocm-smi --showpower --showclocks --showtemp --showmaxpower 2>/dev/null | grep -v "^=\|^$" | head -30
echo "--- idle above; running bench"
(python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-alt "$@" > /tmp/pw_bench.json 2>/dev/null) &
BP=$!
sleep 12
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks -d 0 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/  */ /g'; echo
  sleep 0.7
done
wait $BP
python -c "import json; d=json.loads(open('/tmp/pw_bench.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], 'ms/step')"
