#!/bin/bash
# Round 6, call 25: attention backward as one grid (D = dY . Y) vs the two dependent launches: parity tests, per-launch time, step A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention" > $O/att_tests.log 2>&1; tail -1 $O/att_tests.log
timeout 300 python tools/attention_lab.py 2>/dev/null | tee $O/r06_attention_lab_one_grid.txt
for rep in 1 2 3; do
  TF_ATT_BWD_ONE=0 timeout 200 $B 2>/dev/null | bl "fp32 attention backward: two launches"
  TF_ATT_BWD_ONE=1 timeout 200 $B 2>/dev/null | bl "fp32 attention backward: one grid    "
done
