#!/bin/bash
# Round 6, call 31: GPU_MAX_HW_QUEUES (HIP streams -> hardware queues; default 4) around the captured step, three rounds, same lease
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 200 $B 2>/dev/null | bl "warm-up run (discard)   "
for rep in 1 2 3; do
  timeout 200 $B 2>/dev/null | bl "default (4)             "
  for q in 1 2 3 5 6 8 16; do GPU_MAX_HW_QUEUES=$q timeout 200 $B 2>/dev/null | bl "GPU_MAX_HW_QUEUES=$q     "; done
done
for rep in 1 2; do
  timeout 200 $B --dtype bf16 2>/dev/null | bl "bf16 default            "
  GPU_MAX_HW_QUEUES=8 timeout 200 $B --dtype bf16 2>/dev/null | bl "bf16 GPU_MAX_HW_QUEUES=8"
done
