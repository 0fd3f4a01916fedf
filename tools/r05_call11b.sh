#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "stored_operands_stage" 2>&1 | grep -E "Error|assert|passed|failed" | head -8
for rep in 1 2 3; do
  TF_STORE16_CONV=0 timeout 200 $B --dtype bf16 2>/dev/null | bl "bf16  in-register"
  timeout 200 $B --dtype bf16 2>/dev/null | bl "bf16  stored      "
done
TF_STORE16_CONV=0 timeout 200 $B --dtype fp16 --backbone latentTF 2>/dev/null | bl "latentTF fp16 B=16  in-register"
timeout 200 $B --dtype fp16 --backbone latentTF 2>/dev/null | bl "latentTF fp16 B=16  stored      "
