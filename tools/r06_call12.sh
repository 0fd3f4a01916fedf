#!/bin/bash
# Round 6, call 12: non-temporal policy on AdamW's streamed-once accesses (TF_ADAMW_NT bit 0 loads, bit 1 m / v stores, bit 2 parameter store)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
for v in 0 1 2 3 7; do echo "== TF_ADAMW_NT=$v"; TF_ADAMW_NT=$v timeout 200 python tools/hbm_bench.py 2>&1 | grep -i "adamw"; done
for rep in 1 2; do for v in 0 3 7; do TF_ADAMW_NT=$v timeout 200 $B 2>/dev/null | bl "fp32 TF_ADAMW_NT=$v"; done; done
