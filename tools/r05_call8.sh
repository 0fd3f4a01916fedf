#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
for rep in 1 2 3; do
  TF_FUSE_BN_BWD_STATS=0 timeout 200 $B 2>/dev/null | bl "reduce + finalize + apply                      "
  TF_FUSE_BN_BWD_STATS=2 timeout 200 $B 2>/dev/null | bl "ReLU mask in the dgrad epilogue only           "
  TF_FUSE_BN_BWD_STATS=1 timeout 200 $B 2>/dev/null | bl "mask + BatchNorm sums in the dgrad epilogue    "
done
