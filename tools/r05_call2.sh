#!/bin/bash
# Round 5, second lease: the pair launch and the two k-loop experiments - kernel tests, the lab under each switch, same-lease step A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_pair" 2>&1 | tail -3
( timeout 200 python tools/pair_lab.py; TF_GEMM_PF2=1 timeout 200 python tools/pair_lab.py; TF_GEMM_STAGGER=1 timeout 200 python tools/pair_lab.py; TF_GEMM_PF2=1 TF_GEMM_STAGGER=1 timeout 200 python tools/pair_lab.py ) 2>/dev/null | tee $O/r05_pair_lab.txt
for rep in 1 2; do
  TF_GEMM_PAIR=0 timeout 200 $B 2>/dev/null | bl "pair off               "
  timeout 200 $B 2>/dev/null | bl "pair on (default)      "
  TF_GEMM_PF2=1 timeout 200 $B 2>/dev/null | bl "pair + PF2             "
  TF_GEMM_STAGGER=1 timeout 200 $B 2>/dev/null | bl "pair + stagger         "
  TF_GEMM_PF2=1 TF_GEMM_STAGGER=1 timeout 200 $B 2>/dev/null | bl "pair + PF2 + stagger   "
done
