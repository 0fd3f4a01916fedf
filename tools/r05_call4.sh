#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 1200 python -m pytest tests/test_model_gpu.py -q -x -k "engine or graph or merged_head or train_cli or trajectory" 2>&1 | tail -3
for rep in 1 2 3; do
  TF_EARLY_OPT=0 timeout 200 $B 2>/dev/null | bl "AdamW after the backward          "
  timeout 200 $B 2>/dev/null | bl "first segment beside the backward "
done
TF_GEMM_PAIR=0 TF_AB_MERGE_HEADS=0 TF_AB_WSUM=0 TF_GROUPED_S2=0 TF_IM2COL_GEMM=0 TF_EARLY_OPT=0 timeout 200 $B 2>/dev/null | bl "round-4 behaviour (all round-5 switches off)"
timeout 200 $B 2>/dev/null | bl "round-5 head                      "
TAG=r05c bash tools/gpu_round4.sh trace 2>&1 | head -8
