#!/bin/bash
# Round 5, third lease: merged head convolution + one-launch loss sum: tests, ATen census, plans for the new shapes, same-lease A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "merged_head or engine_graph_replay or decoder_and_head" 2>&1 | tail -3
timeout 300 python tools/aten_census.py 2>/dev/null | tee $O/r05_aten_census.txt | head -30
TF_RETUNE=0 timeout 600 python tools/tune.py $O/mi355x_r05.txt 10 256,160 fp32 2>&1 | tail -3
for rep in 1 2; do
  TF_AB_MERGE_HEADS=0 TF_AB_WSUM=0 timeout 200 $B 2>/dev/null | bl "heads per head, python loss sum   "
  TF_AB_WSUM=0 timeout 200 $B 2>/dev/null | bl "merged heads                      "
  timeout 200 $B 2>/dev/null | bl "merged heads + one-launch loss sum"
  TF_PLANS=$O/mi355x_r05.txt timeout 200 $B 2>/dev/null | bl "  + plans tuned for the new shapes"
done
TF_PLANS=$O/mi355x_r05.txt TAG=r05b bash tools/gpu_round4.sh trace 2>&1 | head -12
