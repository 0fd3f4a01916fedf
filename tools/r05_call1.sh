#!/bin/bash
# Round 5, first GPU lease: the gate of the round-4 verdict (opt-in kernels on the hardware, the new block-gradient tests), same-lease A/B of the two switches,
# counter passes for the trunk contractions, a kernel trace with both switches on.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "im2col or stride2" 2>&1 | tail -3
TF_GROUPED_S2=1 TF_IM2COL_GEMM=1 timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "remaining_block or grouped_s2 or bench_configuration_parity or decoder_and_head" 2>&1 | tail -5
for rep in 1 2; do
  timeout 200 $B 2>/dev/null | bl "default                "
  TF_GROUPED_S2=1 timeout 200 $B 2>/dev/null | bl "TF_GROUPED_S2=1        "
  TF_IM2COL_GEMM=1 timeout 200 $B 2>/dev/null | bl "TF_IM2COL_GEMM=1       "
  TF_GROUPED_S2=1 TF_IM2COL_GEMM=1 timeout 200 $B 2>/dev/null | bl "both                   "
done
TF_STREAM_K=0 timeout 200 $B 2>/dev/null | bl "TF_STREAM_K=0          "
TAG=r05 timeout 900 bash tools/pmc_trunk.sh 2>&1 | tail -20
export TF_GROUPED_S2=1 TF_IM2COL_GEMM=1
TAG=r05a bash tools/gpu_round4.sh trace 2>&1 | head -40
