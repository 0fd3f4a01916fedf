#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "bn_backward_sums or gemm_pair" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_model_gpu.py -q -x -k "ride_on or single_block or remaining_block or bench_configuration_parity or engine_graph_replay" 2>&1 | tail -3
for rep in 1 2 3; do
  TF_FUSE_BN_BWD_STATS=0 timeout 200 $B 2>/dev/null | bl "reduce + finalize + apply          "
  timeout 200 $B 2>/dev/null | bl "mask + sums in the dgrad epilogue  "
done
TAG=r05d bash tools/gpu_round4.sh trace 2>&1 | head -4; sed -n 2,4p $O/r05d_kernel_trace_graph.txt
