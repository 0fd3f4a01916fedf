#!/bin/bash
# Round 6, call 3: pipelined seven-wave grouped weight gradient; H1 slab kernel (LDS counters) parity + time vs the global-atomic form; 16-bit gates
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "grouped_conv or bench_shape_convs or lidar_hist" 2>&1 | tail -3
for v in 0 1; do echo "== TF_GROUPED_WGRAD7=$v"; TF_GROUPED_WGRAD7=$v timeout 300 python tools/grouped_lab.py 2>&1 | grep wgrad; done
for v in 0 1; do echo "== TF_HIST_SLAB=$v"; TF_HIST_SLAB=$v timeout 200 python tools/hbm_bench.py 2>&1 | grep -i "H1\|H2"; done
for rep in 1 2 3; do
  TF_GROUPED_WGRAD7=0 timeout 200 $B 2>/dev/null | bl "fp32 three-wave wgrad + partial panels"
  timeout 200 $B 2>/dev/null | bl "fp32 seven-wave wgrad, atomics        "
done
timeout 2400 python -m pytest tests/test_model_gpu.py -q -s -k "lowp_bench_configuration or fp16_full_size or fp16_training_trajectory" > $O/r06a_lowp_gates.log 2>&1
grep -E "^  (transFuser|latentTF)|passed|failed|Error" $O/r06a_lowp_gates.log | cut -c1-1200 | tail -30
