#!/bin/bash
# Round 6, call 2: seven-wave grouped weight gradient (conv3x3_grouped_wgrad7_kernel): parity on the MI355X, in-graph lab vs the three-wave kernel, step A/B; the 16-bit gates with the print-first helper
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "grouped_conv or bench_shape_convs" 2>&1 | tail -3
for v in 0 1 512 1024; do echo "== TF_GROUPED_WGRAD7=$v"; TF_GROUPED_WGRAD7=$v timeout 300 python tools/grouped_lab.py 2>&1 | grep wgrad; done
for rep in 1 2 3; do
  TF_GROUPED_WGRAD7=0 timeout 200 $B 2>/dev/null | bl "fp32 three-wave wgrad + partial panels"
  timeout 200 $B 2>/dev/null | bl "fp32 seven-wave wgrad, atomics        "
done
timeout 2400 python -m pytest tests/test_model_gpu.py -q -s -k "lowp_bench_configuration or fp16_full_size or fp16_training_trajectory" > $O/r06a_lowp_gates.log 2>&1
grep -E "^  |passed|failed|Error|assert" $O/r06a_lowp_gates.log | cut -c1-900 | tail -30
