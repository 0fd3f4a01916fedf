#!/bin/bash
# Round 6, call 1: the new parity gates (strict B=10 / H=256 gate, 16-bit model-level tests) + baseline numbers of the round-5 kernels on this lease
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 200 $B 2>/dev/null | bl "fp32 head (round-5 kernels)"
timeout 200 $B 2>/dev/null | bl "fp32 head (round-5 kernels)"
timeout 300 python tools/grouped_lab.py > $O/r06a_grouped_lab_baseline.txt 2>&1; cat $O/r06a_grouped_lab_baseline.txt | grep -v Warn
timeout 2400 python -m pytest tests/test_model_gpu.py -q -x -s -k "bench_configuration_parity or lowp_bench_configuration or fp16_full_size or fp16_training_trajectory" > $O/r06a_new_gates.log 2>&1
grep -E "^  |passed|failed|Error|assert" $O/r06a_new_gates.log | cut -c1-400 | tail -60
