#!/bin/bash
# GPU gate + same-lease A/B of the 16-bit stored trunk 1x1 convolutions (TF_STORE16_CONV)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
( timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "lowp16 or producers or stored_operands" 2>&1 | tail -4
  timeout 1200 python -m pytest tests/test_model_gpu.py -q -x -k "round5 or lowp or bf16_mfma or bf16_training or fp16" 2>&1 | tail -4 ) > $O/r05_call10_tests.log 2>&1
cat $O/r05_call10_tests.log
for rep in 1 2 3; do
  TF_STORE16_CONV=0 timeout 200 $B --dtype bf16 2>/dev/null | bl "bf16  trunk 1x1 convolutions: fp32 operands rounded in registers"
  timeout 200 $B --dtype bf16 2>/dev/null | bl "bf16  trunk 1x1 convolutions on 16-bit STORED operands         "
done
for rep in 1 2; do
  TF_STORE16_CONV=0 timeout 200 $B --dtype fp16 --backbone latentTF 2>/dev/null | bl "latentTF fp16 B=16  in-register"
  timeout 200 $B --dtype fp16 --backbone latentTF 2>/dev/null | bl "latentTF fp16 B=16  stored      "
done
timeout 200 $B 2>/dev/null | bl "fp32 (unaffected)"
