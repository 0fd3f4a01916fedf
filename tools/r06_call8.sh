#!/bin/bash
# Round 6, call 8 (second use): softmax / dS reductions of the wave's rows issued together vs row by row (libtransfuser_hip_base.so = this build with the previous attention.cpp)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention" 2>&1 | tail -3
echo "== rows one after the other (previous attention.cpp)"; TF_HIP_LIB=$R/transfuser_amd/libtransfuser_hip_base.so timeout 200 python tools/attention_lab.py 2>&1 | grep -v Warn | grep "C =\|#"
echo "== the wave's six rows together"; timeout 200 python tools/attention_lab.py 2>&1 | grep -v Warn | grep "C =\|#"
for rep in 1 2 3; do
  TF_HIP_LIB=$R/transfuser_amd/libtransfuser_hip_base.so timeout 200 $B 2>/dev/null | bl "fp32 attention attention: rows one after the other   "
  timeout 200 $B 2>/dev/null | bl "fp32 attention attention: six row reductions together"
done
