#!/bin/bash
# Round 6, call 7: F32T statistics by half_sum (DPP), stride-2 seven-wave weight gradient, transposed decoder-tail kernel; A/Bs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "grouped or conv_direct or bench_shape_convs or stride2 or decoder" 2>&1 | tail -3
for v in 0 1; do echo "== TF_GROUPED_F32T=$v"; TF_GROUPED_F32T=$v timeout 300 python tools/grouped_lab.py 2>&1 | grep -E "fwd\+stat" | grep -E "16, 44|16, 16|32, 88|64, 176"; done
for v in 0 1; do echo "== TF_SMALL_TRN=$v"; TF_SMALL_TRN=$v timeout 200 python tools/conv_bench.py 2>&1 | grep "^("; done
for rep in 1 2 3; do
  TF_GROUPED_F32T=0 TF_SMALL_TRN=0 TF_GROUPED_WGRAD7=0 timeout 200 $B 2>/dev/null | bl "fp32 round-5 direct kernels                  "
  TF_SMALL_TRN=0 timeout 200 $B 2>/dev/null | bl "fp32 grouped: wgrad7 (s1 + s2) + F32T          "
  timeout 200 $B 2>/dev/null | bl "fp32 + transposed decoder-tail kernel (head)   "
done
