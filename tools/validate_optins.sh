#!/bin/bash
# First GPU call of the next round: the opt-in kernels written at the end of round 4 (TF_GROUPED_S2: direct stride-2 grouped forward / weight gradient, measured
# -0.4 ms/step in three same-lease pairs; TF_IM2COL_GEMM: im2col + split-K GEMM for the decoders' first convolution, never run on the GPU).
#   1. the kernel tests of both, 2. the FULL -m gpu suite with both switched on, 3. same-lease A/B of each switch.
# If 2 is green and 3 confirms the gains: flip the defaults in transfuser_amd/ops.py (_GROUPED_S2, _IM2COL_GEMM), add a -m gpu test for check_im2col_gemm_conv.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 300 python - <<'PY' 2>&1 | tail -3
import sys; sys.path.insert(0, "tests")
import torch, kernel_cases as kc
kc.check_im2col_gemm_conv("cuda"); kc.check_grouped_s2_modes("cuda"); torch.cuda.synchronize(); print("opt-in kernel checks on the GPU: ok")
PY
TF_GROUPED_S2=1 TF_IM2COL_GEMM=1 timeout 1500 python -m pytest tests -q -m gpu -x > $O/optins_gpu_tests.log 2>&1; tail -3 $O/optins_gpu_tests.log
for rep in 1 2; do
  timeout 200 $B 2>/dev/null | bl "default                "
  TF_GROUPED_S2=1 timeout 200 $B 2>/dev/null | bl "TF_GROUPED_S2=1        "
  TF_IM2COL_GEMM=1 timeout 200 $B 2>/dev/null | bl "TF_IM2COL_GEMM=1       "
  TF_GROUPED_S2=1 TF_IM2COL_GEMM=1 timeout 200 $B 2>/dev/null | bl "both                   "
done
