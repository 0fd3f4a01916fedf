#!/bin/bash
# Round 6, call 22: LayerNorm backward in one pass (dx + dgamma / dbeta) vs the two-kernel form: parity tests, per-op time, step A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "layernorm" > $O/ln_tests.log 2>&1; tail -2 $O/ln_tests.log
for f in 0 1; do echo "== TF_LN_FUSED=$f"; TF_LN_FUSED=$f timeout 300 python tools/hbm_bench.py --iters 200 2>/dev/null | grep "layernorm_bwd"; done
for rep in 1 2 3; do
  TF_LN_FUSED=0 timeout 200 $B 2>/dev/null | bl "fp32 two-kernel LN backward"
  TF_LN_FUSED=1 timeout 200 $B 2>/dev/null | bl "fp32 one-pass LN backward  "
done
