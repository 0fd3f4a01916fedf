#!/bin/bash
# Round 6, call 6: transposed-accumulator fp32 grouped forward / input gradient (16-byte stores) vs the pixel-major form; what the MFMA engine's epilogue stores cost
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "grouped_conv or bench_shape_convs" 2>&1 | tail -3
for v in 0 1; do echo "== TF_GROUPED_F32T=$v"; TF_GROUPED_F32T=$v timeout 300 python tools/grouped_lab.py 2>&1 | grep -E "fwd|dgrad"; done
for v in 0 1; do echo "== TF_GEMM_DBG=$v (1: no epilogue stores in the register-staged engine kernels)"; TF_GEMM_DBG=$v timeout 300 python tools/pair_lab.py 2>&1 | grep -v Warn | grep "^("; done
for rep in 1 2 3; do
  TF_GROUPED_F32T=0 timeout 200 $B 2>/dev/null | bl "fp32 pixel-major grouped fwd / dgrad (16 scalar stores)"
  timeout 200 $B 2>/dev/null | bl "fp32 transposed accumulator (3 x 16-byte stores)      "
done
