#!/bin/bash
# Round 5, second session: GPU gate of the launch fusions (resid_drop in the GEMM epilogue / from ln2's backward, LayerNorm -> 16-bit copies, one-launch
# weight copies) + same-lease A/B on the fp32 headline and the bf16 line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
( timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "dropout or layernorm or lowp16 or producers or dropped_second or misc" 2>&1 | tail -4
  timeout 1200 python -m pytest tests/test_model_gpu.py -q -x -k "round5 or dropout_paths or graph_replay or lowp or bf16_mfma or bf16_training" 2>&1 | tail -4 ) > $O/r05_call9_tests.log 2>&1
cat $O/r05_call9_tests.log
for rep in 1 2; do
  TF_FUSE_DROPOUT=0 timeout 200 $B 2>/dev/null | bl "fp32  separate dropout launches                     "
  timeout 200 $B 2>/dev/null | bl "fp32  resid_drop in the GEMM epilogue / ln2 backward  "
done
for rep in 1 2; do
  TF_FUSE_DROPOUT=0 TF_LN_FWD16=0 TF_CAST16_MULTI=0 timeout 200 $B --dtype bf16 2>/dev/null | bl "bf16  cast launches (start of session)                "
  timeout 200 $B --dtype bf16 2>/dev/null | bl "bf16  LayerNorm -> 16-bit copies, one weight-copy launch"
done
