#!/bin/bash
# Round 6, call 13: the segmentation / depth decoders as two more parallel graph branches (TF_FORK_DECODERS=1) - correctness (graph == eager, parity) and step A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
TF_FORK_DECODERS=1 timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "tiny_model_losses or graph_replay_matches_eager or segmented_graphs or full_size_step_properties" 2>&1 | tail -3
for rep in 1 2 3; do
  timeout 200 $B 2>/dev/null | bl "fp32 decoders on the main stream   "
  TF_FORK_DECODERS=1 timeout 200 $B 2>/dev/null | bl "fp32 decoders forked (two branches)"
done
