#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
TF_FORK_DECODERS=2 timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "tiny_model_losses or graph_replay_matches_eager or segmented_graphs" 2>&1 | grep -E "passed|failed"
for rep in 1 2 3; do for v in 0 1 2; do TF_FORK_DECODERS=$v timeout 200 $B 2>/dev/null | bl "fp32 TF_FORK_DECODERS=$v"; done; done
for v in 0 2; do TF_FORK_DECODERS=$v timeout 200 $B --dtype bf16 2>/dev/null | bl "bf16 TF_FORK_DECODERS=$v"; TF_FORK_DECODERS=$v timeout 200 $B --dtype fp16 --backbone latentTF 2>/dev/null | bl "latentTF fp16 TF_FORK_DECODERS=$v";  TF_FORK_DECODERS=$v timeout 200 $B --backbone geometric_fusion 2>/dev/null | bl "geometric_fusion TF_FORK_DECODERS=$v"; done
