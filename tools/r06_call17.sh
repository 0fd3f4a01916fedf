#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "graph_replay_matches_eager or segmented_graphs or full_size_step_properties or rccl" > $O/r06_zero_fork_tests.log 2>&1; grep -E "passed|failed" $O/r06_zero_fork_tests.log
for rep in 1 2 3 4; do for v in 0 1; do TF_ZERO_FORK=$v timeout 200 $B 2>/dev/null | bl "fp32 TF_ZERO_FORK=$v"; done; done
