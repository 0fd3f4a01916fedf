#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "rccl_single_rank" 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --force-pieces 8 --check --no-cpu-baseline --no-alt 2> $O/r05_fp_$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force-pieces 8:', d['ms_per_step'], d['check']['allreduce_exposed_ms'])"; echo "rc=$?"; done
timeout 300 python bench.py --force-pieces 8 --check --no-cpu-baseline --no-alt --grad-dtype bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('force-pieces 8 bf16 buckets:', d['ms_per_step'], d['check']['allreduce_exposed_ms'])"
TAG=r05 bash tools/gpu_round4.sh trace 2>&1 | head -6; head -3 $O/r05_kernel_trace_graph.txt
