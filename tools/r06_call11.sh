#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -x > $O/r06b_gpu_tests.log 2>&1; tail -12 $O/r06b_gpu_tests.log | cut -c1-300
