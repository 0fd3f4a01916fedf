#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q > $O/r06_kernel_tests.log 2>&1; grep -E "passed|failed|Error" $O/r06_kernel_tests.log | tail -5
