#!/bin/bash
# Round 6, call 23: the four-launch pillar index on the MI355X: parity tests (kernel + model level), per-call time of both forms, PMC traffic
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -k "pillar" > $O/pillar_tests.log 2>&1; tail -3 $O/pillar_tests.log
timeout 300 python tools/hbm_bench.py --iters 50 2>/dev/null | grep "H1\|H2"
bash tools/pmc_hbm.sh r06c > /dev/null 2>&1
grep "H1\|H2\|^#" $O/r06c_pmc_hbm.txt | cut -c1-230
