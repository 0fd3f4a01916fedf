#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
TAG=r06c TRACE_HEAD=8 bash tools/gpu_round4.sh trace | cut -c1-250
grep "^queue" -A14 $O/r06c_timeline.txt | cut -c1-200
