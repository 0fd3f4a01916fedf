#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
TAG=r06a TRACE_HEAD=140 bash tools/gpu_round4.sh trace
