#!/bin/bash
# The round's closing evidence in ONE GPU lease: kernel trace (fp32 + bf16) -> bench (reads the fresh trace summary) -> every other bench line -> HBM kernel
# table -> the full -m gpu suite.  Everything lands in gpurun_out/ (TAG r04); copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
export TAG=${TAG:-r04}
bash tools/gpu_round4.sh trace > $O/${TAG}_trace_head.txt 2>&1
cp $O/${TAG}_kernel_trace_graph.txt profiles/${TAG}_kernel_trace_graph.txt 2>/dev/null
TRACE_TAG=_bf16 BENCH_ARGS="--dtype bf16" bash tools/gpu_round4.sh trace > /dev/null 2>&1
cp $O/${TAG}_kernel_trace_graph_bf16.txt profiles/${TAG}_kernel_trace_graph_bf16.txt 2>/dev/null
bash tools/gpu_round4.sh bench 2>&1 | tail -3 | cut -c1-600
bash tools/bench_all.sh
bash tools/gpu_round4.sh hbm 2>&1 | tail -12
bash tools/gpu_round4.sh tests_all
