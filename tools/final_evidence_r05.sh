#!/bin/bash
# Round 5 closing evidence in ONE GPU lease: kernel trace (fp32 + bf16, each summary names the build it traced) -> the default bench line (reads the fresh
# trace summary; embeds f32x3, BASELINE configs[2..4], the multi-GPU path on one GPU, the CPU baseline) -> same-lease comparison with every round-5
# switch off (= round-4 behaviour of the same library) -> HBM kernel table -> the full -m gpu suite.  Everything lands in gpurun_out/ (TAG r05).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
export TAG=r05
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
bash tools/gpu_round4.sh trace > $O/${TAG}_trace_head.txt 2>&1
cp $O/${TAG}_kernel_trace_graph.txt profiles/${TAG}_kernel_trace_graph.txt 2>/dev/null
TRACE_TAG=_bf16 BENCH_ARGS="--dtype bf16" bash tools/gpu_round4.sh trace > /dev/null 2>&1
cp $O/${TAG}_kernel_trace_graph_bf16.txt profiles/${TAG}_kernel_trace_graph_bf16.txt 2>/dev/null
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err ) 2>&1 | tail -3
tail -12 $O/${TAG}_bench_n1.err
for rep in 1 2 3; do
  TF_GEMM_PAIR=0 TF_AB_MERGE_HEADS=0 TF_AB_WSUM=0 TF_GROUPED_S2=0 TF_IM2COL_GEMM=0 timeout 200 $B 2>/dev/null | bl "round-4 behaviour (pair launch, merged heads, one-launch loss sum, stride-2 grouped, im2col GEMM off)"
  timeout 200 $B 2>/dev/null | bl "round-5 head                                                                                        "
done
bash tools/gpu_round4.sh hbm 2>&1 | tail -14
bash tools/gpu_round4.sh tests_all
