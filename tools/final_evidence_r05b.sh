#!/bin/bash
# Round 5, second session - closing evidence in ONE GPU lease (TAG r05; the first session's files were kept as profiles/r05a_*): kernel traces (fp32 + bf16; each summary names the build it traced) -> the
# default bench line (embeds f32x3, BASELINE configs[2..4], the multi-GPU path on one GPU, the CPU baseline) -> same-lease comparisons:
#   fp32: round-4 behaviour of this library / first-session behaviour (launch fusions off) / head;   bf16: first-session behaviour / head
# -> HBM kernel table -> the full -m gpu suite.  Everything lands in gpurun_out/.
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
S1="TF_FUSE_DROPOUT=0 TF_LN_FWD16=0 TF_CAST16_MULTI=0 TF_STORE16_CONV=0"
for rep in 1 2 3; do
  env $S1 TF_GEMM_PAIR=0 TF_AB_MERGE_HEADS=0 TF_AB_WSUM=0 TF_GROUPED_S2=0 TF_IM2COL_GEMM=0 timeout 200 $B 2>/dev/null | bl "fp32 round-4 behaviour of this library                     "
  env $S1 timeout 200 $B 2>/dev/null | bl "fp32 first session of round 5 (launch fusions off)         "
  timeout 200 $B 2>/dev/null | bl "fp32 round-5 head                                          "
done
for rep in 1 2 3; do
  env $S1 timeout 200 $B --dtype bf16 2>/dev/null | bl "bf16 first session of round 5 (cast launches)              "
  timeout 200 $B --dtype bf16 2>/dev/null | bl "bf16 round-5 head (LayerNorm -> 16-bit copies, 1 weight cast, stored trunk 1x1 convolutions)"
done
for rep in 1 2; do
  env $S1 timeout 200 $B --dtype fp16 --backbone latentTF 2>/dev/null | bl "latentTF fp16 B=16 first session of round 5"
  timeout 200 $B --dtype fp16 --backbone latentTF 2>/dev/null | bl "latentTF fp16 B=16 round-5 head            "
done
bash tools/gpu_round4.sh hbm 2>&1 | tail -16
bash tools/gpu_round4.sh tests_all
