#!/bin/bash
# Round 6 closing evidence in ONE GPU lease (TAG r06): kernel traces (fp32 + bf16; each summary names the build it traced) + timeline -> counter passes (trunk contractions
# incl. the grouped 3x3 kernels, bandwidth-bound kernels incl. H1, the GPT-4 GEMM) -> the default bench line (reads the fresh summaries from profiles/; embeds f32x3, BASELINE
# configs[2..4], the multi-GPU path on one GPU, the CPU baseline incl. the all-cores probe) -> same-lease comparison against the round-5 behaviour of the same library
# (every round-6 switch off) -> labs -> the full -m gpu suite.  Everything lands in gpurun_out/; the summaries bench.py reads are copied to profiles/ on the box.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
export TAG=r06
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
TRACE_HEAD=12 bash tools/gpu_round4.sh trace > $O/${TAG}_trace_head.txt 2>&1; head -4 $O/${TAG}_trace_head.txt | cut -c1-300
cp $O/${TAG}_kernel_trace_graph.txt profiles/${TAG}_kernel_trace_graph.txt 2>/dev/null
TRACE_TAG=_bf16 BENCH_ARGS="--dtype bf16" bash tools/gpu_round4.sh trace > /dev/null 2>&1
cp $O/${TAG}_kernel_trace_graph_bf16.txt profiles/${TAG}_kernel_trace_graph_bf16.txt 2>/dev/null
TAG=$TAG timeout 900 bash tools/pmc_trunk.sh 2>&1 | tail -22 | cut -c1-260
cp $O/${TAG}_pmc_trunk.json $O/${TAG}_pmc_trunk.txt profiles/ 2>/dev/null
timeout 900 bash tools/pmc_hbm.sh $TAG 2>&1 | tail -40 | cut -c1-260
cp $O/${TAG}_pmc_hbm.json $O/${TAG}_pmc_hbm.txt profiles/ 2>/dev/null
TAG=$TAG timeout 700 bash tools/pmc_roofline.sh fp32 2>&1 | tail -8 | cut -c1-260
cp $O/${TAG}_pmc_gemm_roofline.json $O/${TAG}_pmc_gemm_roofline.txt profiles/ 2>/dev/null
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err ) 2>&1 | tail -3
tail -14 $O/${TAG}_bench_n1.err
S5="TF_GROUPED_WGRAD7=0 TF_GROUPED_F32T=0 TF_SMALL_TRN=0 TF_HIST_SLAB=0 TF_FORK_DECODERS=0 TF_ATT_BWD_ONE=0 TF_ATT_ROWLDS=0 TF_PILLAR_V2=0"
for rep in 1 2 3; do
  env $S5 timeout 200 $B 2>/dev/null | bl "fp32 round-5 behaviour of this library (every round-6 switch off)                         "
  timeout 200 $B 2>/dev/null | bl "fp32 round-6 head                                                                         "
done
for rep in 1 2; do
  timeout 200 $B --dtype bf16 2>/dev/null | bl "bf16 round-6 head"
  timeout 200 $B --dtype fp16 --backbone latentTF 2>/dev/null | bl "latentTF fp16 B=16 round-6 head"
done
timeout 300 python tools/hbm_bench.py > $O/${TAG}_hbm_kernels.txt 2>&1; grep -v Warn $O/${TAG}_hbm_kernels.txt | tail -16 | cut -c1-220
timeout 300 python tools/grouped_lab.py 2>&1 | grep -v Warn > $O/${TAG}_grouped_lab.txt; head -9 $O/${TAG}_grouped_lab.txt
timeout 200 python tools/attention_lab.py 2>&1 | grep -v Warn > $O/${TAG}_attention_lab.txt; cat $O/${TAG}_attention_lab.txt
timeout 300 python tools/pair_lab.py 2>&1 | grep -v Warn > $O/${TAG}_pair_lab.txt; cat $O/${TAG}_pair_lab.txt | cut -c1-220
timeout 3000 python -m pytest tests -q -m gpu > $O/${TAG}_gpu_tests.log 2>&1; tail -3 $O/${TAG}_gpu_tests.log | cut -c1-200
