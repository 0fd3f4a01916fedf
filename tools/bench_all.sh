#!/bin/bash
# Every bench line of the round (BASELINE configs[1..4] + the precision variants) in one GPU lease -> gpurun_out/<TAG>_bench_*.json
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; T=${TAG:-r04}; cd $R; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt "$@" > $O/${T}_bench_$name.json 2> $O/${T}_bench_$name.err; python -c "import json; d=json.loads(open('$O/${T}_bench_$name.json').read().strip().splitlines()[-1]); print('%-34s %8.3f ms/step %8.2f samples/s  loss %s' % ('$name', d['ms_per_step'], d['value'], d['config']['final_loss']))"; }
run transfuser_b10_h256_f32x3 --dtype f32x3
run transfuser_b10_h256_bf16 --dtype bf16
run transfuser_b10_h256_fp16 --dtype fp16
run transfuser_b10_h160 --height 160
run geometric_fusion_b12_h160 --backbone geometric_fusion
run latentTF_b16_h256 --backbone latentTF
run latentTF_b16_h256_fp16 --backbone latentTF --dtype fp16
run latentTF_b16_h256_bf16 --backbone latentTF --dtype bf16
