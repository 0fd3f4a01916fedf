#!/bin/bash
# One GPU-box pass of the round-2 evidence (run from the repo root through gpurun); every step bounded by its own timeout.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for what in "$@"; do
case $what in
tests_new)  timeout 900 python -m pytest tests/test_model_gpu.py -q -k "graph or segmented or bench_configuration or single_block" -s 2>&1 | tail -40 > $O/r02_tests_new.log; tail -15 $O/r02_tests_new.log ;;
tests_all)  timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 > $O/r02_gpu_tests.log; tail -8 $O/r02_gpu_tests.log ;;
bench)      timeout 420 python bench.py --steps 20 --warmup 5 > $O/r02_bench_n1.json 2> $O/r02_bench_n1.err; tail -4 $O/r02_bench_n1.err; cat $O/r02_bench_n1.json ;;
bench_fast) timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r02_bench_fast.json 2> $O/r02_bench_fast.err; tail -2 $O/r02_bench_fast.err; cat $O/r02_bench_fast.json ;;
pmc)        timeout 600 bash tools/pmc_roofline.sh fp32 2>&1 | tail -12 ;;
pmc_x3)     timeout 600 bash tools/pmc_roofline.sh f32x3 2>&1 | tail -12 ;;
hbm)        timeout 200 python tools/hbm_bench.py > $O/r02_hbm_kernels.txt 2>&1; cat $O/r02_hbm_kernels.txt ;;
trace)      (cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace_r02 -o step --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline $BENCH_ARGS > $O/r02_trace_bench.log 2>&1)
            python tools/trace_csv_stats.py $O/trace_r02 > $O/r02_kernel_trace_graph${TRACE_TAG}.txt 2>&1; head -45 $O/r02_kernel_trace_graph${TRACE_TAG}.txt
            cp $O/trace_r02/*kernel_stats.csv $O/r02_kernel_stats.csv 2>/dev/null; rm -rf $O/trace_r02 ;;
tune)       timeout 400 python tools/tune.py $O/mi355x_r02.txt 10 256,160 fp32,bf16 2>&1 | tail -6; cp $O/mi355x_r02.txt transfuser_amd/plans/mi355x.txt ;;
tests_grouped) timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "grouped_conv or bench_shape_convs" 2>&1 | tail -5 ;;
conv_bench) timeout 200 python tools/grouped_bench.py 2>&1 | tail -20 ;;
ab_grouped) for v in 0 1 0 1; do TF_GROUPED_CONV=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('TF_GROUPED_CONV=$v', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; GPT4 fc1', d['roofline']['avg_launch_us'], 'us')"; done ;;
ab_bf16)    for v in "1 1" "0 1" "1 0" "0 0"; do set -- $v; TF_GROUPED_CONV=$1 TF_DIRECT_CONV=$2 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 TF_GROUPED_CONV=$1 TF_DIRECT_CONV=$2', d['ms_per_step'], 'ms/step', d['value'], 'samples/s')"; done ;;
test_graph) timeout 300 python -m pytest tests/test_model_gpu.py -q -k "graph_replay" -s 2>&1 | grep -v "Warning\|warn" | tail -30 ;;
bench_cfgs) for a in "--backbone geometric_fusion" "--backbone latentTF" "--height 160" "--backbone late_fusion_skip"; do
              case "$a" in *skip) continue;; esac
              tag=$(echo $a | tr -d ' -' ); timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $a > $O/r02_bench_$tag.json 2> $O/r02_bench_$tag.err; tail -1 $O/r02_bench_$tag.err; cat $O/r02_bench_$tag.json; done ;;
tests_bf16) timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -k "bf16" -s 2>&1 | tail -12 ;;
bench_bf16) timeout 300 python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline > $O/r02_bench_bf16.json 2> $O/r02_bench_bf16.err; tail -3 $O/r02_bench_bf16.err; cat $O/r02_bench_bf16.json ;;
x3_tests)   timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "f32x3 or gemm_dma" 2>&1 | tail -8 ;;
x3_bench)   timeout 400 python tools/x3_bench.py 10 > $O/r02_x3_bench.txt 2>&1; grep BEST $O/r02_x3_bench.txt ;;
x3_tune)    TF_RETUNE=0 timeout 400 python tools/tune.py $O/mi355x_x3.txt 10 256,160 f32x3 2>&1 | tail -4 ;;
x3_ab)      for d in f32 f32x3 f32 f32x3; do TF_PLANS=$O/mi355x_x3.txt timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype $d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dtype $d', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'], '; GPT4 fc1', d['roofline']['avg_launch_us'], 'us', d['roofline']['achieved'], 'TF/s; engine', d['roofline']['engine_ms_per_step'], 'ms')"; done ;;
x3_model)   timeout 900 python -m pytest tests/test_model_gpu.py -q -k "f32x3" -s 2>&1 | grep -v "Warning\|warn" | tail -25 ;;
x3_retune)  awk -F';' '/^#/ || $6<8' transfuser_amd/plans/mi355x.txt > $O/plans_nox3.txt; cp $O/plans_nox3.txt transfuser_amd/plans/mi355x.txt
            TF_RETUNE=0 timeout 500 python tools/tune.py $O/mi355x_x3.txt 10 256,160 f32x3 2>&1 | tail -4 ;;
x3_ab2)     for v in "f32x3 1" "f32x3 0" "f32 1" "f32x3 1" "f32x3 0"; do set -- $v; TF_X3_DIRECT=$2 TF_PLANS=$O/mi355x_x3.txt timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --dtype $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dtype $1 TF_X3_DIRECT=$2', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'], '; GPT4 fc1', d['roofline']['avg_launch_us'], 'us', d['roofline']['achieved'], 'TF/s; engine', d['roofline']['engine_ms_per_step'], 'ms')"; done ;;
x3_tests2)  timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "f32x3 or gemm_dma or bf16" 2>&1 | tail -8 ;;
census_x3)  timeout 300 python tools/census.py 10 256 f32x3 > $O/r02_census_f32x3.txt 2>&1; head -70 $O/r02_census_f32x3.txt ;;
tp_tests)   timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "two_pass or gemm_dma or test_gemm or bench_shape" 2>&1 | tail -8 ;;
retune_all) TF_RETUNE=1 timeout 900 python tools/tune.py $O/mi355x_r02b.txt 10 256,160 fp32,f32x3,bf16 2>&1 | tail -8 ;;
ab_tp)      for v in "f32 1" "f32 0" "f32x3 1" "f32x3 0" "f32 1" "f32x3 1"; do set -- $v; TF_TWO_PASS_SPLITK=$2 TF_PLANS=$O/mi355x_r02b.txt timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --dtype $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dtype $1 TWO_PASS=$2', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'], '; GPT4 fc1', d['roofline']['avg_launch_us'], 'us', d['roofline']['achieved'], 'TF/s; engine', d['roofline']['engine_ms_per_step'], 'ms')"; done ;;
diag_tp)    TF_TWO_PASS_SPLITK=0 TF_RETUNE=1 timeout 150 python tools/tune.py $O/diag_notp.txt 10 256 fp32 2>&1 | tail -3
            TF_TRACE_GEMM=1 TF_RETUNE=1 timeout 200 python tools/tune.py $O/diag_tp.txt 10 256 fp32 > $O/diag_tp.log 2>&1; tail -6 $O/diag_tp.log | cut -c1-300 ;;
diag_tp2)   TF_TRACE_CALLS=1 TF_TRACE_GEMM=1 TF_RETUNE=1 timeout 250 python tools/tune.py $O/diag_tp.txt 10 256 fp32 > $O/diag_tp.log 2>&1; tail -12 $O/diag_tp.log | cut -c1-300 ;;
diag_tp3)   TF_TRACE_TUNE=1 TF_RETUNE=1 timeout 250 python tools/tune.py $O/diag_tp.txt 10 256 fp32 > $O/diag_tp.log 2>&1; grep -v "ok$" $O/diag_tp.log | tail -8 | cut -c1-300; tail -5 $O/diag_tp.log | cut -c1-300 ;;
ps_tests)   timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "direct" 2>&1 | tail -4 ;;
ab_ps)      for v in 1 0 1 0; do TF_X3_PRESPLIT=$v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --dtype f32x3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32x3 PRESPLIT=$v', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; done
            for v in 1 0; do TF_X3_PRESPLIT=$v timeout 100 python tools/conv_bench_x3.py 2>&1 | tail -4; done ;;
tune_cfgs)  cp transfuser_amd/plans/mi355x.txt $O/mi355x_r02c.txt
            for a in "12 160 geometric_fusion" "16 256 latentTF"; do set -- $a; TF_PLANS=$O/mi355x_r02c.txt TF_RETUNE=0 timeout 300 python tools/tune.py $O/mi355x_r02c.txt $1 $2 fp32,f32x3,bf16 $3 2>&1 | grep "H=\|saved"; done ;;
bench_cfgs2) for a in "--backbone geometric_fusion" "--backbone latentTF" "--height 160" "--dtype bf16"; do
              tag=$(echo $a | tr -d ' -' ); TF_PLANS=$O/mi355x_r02c.txt timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $a > $O/r02_bench_$tag.json 2> $O/r02_bench_$tag.err; tail -1 $O/r02_bench_$tag.err; python -c "import sys,json; d=json.load(open('$O/r02_bench_$tag.json')); print('$a', d['ms_per_step'], 'ms/step', d['value'], 'samples/s', ('; f32x3 %s ms %s samples/s' % (d['f32x3']['ms_per_step'], d['f32x3']['value'])) if 'f32x3' in d and 'value' in d['f32x3'] else '')"; done ;;
esac
done
