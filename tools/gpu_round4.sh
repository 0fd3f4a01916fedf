#!/bin/bash
# One GPU-box pass of round-4 evidence (run from the repo root through gpurun); every step bounded by its own timeout.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
T=${TAG:-r04}
bench_line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'], '; GPT4 fc1', d['roofline'].get('avg_launch_us'), 'us; engine', d['roofline']['engine_ms_per_step'], 'ms')"; }
for what in "$@"; do
case $what in
ktests)     timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x ${KARGS:+-k "$KARGS"} 2>&1 | grep -v "Warning\|warn" | tail -15 ;;
mtests)     timeout 2400 python -m pytest tests/test_model_gpu.py -v -x ${MARGS:+-k "$MARGS"} > $O/${T}_mtests.log 2>&1; grep -E "PASSED|FAILED|ERROR|passed|failed|Error|error:" $O/${T}_mtests.log | head -60 ;;
tests_all)  timeout 2700 python -m pytest tests -q -m gpu > $O/${T}_gpu_tests.log 2>&1; tail -8 $O/${T}_gpu_tests.log ;;
bench)      timeout 600 python bench.py --steps 20 --warmup 5 > $O/${T}_bench_n1.json 2> $O/${T}_bench_n1.err; tail -4 $O/${T}_bench_n1.err; cat $O/${T}_bench_n1.json ;;
bench_fast) timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt $BENCH_ARGS 2> $O/${T}_bench_fast.err | tee $O/${T}_bench_fast.json | bench_line fast ;;
ab_lib)     # A/B of two builds of the library inside this lease: the build of the previous commit (libtransfuser_hip_base.so) vs the current one
            for rep in 1 2; do
              TF_HIP_LIB=$R/transfuser_amd/libtransfuser_hip_base.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt $BENCH_ARGS 2>/dev/null | bench_line "base"
              timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt $BENCH_ARGS 2>/dev/null | bench_line "new "
            done ;;
ab)         for v in $AB_VALUES; do env $AB_VAR=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt $BENCH_ARGS 2>/dev/null | bench_line "$AB_VAR=$v"; done ;;
trace)      (cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace_$T -o step --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-alt $BENCH_ARGS > $O/${T}_trace_bench.log 2>&1)
            python tools/trace_csv_stats.py $O/trace_$T $O/${T}_trace_bench.log > $O/${T}_kernel_trace_graph${TRACE_TAG}.txt 2>&1; python tools/trace_timeline.py $O/trace_$T > $O/${T}_timeline${TRACE_TAG}.txt 2>&1; head -${TRACE_HEAD:-60} $O/${T}_timeline${TRACE_TAG}.txt
            cp $O/trace_$T/*kernel_stats.csv $O/${T}_kernel_stats${TRACE_TAG}.csv 2>/dev/null; rm -rf $O/trace_$T ;;
check)      timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --check 2>&1 | tail -6 ;;
pmc_hbm)    timeout 600 bash tools/pmc_hbm.sh $T 2>&1 | tail -24 ;;
pmc)        for pr in ${PMC_PRECS:-fp32 f32x3 bf16 fp16}; do TAG=$T timeout 600 bash tools/pmc_roofline.sh $pr 2>&1 | tail -9; done ;;
hbm)        timeout 300 python tools/hbm_bench.py > $O/${T}_hbm_kernels.txt 2>&1; cat $O/${T}_hbm_kernels.txt ;;
lab)        timeout 300 python tools/launch_lab.py > $O/${T}_launch_lab.txt 2>&1; cat $O/${T}_launch_lab.txt ;;
grouped)    timeout 300 python tools/grouped_lab.py > $O/${T}_grouped_lab.txt 2>&1; cat $O/${T}_grouped_lab.txt ;;
tune)       TF_RETUNE=${TF_RETUNE:-0} timeout 900 python tools/tune.py $O/mi355x_$T.txt 10 256,160 ${TUNE_PREC:-fp32} 2>&1 | tail -6 ;;
census)     timeout 300 python tools/census.py 10 256 ${CENSUS_PREC:-fp32} > $O/${T}_census_${CENSUS_PREC:-fp32}.txt 2>&1; head -${CENSUS_HEAD:-70} $O/${T}_census_${CENSUS_PREC:-fp32}.txt ;;
cmd)        bash -c "$CMD" ;;
esac
done
