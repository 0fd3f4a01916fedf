#!/bin/bash
# One GPU-box pass of round-3 evidence (run from the repo root through gpurun); every step bounded by its own timeout.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
bench_line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'], '; GPT4 fc1', d['roofline'].get('avg_launch_us'), 'us; engine', d['roofline']['engine_ms_per_step'], 'ms')"; }
for what in "$@"; do
case $what in
ktests)     timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x ${KARGS:+-k "$KARGS"} 2>&1 | grep -v "Warning\|warn" | tail -15 ;;
mtests)     timeout 2400 python -m pytest tests/test_model_gpu.py -v -x ${MARGS:+-k "$MARGS"} > $O/r03_mtests.log 2>&1; grep -E "PASSED|FAILED|ERROR|passed|failed|Error|error:" $O/r03_mtests.log | head -60 ;;
tests_all)  timeout 2700 python -m pytest tests -q -m gpu > $O/r03_gpu_tests.log 2>&1; tail -8 $O/r03_gpu_tests.log ;;
bench)      timeout 600 python bench.py --steps 20 --warmup 5 > $O/r03_bench_n1.json 2> $O/r03_bench_n1.err; tail -4 $O/r03_bench_n1.err; cat $O/r03_bench_n1.json ;;
bench_fast) timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt $BENCH_ARGS 2> $O/r03_bench_fast.err | tee $O/r03_bench_fast.json | bench_line fast ;;
ab_bn)      for v in 1 0 1 0; do TF_FUSE_BN_STATS=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt 2>/dev/null | bench_line "TF_FUSE_BN_STATS=$v"; done ;;
ab)         for v in $AB_VALUES; do env $AB_VAR=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt $BENCH_ARGS 2>/dev/null | bench_line "$AB_VAR=$v"; done ;;
trace)      (cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace_r03 -o step --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-alt $BENCH_ARGS > $O/r03_trace_bench.log 2>&1)
            python tools/trace_csv_stats.py $O/trace_r03 > $O/r03_kernel_trace_graph${TRACE_TAG}.txt 2>&1; head -60 $O/r03_kernel_trace_graph${TRACE_TAG}.txt
            cp $O/trace_r03/*kernel_stats.csv $O/r03_kernel_stats${TRACE_TAG}.csv 2>/dev/null; rm -rf $O/trace_r03 ;;
check)      timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --check 2>&1 | tail -6 ;;
pmc_hbm)    timeout 600 bash tools/pmc_hbm.sh r03 2>&1 | tail -24 ;;
pmc)        timeout 600 bash tools/pmc_roofline.sh fp32 2>&1 | tail -12 ;;
hbm)        timeout 300 python tools/hbm_bench.py > $O/r03_hbm_kernels.txt 2>&1; cat $O/r03_hbm_kernels.txt ;;
tune)       TF_RETUNE=${TF_RETUNE:-0} timeout 900 python tools/tune.py $O/mi355x_r03.txt 10 256,160 ${TUNE_PREC:-fp32} 2>&1 | tail -6 ;;
census)     timeout 300 python tools/census.py 10 256 ${CENSUS_PREC:-fp32} > $O/r03_census_${CENSUS_PREC:-fp32}.txt 2>&1; head -70 $O/r03_census_${CENSUS_PREC:-fp32}.txt ;;
cmd)        bash -c "$CMD" ;;
esac
done
