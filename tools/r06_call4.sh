#!/bin/bash
# Round 6, call 4: operand-read pipelining in the grouped forward / input-gradient kernel; H1 slab kernel with 16 loads in flight
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "grouped_conv or bench_shape_convs or lidar_hist" 2>&1 | tail -3
timeout 300 python tools/grouped_lab.py 2>&1 | grep -v Warn > $O/r06b_grouped_lab.txt; cat $O/r06b_grouped_lab.txt
for v in 0 1; do echo "== TF_HIST_SLAB=$v"; TF_HIST_SLAB=$v timeout 200 python tools/hbm_bench.py 2>&1 | grep -i "H1"; done
for rep in 1 2 3; do
  timeout 200 $B 2>/dev/null | bl "fp32 head"
done
