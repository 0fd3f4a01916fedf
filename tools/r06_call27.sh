#!/bin/bash
# Round 6, call 27: attention phase A with the row operand staged in LDS (TF_ATT_ROWLDS) vs fragments from global memory: tests, per-launch time, step A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention" > $O/att_tests.log 2>&1; tail -1 $O/att_tests.log
for f in 0 1; do echo "== TF_ATT_ROWLDS=$f"; TF_ATT_ROWLDS=$f timeout 300 python tools/attention_lab.py 2>/dev/null; done | tee $O/r06_attention_lab_row_lds.txt
for rep in 1 2 3; do
  TF_ATT_ROWLDS=0 TF_ATT_BWD_ONE=0 timeout 200 $B 2>/dev/null | bl "fp32 attention: rows from global, two backward launches"
  TF_ATT_ROWLDS=1 TF_ATT_BWD_ONE=0 timeout 200 $B 2>/dev/null | bl "fp32 attention: rows in LDS, two backward launches    "
  TF_ATT_ROWLDS=1 TF_ATT_BWD_ONE=1 timeout 200 $B 2>/dev/null | bl "fp32 attention: rows in LDS, one backward grid        "
done
