#!/bin/bash
# Round 6, call 29: attention phase B with the wave's two column tiles multiplied together: parity tests, per-launch time
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention" > $O/att_tests.log 2>&1; tail -1 $O/att_tests.log
timeout 300 python tools/attention_lab.py 2>/dev/null | tee $O/r06_attention_phase_b_two_tiles.txt
for d in 1 2; do echo "== TF_ATT_DBG=$d"; TF_ATT_DBG=$d timeout 300 python tools/attention_lab.py 2>/dev/null | grep "C ="; done | tee -a $O/r06_attention_phase_b_two_tiles.txt
