#!/bin/bash
# Round 6, call 28: phase ablation of the fused attention kernels (TF_ATT_DBG: 1 no phase A, 2 no phase B; 3 = launch + score stores + softmax / dS middle only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
for rl in 0 1; do for d in 0 1 2 3; do echo "== TF_ATT_ROWLDS=$rl TF_ATT_DBG=$d"; TF_ATT_ROWLDS=$rl TF_ATT_DBG=$d timeout 300 python tools/attention_lab.py 2>/dev/null | grep "C ="; done; done | tee $O/r06_attention_phase_ablation.txt
