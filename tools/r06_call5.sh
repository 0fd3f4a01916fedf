#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
for v in 0 1 2 3 4 7; do echo "== TF_GC_DBG=$v (1: no MFMA loop, 2: no epilogue stores, 4: no next-tile fetch)"; TF_GC_DBG=$v timeout 300 python tools/grouped_lab.py 2>&1 | grep -E "fwd  |dgrad" | grep -E "16, 44|64, 176|16, 16"; done
