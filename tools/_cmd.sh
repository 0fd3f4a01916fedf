cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_dataprep.py -q -x -k "hist or prep or corr" 2>&1 | tail -2
timeout 300 python tools/hbm_bench.py 2>&1 | grep -i "H1\|H2\|pillar" | head
