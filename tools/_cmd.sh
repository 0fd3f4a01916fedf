cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "consumer or grouped" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "folded or single_block" 2>&1 | tail -2
for rep in 1 2; do
 TF_FUSE_SE_BN_BWD=0 timeout 200 $B 2>/dev/null | bl "se bwd separate"
 timeout 200 $B 2>/dev/null | bl "se bwd folded  "
done
