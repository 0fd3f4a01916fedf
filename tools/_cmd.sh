cd $GRAFT_REPO_ROOT
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "se_" 2>&1 | tail -2
for rep in 1 2; do
 TF_HIP_LIB=$PWD/transfuser_amd/libtransfuser_hip_prev.so timeout 200 $B 2>/dev/null | bl "prev"
 timeout 200 $B 2>/dev/null | bl "new "
done
for a in bn_apply_kernel se_bwd_apply bn_bwd_finalize bn_fwd_finalize splitk_fixup se_excite dropout_kernel layernorm; do TF_ABLATE=$a timeout 200 $B 2>/dev/null | bl "ablate $a"; done
for e in AMD_OPT_FLUSH=0 DEBUG_HIP_FORCE_GRAPH_QUEUES=2 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 GPU_MAX_HW_QUEUES=8 GPU_MAX_HW_QUEUES=2 DEBUG_HIP_GRAPH_BATCH_SIZE=64 DEBUG_CLR_MAX_BATCH_SIZE=256 AMD_DIRECT_DISPATCH=0; do env $e timeout 200 $B 2>/dev/null | bl "env $e"; done
timeout 200 $B 2>/dev/null | bl "new "
