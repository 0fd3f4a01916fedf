cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step; loss', d['config']['final_loss'], 'engine', d['roofline']['engine_ms_per_step'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --dtype f32x3"
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "f32x3" 2>&1 | tail -2
TF_HIP_LIB=$PWD/transfuser_amd/libtransfuser_hip_prev.so timeout 200 $B 2>/dev/null | bl "prev lib, old plans"
timeout 200 $B 2>/dev/null | bl "new lib, old plans "
awk -F';' '/^#/ || ($6 < 8 || $6 > 11)' transfuser_amd/plans/mi355x.txt > /tmp/plans_no_x3.txt; wc -l /tmp/plans_no_x3.txt
TF_PLANS=/tmp/plans_no_x3.txt TF_RETUNE=0 timeout 600 python tools/tune.py $O/mi355x_r04_x3.txt 10 256,160 f32x3 2>&1 | tail -4
wc -l $O/mi355x_r04_x3.txt
TF_PLANS=$PWD/$O/mi355x_r04_x3.txt timeout 200 $B 2>/dev/null | bl "new lib, new plans "
TF_PLANS=$PWD/$O/mi355x_r04_x3.txt timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt 2>/dev/null | bl "fp32 with the new plan file"
TF_PLANS=$PWD/$O/mi355x_r04_x3.txt timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "f32x3" 2>&1 | tail -2
