#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -q -x -k "pillar" 2>&1 | grep -v "Warning\|warn(" | tail -6
timeout 300 python - <<'PY' 2>&1 | tail -6
import sys, torch, time
sys.path.insert(0, "tests")
from transfuser_amd import ops
from transfuser_amd.data import synthetic_cloud
pts = torch.from_numpy(synthetic_cloud(10, 40000, 0)).cuda(); num = torch.full((10,), 40000, dtype=torch.int32).cuda()
def t_eager(static):
    for _ in range(3): ops.pillar_index(pts, num, -16, 32, -32, 32, 4 if False else 8, static=static) if False else ops.pillar_index(pts, num, -16, 16, -32, 0, 8, static=static)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.pillar_index(pts, num, -16, 16, -32, 0, 8, static=static)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 20 * 1e6
print("pillar_index 10 x 40000 points: eager with host read %.1f us, static eager %.1f us" % (t_eager(False), t_eager(True)))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): ops.pillar_index(pts, num, -16, 16, -32, 0, 8, static=True)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20): ix = ops.pillar_index(pts, num, -16, 16, -32, 0, 8, static=True)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); g.replay(); e1.record(s); e1.synchronize()
print("pillar_index static, hipGraph replay of 20 calls: %.1f us per call; totals %s" % (e0.elapsed_time(e1) * 1e3 / 20, ix["totals"].tolist()))
PY
