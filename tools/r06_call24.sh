#!/bin/bash
# Round 6, call 24: where the pillar slab kernel spends its time (TF_PL_DBG: 1 no cell computation / queueing, 2 no accumulation, 4 no output phase, 8 no key writes / counts)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
cat > /tmp/pl_lab.py <<'PY'
import torch, sys, ctypes
sys.path.insert(0, '.')
from transfuser_amd import ops
from transfuser_amd.ops import L, ptr, check, stream_of
from transfuser_amd.data import synthetic_cloud
dev = 'cuda'
pts = torch.from_numpy(synthetic_cloud(10, 32768, 0)).to(dev)
raw = torch.zeros(10, 40000, 4, device=dev); raw[:, :32768] = pts
num = torch.full((10,), 32768, dtype=torch.int32, device=dev)
B, Nmax, GX, GY = 10, 40000, 257, 257
L().tf_pillar_padded_cells.restype = ctypes.c_long
CP = L().tf_pillar_padded_cells(GX, GY)
i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)
keys, bitmap, blockcnt = i32(B * Nmax), i32(B * CP // 32), i32(B * 40)
cellsums = torch.empty(B * CP, 4, dtype=torch.int64, device=dev)
f = ctypes.c_float
def run():
    check(L().tf_pillar_mark_f32(ptr(raw), ptr(num), B, Nmax, 4, f(-16), f(16), f(-32), f(0), f(8), GX, GY, ptr(keys), ptr(bitmap), ptr(cellsums), ptr(blockcnt),
                                 stream_of(raw)), "mark")
for _ in range(3): run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): run()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): g.replay()
e1.record(); e1.synchronize()
print("pillar slab kernel, in-graph: %.1f us / launch" % (e0.elapsed_time(e1) * 1e3 / 200))
PY
for d in 0 1 2 4 8 3 7 15; do echo -n "TF_PL_DBG=$d  "; TF_PL_DBG=$d timeout 120 python /tmp/pl_lab.py 2>/dev/null | tail -1; done
