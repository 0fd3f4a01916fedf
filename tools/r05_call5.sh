#!/bin/bash
# Round 5, fifth lease: the reworked bench line end to end (default invocation, as the driver runs it), the multi-GPU path on one GPU, the RCCL single-rank test.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "rccl_single_rank" 2>&1 | tail -3
timeout 300 python bench.py --force-pieces 8 --check --no-cpu-baseline --no-alt > $O/r05_force_pieces.json 2> $O/r05_force_pieces.err; tail -3 $O/r05_force_pieces.err; cut -c1-600 $O/r05_force_pieces.json
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_default.json 2> $O/r05_bench_default.err ) 2>&1 | tail -3
tail -12 $O/r05_bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
r = d["roofline"]
print("kernel:", r["kernel"][:200]); print("achieved", r["achieved"], "frac", r["frac"], "traffic", r.get("traffic"))
for row in r["top5"]: print("  ", row)
print("gpt4:", {k: v for k, v in r.get("gpt4_mlp0", {}).items() if k != "kernel"})
print("engine_graph", r.get("engine_graph"))
print("f32x3", d.get("f32x3", {}).get("ms_per_step"))
for k, v in d.get("configs", {}).items(): print(k, {kk: vv for kk, vv in v.items() if kk in ("value", "ms_per_step", "error", "roofline_frac")})
print("multi", d.get("multi_gpu_path_on_one_gpu"))
print("cpu", d.get("cpu_baseline", {}).get("value"))
PY
