#!/bin/bash
# Round 6, call 30: HIP runtime knobs around the captured step (same lease, two rounds each)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R; mkdir -p $O
bl() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms/step', d['value'], 'samples/s; loss', d['config']['final_loss'])"; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt"
for rep in 1 2; do
  timeout 200 $B 2>/dev/null | bl "default                              "
  HIP_FORCE_DEV_KERNARG=1 timeout 200 $B 2>/dev/null | bl "HIP_FORCE_DEV_KERNARG=1              "
  HIP_FORCE_DEV_KERNARG=0 timeout 200 $B 2>/dev/null | bl "HIP_FORCE_DEV_KERNARG=0              "
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 timeout 200 $B 2>/dev/null | bl "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1     "
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 200 $B 2>/dev/null | bl "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0     "
  GPU_MAX_HW_QUEUES=8 timeout 200 $B 2>/dev/null | bl "GPU_MAX_HW_QUEUES=8                  "
  GPU_MAX_HW_QUEUES=2 timeout 200 $B 2>/dev/null | bl "GPU_MAX_HW_QUEUES=2                  "
  DEBUG_HIP_GRAPH_DOT_PRINT=0 HIP_GRAPH_MAX_STREAMS=8 timeout 200 $B 2>/dev/null | bl "HIP_GRAPH_MAX_STREAMS=8 (if it exists)"
done
env | grep -i "^HIP_\|^HSA_\|^GPU_\|^AMD_\|^ROC" | head
