#!/bin/bash
# PMC recipe for bench.py's roofline.traffic (MI355X_MICROARCH.md, HBM / rocprofv3 section): the dominant kernel (GPT-4 mlp.0 forward GEMM,
# tuned plan) is profiled in SEPARATE counter passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2), --kernel-trace only (no sys/hip traces),
# and summarised by tools/pmc_summary.py into gpurun_out/<TAG>_pmc_gemm_roofline.{txt,json} (TAG default r03; copy to profiles/).  Run on the GPU box from the repo root:
#   bash tools/pmc_roofline.sh [fp32|f32x3] (writes gpurun_out/pmc_<TAG>[_f32x3]/ + gpurun_out/<TAG>_pmc_gemm_roofline[_f32x3].{txt,json})
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PREC=${1:-fp32}
SUF=""; [ "$PREC" != "fp32" ] && SUF="_$PREC"
TAG=${TAG:-r03}
OUT=$R/gpurun_out/pmc_$TAG$SUF
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES"; do
    tag=$(echo "$pass" | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $pass --kernel-trace -d "$OUT" -o "$tag" --output-format csv -- python "$R/tools/gemm_tuned.py" 1740 6048 1512 nt 10 $PREC > "$OUT/$tag.log" 2>&1
done
python "$R/tools/pmc_summary.py" "$OUT" "$R/gpurun_out/${TAG}_pmc_gemm_roofline$SUF" $PREC
