#!/bin/bash
# Counter passes for the trunk contractions (tools/pmc_trunk.py), SEPARATE passes per counter group, --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 section).
# bash tools/pmc_trunk.sh   (TAG default r05) -> gpurun_out/<TAG>_pmc_trunk.{txt,json}
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${TAG:-r05}
OUT=$R/gpurun_out/pmc_trunk_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo "$pass" | tr ' ' '_')
    timeout 240 rocprofv3 --pmc $pass --kernel-trace -d "$OUT" -o "$tag" --output-format csv -- python "$R/tools/pmc_trunk.py" > "$OUT/$tag.log" 2>&1
done
python "$R/tools/pmc_trunk_summary.py" "$OUT" "$R/gpurun_out/${TAG}_pmc_trunk"
rm -rf "$OUT"/*.log
