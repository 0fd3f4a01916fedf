#!/bin/bash
# Counter-based HBM evidence for the bandwidth-bound kernels (MI355X_MICROARCH.md "HBM" + "rocprofv3 PMC slots"): FETCH_SIZE and
# WRITE_SIZE do not fit one pass (3 + 2 of the 4 TCC slots), so each gets its own rocprofv3 run with --kernel-trace only (no sys / hip traces).
# tools/hbm_bench.py --pmc launches a marker kernel before every case; tools/pmc_hbm_summary.py cuts the dispatch list there, applies the
# guide's gfx950 correction (FETCH_SIZE x 2 for wide coalesced reads) and prints traffic / algorithmic bytes per case.
#   bash tools/pmc_hbm.sh [tag]     -> gpurun_out/<tag>_pmc_hbm.{txt,json}   (tag default r03)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03}
OUT=$R/gpurun_out/pmc_hbm_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $pass --kernel-trace -d "$OUT" -o "$pass" --output-format csv -- python "$R/tools/hbm_bench.py" --pmc --iters 8 --meta "$OUT/cases_$pass.json" > "$OUT/$pass.log" 2>&1
done
python "$R/tools/pmc_hbm_summary.py" "$OUT" "$R/gpurun_out/${TAG}_pmc_hbm"
