#!/bin/bash
# What bounds the fused attention kernels (csrc/attention.cpp)?  SQ / TA / TCP counters of tools/attention_lab.py --pmc (eager launches at the four GPT stages), one
# rocprofv3 pass per counter group, --kernel-trace only.   bash tools/pmc_attention.sh [tag]  -> gpurun_out/<tag>_pmc_attention.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r06}
OUT=$R/gpurun_out/pmc_attention_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
            "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL" "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAIT_ANY"; do
    i=$((i + 1))
    timeout 200 rocprofv3 --pmc $pass --kernel-trace -d "$OUT" -o "p$i" --output-format csv -- python "$R/tools/attention_lab.py" --pmc > "$OUT/p$i.log" 2>&1
done
python - "$OUT" > "$R/gpurun_out/${TAG}_pmc_attention.txt" <<'PY'
import csv, glob, os, re, sys, collections
d = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
durs = collections.defaultdict(list)
for cc in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    tag = os.path.basename(cc)[:-len("_counter_collection.csv")]
    kt = {r["Dispatch_Id"]: r for r in csv.DictReader(open(os.path.join(d, tag + "_kernel_trace.csv")))}
    seen = set()
    for r in csv.DictReader(open(cc)):
        n = r["Kernel_Name"]
        if "attention" not in n:
            continue
        key = (re.search(r"attention_\w+", n).group(0), r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
        vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if (tag, r["Dispatch_Id"]) not in seen:
            seen.add((tag, r["Dispatch_Id"]))
            k = kt[r["Dispatch_Id"]]
            durs[key].append((int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e3)
print("# rocprofv3 --pmc <group> --kernel-trace -- python tools/attention_lab.py --pmc : per-launch averages, eager launches (tools/pmc_attention.sh)")
for key in sorted(vals):
    print("\n%s grid %s: avg duration %.1f us over %d launches" % (key[0], key[1], sum(durs[key]) / len(durs[key]), len(durs[key])))
    for c in sorted(vals[key]):
        v = vals[key][c]
        print("    %-40s %16.1f" % (c, sum(v) / len(v)))
PY
cat "$R/gpurun_out/${TAG}_pmc_attention.txt" | head -150
