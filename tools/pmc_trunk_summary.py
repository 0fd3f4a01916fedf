#!/usr/bin/env python
"""Summary of the counter passes of tools/pmc_trunk.sh: per case (cut at the marker launches of tools/pmc_trunk.py) the per-launch kernel time,
TFLOP/s, fabric-side traffic 2 x FETCH_SIZE + WRITE_SIZE (KB counters; gfx950: FETCH_SIZE under-reports wide coalesced reads 2x,
MI355X_MICROARCH.md) against the algorithmic bytes, MFMA-busy fraction and effective clock.  python tools/pmc_trunk_summary.py <dir> <out prefix>"""
import csv, glob, json, os, subprocess, sys
d, out = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
cases = json.loads(subprocess.run([sys.executable, os.path.join(here, "pmc_trunk.py"), "list"], capture_output=True, text=True).stdout)
per = [dict() for _ in cases]       # case -> counter -> [sum of values per launch-group..], plus durations
for cc in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    tag = os.path.basename(cc)[:-len("_counter_collection.csv")]
    kt = sorted(csv.DictReader(open(os.path.join(d, tag + "_kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
    seg, idx = {}, -1
    for r in kt:        # dispatch id -> case index (markers: the cos_ launches)
        if "cos_kernel" in r["Kernel_Name"]:
            idx += 1
            continue
        seg[r["Dispatch_Id"]] = idx
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in kt}
    seen = set()
    for r in csv.DictReader(open(cc)):
        i = seg.get(r["Dispatch_Id"], -1)
        if i < 0 or i >= len(cases):
            continue
        c = per[i].setdefault(r["Counter_Name"], dict(val=0.0, us=0.0, main_us=0.0, n=0, kernels=set()))
        c["val"] += float(r["Counter_Value"])
        key = (r["Counter_Name"], r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            c["us"] += dur[r["Dispatch_Id"]]
            if cases[i]["match"] in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"] and "fixup" not in r["Kernel_Name"]:
                c["main_us"] += dur[r["Dispatch_Id"]]; c["n"] += 1
            c["kernels"].add(r["Kernel_Name"].split("(")[0][-90:])
sys.path.insert(0, here)
from pmc_trunk import ITERS
lines = ["# rocprofv3 --pmc <one pass per counter group> --kernel-trace -- python tools/pmc_trunk.py  (tools/pmc_trunk.sh); shipped plans, fp32 MFMA",
         "# per call (= all kernels one call launches: split-K partials / reduce included), averages over %d calls; peak 157.3 TF/s" % ITERS,
         "%-34s %8s %8s %7s %9s %9s %7s %6s %6s  %s" % ("case", "us/call", "TFLOP/s", "frac", "algo MB", "fabric MB", "x algo", "MFMA", "GHz", "kernels")]
res = []
for cs, p in zip(cases, per):
    g = lambda k: p.get(k)
    us = (g("GRBM_GUI_ACTIVE") or g("FETCH_SIZE") or dict(us=0))["us"] / ITERS
    row = dict(case=cs["name"], us_per_call=us, tflops=cs["flops"] / us / 1e6 if us else None, algorithmic_bytes=cs["bytes"])
    if g("FETCH_SIZE") and g("WRITE_SIZE"):
        row["traffic_bytes"] = (2 * g("FETCH_SIZE")["val"] + g("WRITE_SIZE")["val"]) * 1024 / ITERS
    if g("GRBM_GUI_ACTIVE") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        ga = g("GRBM_GUI_ACTIVE")
        row["clock_ghz"] = ga["val"] / 8 / (ga["us"] * 1e3)
        row["mfma_busy_frac"] = g("SQ_VALU_MFMA_BUSY_CYCLES")["val"] / (ga["val"] / 8 * 1024)
    row["kernels"] = sorted(set().union(*[c["kernels"] for c in p.values()])) if p else []
    res.append(row)
    f = lambda v, fmt: (fmt % v) if v is not None else "-"
    lines.append("%-34s %8.1f %8s %7s %9.1f %9s %7s %6s %6s  %s" % (
        cs["name"], us, f(row["tflops"], "%.1f"), f(row["tflops"] / 157.3 if row["tflops"] else None, "%.3f"), cs["bytes"] / 1e6,
        f(row.get("traffic_bytes") and row["traffic_bytes"] / 1e6, "%.1f"), f(row.get("traffic_bytes") and row["traffic_bytes"] / cs["bytes"], "%.2f"),
        f(row.get("mfma_busy_frac"), "%.2f"), f(row.get("clock_ghz"), "%.2f"), "; ".join(k.split("::")[-1][:48] for k in row["kernels"])))
open(out + ".txt", "w").write("\n".join(lines) + "\n")
json.dump(res, open(out + ".json", "w"), indent=1)
print("\n".join(lines))
