#!/usr/bin/env python
"""Summary of tools/pmc_hbm.sh: per bandwidth-bound case, the HBM-side bytes the counters saw per call vs the algorithmic bytes.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
reports exactly half of the bytes of a wide (16 B / lane) coalesced streaming read -> read bytes = 2 x FETCH_SIZE; WRITE_SIZE is used as
reported (the guide calls it uncalibrated: the AdamW case - 12 B written per parameter, a known byte count - is printed as its calibration
row).  Both counters sit on the L2's fabric side, so Infinity-Cache hits are counted: for working sets under ~200 MB "traffic" is fabric
traffic, an upper bound of the HBM traffic.  Durations are the kernel-trace timestamps of the same (profiled) pass.
python tools/pmc_hbm_summary.py <dir> <output prefix>"""
import csv, glob, json, os, sys
d, out = sys.argv[1], sys.argv[2]
MARK = "sigmoid_kernel"


def load(tag):
    cc = glob.glob(os.path.join(d, "**", "%s_counter_collection.csv" % tag), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "%s_kernel_trace.csv" % tag), recursive=True)
    if not cc or not kt:
        return None
    trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(kt[0]))}
    rows = []
    for r in csv.DictReader(open(cc[0])):
        if r["Counter_Name"] != tag:
            continue
        k = trace.get(r["Dispatch_Id"])
        if k is None:
            continue
        rows.append((int(k["Start_Timestamp"]), r["Kernel_Name"], float(r["Counter_Value"]), (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e3))
    rows.sort()
    cases, cur = [], None
    for _, name, val, us in rows:
        if MARK in name:
            cur = []
            cases.append(cur)
        elif cur is not None:
            cur.append((name, val, us))
    return cases[0::2]        # marker pairs: [case 0][set-up of case 1][case 1]...


res, lines = [], []
fetch, write = load("FETCH_SIZE"), load("WRITE_SIZE")
meta = json.load(open(os.path.join(d, "cases_FETCH_SIZE.json")))
lines.append("# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python tools/hbm_bench.py --pmc   (tools/pmc_hbm.sh)")
lines.append("# per CALL of the case (all kernels of the op): read = 2 x FETCH_SIZE (gfx950 correction), written = WRITE_SIZE; kernel time = sum of the "
             "op's kernel durations in the FETCH pass (profiled passes run at a lower clock than un-profiled ones)")
lines.append("%-74s %9s %9s %9s %7s %9s %8s %8s" % ("# case", "algo MB", "read MB", "write MB", "ratio", "kern us", "TB/s", "of 8TB/s"))
for i, m in enumerate(meta):
    if fetch is None or write is None or i >= len(fetch) or i >= len(write):
        break
    calls = m["calls"]
    rd = 2 * sum(v for _, v, _ in fetch[i]) * 1024 / calls
    wr = sum(v for _, v, _ in write[i]) * 1024 / calls
    us = sum(u for _, _, u in fetch[i]) / calls
    kern = sorted(set(n.split("(")[0][-60:] for n, _, _ in fetch[i]))
    tr = rd + wr
    algo = m["algorithmic_bytes"]
    res.append(dict(case=m["name"], algorithmic_bytes=algo, read_bytes=rd, written_bytes=wr, traffic_over_algorithmic=tr / algo, kernel_us_per_call=us,
                    event_us_per_call=m["event_us"], achieved_tbps_algorithmic=algo / us / 1e6, kernels=kern, launches_per_call=len(fetch[i]) / calls))
    lines.append("%-74s %9.2f %9.2f %9.2f %7.2f %9.1f %8.2f %7.1f%%" % (m["name"], algo / 1e6, rd / 1e6, wr / 1e6, tr / algo, us, algo / us / 1e6, 100 * algo / us / 8e6))
open(out + ".txt", "w").write("\n".join(lines) + "\n")
json.dump(res, open(out + ".json", "w"), indent=1)
print("\n".join(lines))
