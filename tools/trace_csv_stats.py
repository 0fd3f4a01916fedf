#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV of bench.py (hipGraph replay): ms per training step, launches per step, average us,
for the LAST full steps (delimited by adamw_kernel launches), plus GPU-busy (union of kernel intervals) vs wall.  python tools/trace_csv_stats.py <dir>"""
import collections, csv, glob, json, os, sys
if len(sys.argv) > 2:       # the traced bench.py run's stdout: its JSON line names the library that was traced (bench.py only trusts a summary of ITS build)
    try:
        line = [l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1]
        print("# build_id %s" % json.loads(line)["config"]["build_id"])
    except Exception:
        pass
f = glob.glob(os.path.join(sys.argv[1], "*kernel_trace.csv"))[0]
rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))), key=lambda r: r[1])
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
print("kernels", len(rows), "adamw launches", len(ad))
n = min(4, len(ad) - 2)
a, b = ad[-1 - n], ad[-1]
seg = rows[a + 1:b + 1]
wall = seg[-1][2] - rows[a][2]
ev = sorted((r[1], r[2]) for r in seg)
busy, cs, ce = 0, ev[0][0], ev[0][1]
for s, e in ev[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
tot = sum(r[2] - r[1] for r in seg)
print("last %d steps: wall %.2f ms/step, GPU busy (union) %.2f ms/step, sum of kernel durations %.2f ms/step, %d kernels/step" % (n, wall / n / 1e6, busy / n / 1e6, tot / n / 1e6, len(seg) // n))
agg = collections.defaultdict(lambda: [0, 0, 1 << 60, 0])
for name, s, e in seg:
    a = agg[name]
    a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
fam = collections.defaultdict(float)
for name, (c, t, _, _) in agg.items():
    key = "gemm engine (gemm_kernel / gemm_pair_kernel / gemm_dma_kernel)" if "gemm_kernel" in name or "gemm_dma" in name or "gemm_pair" in name else "direct conv (conv3x3_grouped / conv3x3_small / conv3x3_thin)" if ("conv3x3_small" in name or "conv3x3_grouped" in name or "conv3x3_thin" in name) else \
        "batchnorm" if ("bn_" in name or "BnStat" in name or "BnBwd" in name) else "other"
    fam[key] += t / n / 1e6
print("families (ms/step):", ", ".join("%s %.2f" % kv for kv in sorted(fam.items(), key=lambda kv: -kv[1])))
print("per kernel (ms/step, calls/step, avg us, min us, max us):")
for name, (c, t, lo, hi) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
    print("%9.3f %6d %8.1f %7.1f %7.1f  %s" % (t / n / 1e6, c // n, t / c / 1e3, lo / 1e3, hi / 1e3, name[:130]))
