#!/usr/bin/env python
"""Who owns the wall time of a replayed training step?  From a rocprofv3 --kernel-trace CSV of bench.py: the last step (between two adamw_kernel
launches) is cut at every kernel start / end; each slice of wall time is charged to the kernels running in it (1 / n each when n overlap) and
to "idle" when none is.  Per kernel name: charged wall ms, the part of it spent running ALONE (nothing else on the chip: shortening the kernel
shortens the step one for one), launches.  Also per queue: busy time.   python tools/trace_timeline.py <dir> [steps]"""
import collections, csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "*kernel_trace.csv"))[0]
rd = list(csv.DictReader(open(f)))
qk = "Queue_Id" if "Queue_Id" in rd[0] else None
rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get(qk, "0") if qk else "0") for r in rd), key=lambda r: r[1])
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
n = min(int(sys.argv[2]) if len(sys.argv) > 2 else 3, len(ad) - 2)
a, b = ad[-1 - n], ad[-1]
seg = rows[a + 1:b + 1]
t0, t1 = rows[a][2], seg[-1][2]
ev = []
for i, (name, s, e, q) in enumerate(seg):
    ev.append((s, 1, i)); ev.append((e, 0, i))
ev.sort()
charged = collections.defaultdict(float); solo = collections.defaultdict(float); cnt = collections.Counter(); dur = collections.defaultdict(float)
for name, s, e, q in seg:
    cnt[name] += 1; dur[name] += e - s
active = set(); last = t0; idle = 0.0; hist = collections.defaultdict(float)
for t, kind, i in ev:
    dt = t - last
    if dt > 0:
        k = len(active)
        hist[min(k, 4)] += dt
        if k == 0:
            idle += dt
        else:
            for j in active:
                charged[seg[j][0]] += dt / k
            if k == 1:
                solo[seg[next(iter(active))][0]] += dt
    last = t
    if kind: active.add(i)
    else: active.discard(i)
wall = (t1 - t0) / n / 1e6
print("last %d steps: wall %.2f ms/step; idle %.2f ms; time with 1 / 2 / 3 / >=4 kernels running: %s ms" %
      (n, wall, idle / n / 1e6, " / ".join("%.2f" % (hist[k] / n / 1e6) for k in (1, 2, 3, 4))))
qb = collections.defaultdict(float)
for name, s, e, q in seg: qb[q] += e - s
print("per queue busy ms/step:", ", ".join("%s: %.2f" % (q, v / n / 1e6) for q, v in sorted(qb.items(), key=lambda kv: -kv[1])))
fam = lambda name: ("gemm engine" if "gemm_kernel" in name or "gemm_dma" in name or "gemm_pair" in name else "grouped conv" if "conv3x3_grouped" in name else "direct conv" if ("conv3x3_small" in name or "conv3x3_thin" in name or "stem_direct" in name) else
                    "batchnorm" if ("bn_" in name or "BnStat" in name or "BnBwd" in name or "BnRelu" in name) else "attention" if "attention" in name else "se" if "se_" in name or "SeGate" in name else "adamw" if "adamw" in name else "other")
fc = collections.defaultdict(float); fs = collections.defaultdict(float)
for name in charged: fc[fam(name)] += charged[name]; fs[fam(name)] += solo[name]
print("families: charged wall ms (of which alone):", ", ".join("%s %.2f (%.2f)" % (k, v / n / 1e6, fs[k] / n / 1e6) for k, v in sorted(fc.items(), key=lambda kv: -kv[1])))
print("per kernel: charged wall ms/step, alone ms/step, sum of durations ms/step, launches/step, name")
for name, c in sorted(charged.items(), key=lambda kv: -kv[1])[:70]:
    print("%8.3f %8.3f %8.3f %5d  %s" % (c / n / 1e6, solo[name] / n / 1e6, dur[name] / n / 1e6, cnt[name] // n, name[:140]))
# per queue: the kernels that make up its busy time (the queue with the most busy time is the step's critical path: kernel time saved on the others is hidden)
# and the GAPS between consecutive kernels of the busiest queue (dependent-launch boundaries + waits for the other queues)
for q, v in sorted(qb.items(), key=lambda kv: -kv[1])[:3]:
    per = collections.defaultdict(float); pc = collections.Counter()
    ks = sorted((s, e, name) for name, s, e, qq in seg if qq == q)
    for s, e, name in ks:
        per[name] += e - s; pc[name] += 1
    gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
    small = sum(g for g in gaps if 0 < g < 5000); big = sum(g for g in gaps if g >= 5000)
    print("queue %s: %.2f ms busy, %d launches / step; gaps between its consecutive kernels: %.2f ms in %d gaps < 5 us (boundaries), %.2f ms in %d longer ones (waits)" %
          (q, v / n / 1e6, len(ks) // n, small / n / 1e6, sum(1 for g in gaps if 0 < g < 5000) // n, big / n / 1e6, sum(1 for g in gaps if g >= 5000) // n))
    for name, t in sorted(per.items(), key=lambda kv: -kv[1])[:25]:
        print("    %8.3f ms %5d  %s" % (t / n / 1e6, pc[name] // n, name[:130]))
