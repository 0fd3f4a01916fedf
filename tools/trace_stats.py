#!/usr/bin/env python
"""Summarises a rocprofv3 --kernel-trace sqlite db: per-kernel totals, and for the LAST n steps the GPU busy time (union of
kernel intervals) vs wall time, i.e. how much of a hipGraph-replayed step is idle gaps.  python tools/trace_stats.py db [top]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
# steps are delimited by adamw_kernel launches
ad = [i for i, r in enumerate(rows) if 'adamw_kernel' in r[0]]
print("kernels", len(rows), "adamw launches", len(ad))
if len(ad) >= 3:
    a, b = ad[-3], ad[-1]   # two full steps
    seg = rows[a + 1:b + 1]
    wall = seg[-1][2] - rows[a][2]
    ev = sorted((r[1], r[2]) for r in seg)
    busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(r[2] - r[1] for r in seg)
    print("last 2 steps: wall %.2f ms/step, GPU busy (union) %.2f ms/step, sum of kernel durations %.2f ms/step, %d kernels/step" % (wall / 2e6, busy / 2e6, tot / 2e6, len(seg) // 2))
    gaps = collections.Counter()
    prev_e, prev_n = None, None
    cur_e = None
    for n, s, e, q in seg:
        if cur_e is not None and s > cur_e:
            gaps[prev_n[:60]] += s - cur_e
        if cur_e is None or e > cur_e: cur_e, prev_n = e, n
    print("idle time following kernel (top 12, ms/step):")
    for n, g in gaps.most_common(12): print("   %.3f  %s" % (g / 2e6, n))
    agg = collections.defaultdict(lambda: [0, 0])
    for n, s, e, q in seg:
        agg[n][0] += 1; agg[n][1] += e - s
    print("per kernel (ms/step, calls/step, avg us):")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("  %7.3f %5d %8.1f  %s" % (t / 2e6, c // 2, t / c / 1e3, n[:130]))
