#!/usr/bin/env python3
"""GPU timeline of the LAST `window_ms` of activity in a rocprofv3 rocpd database (the last proof of tools/prove_bench.py):
busy time (union of kernel / copy intervals), idle gaps, and the per-kernel totals inside the window.
    python tools/timeline.py <results.db> [window_ms]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 60e6
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
ev = [(s, e, n) for n, s, e in cur.execute("select %s, start, end from kernels" % name_col)]
if "memory_copies" in tables:
    mc = [r[1] for r in cur.execute("pragma table_info(memory_copies)")]
    if "start" in mc and "end" in mc:
        ev += [(s, e, "<copy>") for s, e in cur.execute("select start, end from memory_copies")]
t1 = max(e for _, e, _ in ev)
ev = sorted(x for x in ev if x[1] >= t1 - win)
t0 = ev[0][0]
busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
for s, e, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - t0
print("window %.2f ms: %d events, busy %.2f ms (%.1f %%), idle %.2f ms in %d gaps" % (span / 1e6, len(ev), busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps)))
for lo, hi in ((0, 5e3), (5e3, 20e3), (20e3, 50e3), (50e3, 200e3), (200e3, 1e12)):
    g = [x for x in gaps if lo <= x < hi]
    print("  gaps %6.0f..%-8.0f us: %5d, total %.2f ms" % (lo / 1e3, min(hi, 1e9) / 1e3, len(g), sum(g) / 1e6))
agg = {}
for s, e, n in ev:
    a = agg.setdefault(n.split("(")[0][:60], [0, 0])
    a[0] += 1; a[1] += e - s
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print("  %-60s %5d %9.2f ms" % (n, a[0], a[1] / 1e6))
