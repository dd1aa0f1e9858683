#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7 rocpd sqlite) kernel trace into the --stats style table:
    python tools/rocpd_stats.py gpurun_out/prof/.../*_results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
agg = {}
for n, s, e in rows:
    a = agg.setdefault(n, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values()) or 1
print("%-70s %7s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    short = n if len(n) <= 70 else n[:67] + "..."
    print("%-70s %7d %12.1f %12.2f %12.2f %12.2f %6.2f%%" % (short, a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))
