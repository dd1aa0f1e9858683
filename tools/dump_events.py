#!/usr/bin/env python3
"""dump (start_us, end_us, name) of the last `window_ms` of a rocprofv3 rocpd database as CSV (kernels + memory copies)
    python tools/dump_events.py <results.db> <window_ms> > events.csv"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); win = float(sys.argv[2]) * 1e6
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
extra = [c for c in cols if c in ("queue_id", "stream_id")]
ev = [(s, e, n.split("(")[0][:40], q) for n, s, e, q in cur.execute("select %s, start, end, %s from kernels" % (name_col, extra[0] if extra else "0"))]
if "memory_copies" in tables:
    ev += [(s, e, "<copy>", -1) for s, e in cur.execute("select start, end from memory_copies")]
t1 = max(e for _, e, _, _ in ev)
ev = sorted(x for x in ev if x[1] >= t1 - win)
t0 = ev[0][0]
for s, e, n, q in ev:
    print("%.1f,%.1f,%s,%s" % ((s - t0) / 1e3, (e - t0) / 1e3, n, q))
