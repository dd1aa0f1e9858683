#!/usr/bin/env python3
"""Host-side view of the LAST `window_ms` of a rocprofv3 run made with --kernel-trace --marker-trace: the library opens a roctx range per
C-ABI call (csrc/common.hpp RoctxRange), so the `regions` view is the sequence of ezkl_hip_* calls the host prover made, with host
timestamps.  Prints (i) every call or gap between calls longer than `min_us`, with the GPU-busy fraction during it, and (ii) totals per
entry point: where the host waits, and where the GPU idles because the host is busy.
    python tools/hosttrace.py <results.db> [window_ms=100] [min_us=150]"""
import sqlite3, sys, bisect
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 100e6
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 150.0
cur = db.cursor()
k = [(s, e) for s, e in cur.execute("select start, end from kernels")]
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "memory_copies" in tables:
    k += [(s, e) for s, e in cur.execute("select start, end from memory_copies")]
k.sort()
t1 = max(e for _, e in k)
t0 = t1 - win
# union of GPU-busy intervals
iv = []
for s, e in k:
    if iv and s <= iv[-1][1]:
        iv[-1][1] = max(iv[-1][1], e)
    else:
        iv.append([s, e])
starts = [a for a, _ in iv]
def busy(a, b):
    if b <= a: return 0.0
    i = max(0, bisect.bisect_right(starts, a) - 1)
    tot = 0
    while i < len(iv) and iv[i][0] < b:
        tot += max(0, min(b, iv[i][1]) - max(a, iv[i][0]))
        i += 1
    return tot / (b - a)
import json as _json
def _label(name, ext):
    try:
        return _json.loads(ext).get("message") or name          # roctx ranges: the message is the C-ABI entry point
    except Exception:
        return name
regs = sorted((s, e, _label(n, x)) for n, s, e, x in cur.execute("select name, start, end, extdata from regions") if e >= t0)
# keep outermost ranges only (nested calls: batch entry points call others)
outer, last_end = [], 0
for s, e, n in regs:
    if s >= last_end:
        outer.append((s, e, n)); last_end = e
print("window: last %.1f ms; %d outer C-ABI calls" % (win / 1e6, len(outer)))
agg, prev_end = {}, None
for s, e, n in outer:
    if prev_end is not None and (s - prev_end) / 1e3 >= min_us:
        print("  %9.3f ms  %8.0f us  (host between calls)                      gpu busy %3.0f %%" % ((prev_end - t0) / 1e6, (s - prev_end) / 1e3, 100 * busy(prev_end, s)))
    if (e - s) / 1e3 >= min_us:
        print("  %9.3f ms  %8.0f us  %-44s gpu busy %3.0f %%" % ((s - t0) / 1e6, (e - s) / 1e3, n[:44], 100 * busy(s, e)))
    a = agg.setdefault(n, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += (e - s) / 1e3 * (1 - busy(s, e))
    if prev_end is not None:
        g = agg.setdefault("(host between calls)", [0, 0.0, 0.0])
        g[0] += 1; g[1] += (s - prev_end) / 1e3; g[2] += (s - prev_end) / 1e3 * (1 - busy(prev_end, s))
    prev_end = e
print("%-46s %6s %10s %12s" % ("entry point", "calls", "host ms", "gpu-idle ms"))
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-46s %6d %10.2f %12.2f" % (n[:46], a[0], a[1] / 1e3, a[2] / 1e3))
