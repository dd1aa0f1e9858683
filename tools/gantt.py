#!/usr/bin/env python3
"""ASCII swim lanes of the LAST `window_ms` of GPU activity in a rocprofv3 rocpd database: one row per HIP stream (+ one for the
copy engines), one character per `bin_us` microseconds showing what ran there:
   A msm_accumulate   s msm sort (hist / scans / partition / binsort / bigsort)   f msm fixup   r msm reduce / planes
   N ntt_pass   E evalh sweep   C memory copy   v everything else (vec ops, scans, inversions, fills)   . idle
followed by the busy fraction of every lane and of the whole device.
    python tools/gantt.py <results.db> [window_ms=100] [bin_us=250]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 100e6
binw = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 250e3
cur = db.cursor()


def cls(n):
    if "msm_accumulate" in n: return "A"
    if any(x in n for x in ("msm_hist", "msm_part", "msm_binsort", "msm_bigsort")): return "s"
    if "msm_fixup" in n: return "f"
    if "msm_reduce" in n or "msm_planes" in n: return "r"
    if "ntt_pass" in n or "ntt_" in n: return "N"
    if "evalh" in n or "eval_program" in n: return "E"
    return "v"


ev = [(s, e, cls(n), "stream %s" % st) for n, s, e, st in cur.execute("select name, start, end, stream_id from kernels")]
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "memory_copies" in tables:
    ev += [(s, e, "C", "copies") for s, e in cur.execute("select start, end from memory_copies")]
t1 = max(e for _, e, _, _ in ev)
ev = sorted(x for x in ev if x[1] >= t1 - win)
t0 = ev[0][0]
nb = int((t1 - t0) / binw) + 1
lanes = {}
for s, e, c, lane in ev:
    row = lanes.setdefault(lane, [dict() for _ in range(nb)])
    b0, b1 = int((s - t0) / binw), int((e - t0) / binw)
    for b in range(b0, b1 + 1):
        lo, hi = max(s, t0 + b * binw), min(e, t0 + (b + 1) * binw)
        if hi > lo:
            row[b][c] = row[b].get(c, 0) + (hi - lo)
print("window %.2f ms, %d bins of %.0f us" % ((t1 - t0) / 1e6, nb, binw / 1e3))
print("%-10s %s" % ("ms", "".join(("%-8d" % int(i * binw * 8 / 1e6 * 1)) if False else "" for i in range(0))))
ruler = ""
for b in range(nb):
    ms = b * binw / 1e6
    ruler += ("|" if abs(ms - round(ms)) < 1e-9 and int(round(ms)) % 5 == 0 else " ")
print("%-10s %s" % ("5 ms marks", ruler))
def busy_of(row): return sum(min(binw, sum(d.values())) for d in row)
for lane in sorted(lanes, key=lambda l: -busy_of(lanes[l])):
    row = lanes[lane]
    line = "".join((max(d, key=d.get) if sum(d.values()) > 0.15 * binw else ("," if d else ".")) for d in row)
    print("%-10s %s  %4.1f%%" % (lane[:10], line, 100.0 * busy_of(row) / (t1 - t0)))
# whole device: union of intervals
iv = sorted((s, e) for s, e, _, _ in ev)
busy, cs, ce = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print("device busy (union) %.2f of %.2f ms = %.1f %%" % (busy / 1e6, (t1 - t0) / 1e6, 100.0 * busy / (t1 - t0)))
