#!/usr/bin/env python3
"""Per-kernel HBM traffic of ONE k = 20 MLP proof from the two rocprofv3 PMC passes over tools/pmc_prove.py (FETCH_SIZE and WRITE_SIZE in
separate runs, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; KiB units; FETCH_SIZE counts half of a coalesced streaming read and
64-byte gathers in full -- both factors re-calibrated in the same run with kernels of known byte counts).
    python tools/pmc_prove_reduce.py <fetch csv> <write csv> <out json> <label>"""
import csv, hashlib, json, os, subprocess, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r.get("Dispatch_Id", len(rows))), r["Kernel_Name"].split("(")[0].replace("ezkl::", "").replace("void ", ""), float(r["Counter_Value"]) * 1024.0))
    rows.sort()
    return rows


fetch, write = load(sys.argv[1]), load(sys.argv[2])
GiB = float(1 << 30)
def first(rows, name, which=-1):
    v = [b for _, n, b in rows if n == name]
    return v[which]
copy_f, copy_w = first(fetch, "ub_copy_kernel"), first(write, "ub_copy_kernel")
gath_f = first(fetch, "ub_gather_kernel", 0)
stream_factor, write_factor = GiB / copy_f, GiB / copy_w
gather_factor = 256 * 16 * 256 * 32 * 64.0 / gath_f
def tail(rows):                      # the measured proof: everything after the last marker launch
    last = max(i for i, (_, n, _) in enumerate(rows) if n == "ub_gather_kernel")
    return rows[last + 1:]
tf, tw = tail(fetch), tail(write)
assert [n for _, n, _ in tf] == [n for _, n, _ in tw], "the two passes launched different kernel sequences"
per = {}
for (_, n, f), (_, _, w) in zip(tf, tw):
    d = per.setdefault(n, [])
    d.append((f, w))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/pmc_prove.py (one k = " + os.environ.get("K", "20") + " MLP proof, C++ host), " + (sys.argv[4] if len(sys.argv) > 4 else ""),
       "units": "bytes per launch (HBM read + write), corrected", "calibration": {"stream_factor": stream_factor, "gather_factor": gather_factor, "write_factor": write_factor},
       "kernels": {}}
n_ext, n = 1 << 22, 1 << 20
for name, v in sorted(per.items(), key=lambda kv: -sum(f for f, _ in kv[1])):
    ff = gather_factor if name == "msm_accumulate_kernel" else stream_factor
    tot = [f * ff + w * write_factor for f, w in v]
    out["kernels"][name] = {"launches": len(v), "bytes_total": sum(tot), "bytes_per_launch_mean": sum(tot) / len(tot), "bytes_per_launch_max": max(tot),
                            "read_factor_applied": "gather" if ff == gather_factor else "stream"}
# the sweep: the evalh_jit launches with the most traffic are the cosets of the quotient numerator (the others are n-row helper programs)
if "evalh_jit" in per:
    # (SWEEP_LAUNCHES: cosets x kernels per coset when the program is cut -- 8 for the k = 22 / 30-column circuit; K: its rows)
    sw = sorted((f * stream_factor + w * write_factor for f, w in per["evalh_jit"]), reverse=True)[:int(os.environ.get("SWEEP_LAUNCHES", "4"))]
    out["evalh_jit_sweep"] = {"launches": len(sw), "bytes_per_launch_mean": sum(sw) / len(sw), "bytes_total": sum(sw), "rows_per_launch": 1 << int(os.environ.get("K", "20"))}
for name in list(out["kernels"]):                       # a templated kernel with one instantiation is also listed under its plain name
    base = name.split("<")[0]
    if base != name and sum(1 for n_ in out["kernels"] if n_.split("<")[0] == base) == 1:
        out["kernels"][base] = out["kernels"][name]
# what ties the figures to the kernels they were measured on (bench.py refuses them when msm.hip / ntt.hip / evalh.hip have changed since)
out["kernel_sources_sha256"] = {f: hashlib.sha256(open(os.path.join(ROOT, "ezkl_amd", "csrc", f), "rb").read()).hexdigest() for f in ("msm.hip", "ntt.hip", "evalh.hip", "field29.hpp", "curve29.hpp")}
try:
    out["commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("EZKL_COMMIT")
except OSError:
    out["commit"] = os.environ.get("EZKL_COMMIT")
json.dump(out, open(sys.argv[3], "w"), indent=1)
for name, d in list(out["kernels"].items())[:14]:
    print("%-34s %5d launches  %10.1f MB per launch  %10.1f MB total" % (name, d["launches"], d["bytes_per_launch_mean"] / 1e6, d["bytes_total"] / 1e6))
print("evalh sweep:", out.get("evalh_jit_sweep"))
