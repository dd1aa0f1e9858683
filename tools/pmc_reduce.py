#!/usr/bin/env python3
"""Reduce the two rocprofv3 PMC passes over tools/pmc_probe.py into profiles/pmc_traffic.json (read by bench.py for
roofline.traffic).  Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md's HBM section and are
re-calibrated IN THE SAME RUN with kernels of known byte counts:
  FETCH_SIZE / WRITE_SIZE are reported in KiB;
  FETCH_SIZE counts one half of a coalesced streaming read   -> stream_factor (1 GiB copy kernel, expected ~2.0);
  FETCH_SIZE counts 64-byte random gathers in full            -> gather_factor (gather64 kernel, expected ~1.0);
  WRITE_SIZE is exact                                          -> write_factor (1 GiB copy kernel, expected ~1.0).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/tools/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python $R/tools/pmc_probe.py
    python tools/pmc_reduce.py <fetch csv> <write csv> profiles/pmc_traffic.json "<label>"
"""
import csv, hashlib, json, os, subprocess, sys
from collections import defaultdict

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def stamp():
    """what ties the figures to the kernels they were measured on: the SHA-256 of the kernel sources the probe's library was built from, and
    the commit (bench.py refuses a figure whose msm.hip / ntt.hip hash differs from the tree it runs in)"""
    sha = {f: hashlib.sha256(open(os.path.join(ROOT, "ezkl_amd", "csrc", f), "rb").read()).hexdigest() for f in ("msm.hip", "ntt.hip", "field29.hpp", "curve29.hpp")}
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        commit = None
    return sha, commit


def load(path):
    d = defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0]
        if name.startswith("void "):                 # templated kernels are reported with their return type and arguments
            name = name[5:]
        name = name.split("<")[0]
        d[name].append(float(r["Counter_Value"]) * 1024.0)      # KiB -> bytes
    return {k: v for k, v in d.items()}


def mean_last(v, n):          # the probe launches every measured kernel after a warm-up: keep the last n launches
    v = v[-n:]
    return sum(v) / len(v)


fetch, write = load(sys.argv[1]), load(sys.argv[2])
GiB = float(1 << 30)
copy_f = mean_last(fetch["ezkl::ub_copy_kernel"], 2)
gath_f = mean_last(fetch["ezkl::ub_gather_kernel"], 1)
copy_w = mean_last(write["ezkl::ub_copy_kernel"], 2)
stream_factor, write_factor = GiB / copy_f, GiB / copy_w
# the gather probe: nthreads * 32 iterations * 64 B, nthreads = CUs * 16 * 256
gather_bytes = 256 * 16 * 256 * 32 * 64.0
gather_factor = gather_bytes / gath_f
acc_f, acc_w = mean_last(fetch["ezkl::msm_accumulate_kernel"], 3), mean_last(write["ezkl::msm_accumulate_kernel"], 3)
ntt_f, ntt_w = mean_last(fetch["ezkl::ntt_pass_kernel"], 9), mean_last(write["ezkl::ntt_pass_kernel"], 9)
_sha, _commit = stamp()
out = {
    "kernel_sources_sha256": _sha,
    "commit": _commit or os.environ.get("EZKL_COMMIT"),
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/pmc_probe.py, " + (sys.argv[4] if len(sys.argv) > 4 else ""),
    "units": "bytes per launch",
    "calibration": {"stream_copy_1GiB_FETCH_SIZE_bytes": copy_f, "stream_factor": stream_factor, "gather64_FETCH_SIZE_bytes": gath_f,
                    "gather64_expected_bytes": gather_bytes, "gather_factor": gather_factor, "WRITE_SIZE_copy_1GiB_bytes": copy_w, "write_factor": write_factor},
    "msm_accumulate_kernel_fetch_raw": acc_f,
    "msm_accumulate_kernel_write": acc_w * write_factor,
    # gather-dominated: table records count at the gather factor; the sorted-index stream (a few % of the bytes) would count x2
    "msm_accumulate_kernel_bytes_per_launch": acc_f * gather_factor + acc_w * write_factor,
    "msm_accumulate_note": "gather-dominated (one 64-byte table record per pair at factor ~1.0); the sorted-index stream would count x2, so this is a lower bound within ~5%",
    "ntt_pass_kernel_fetch_raw_mean": ntt_f,
    "ntt_pass_kernel_write": ntt_w * write_factor,
    "ntt_2p22_bytes_per_transform": 3 * (ntt_f * stream_factor + ntt_w * write_factor),
    "algorithmic": {"msm_2p20": 96 << 20, "ntt_2p22": 64 << 22},
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
