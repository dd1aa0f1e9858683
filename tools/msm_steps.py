#!/usr/bin/env python3
"""STEPS (default 40) synchronous 2^20-point MSMs and 2^22 NTTs after a clock warm-up: the workload of bench.py's two timed regions, alone,
for `rocprofv3 --kernel-trace --stats` (per-kernel averages of exactly these launches)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
rng = np.random.default_rng(0x657a6b6c)
def rand(n):
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 61) - 1); return a
steps = int(os.environ.get("STEPS", "40"))
what = os.environ.get("WHAT", "msm,ntt").split(",")
if "msm" in what:
    n = 1 << 20
    bases = B.Bases.generate(0x657a6b6c, n)
    sc = B.DeviceBuffer.from_numpy(rand(n))
    for _ in range(150): B.msm_g1_dev(bases, sc.ptr, n)
    B.synchronize(); t0 = time.perf_counter(); ms = []
    for _ in range(steps): B.msm_g1_dev(bases, sc.ptr, n); ms.append(B.last_kernel_ms("msm"))
    B.synchronize(); print("msm 2^20: wall %.4f ms/step, device %.4f ms (min %.4f)" % ((time.perf_counter() - t0) / steps * 1e3, np.mean(ms), np.min(ms)))
if "ntt" in what:
    k = 22
    dom = ezkl_amd.EvaluationDomain(2, k)
    col = B.DeviceBuffer.from_numpy(rand(1 << k))
    for _ in range(200): B.ntt_dev(col.ptr, k, dom.omega)
    B.synchronize(); t0 = time.perf_counter(); ms = []
    for _ in range(steps): B.ntt_dev(col.ptr, k, dom.omega); ms.append(B.last_kernel_ms("ntt"))
    B.synchronize(); print("ntt 2^22: wall %.4f ms/step, device %.4f ms (min %.4f)" % ((time.perf_counter() - t0) / steps * 1e3, np.mean(ms), np.min(ms)))
