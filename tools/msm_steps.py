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
    B.synchronize(); B.kernel_ms_stats("msm", reset=True); B.kernel_ms_stats("msm_accumulate", reset=True); t0 = time.perf_counter()
    for _ in range(steps): B.msm_g1_dev(bases, sc.ptr, n)
    B.synchronize(); wall = (time.perf_counter() - t0) / steps * 1e3
    (ms, cnt), (acc, _) = B.kernel_ms_stats("msm"), B.kernel_ms_stats("msm_accumulate")
    print("msm 2^20: wall %.4f ms/step, device %.4f ms, accumulate %.4f ms (%d steps)" % (wall, ms / cnt, acc / cnt, cnt))
if "ntt" in what:
    k = 22
    dom = ezkl_amd.EvaluationDomain(2, k)
    col = B.DeviceBuffer.from_numpy(rand(1 << k))
    for _ in range(200): B.ntt_dev(col.ptr, k, dom.omega)
    B.synchronize(); B.kernel_ms_stats("ntt", reset=True); t0 = time.perf_counter()
    for _ in range(steps): B.ntt_dev(col.ptr, k, dom.omega)
    B.synchronize(); wall_sync = (time.perf_counter() - t0) / steps * 1e3
    ms, cnt = B.kernel_ms_stats("ntt", reset=True)
    was = B.set_async(True); t0 = time.perf_counter()                       # queued stream-ordered, one synchronise at the end (bench.py's NTT region)
    for _ in range(steps): B.ntt_dev(col.ptr, k, dom.omega)
    B.synchronize(); wall_async = (time.perf_counter() - t0) / steps * 1e3
    B.set_async(was)
    ms2, cnt2 = B.kernel_ms_stats("ntt")
    print("ntt 2^22: wall %.4f ms/step synchronous, %.4f ms/step queued; device %.4f / %.4f ms (%d + %d steps)" % (wall_sync, wall_async, ms / cnt, ms2 / cnt2, cnt, cnt2))
