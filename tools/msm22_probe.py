#!/usr/bin/env python3
"""where a 2^22-point uniform MSM spends its time: accumulate (HIP events) vs the whole chain, next to 2^20 and 2^21"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
rng = np.random.default_rng(1)
def rand(n):
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
for k in (20, 21, 22):
    n = 1 << k
    bases = B.Bases.generate(0x657a6b6c, n)
    s = B.DeviceBuffer.from_numpy(rand(n))
    for _ in range(20): B.msm_g1_dev(bases, s.ptr, n)
    B.synchronize(); B.kernel_ms_stats("msm", reset=True); B.kernel_ms_stats("msm_accumulate", reset=True)
    t0 = time.perf_counter()
    for _ in range(10): B.msm_g1_dev(bases, s.ptr, n)
    B.synchronize(); wall = (time.perf_counter() - t0) / 10 * 1e3
    (ms, c1), (acc, c2) = B.kernel_ms_stats("msm"), B.kernel_ms_stats("msm_accumulate")
    print("k=%d: wall %.3f ms, chain %.3f ms, accumulate %.3f ms (%.3f us per 2^20 points)" % (k, wall, ms / c1, acc / c2, acc / c2 / (n >> 20) * 1e3), flush=True)
    bases.free()
