#!/usr/bin/env python3
"""wall time per synchronous 2^20-point MSM call, nothing else (no event reads): for A/B of switches that remove the timing events"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
rng = np.random.default_rng(0x657a6b6c)
a = rng.integers(0, 1 << 63, size=(1 << 20, 4), dtype=np.uint64) * np.uint64(2); a[:, 3] &= np.uint64((1 << 61) - 1)
n = 1 << 20
bases = B.Bases.generate(0x657a6b6c, n); sc = B.DeviceBuffer.from_numpy(a)
for _ in range(200): B.msm_g1_dev(bases, sc.ptr, n)
best = []
for rep in range(5):
    B.synchronize(); t0 = time.perf_counter()
    for _ in range(50): B.msm_g1_dev(bases, sc.ptr, n)
    B.synchronize(); best.append((time.perf_counter() - t0) / 50 * 1e3)
print("wall ms per MSM: %s" % ["%.4f" % x for x in best])
