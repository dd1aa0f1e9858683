#!/usr/bin/env python3
"""One 2^20-point MSM of small (20-bit) witness values, repeated: the profile target for the skew the signed-digit carry
creates (half of the rows carry a digit +1 into window 1, i.e. one bucket with 2^19 entries)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
n = 1 << 20
bits = int(os.environ.get("BITS", "20"))
rng = np.random.default_rng(0)
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
R2 = np.frombuffer(((1 << 512) % R).to_bytes(32, "little"), np.uint64).copy()
c = np.zeros((n, 4), np.uint64); c[:, 0] = rng.integers(1, 1 << bits, size=n).astype(np.uint64)
d = B.DeviceBuffer.from_numpy(c); B.vec_scale(d.ptr, R2, d.ptr, n)
bases = B.Bases.generate(1, n)
B.msm_g1_dev(bases, d.ptr, n)
t = []
for _ in range(10):
    t0 = time.perf_counter(); B.msm_g1_dev(bases, d.ptr, n); t.append(time.perf_counter() - t0)
print("bits", bits, "msm ms", round(min(t) * 1e3, 3), "device", B.last_kernel_ms("msm"), "accumulate", B.last_kernel_ms("msm_accumulate"))
