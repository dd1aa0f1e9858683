#!/usr/bin/env python3
"""Print the on-device microbenchmarks (compute/HBM ceilings) and first kernel timings."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
out = {}
for w in ("modmul", "modmul_c", "modmul_inl", "addsub", "mad64", "dfma", "copy"):
    out[w] = B.ubench(w)
print(json.dumps(out))
rng = np.random.default_rng(1)
def rand(n):
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
for k in (16, 20, 22):
    d = ezkl_amd.EvaluationDomain(2, k)
    buf = B.DeviceBuffer.from_numpy(rand(1 << k))
    for _ in range(3):
        B.ntt_dev(buf.ptr, k, d.omega)
    print("ntt 2^%d: %.3f ms" % (k, B.last_kernel_ms("ntt")))
for k in (16, 20):
    n = 1 << k
    t = time.time(); bases = B.Bases.generate(0x657a6b6c, n); print("gen bases 2^%d %.2fs" % (k, time.time() - t))
    sc = B.DeviceBuffer.from_numpy(rand(n))
    t = time.time(); B.msm_g1_dev(bases, sc.ptr, n); print("first msm (table precompute) %.2fs" % (time.time() - t))
    for _ in range(3):
        t = time.time(); B.msm_g1_dev(bases, sc.ptr, n); wall = time.time() - t
    print("msm 2^%d: total %.3f ms, accumulate %.3f ms, wall %.3f ms" % (k, B.last_kernel_ms("msm"), B.last_kernel_ms("msm_accumulate"), wall * 1e3))
    bases.free()
