#!/usr/bin/env python3
"""MSM / NTT throughput versus size (single call and pipelined batch of 8)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
rng = np.random.default_rng(1)
def rand(n):
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
for k in (12, 14, 16, 17, 18, 20, 22):
    n = 1 << k
    bases = B.Bases.generate(0x657a6b6c, n)
    cols = [B.DeviceBuffer.from_numpy(rand(n)) for _ in range(8 if k <= 20 else 2)]
    B.msm_g1_dev(bases, cols[0].ptr, n)
    t0 = time.perf_counter()
    for _ in range(5):
        B.msm_g1_dev(bases, cols[0].ptr, n)
    single = (time.perf_counter() - t0) / 5
    B.msm_g1_batch_dev(bases, [c.ptr for c in cols], n)
    t0 = time.perf_counter()
    for _ in range(3):
        B.msm_g1_batch_dev(bases, [c.ptr for c in cols], n)
    batch = (time.perf_counter() - t0) / (3 * len(cols))
    d = ezkl_amd.EvaluationDomain(2, k)
    for _ in range(3):
        B.ntt_dev(cols[0].ptr, k, d.omega)
    ntt1 = B.last_kernel_ms("ntt")
    nb = len(cols)
    big = B.DeviceBuffer(nb * n * 32)
    for _ in range(3):
        B.ntt_dev(big.ptr, k, d.omega, batch=nb)
    nttb = B.last_kernel_ms("ntt") / nb
    print("k=%2d  msm single %.3f ms (%.2e pts/s)  batch %.3f ms (%.2e pts/s) | ntt single %.3f ms (%.2e el/s)  batch %.3f ms (%.2e el/s)"
          % (k, single * 1e3, n / single, batch * 1e3, n / batch, ntt1, n / ntt1 * 1e3, nttb, n / nttb * 1e3), flush=True)
    bases.free()
# BASELINE.md §3's last NTT scaling point: 2^24 (four passes); no MSM at this size (the bench's base sets stop at 2^22)
k = 24
n = 1 << k
d = ezkl_amd.EvaluationDomain(2, k)
col = B.DeviceBuffer.from_numpy(rand(n))
for _ in range(5):
    B.ntt_dev(col.ptr, k, d.omega)
t = B.last_kernel_ms("ntt")
print("k=24  ntt single %.3f ms (%.2e el/s)" % (t, n / t * 1e3), flush=True)
