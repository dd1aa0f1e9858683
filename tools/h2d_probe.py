#!/usr/bin/env python3
"""Host-to-device copy rates seen through the C ABI: blocking copies from pageable / page-locked memory into pooled blocks,
with and without a device synchronisation before each copy (NOTEBOOK.md §6, PCIe note)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
n, m = 1 << 20, 14
rng = np.random.default_rng(0)
cols = [rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64) for _ in range(m)]
pinned = [B.PinnedArray((n, 4)) for _ in range(m)]
for pa, c in zip(pinned, cols):
    pa.array[:] = c
def run(src, sync_first, label):
    best = 1e9
    for rep in range(4):
        devs = []
        t0 = time.perf_counter()
        for a in src:
            if sync_first:
                B.synchronize()
            devs.append(B.DeviceBuffer.from_numpy(a))
        dt = time.perf_counter() - t0
        best = min(best, dt)
        del devs
    print("%-40s %6.2f ms  %5.1f GB/s" % (label, best * 1e3, m * n * 32 / best / 1e9))
run(cols, False, "pageable, blocking")
run(cols, True, "pageable, blocking, device sync first")
run([pa.array for pa in pinned], False, "page-locked, blocking")
run([pa.array for pa in pinned], True, "page-locked, blocking, device sync first")
bases = B.Bases.generate(1, n)
for src, label in ((cols, "pageable"), ([pa.array for pa in pinned], "page-locked")):
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        devs, _ = B.upload_commit_batch(bases, src)
        best = min(best, time.perf_counter() - t0)
        del devs
    print("%-40s %6.2f ms" % ("upload_commit_batch, " + label, best * 1e3))
best = 1e9
ds = [B.DeviceBuffer.from_numpy(a) for a in cols]
for rep in range(4):
    t0 = time.perf_counter()
    B.msm_g1_batch_dev(bases, [d.ptr for d in ds], n)
    best = min(best, time.perf_counter() - t0)
print("%-40s %6.2f ms" % ("msm batch of 14 resident (uniform)", best * 1e3))
R2 = np.frombuffer(((1 << 512) % 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001).to_bytes(32, "little"), np.uint64).copy()
small = []
for j in range(m):                                  # witness-like: 20-bit values in Montgomery form
    c = np.zeros((n, 4), np.uint64); c[:, 0] = rng.integers(1, 1 << 20, size=n).astype(np.uint64)
    d = B.DeviceBuffer.from_numpy(c); B.vec_scale(d.ptr, R2, d.ptr, n); small.append(d)
for rep in range(5):
    t0 = time.perf_counter()
    B.msm_g1_batch_dev(bases, [d.ptr for d in small], n)
    dt = time.perf_counter() - t0
    print("%-40s %6.2f ms" % ("msm batch of 14 resident (20-bit values)", dt * 1e3))
hs = [d.to_numpy(shape=(n, 4)) for d in small]
for rep in range(3):
    t0 = time.perf_counter()
    devs, _ = B.upload_commit_batch(bases, hs)
    print("%-40s %6.2f ms" % ("upload_commit_batch, pageable, 20-bit", (time.perf_counter() - t0) * 1e3))
    del devs
