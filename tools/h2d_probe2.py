#!/usr/bin/env python3
"""H2D copy rate vs copy size, pinned vs pageable, blocking (ezkl_hip_memcpy_h2d) and through the upload phase (async copies on the copy stream)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
for logn in (20, 21, 22):
    n, m = 1 << logn, 6
    src = np.zeros((n, 4), np.uint64); src[:, 0] = np.arange(n)
    pin = [B.PinnedArray((n, 4)) for _ in range(m)]
    for p in pin: p.array[:] = src
    pag = [src.copy() for _ in range(m)]
    dev = [B.DeviceBuffer(n * 32) for _ in range(m)]
    for label, cols in (("pinned", [p.array for p in pin]), ("pageable", pag)):
        best = 1e9
        for rep in range(3):
            B.synchronize(); t0 = time.perf_counter()
            for d, a in zip(dev, cols): B.memcpy_h2d(d.ptr, a)
            B.synchronize(); best = min(best, time.perf_counter() - t0)
        print("2^%d x %d  %-9s blocking      %7.2f ms  %5.1f GB/s" % (logn, m, label, best * 1e3, m * n * 32 / best / 1e9), flush=True)
    bases = B.Bases.generate(1, n)
    B.upload_commit_batch(bases, [p.array for p in pin])
    for label, cols in (("pinned", [p.array for p in pin]), ("pageable", pag)):
        best = 1e9
        for rep in range(3):
            B.synchronize(); t0 = time.perf_counter()
            devs, _ = B.upload_commit_batch(bases, cols)
            best = min(best, time.perf_counter() - t0); del devs
        print("2^%d x %d  %-9s upload+commit %7.2f ms" % (logn, m, label, best * 1e3), flush=True)
    bases.free()
