#!/usr/bin/env python3
"""n-point NTT and coset-major extended NTT timings (device ms from HIP events), batch of 8 columns: the transforms a proof runs.
EZKL_NTT_MAXR=8 selects the three-pass plans of round 2 for an A/B."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
rng = np.random.default_rng(1)
def rand(n):
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
for k, ek in ((16, 18), (18, 20), (20, 22), (22, 24)):
    n, batch = 1 << k, 8 if k <= 20 else 2
    d = ezkl_amd.EvaluationDomain(5, k)
    buf = B.DeviceBuffer.from_numpy(rand(n * batch).reshape(batch, n, 4))
    ext = B.DeviceBuffer(batch * (1 << ek) * 32)
    for _ in range(3): B.ntt_dev(buf.ptr, k, d.omega, batch=batch)
    t_ntt = B.last_kernel_ms("ntt") / batch
    for _ in range(3): B.coeff_to_cosets_dev(buf.ptr, ext.ptr, k, ek, batch=batch)
    t_cm = B.last_kernel_ms("coset_ntt") / batch
    for _ in range(3): B.coset_ntt_dev(buf.ptr, ext.ptr, k, ek, batch=batch)
    t_nat = B.last_kernel_ms("coset_ntt") / batch
    print("k=%d: n-point NTT %.4f ms/col | coset-major 2^%d -> 2^%d %.4f ms/col | natural-order (one 2^%d transform) %.4f ms/col" % (k, t_ntt, k, ek, t_cm, ek, t_nat), flush=True)
