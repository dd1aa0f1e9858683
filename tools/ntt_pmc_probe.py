#!/usr/bin/env python3
"""the workload of tools/ntt_pmc.sh: 3 resident transforms of 2^20 points, then 3 of 2^22, then 3 coset-major 2^20 -> 2^22 (what a proof runs)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
rng = np.random.default_rng(1)
for k in (20, 22):
    a = rng.integers(0, 1 << 62, size=(1 << k, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1)
    d = ezkl_amd.EvaluationDomain(2, k); buf = B.DeviceBuffer.from_numpy(a)
    for _ in range(3):
        B.ntt_dev(buf.ptr, k, d.omega)
    print("ntt 2^%d ms" % k, B.last_kernel_ms("ntt"))
a = rng.integers(0, 1 << 62, size=(1 << 20, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1)
src, dst = B.DeviceBuffer.from_numpy(a), B.DeviceBuffer((1 << 22) * 32)
for _ in range(3):
    B.coeff_to_cosets_dev(src.ptr, dst.ptr, 20, 22)
print("coset-major 2^20 -> 2^22 ms", B.last_kernel_ms("coset_ntt"))
