#!/usr/bin/env python3
"""Workload for the PMC passes (run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, separately):
two calibration kernels with KNOWN byte counts (streaming copy: 1 GiB read + 1 GiB written; random 64-byte
gathers: threads*32*64 B) followed by the measured kernels (one 2^20 MSM, one 2^22 NTT)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
print("copy", B.ubench("copy"), "gather64", B.ubench("gather64"))
rng = np.random.default_rng(1)
def rand(n):
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
n = 1 << 20
bases = B.Bases.generate(0x657a6b6c, n)
sc = B.DeviceBuffer.from_numpy(rand(n))
for _ in range(3):
    B.msm_g1_dev(bases, sc.ptr, n)
d = ezkl_amd.EvaluationDomain(2, 22)
buf = B.DeviceBuffer.from_numpy(rand(1 << 22))
for _ in range(3):
    B.ntt_dev(buf.ptr, 22, d.omega)
print("done")
