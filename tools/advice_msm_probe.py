#!/usr/bin/env python3
"""MSM timing on the REAL advice columns of an ezkl circuit (tools/bench_circuits.py): what the prover's advice / lookup commit phases
actually multiply -- small signed values, decomposition digits, 0 / 1 masks, long constant runs -- column by column and as one batch.
    K=20 python tools/advice_msm_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ezkl_amd
from ezkl_amd import backend as B
import bench_circuits as BC
ezkl_amd.init(0)
k = int(os.environ.get("K", "17"))
n = 1 << k
built = BC.build(os.environ.get("CIRCUIT", "mlp"), k, gpu=B)
adv = built["advice"]
bases = B.Bases.generate(0x657a6b6c, n)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rand_fr
rng = np.random.default_rng(1)
d_uni = B.DeviceBuffer.from_numpy(rand_fr(rng, n))
for _ in range(3):
    B.msm_g1_dev(bases, d_uni.ptr, n)
print("uniform            : device %.3f ms (accumulate %.3f)" % (B.last_kernel_ms("msm"), B.last_kernel_ms("msm_accumulate")), flush=True)
cols = [B.DeviceBuffer.from_numpy(np.ascontiguousarray(a)) for a in adv]
M, Rm = 1 << 256, 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
for c, (a, d) in enumerate(zip(adv, cols)):
    for _ in range(3):
        B.msm_g1_dev(bases, d.ptr, n)
    nz = int((np.asarray(a) != 0).any(axis=1).sum())
    distinct = len(np.unique(np.asarray(a), axis=0))
    print("advice column %2d   : device %.3f ms (accumulate %.3f)   nonzero rows %d, distinct values %d" %
          (c, B.last_kernel_ms("msm"), B.last_kernel_ms("msm_accumulate"), nz, distinct), flush=True)
B.msm_g1_batch_dev(bases, [c.ptr for c in cols], n)
t0 = time.perf_counter()
for _ in range(3):
    B.msm_g1_batch_dev(bases, [c.ptr for c in cols], n)
print("batch of %d advice columns: %.3f ms per column" % (len(cols), (time.perf_counter() - t0) / 3 / len(cols) * 1e3))
ucols = [B.DeviceBuffer.from_numpy(rand_fr(rng, n)) for _ in range(len(cols))]
B.msm_g1_batch_dev(bases, [c.ptr for c in ucols], n)
t0 = time.perf_counter()
for _ in range(3):
    B.msm_g1_batch_dev(bases, [c.ptr for c in ucols], n)
print("batch of %d uniform columns: %.3f ms per column" % (len(cols), (time.perf_counter() - t0) / 3 / len(cols) * 1e3))
