#!/usr/bin/env python3
"""MSM timing under skewed scalar distributions (SURVEY.md §8(d) 'W' + pathological cases)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ezkl_amd
from ezkl_amd import backend as B
from conftest import witness_like, rand_fr, fe_from_int
from oracle import binding as ob
ezkl_amd.init(0)
n = 1 << 20
rng = np.random.default_rng(3)
bases = B.Bases.generate(0x657a6b6c, n)
pts = bases.download()
cases = {"uniform": rand_fr(rng, n), "witness_like": witness_like(rng, n), "all_equal": np.tile(fe_from_int(12345), (n, 1)),
         "two_values": np.where((np.arange(n) % 2 == 0)[:, None], fe_from_int(7), fe_from_int(2**200 + 5)).astype(np.uint64),
         "all_zero": np.zeros((n, 4), np.uint64)}
for name, s in cases.items():
    d = B.DeviceBuffer.from_numpy(s)
    got = B.msm_g1_dev(bases, d.ptr, n)
    for _ in range(3):
        got = B.msm_g1_dev(bases, d.ptr, n)
    ok = (got == ob.msm(s, pts)).all()
    print("%-14s device %.3f ms (accumulate %.3f)  parity=%s" % (name, B.last_kernel_ms("msm"), B.last_kernel_ms("msm_accumulate"), ok), flush=True)
# a prover phase: 14 witness-like columns committed as one pipelined batch
cols = [B.DeviceBuffer.from_numpy(witness_like(rng, n)) for _ in range(14)]
B.msm_g1_batch_dev(bases, [c.ptr for c in cols], n)
t0 = time.perf_counter()
for _ in range(3):
    B.msm_g1_batch_dev(bases, [c.ptr for c in cols], n)
print("batch of 14 witness-like columns: %.3f ms per column" % ((time.perf_counter() - t0) / 3 / 14 * 1e3))
