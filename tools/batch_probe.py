import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import bench, ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
n = 1 << 20
bases = B.Bases.generate(bench.SEED, n)
cols = [B.DeviceBuffer.from_numpy(bench.witness_like_fr(np.random.default_rng(bench.SEED + 17 + i), n)) for i in range(12)]
ptrs = [c.ptr for c in cols]
for nb in (1, 4, 6, 12):
    for _ in range(10): r = B.msm_g1_batch_dev(bases, ptrs[:nb], n)
    B.synchronize(); t0 = time.perf_counter()
    reps = 20
    for _ in range(reps): r = B.msm_g1_batch_dev(bases, ptrs[:nb], n)
    B.synchronize(); dt = (time.perf_counter() - t0) / reps
    print("W batch of %2d: %.3f ms per batch, %.4f ms per MSM, %.3e pts/s" % (nb, dt * 1e3, dt * 1e3 / nb, nb * n / dt))
u = [B.DeviceBuffer.from_numpy(bench.rand_fr(np.random.default_rng(5 + i), n)) for i in range(8)]
for nb in (1, 4, 8):
    pu = [c.ptr for c in u[:nb]]
    for _ in range(10): r = B.msm_g1_batch_dev(bases, pu, n)
    B.synchronize(); t0 = time.perf_counter()
    for _ in range(20): r = B.msm_g1_batch_dev(bases, pu, n)
    B.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("U batch of %2d: %.3f ms per batch, %.4f ms per MSM, %.3e pts/s" % (nb, dt * 1e3, dt * 1e3 / nb, nb * n / dt))
