import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
rng = np.random.default_rng(1)
n = 1 << 22
a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1)
bases = B.Bases.generate(0x657a6b6c, n)
s = B.DeviceBuffer.from_numpy(a)
for _ in range(25): B.msm_g1_dev(bases, s.ptr, n)
B.synchronize()
