import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
rng = np.random.default_rng(7)
n = 1 << int(os.environ.get("LOGN", "20"))
a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1)
bases = B.Bases.generate(0x657a6b6c, n)
sc = B.DeviceBuffer.from_numpy(a)
print(B.msm_g1_dev(bases, sc.ptr, n))
