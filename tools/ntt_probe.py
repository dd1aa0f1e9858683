#!/usr/bin/env python3
"""Device time of one resident 2^22 NTT (HIP events)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
rng=np.random.default_rng(1)
a=rng.integers(0,1<<62,size=(1<<22,4),dtype=np.uint64); a[:,3]&=np.uint64((1<<61)-1)
d=ezkl_amd.EvaluationDomain(2,22); buf=B.DeviceBuffer.from_numpy(a)
for _ in range(4): B.ntt_dev(buf.ptr,22,d.omega)
print("ntt 2^22 ms", B.last_kernel_ms("ntt"))
