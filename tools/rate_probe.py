#!/usr/bin/env python3
"""Montgomery product rates: radix 2^32 (shipped NTT / sweep product) and the carry-free radix 2^29 product of the MSM,
at full occupancy and at N waves per SIMD (_oN), plus the on-device self-check of the generated radix-2^29 code."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
for w in ["modmul","modmul_o4","modmul_o2","modmul_o1","modmul29","modmul29_o4","modmul29_o3","modmul29_o2","modmul29_o1","modmul29_check"]: print(w, "%.3e"%B.ubench(w))
