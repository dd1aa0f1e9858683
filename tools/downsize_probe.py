#!/usr/bin/env python3
"""ParamsKZG::downsize on the device at the sizes a prover meets: a k = 22 coefficient basis downsized to k = 20 / 17, and the full-size
rebuild k = 20 -> 20 checked against the closed-form Lagrange basis of the same secret (gen_srs)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, ezkl_amd
from ezkl_amd import backend as B, plonk as P
ezkl_amd.init(0)
s = 0x1234567890abcdef1234567890abcdef % P.R
g22, gl22 = B.gen_srs(22, s)
gl22.free()
for k in (17, 20):
    want_g, want_gl = B.gen_srs(k, s)
    B.synchronize(); t0 = time.time()
    g, gl = g22.downsize(k)
    B.synchronize(); dt = time.time() - t0
    ok = bool((gl.download() == want_gl.download()).all() and (g.download() == want_g.download()).all())
    print("downsize 2^22 -> 2^%d: %.3f s, equals the closed-form SRS of the same secret: %s" % (k, dt, ok))
    for b in (g, gl, want_g, want_gl): b.free()
