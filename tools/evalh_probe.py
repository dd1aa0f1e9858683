#!/usr/bin/env python3
"""Quotient-sweep timing on an ezkl-shaped synthetic gate program:
per (block, inner column) gates sel*(out - a*b), dot-product accumulators with rotation -1, folded with y."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B
ezkl_amd.init(0)
k = int(os.environ.get("K", "17")); ek = k + 3
ne = 1 << ek
rng = np.random.default_rng(1)
NCOL = 24            # 6 blocks x (a, b, out) + 6 selectors
def rand(n):
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
cols = [B.DeviceBuffer.from_numpy(rand(ne)) for _ in range(NCOL)]
prog = B.GraphProgram(k, ek)
gates = []
for blk in range(6):
    a, b, out, sel = prog.column(3 * blk), prog.column(3 * blk + 1), prog.column(3 * blk + 2), prog.column(18 + blk)
    ab = prog.calc("mul", a, b)
    gates.append(prog.calc("mul", sel, prog.calc("sub", out, ab)))                                      # MULT gate
    gates.append(prog.calc("mul", sel, prog.calc("sub", out, prog.calc("add", prog.column(3 * blk + 2, -1), ab))))   # DOT gate
    gates.append(prog.calc("mul", sel, prog.calc("sub", out, prog.calc("add", a, b))))                 # ADD gate
prog.horner(prog.previous(), gates, prog.challenge(0))
code, _, _ = prog.arrays()
nmul = int(((code[:, 0] == 2) | (code[:, 0] == 3) | (code[:, 0] == 7)).sum())
chal = rand(1)
out = B.DeviceBuffer.from_numpy(np.zeros((ne, 4), np.uint64))
for _ in range(3):
    prog.evaluate_h([c.ptr for c in cols], chal, out.ptr)
ms = B.last_kernel_ms("eval_h")
alg = 32.0 * (NCOL + 1) * ne
print("eval_h k=%d ext_k=%d rows=%d cols=%d instr=%d (%d products/row): %.3f ms  -> %.3e rows/s, algorithmic %.1f GB/s, %.3e products/s"
      % (k, ek, ne, NCOL, code.shape[0], nmul, ms, ne / ms * 1e3, alg / ms / 1e6, nmul * ne / ms * 1e3))
