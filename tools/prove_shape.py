#!/usr/bin/env python3
"""Prove-SHAPED kernel pipeline: runs, back to back on resident synthetic data, the MSM / NTT / coset-NTT /
quotient-sweep calls that halo2's create_proof issues for one proof of a k-row circuit (call counts from
SURVEY.md §3.1), and reports device wall time.  It is NOT a proof (no transcript, no witness synthesis, no
SHPLONK host work): it measures how long the kernels of a prove take at the reference's call counts, the
quantity SURVEY.md §8(d) asks for until a host prover exists.

    K=20 A=12 L=6 PCOLS=14 DEG=6 FIXED=10 python tools/prove_shape.py
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ezkl_amd
from ezkl_amd import backend as B

k = int(os.environ.get("K", "20"))
A = int(os.environ.get("A", "12"))          # advice columns
I = 1                                       # instance columns
L = int(os.environ.get("L", "6"))           # lookup arguments (after chunking)
PCOLS = int(os.environ.get("PCOLS", "14"))  # columns in the permutation argument
DEG = int(os.environ.get("DEG", "6"))       # cs.degree()
FIXED = int(os.environ.get("FIXED", "10"))
P = -(-PCOLS // (DEG - 2))                  # permutation chunks
Q = DEG - 1                                 # quotient pieces
dom = ezkl_amd.EvaluationDomain(DEG, k)
ek, n, ne = dom.ext_k, 1 << k, 1 << dom.ext_k
ezkl_amd.init(0)
rng = np.random.default_rng(1)
def rand(m):
    a = rng.integers(0, 1 << 62, size=(m, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 61) - 1); return a
bases = B.Bases.generate(0x657a6b6c, n)
col_n = B.DeviceBuffer.from_numpy(rand(n))
NC = A + I + FIXED
ext_cols = [B.DeviceBuffer.from_numpy(rand(ne)) for _ in range(min(NC, 8))]      # reuse 8 physical columns
ext_ptrs = [ext_cols[i % len(ext_cols)].ptr for i in range(NC)]
out_ext = B.DeviceBuffer.from_numpy(np.zeros((ne, 4), np.uint64))
# an ezkl-shaped gate program over NC columns: per advice triple MULT/ADD/DOT gates folded with y
prog = B.GraphProgram(k, ek)
gates = []
for t in range(max(1, A // 3)):
    a, b, o = prog.column(3 * t), prog.column(3 * t + 1), prog.column(3 * t + 2)
    sel = prog.column(A + I + (t % FIXED))
    ab = prog.calc("mul", a, b)
    gates += [prog.calc("mul", sel, prog.calc("sub", o, ab)),
              prog.calc("mul", sel, prog.calc("sub", o, prog.calc("add", prog.column(3 * t + 2, -1), ab))),
              prog.calc("mul", sel, prog.calc("sub", o, prog.calc("add", a, b)))]
prog.horner(prog.previous(), gates, prog.challenge(0))
chal = rand(1)
counts = {"msm_n": A + 2 * L + P + 1 + Q + 2, "intt_n": A + I + 2 * L + P, "coset_ntt_ext": A + I + 2 * L + P, "inverse_coset_ntt_ext": 1,
          "eval_h_rows": ne, "eval_h_columns": NC}
import ctypes as C
from ezkl_amd import lib as L_
lib = L_.load()
def coset(inverse):
    L_.check(lib.ezkl_hip_coset_ntt_dev(C.c_void_p(col_n.ptr if not inverse else out_ext.ptr), C.c_void_p(out_ext.ptr), C.c_size_t(1),
                                        C.c_size_t(n if not inverse else ne), C.c_size_t(ne), C.c_uint32(k), C.c_uint32(ek),
                                        C.c_int(1 if inverse else 0), C.c_void_p(None)), "coset")
def run():
    t = {}
    t0 = time.perf_counter()
    for _ in range(counts["msm_n"]):
        B.msm_g1_dev(bases, col_n.ptr, n)
    t["msm"] = time.perf_counter() - t0; t0 = time.perf_counter()
    for _ in range(counts["intt_n"]):
        B.ntt_dev(col_n.ptr, k, dom.omega_inv, inverse=True)
    t["intt"] = time.perf_counter() - t0; t0 = time.perf_counter()
    for _ in range(counts["coset_ntt_ext"]):
        coset(False)
    t["coset_ntt"] = time.perf_counter() - t0; t0 = time.perf_counter()
    prog.evaluate_h(ext_ptrs, chal, out_ext.ptr)
    L_.check(lib.ezkl_hip_divide_by_vanishing_dev(C.c_void_p(out_ext.ptr), C.c_uint32(k), C.c_uint32(ek), C.c_void_p(None)), "vanish")
    t["eval_h+vanishing"] = time.perf_counter() - t0; t0 = time.perf_counter()
    coset(True)
    t["inverse_coset_ntt"] = time.perf_counter() - t0
    return t
run()                       # warm-up: tables, twiddles, JIT
B.synchronize()
t = run()
B.synchronize()
total = sum(t.values())
print(json.dumps({"what": "prove-shaped kernel pipeline (NOT a proof)", "k": k, "ext_k": ek, "advice": A, "lookups": L, "perm_chunks": P,
                  "quotient_pieces": Q, "columns": NC, "counts": counts, "seconds": {a: round(b, 5) for a, b in t.items()},
                  "total_seconds": round(total, 5)}))
