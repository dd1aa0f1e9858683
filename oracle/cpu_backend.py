"""CPU backend for ezkl_amd.plonk built on the oracle (TEST INFRASTRUCTURE ONLY).

It implements the same column-handle interface as ezkl_amd.plonk.GpuBackend with numpy arrays as handles and the C
oracle doing the arithmetic, so that (a) the protocol logic can be tested without a GPU and (b) a proof made on the
GPU can be compared BYTE FOR BYTE with the proof the CPU restatement makes from the same witness and randomness."""
import numpy as np
from . import binding as ob
from ezkl_amd.plonk import to_mont, from_mont, point_to_ints, R, ROOT, DELTA


class OracleBackend:
    name = "oracle"

    def __init__(self, params_g, params_g_lagrange, k):
        self.k, self.n = k, 1 << k
        self.g = np.ascontiguousarray(params_g, np.uint64)
        self.gl = np.ascontiguousarray(params_g_lagrange, np.uint64)

    def upload(self, a): return np.array(a, np.uint64, copy=True)
    def download(self, h, n): return h[:n]
    def clone(self, h): return h.copy()
    def commit_lagrange(self, hs): return [point_to_ints(ob.msm(h[: self.n], self.gl)) for h in hs]
    def commit(self, hs): return [point_to_ints(ob.msm(h[: self.n], self.g)) for h in hs]
    def lagrange_to_coeff(self, h): return ob.lagrange_to_coeff(h, self.k)
    def coeff_to_extended(self, h, ext_k): return ob.coeff_to_extended(h[: self.n], self.k, ext_k)
    def extended_to_coeff(self, h, ext_k):
        h[:] = ob.extended_to_coeff(h, ext_k)
        return h
    def divide_by_vanishing(self, h, ext_k): h[:] = ob.divide_by_vanishing(h, self.k, ext_k)
    def eval_program(self, prog, cols, challenges, out):
        code, consts, rots = prog.arrays()
        ch = np.stack([to_mont(c) for c in challenges]) if challenges else np.zeros((1, 4), np.uint64)
        out[:] = ob.eval_program(code, prog.n_intermediates, consts, rots, cols, ch, prog.k, prog.ext_k, previous=out)
    def zeros(self, n): return np.zeros((n, 4), np.uint64)
    def eval_poly(self, h, n, x, offset=0): return from_mont(ob.eval_poly(h[offset:offset + n], to_mont(x)))
    def slice_copy(self, h, offset, n): return h[offset:offset + n].copy()
    def axpy(self, acc, s, h, n): acc[:n] = ob.vec_op("add", acc[:n], ob.vec_scale(h[:n], to_mont(s)))
    def sub_low(self, h, coeffs):
        m = len(coeffs)
        h[:m] = ob.vec_op("sub", h[:m], np.stack([to_mont(c) for c in coeffs]))
    def scale(self, h, s, n): h[:n] = ob.vec_scale(h[:n], to_mont(s))
    def omega_powers(self): return None
    def permutation_product(self, value_cols, sigma_cols, beta, gamma, first_index, z0, omega_col):
        n, w = self.n, pow(ROOT, 1 << (28 - self.k), R)
        V = [[from_mont(x) for x in c] for c in value_cols]
        S = [[from_mont(x) for x in c] for c in sigma_cols]
        num, den = np.empty((n, 4), np.uint64), np.empty((n, 4), np.uint64)
        wi = 1
        for i in range(n):
            a = b = 1
            for j in range(len(V)):
                a = a * (V[j][i] + beta * pow(DELTA, first_index + j, R) % R * wi + gamma) % R
                b = b * (V[j][i] + beta * S[j][i] + gamma) % R
            num[i], den[i] = to_mont(a), to_mont(b)
            wi = wi * w % R
        ratio = ob.vec_op("mul", num, ob.batch_invert(den))
        z = ob.prefix_scan(ratio, "mul", exclusive=True)
        if z0 is not None:
            z = ob.vec_scale(z, to_mont(z0))
        return z
    def lookup_multiplicity(self, inputs, table, usable):
        first = {}
        for i in range(usable):
            first.setdefault(table[i].tobytes(), i)
        m = np.zeros((self.n, 4), np.uint64)
        cnt, missing = {}, 0
        for x in inputs:
            for r in range(usable):
                idx = first.get(x[r].tobytes())
                if idx is None: missing += 1
                else: cnt[idx] = cnt.get(idx, 0) + 1
        for idx, c in cnt.items():
            m[idx] = to_mont(c)
        return m, missing
    def lookup_grand_sum(self, inputs, table, m, beta):
        n = self.n
        b = to_mont(beta)
        acc = np.zeros((n, 4), np.uint64)
        bcol = np.tile(b, (n, 1))
        for x in inputs:
            acc = ob.vec_op("add", acc, ob.batch_invert(ob.vec_op("add", x[:n], bcol)))
        tinv = ob.batch_invert(ob.vec_op("add", table[:n], bcol))
        acc = ob.vec_op("sub", acc, ob.vec_op("mul", tinv, m[:n]))
        return ob.prefix_scan(acc, "add", exclusive=True)
    def set_rows(self, h, start, mont_rows):
        h[start:start + len(mont_rows)] = mont_rows
    def get_row(self, h, i): return from_mont(h[i])
    def kate_div(self, h, z, n):
        h[:n] = ob.kate_div(h[:n], to_mont(z))
        return h
