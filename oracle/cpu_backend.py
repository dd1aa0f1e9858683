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
    def _omega_col(self):
        if getattr(self, "_wcol", None) is None:
            w = pow(ROOT, 1 << (28 - self.k), R)
            self._wcol = ob.prefix_scan(np.tile(to_mont(w), (self.n, 1)), "mul", exclusive=True)       # omega^i
        return self._wcol

    def permutation_product(self, value_cols, sigma_cols, beta, gamma, first_index, z0, omega_col):
        """z[i] = prod_{r<i} prod_j (v_j + beta delta^j omega^r + gamma) / (v_j + beta sigma_j + gamma), all O(n) steps in the C oracle"""
        n = self.n
        wcol, gcol = self._omega_col(), np.tile(to_mont(gamma), (n, 1))
        num = den = None
        for j in range(len(value_cols)):
            v = ob.vec_op("add", value_cols[j][:n], gcol)
            a = ob.vec_op("add", v, ob.vec_scale(wcol, to_mont(beta * pow(DELTA, first_index + j, R) % R)))
            b = ob.vec_op("add", v, ob.vec_scale(sigma_cols[j][:n], to_mont(beta)))
            num = a if num is None else ob.vec_op("mul", num, a)
            den = b if den is None else ob.vec_op("mul", den, b)
        ratio = ob.vec_op("mul", num, ob.batch_invert(den))
        z = ob.prefix_scan(ratio, "mul", exclusive=True)
        if z0 is not None:
            z = ob.vec_scale(z, to_mont(z0))
        return z
    def lookup_multiplicity(self, inputs, table, usable):
        """m[first row holding t] = number of input cells equal to t (rows compared as 32-byte strings, numpy sort / search)"""
        tv = np.ascontiguousarray(table[:usable]).view("V32").reshape(-1)
        uniq, first = np.unique(tv, return_index=True)
        m = np.zeros(self.n, np.int64)
        missing = 0
        for x in inputs:
            xv = np.ascontiguousarray(x[:usable]).view("V32").reshape(-1)
            pos = np.searchsorted(uniq, xv)
            pos[pos >= len(uniq)] = 0
            hit = uniq[pos] == xv
            missing += int((~hit).sum())
            np.add.at(m, first[pos[hit]], 1)
        out = np.zeros((self.n, 4), np.uint64)
        nz = np.nonzero(m)[0]
        for i in nz:
            out[i] = to_mont(int(m[i]))
        return out, missing
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


class DistOracleBackend(__import__("ezkl_amd.plonk", fromlist=["ColumnShardMixin"]).ColumnShardMixin, OracleBackend):
    """The multi-rank decomposition on CPU (TEST INFRASTRUCTURE: gloo, world 2 / 4): MSMs sharded by points, NTTs by columns, the
    sweep by rows with the all-to-all of ezkl_amd/dist.py -- the oracle does the arithmetic of every shard, so what is tested is the
    sharding, the exchanges and the bookkeeping: the ranks must emit the single-rank proof."""
    name = "oracle-dist"

    def __init__(self, params_g, params_g_lagrange, k, dist, device):
        from ezkl_amd import dist as D
        OracleBackend.__init__(self, params_g, params_g_lagrange, k)
        self.D = D
        self._dist_init(dist, device)
        self.lo, self.hi = D.shard_range(self.n, self.rank, self.world)

    def _commit(self, bases, hs):
        part = np.stack([ob.msm(h[self.lo:self.hi], bases[self.lo:self.hi]) for h in hs])
        full = self.D.fold_columns(self.D.allgather_points(part, self.dist, self.device))
        return [point_to_ints(p) for p in full]
    def commit_lagrange(self, hs): return self._commit(self.gl, hs) if hs else []
    def commit(self, hs): return self._commit(self.g, hs) if hs else []
    # row-shard primitives
    def window(self, h, start, length, total):
        return h[start:start + length] if start + length <= total else np.concatenate([h[start:total], h[: start + length - total]])
    def eval_rows(self, sub, handles, challenges, out, lo, hi):
        code, consts, rots = sub.arrays()
        ch = np.stack([to_mont(c) for c in challenges]) if challenges else np.zeros((1, 4), np.uint64)
        out[lo:hi] = ob.eval_program(code, sub.n_intermediates, consts, rots, [np.ascontiguousarray(h) for h in handles], ch, sub.k, sub.ext_k,
                                     previous=np.ascontiguousarray(out[lo:hi]))
    def gather_rows(self, out, lo, hi, total):
        parts = self.D.allgather_array(np.ascontiguousarray(out[lo:hi]), self.dist, self.device)
        rows = hi - lo
        for r, p_ in enumerate(parts):
            out[r * rows:(r + 1) * rows] = p_
