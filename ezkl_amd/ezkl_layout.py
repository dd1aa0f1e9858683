"""Witness layout (halo2 `synthesize`) for the ezkl circuits this repo proves end to end: which cell holds what, which selector is
on at which row, which cells are copy-constrained.  It follows the reference's region logic for the ops it covers so that the
constraint system produced by ezkl_circuit.py is exercised the way ezkl exercises it (every gate kind used at its real rotation,
second-phase columns filled from post-commitment challenges, constants through the fixed column), without re-implementing
ezkl's 6.8k-line layouts.rs (stays in Rust).

Covered:
  * `EinsumMatmulCircuit` = the reference's criterion bench circuit, /root/reference/benches/accum_einsum_matmul.rs:32-118
    ("ij,jk->ik", configure_einsums with one inner column + two constant cells), laid out by `Einsums::assign_einsum`
    (src/circuit/ops/chip/einsum/mod.rs:96-309): Freivalds' argument -- the output is squashed by random linear combinations
    along its axes (assign_output :311-356, RLCConfig::assign_rlc :785-866), the inputs are reduced in the order of
    reduction_planner.rs:87-205 (RLC "ij,i->j", RLC "jk,k->j", contraction "j,j->" = einsum/layouts.rs:234-318 `dot`), the
    remaining scalar goes through `prod` (:154-231) and is copy-constrained to the squashed output.
One linear coordinate (`einsum_col_coord`) is shared by the six einsum VarTensors; every op occupies the same rows in the
columns it touches.  Column overflow into a second block (assign_einsum_with_duplication) is not needed at the bench sizes and is
refused loudly.
"""
import numpy as np

from . import ezkl_circuit as EC
from . import plonk as P
from .halo2_cs import ConstraintSystem

R = P.R


class Region:
    """cells, selector activations and copy constraints of one synthesis pass"""

    def __init__(self, cs, k):
        self.cs, self.k, self.n = cs, k, 1 << k
        self.usable = self.n - cs.blinding_factors() - 1
        self.advice = {}                               # advice column index -> list of n ints
        self.fixed = {}
        self.activations = [set() for _ in cs.selectors]
        self.copies = []                               # ((kind, col, row), (kind, col, row))
        self.coord = 0                                 # einsum_col_coord
        self.const_cells = {}                          # value -> (fixed col, row)
        self.const_next = 0

    def column(self, col):
        store = self.advice if col.kind == "adv" else self.fixed
        if col.index not in store:
            store[col.index] = [0] * self.n
        return store[col.index]

    def enable(self, selector, row):
        assert row < self.usable
        self.activations[selector.index].add(row)

    def copy(self, a, b):
        if a != b:
            self.copies.append((a, b))

    def constant(self, const_cols, value):
        """a cell of the constant fixed column holding `value` (assign_constant: one cell per distinct value)"""
        value %= R
        if value not in self.const_cells:
            col = const_cols[self.const_next // self.usable]
            row = self.const_next % self.usable
            self.const_next += 1
            self.column(col)[row] = value
            self.const_cells[value] = ("fix", col.index, row)
        return self.const_cells[value]

    def selector_rows(self):
        out = []
        for rows in self.activations:
            a = [False] * self.n
            for r in rows:
                a[r] = True
            out.append(a)
        return out


class Val:
    """a value with the cell it was last assigned to (ValType::PrevAssigned) or the constant it is (ValType::Constant)"""
    __slots__ = ("v", "cell", "const")

    def __init__(self, v, cell=None, const=False):
        self.v, self.cell, self.const = v % R, cell, const


class EinsumMatmulCircuit:
    """benches/accum_einsum_matmul.rs MyCircuit for "ij,jk->ik" with square len x len inputs"""

    def __init__(self, k, length, num_inner_cols=1):
        self.k, self.len, self.w = k, length, num_inner_cols
        dims = {"i": length, "j": length, "k": length}
        # analyze_single_equation (einsum/analysis.rs:74-209) for this equation
        out_red = length * length + length                 # RLC over k for every i, then over i
        in_red = 2 * length * length + length              # RLC ij,i->j ; RLC jk,k->j ; dot j,j->
        self.reduction_length = out_red + in_red
        cs = self.cs = ConstraintSystem()
        self.base = type("Cfg", (), {})()
        self.einsums = EC.Einsums(cs, self.reduction_length, 2, num_inner_cols, k)
        self.const_cols = EC.VarTensor.constant_cols(cs, k, 2)
        cs.chunk_lookups()
        assert all(v.num_blocks() == 1 for v in self.einsums.inputs + self.einsums.outputs), "column overflow (duplication) is not laid out here"
        assert num_inner_cols == 1, "one inner column, as in the reference bench"

    # ---- the pieces of assign_einsum -------------------------------------------------------------------------------
    def _assign(self, region, var, vals, live):
        """region.assign_einsum: vals (list of Val) go to rows coord.. of `var`; a previously assigned value is copy-constrained,
        a constant is copy-constrained to the fixed column.  `live`: fill values (False in the phase that does not own the column)."""
        col = var.inner[0][0]
        store = region.column(col) if live else None
        out = []
        for t, val in enumerate(vals):
            row = region.coord + t
            assert row < region.usable, "einsum column overflow"
            if live:
                store[row] = val.v
            cell = ("adv", col.index, row)
            if val.const:
                region.copy(cell, region.constant(self.const_cols, val.v))
            elif val.cell is not None:
                region.copy(cell, val.cell)
            out.append(Val(val.v, cell))
        return out

    def _rlc(self, region, gate_idx, vals, challenge, rlc_len, phase, cur_phase):
        """RLCConfig::assign_rlc with block width 1: out[0] = c*v0 (init), out[t] = out[t-1]*c + c*v[t] (acc)"""
        g = self.einsums.rlc[gate_idx]
        in_var = [self.einsums.inputs[0], self.einsums.inputs[2]][phase]
        out_var = self.einsums.outputs[1]
        results = []
        c = challenge
        for s in range(0, len(vals), rlc_len):
            chunk = vals[s:s + rlc_len]
            self._assign(region, in_var, chunk, live=(phase == 0 or cur_phase == 1))
            run, acc = [], 0
            if cur_phase == 1:
                for v in chunk:
                    acc = (acc * c + c * v.v) % R
                    run.append(Val(acc))
            else:
                run = [Val(0)] * len(chunk)
            outs = self._assign(region, out_var, run, live=(cur_phase == 1))
            init_s, acc_s = g["selectors"][(phase, 0)]
            region.enable(init_s, region.coord)
            for t in range(1, len(chunk)):
                region.enable(acc_s, region.coord + t)
            results.append(outs[-1])
            region.coord += len(chunk)
        return results

    def synthesize(self, a, b, challenges=None):
        """a, b: len x len integer arrays (values mod r).  challenges=None: first phase (first-phase columns, selectors, copies);
        else the two challenges: everything.  Returns the Region."""
        L = self.len
        cur = 0 if challenges is None else 1
        c0, c1 = (0, 0) if challenges is None else challenges
        region = Region(self.cs, self.k)
        E = self.einsums
        A = [[Val(int(a[i][j])) for j in range(L)] for i in range(L)]
        B = [[Val(int(b[j][kk])) for kk in range(L)] for j in range(L)]
        O = [Val(int(v)) for v in self.matmul(a, b).reshape(-1)]
        # assign_output: RLC along k (challenge 1, first-phase input), then along i (challenge 0, second-phase input)
        inter = self._rlc(region, 1, O, c1, L, 0, cur)
        inter = self._rlc(region, 0, inter, c0, L, 1, cur)
        squashed_out = self._assign(region, E.outputs[1], inter, live=(cur == 1))[0]
        region.coord += 1
        # input reductions (reduction_planner::input_reductions("ij,jk->ik")): axis i, axis k, then the common axis j
        colsA = [A[i][j] for j in range(L) for i in range(L)]          # for every j: the slice A[:, j]
        ra = self._rlc(region, 0, colsA, c0, L, 0, cur)
        rowsB = [B[j][kk] for j in range(L) for kk in range(L)]        # for every j: the slice B[j, :]
        rb = self._rlc(region, 1, rowsB, c1, L, 0, cur)
        # contraction "j,j->": dot of two second-phase vectors (einsum/layouts.rs `dot`, BothSecondPhase)
        sel = E.contraction_selectors
        self._assign(region, E.inputs[2], ra, live=(cur == 1))
        self._assign(region, E.inputs[3], rb, live=(cur == 1))
        acc, run = 0, []
        for x, y in zip(ra, rb):
            acc = (acc + x.v * y.v) % R
            run.append(Val(acc))
        outs = self._assign(region, E.outputs[1], run, live=(cur == 1))
        region.enable(sel[((EC.DOTINIT, EC.BOTH_SECOND), 0, 0)], region.coord)
        for t in range(1, L):
            region.enable(sel[((EC.DOT, EC.BOTH_SECOND), 0, 0)], region.coord + t)
        region.coord += L
        # prod over the remaining scalars (one), second phase: CUMPRODINIT
        self._assign(region, E.inputs[2], [outs[-1]], live=(cur == 1))
        squashed_in = self._assign(region, E.outputs[1], [Val(outs[-1].v)], live=(cur == 1))[0]
        region.enable(sel[((EC.CUMPRODINIT, EC.SECOND_PHASE), 0, 0)], region.coord)
        region.coord += 1
        region.copy(squashed_in.cell, squashed_out.cell)
        return region

    @staticmethod
    def matmul(a, b):
        """the einsum output the prover witnesses: exact integer product of quantised (small) inputs"""
        a64, b64 = np.asarray(a, np.int64), np.asarray(b, np.int64)
        bound = int(np.abs(a64).max(initial=0)) * int(np.abs(b64).max(initial=0)) * a64.shape[1]
        assert bound < 1 << 62, "inputs too large for the exact int64 product"
        return a64 @ b64

    # ---- what keygen / the prover need ------------------------------------------------------------------------------
    def keygen_inputs(self, a, b):
        """-> (plonk.ConstraintSystem, fixed columns (ints), copies over cs.perm positions, rows used)"""
        region = self.synthesize(a, b)
        n = 1 << self.k
        sel_cols = self.cs.compress_selectors(region.selector_rows())
        cs = self.cs.to_plonk(self.k)
        n_pre = cs.n_fixed - len(sel_cols)
        fixed = [region.fixed.get(c, [0] * n) for c in range(n_pre)] + sel_cols
        pos = {kc: i for i, kc in enumerate(cs.perm)}
        copies = [((pos[(x[0], x[1])], x[2]), (pos[(y[0], y[1])], y[2])) for x, y in region.copies]
        return cs, fixed, copies, region.coord

    def advice_fn(self, a, b, n_advice):
        """the per-phase witness callback of create_proof: phase 0 -> first-phase columns, phase 1 (with the challenges) -> the
        second-phase columns"""
        n = 1 << self.k
        def fn(phase, challenges):
            region = self.synthesize(a, b, None if phase == 0 else tuple(challenges[:2]))
            return {c.index: region.advice.get(c.index, [0] * n) for c in self.cs.advice if c.phase == phase}
        return fn


def ints_to_mont(colv):
    """list of n canonical ints -> (n, 4) u64 Montgomery array"""
    M = 1 << 256
    return np.frombuffer(b"".join((v * M % R).to_bytes(32, "little") for v in colv), np.uint64).reshape(len(colv), 4).copy()
