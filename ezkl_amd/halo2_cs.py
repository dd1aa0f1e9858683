"""halo2's `ConstraintSystem` as data: the configure-time API ezkl's circuits are written against, restated so that the
constraint system of an ezkl circuit (columns, gates, lookups, permutation, QUERY ORDER, selector compression, lookup
chunking, degree, blinding factors) can be produced here and handed to the provers (`plonk.ConstraintSystem`, the EZCS
blob of include/ezkl_prover.h).

What it restates ([UPSTREAM] zkonduit/halo2 @ 01c8884, halo2_proofs/src/plonk/circuit.rs + circuit/compress_selectors.rs,
not on disk; semantics recalled and PINNED on the reference's fixtures, tests/test_ezkl_circuit.py):
  * query bookkeeping: `advice_queries` / `fixed_queries` / `instance_queries` hold (column, rotation) in order of FIRST
    query, `enable_equality` queries the column at rotation 0 (this order is the order of the evaluations in a proof:
    pinned on /root/reference/tests/assets/proof.json, whose 38 fixed and 32 sigma evaluations match pk.key's polynomials
    at the recovered challenge x in exactly this order);
  * `blinding_factors() = max(3, max #queries of an advice column) + 2`;
  * mv-lookup `lookup` / `lookup_any` arguments grouped by table expression, `chunk_lookups()` splitting the inputs of a
    table into arguments of degree <= max(max gate degree, max single-lookup degree) (pinned: the fixture's 40 range-check
    lookups become exactly the 35 arguments its proof commits to, and cs.degree() = 7);
  * `compress_selectors`: simple selectors that are never active together share one fixed column with values 1..len and
    are replaced by q * prod_{j != i} (j - q); complex / unused selectors get a column each (pinned: from the fixture vk's
    80 selector activations it reproduces the 33 selector-derived fixed columns of pk.key bit for bit).
"""
from . import plonk as P
from .plonk import Expr, R

CUR = 0


def sel_expr(index):
    return Expr(("sel", index))


class Column:
    __slots__ = ("kind", "index", "phase", "blinded")

    def __init__(self, kind, index, phase=0, blinded=True):
        self.kind, self.index, self.phase, self.blinded = kind, index, phase, blinded

    def __repr__(self):
        return "%s%d" % (self.kind, self.index)


class Selector:
    __slots__ = ("index", "simple")

    def __init__(self, index, simple):
        self.index, self.simple = index, simple


def degree(e):
    op = e.node[0]
    if op in ("const", "chal"): return 0
    if op in ("adv", "fix", "inst", "sel"): return 1
    if op == "neg": return degree(e.node[1])
    if op in ("add", "sub"): return max(degree(e.node[1]), degree(e.node[2]))
    return degree(e.node[1]) + degree(e.node[2])


def identifier(e):
    """Expression::identifier (the key of halo2's lookups_map: lookups with the same table expressions share a tracker)"""
    op = e.node[0]
    if op == "const": return "0x%064x" % e.node[1]
    if op == "sel": return "selector[%d]" % e.node[1]
    if op == "fix": return "fixed[%d][%d]" % (e.node[1], e.node[2])
    if op == "adv": return "advice[%d][%d]" % (e.node[1], e.node[2])
    if op == "inst": return "instance[%d][%d]" % (e.node[1], e.node[2])
    if op == "chal": return "challenge[%d]" % e.node[1]
    if op == "neg": return "(-" + identifier(e.node[1]) + ")"
    sym = {"add": "+", "sub": "-", "mul": "*"}[op]
    return "(" + identifier(e.node[1]) + sym + identifier(e.node[2]) + ")"


def simple_selectors(e, sels, out):
    op = e.node[0]
    if op == "sel":
        if sels[e.node[1]].simple:
            out.add(e.node[1])
    elif op in ("neg", "add", "sub", "mul"):
        for c in e.node[1:]:
            simple_selectors(c, sels, out)
    return out


def substitute(e, repl, memo):
    """replace ("sel", i) nodes by repl[i]; shared sub-expressions stay shared"""
    key = id(e)
    if key in memo:
        return memo[key]
    op = e.node[0]
    if op == "sel":
        r = repl[e.node[1]]
    elif op == "neg":
        c = substitute(e.node[1], repl, memo)
        r = e if c is e.node[1] else Expr(("neg", c))
    elif op in ("add", "sub", "mul"):
        a, b = substitute(e.node[1], repl, memo), substitute(e.node[2], repl, memo)
        r = e if (a is e.node[1] and b is e.node[2]) else Expr((op, a, b))
    else:
        r = e
    memo[key] = r
    return r


class VirtualCells:
    """the `meta` handed to create_gate / lookup closures: every query registers (column, rotation) in first-query order"""

    def __init__(self, cs):
        self.cs = cs

    def query_selector(self, s): return sel_expr(s.index)
    def query_advice(self, col, rot=0):
        self.cs._query(col, rot)
        return P.adv(col.index, rot)
    def query_fixed(self, col, rot=0):
        self.cs._query(col, rot)
        return P.fix(col.index, rot)
    def query_instance(self, col, rot=0):
        self.cs._query(col, rot)
        return P.inst(col.index, rot)
    def query_any(self, col, rot=0):
        return {"adv": self.query_advice, "fix": self.query_fixed, "inst": self.query_instance}[col.kind](col, rot)
    def query_challenge(self, ch): return P.chal(ch)


class Gate:
    __slots__ = ("name", "polys")

    def __init__(self, name, polys):
        self.name, self.polys = name, polys


class LookupArgument:
    __slots__ = ("name", "table", "inputs")

    def __init__(self, name, table, inputs):
        self.name, self.table, self.inputs = name, list(table), [list(i) for i in inputs]

    def required_degree(self):
        """mv_lookup::Argument::required_degree: (1 - (l_last + l_blind)) * tau * prod(phi_i) * (phi(wX) - phi(X))"""
        ins = sum(max(degree(e) for e in t) for t in self.inputs)
        tab = max(degree(e) for e in self.table)
        return max(3, tab + ins + 2)


class ConstraintSystem:
    def __init__(self):
        self.advice, self.fixed, self.instance = [], [], []
        self.selectors = []
        self.challenges = []                          # phase after which each challenge is usable
        self.gates = []
        self.advice_queries, self.fixed_queries, self.instance_queries = [], [], []
        self._qidx = {}
        self.num_advice_queries = []
        self.permutation = []                         # columns, in enable_equality order
        self._perm_set = set()
        self.lookups_map = {}                         # table identifier -> LookupArgument (all inputs)
        self.lookups = []                             # after chunk_lookups()
        self.constants = []
        self.minimum_degree = None

    # ---- columns
    def advice_column(self, phase=0, blinded=True):
        c = Column("adv", len(self.advice), phase, blinded)
        self.advice.append(c)
        self.num_advice_queries.append(0)
        return c
    def advice_column_in(self, phase): return self.advice_column(phase)
    def unblinded_advice_column(self): return self.advice_column(0, False)
    def fixed_column(self):
        c = Column("fix", len(self.fixed))
        self.fixed.append(c)
        return c
    lookup_table_column = fixed_column
    def instance_column(self):
        c = Column("inst", len(self.instance))
        self.instance.append(c)
        return c
    def selector(self):
        s = Selector(len(self.selectors), True)
        self.selectors.append(s)
        return s
    def complex_selector(self):
        s = Selector(len(self.selectors), False)
        self.selectors.append(s)
        return s
    def challenge_usable_after(self, phase):
        self.challenges.append(phase)
        return len(self.challenges) - 1

    def _query(self, col, rot):
        key = (col.kind, col.index, rot)
        if key in self._qidx:
            return self._qidx[key]
        lst = {"adv": self.advice_queries, "fix": self.fixed_queries, "inst": self.instance_queries}[col.kind]
        self._qidx[key] = len(lst)
        lst.append((col.index, rot))
        if col.kind == "adv":
            self.num_advice_queries[col.index] += 1
        return self._qidx[key]

    def enable_equality(self, col):
        self._query(col, CUR)
        if (col.kind, col.index) not in self._perm_set:
            self._perm_set.add((col.kind, col.index))
            self.permutation.append((col.kind, col.index))
    def enable_constant(self, col):
        if col.index not in self.constants:
            self.constants.append(col.index)
            self.enable_equality(col)

    # ---- gates / lookups
    def create_gate(self, name, f):
        """f(meta) -> (selector expression or None, [constraint polynomials]) -- Constraints::with_selector"""
        sel, polys = f(VirtualCells(self))
        polys = [p if sel is None else sel * p for p in polys]
        assert polys, "gates must contain at least one constraint"
        self.gates.append(Gate(name, polys))

    def lookup(self, name, f):
        """f(meta) -> [(input expression, table column)]"""
        meta = VirtualCells(self)
        pairs = f(meta)
        ins, tab = [], []
        for e, tcol in pairs:
            assert not simple_selectors(e, self.selectors, set()), "expression containing simple selector supplied to lookup argument"
            tab.append(meta.query_fixed(tcol, CUR))
            ins.append(e)
        self._track(name, ins, tab)

    def lookup_any(self, name, f):
        """f(meta) -> [(input expression, table expression)]"""
        pairs = f(VirtualCells(self))
        for e, t in pairs:
            assert not simple_selectors(e, self.selectors, set()) and not simple_selectors(t, self.selectors, set())
        self._track(name, [e for e, _ in pairs], [t for _, t in pairs])

    def _track(self, name, ins, tab):
        key = "".join(identifier(t) for t in tab)
        if key in self.lookups_map:
            self.lookups_map[key].inputs.append(list(ins))
        else:
            self.lookups_map[key] = LookupArgument(name, tab, [ins])

    # ---- derived quantities
    def blinding_factors(self):
        return max(3, max(self.num_advice_queries, default=1)) + 2

    def max_gate_degree(self):
        return max((degree(p) for g in self.gates for p in g.polys), default=0)

    def degree(self):
        d = 3                                          # the permutation argument: l_last * (z^2 - z)
        for l in (self.lookups if self.lookups else self.lookups_map.values()):
            d = max(d, l.required_degree())
        d = max(d, self.max_gate_degree())
        return max(d, self.minimum_degree or 1)

    def chunk_lookups(self):
        """split the inputs of every table into arguments whose degree stays within the degree the circuit needs anyway
        (max gate degree, or the degree of the widest single-input lookup).  Tables iterate in BTreeMap (string) order."""
        if not self.lookups_map:
            return self
        single = 0
        for l in self.lookups_map.values():
            tab = max(degree(e) for e in l.table)
            base = max(3, tab + 2)
            single = max(single, base + max(max(degree(e) for e in t) for t in l.inputs))
        self.minimum_degree = max(self.minimum_degree or 1, max(self.max_gate_degree(), single))
        out = []
        for key in sorted(self.lookups_map):
            l = self.lookups_map[key]
            args = [LookupArgument(l.name, l.table, [])]
            for t in l.inputs:
                dt = max(degree(e) for e in t)
                for a in args:
                    if a.required_degree() + dt <= self.minimum_degree:
                        a.inputs.append(list(t))
                        break
                else:
                    args.append(LookupArgument(l.name, l.table, [t]))
            out += args
        self.lookups = out
        return self

    # ---- selectors -> fixed columns (keygen)
    def compress_selectors(self, activations, grouping=None):
        """activations: one boolean row-vector per selector (list / numpy array; None = never enabled).  Returns the fixed-column
        assignments (numpy int64 arrays of small integers, one per NEW fixed column, in allocation order); gates and lookups are
        rewritten in place.  grouping: optional {first selector of a combination: [selectors]} that overrides the greedy choice
        (must be a valid one: disjoint activations, degree bound) -- used to rebuild the constraint system of a key whose selector
        activations are not known but whose combinations are (tests/test_evm_verifier.py)."""
        import numpy as np
        assert len(activations) == len(self.selectors)
        nsel = len(self.selectors)
        n = next((len(a) for a in activations if a is not None), 0)
        act = [None if a is None else np.asarray(a, dtype=bool) for a in activations]
        degrees = [0] * nsel
        for g in self.gates:
            for p in g.polys:
                ss = simple_selectors(p, self.selectors, set())
                assert len(ss) <= 1, "at most one simple selector per constraint"
                for s in ss:
                    degrees[s] = max(degrees[s], degree(p))
        max_degree = self.degree()
        meta = VirtualCells(self)
        new_cols, repl, smap = [], [None] * nsel, [None] * nsel

        def alloc():
            col = self.fixed_column()
            return col, meta.query_fixed(col, CUR)

        rest = []
        for i in range(nsel):                          # complex selectors and selectors of no gate: a column of their own
            if degrees[i] == 0:
                col, q = alloc()
                new_cols.append(np.zeros(n, np.int64) if act[i] is None else act[i].astype(np.int64))
                repl[i], smap[i] = q, col.index
            else:
                rest.append(i)
        added = set()
        for pos, i in enumerate(rest):
            if i in added:
                continue
            added.add(i)
            assert degrees[i] <= max_degree
            d = degrees[i] - 1
            combo = [i]
            union = None if act[i] is None else act[i].copy()     # rows where some selector of the combination is on
            if grouping is not None:
                combo = list(grouping[i])
                assert combo[0] == i and max(degrees[t] - 1 for t in combo) + len(combo) <= max_degree
                for t in combo[1:]:
                    assert t not in added and (union is None or act[t] is None or not (union & act[t]).any()), "invalid selector grouping"
                    added.add(t)
                    if act[t] is not None:
                        union = act[t].copy() if union is None else (union | act[t])
            for j in (rest[pos + 1:] if grouping is None else ()):
                if d + len(combo) == max_degree:
                    break
                if j in added:
                    continue
                if union is not None and act[j] is not None and (union & act[j]).any():
                    continue
                nd = max(d, degrees[j] - 1)
                if nd + len(combo) + 1 > max_degree:
                    continue
                d = nd
                combo.append(j)
                added.add(j)
                if act[j] is not None:
                    union = act[j].copy() if union is None else (union | act[j])
            col, q = alloc()
            assign = np.zeros(n, np.int64)
            for root, s in enumerate(combo, start=1):
                e = q
                for other in range(1, len(combo) + 1):
                    if other != root:
                        e = e * (P.const(other) - q)
                repl[s], smap[s] = e, col.index
                if act[s] is not None:
                    assign[act[s]] = root
            new_cols.append(assign)
        self.selector_map = smap
        self._replace_selectors(repl)
        return new_cols

    def directly_convert_selectors_to_fixed(self, activations):
        meta = VirtualCells(self)
        repl, cols, smap = [], [], []
        for i in range(len(self.selectors)):
            col = self.fixed_column()
            repl.append(meta.query_fixed(col, CUR))
            smap.append(col.index)
            cols.append([0] * 0 if activations[i] is None else [1 if b else 0 for b in activations[i]])
        self.selector_map = smap
        self._replace_selectors(repl)
        return cols

    def _replace_selectors(self, repl):
        memo = {}
        for g in self.gates:
            g.polys = [substitute(p, repl, memo) for p in g.polys]
        for l in list(self.lookups_map.values()) + list(self.lookups):
            l.inputs = [[substitute(e, repl, memo) for e in t] for t in l.inputs]
            l.table = [substitute(e, repl, memo) for e in l.table]

    # ---- hand-over to the provers
    def to_plonk(self, k):
        """the prover-side description: selectors must have been converted (compress_selectors), lookups chunked"""
        lookups = self.lookups if (self.lookups or not self.lookups_map) else self.chunk_lookups().lookups
        gates = [p for g in self.gates for p in g.polys]
        cs = P.ConstraintSystem(k, len(self.advice), len(self.fixed), gates, self.permutation,
                                lookups=[(l.inputs, l.table) for l in lookups], n_instance=len(self.instance),
                                advice_phase=[c.phase for c in self.advice], n_challenges=len(self.challenges),
                                query_order=(list(self.advice_queries), list(self.fixed_queries), list(self.instance_queries)),
                                blinding=self.blinding_factors(), minimum_degree=self.minimum_degree,
                                unblinded=[c.index for c in self.advice if not c.blinded], n_selectors=len(self.selectors))
        assert cs.degree == self.degree(), (cs.degree, self.degree())
        return cs
