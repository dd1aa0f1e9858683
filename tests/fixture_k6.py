"""The reference's fixture circuit (tests/assets/{settings.json, pk.key, vk.key, witness.json, proof.json}, copied to tests/golden
by make_golden.py): a 1l_relu model (Gemm 3->4 + ReLU, scale 0) at k = 6 -- the only complete ezkl circuit + key + proof in the
reference tree.  This module rebuilds its constraint system with the generator (ezkl_amd/ezkl_circuit.py), takes the fixed
columns, selector activations and the permutation from the reference's OWN pk.key, and builds a satisfying witness for the
reference's own input (witness.json: x = [2, 1, 1]).

The witness layout was read off pk.key: selector activations give the op of every row, the permutation's cycles give the data
flow.  Linear cell L of the output VarTensor is (block L // 116, inner column L % 2, row (L % 116) // 2)
(/root/reference/src/tensor/var.rs:319-327).  Every decomposition (src/circuit/ops/layouts.rs:6321 `decompose`, base 128, 2 legs)
writes [sign, d1, d0] to three consecutive cells; relu(x) = x * [sign(x) == 1] with an is-zero gadget on (sign - 1) whose inverse hint
sits at rows 47-48 of the second input VarTensor."""
import json
import os

import numpy as np

from ezkl_amd import codecs, ezkl_circuit as EC, plonk as P

R = P.R
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N_PERM, N_SEL, K = 32, 80, 6


def col_ints(a):
    return [int.from_bytes(a[i].tobytes(), "little") * P.RINV % R for i in range(a.shape[0])]


def load():
    """-> dict(gc (GraphConfig with selectors compressed), cs (plonk.ConstraintSystem), pk (parsed pk.key), fixed (38 int columns),
    selector_columns (the 33 generated ones), table_columns)"""
    st = EC.GraphSettings.from_json(json.load(open(os.path.join(G, "settings_k6.json"))))
    gc = EC.GraphConfig(st)
    pk = codecs.read_pk(open(os.path.join(G, "pk_k6.key"), "rb").read(), N_PERM, N_SEL)
    pre = dict(n_fixed=len(gc.cs.fixed), n_selectors=len(gc.cs.selectors), n_lookups=len(gc.cs.lookups), degree=gc.cs.degree())
    sel_cols = [c.tolist() for c in gc.cs.compress_selectors(list(pk["vk"]["selectors"]))]
    cs = gc.cs.to_plonk(K)
    return dict(gc=gc, cs=cs, pk=pk, pre=pre, selector_columns=sel_cols, table_columns=gc.table_columns(),
                fixed=[col_ints(p) for p in pk["fixed_values"]])


def copy_cycles(pk, n_perm=N_PERM, k=K):
    """decode pk.key's `permutations` (sigma[c][r] = delta^c' * omega^r') into the copy-constraint cycles"""
    n, w = 1 << k, P.omega(k)
    lab = {}
    for c in range(n_perm):
        d = pow(P.DELTA, c, R)
        for r in range(n):
            lab[d * pow(w, r, R) % R] = (c, r)
    sig = [col_ints(p) for p in pk["permutations"]]
    nxt = {(c, r): lab[sig[c][r]] for c in range(n_perm) for r in range(n)}
    seen, cycles = set(), []
    for c in range(n_perm):
        for r in range(n):
            cyc, cur = [], (c, r)
            while cur not in seen:
                seen.add(cur); cyc.append(cur); cur = nxt[cur]
            if len(cyc) > 1:
                cycles.append(cyc)
    return cycles


def copies_of(cycles):
    return [(cyc[i], cyc[i + 1]) for cyc in cycles for i in range(len(cyc) - 1)]


_NONACC, _ACC = ["ADD", "SUB", "MULT"], ["DOTINIT", "DOT", "CUMPROD", "CUMPRODINIT", "SUM", "SUMINIT"]     # selector creation order, chip.rs:343-359


def witness(fx, x=(2, 1, 1)):
    """advice columns (30 lists of 64 ints) + instance column for input x, by forward evaluation of the layout"""
    pk, fixed = fx["pk"], fx["fixed"]
    n, u, W = 1 << K, fx["cs"].usable, 2
    act = pk["vk"]["selectors"]
    parent = {}
    def find(c):
        while parent.setdefault(c, c) != c:
            parent[c] = parent[parent[c]]; c = parent[c]
        return c
    for a, b in copies_of(copy_cycles(pk)):
        parent[find(a)] = find(b)
    val = {}
    def put(cell, v): val[find(cell)] = v % R
    def get(cell): return val.get(find(cell), 0)
    def coord(L):
        blk = L // (u * W)
        return (20 + W * blk + L % W, (L - blk * u * W) // W)
    def decomp(v):
        s = (v > 0) - (v < 0)
        return [s % R, abs(v) // 128, abs(v) % 128]
    wts = [[0, -1, 0], [0, -1, 0], [0, 0, -1], [0, 0, 0]]      # round(dense.weight), round(dense.bias) of tests/assets/network.onnx at scale 0
    bias = [0, 1, 0, 0]
    z = [sum(wts[i][j] * x[j] for j in range(3)) + bias[i] for i in range(4)]
    o = [max(0, t) for t in z]
    put((0, 0), x[0]); put((1, 0), x[1]); put((0, 1), x[2])
    for i, v in enumerate(x):
        for t, h in enumerate(decomp(v)): put(coord(3 * i + t), h)
    for i in range(4):
        put((10, 15 + 2 * i), wts[i][0]); put((11, 15 + 2 * i), wts[i][1]); put((10, 16 + 2 * i), wts[i][2])
    for i in range(4):
        put((10 + i % 2, 23 + i // 2), bias[i])
    for i, v in enumerate(z):
        for t, h in enumerate(decomp(v)): put(coord(50 + 3 * i + t), h)
        s = (v > 0) - (v < 0)
        put((10 + i % 2, 47 + i // 2), pow((s - 1) % R, -1, R) if s != 1 else 0)
    for base in (114, 154):                                   # the ReLU outputs are decomposed twice (range check of op output, of the public output)
        for i, v in enumerate(o):
            for t, h in enumerate(decomp(v)): put(coord(base + 3 * i + t), h)
    for r in range(n):
        if (30, r) in parent:
            put((30, r), fixed[0][r])
    for blk in range(5):
        for row in range(u):
            for s in range(60):
                if not act[s][row]:
                    continue
                if s < 30:
                    i, rem = divmod(s, 6); j, o_ = divmod(rem, 3); op = _NONACC[o_]
                else:
                    i, o_ = divmod(s - 30, 6); j, op = 0, _ACC[o_]
                if i != blk:
                    continue
                a = [get((2 * i + t, row)) for t in range(W)]
                b = [get((10 + 2 * i + t, row)) for t in range(W)]
                prev = get((20 + 2 * i, row - 1)) if row else 0
                out = {"ADD": lambda: a[j] + b[j], "SUB": lambda: a[j] - b[j], "MULT": lambda: a[j] * b[j],
                       "DOTINIT": lambda: a[0] * b[0] + a[1] * b[1], "DOT": lambda: prev + a[0] * b[0] + a[1] * b[1],
                       "SUMINIT": lambda: b[0] + b[1], "SUM": lambda: prev + b[0] + b[1],
                       "CUMPRODINIT": lambda: b[0] * b[1], "CUMPROD": lambda: prev * b[0] * b[1]}[op]()
                put((20 + 2 * i + j, row), out)
    # range check (0,127) spans three 56-row table columns: the `index` cell beside a checked value selects its column
    # (src/circuit/ops/chip.rs:864-956 synthetic selector; layouts.rs range_check assigns table.get_col_index)
    for blk in range(5):
        for j in range(W):
            s = 70 + 2 * blk + j
            for row in range(u):
                if act[s][row]:
                    put((10 + 2 * blk + j, row), get((2 * blk + j, row)) // 56)
    adv = [[get((c, r)) if r < u else 0 for r in range(n)] for c in range(30)]
    inst = [get((31, r)) for r in range(4)]
    return adv, [inst], o


def mont_cols(cols):
    return [np.stack([P.to_mont(v) for v in c]) for c in cols]
