"""The parts of the ezkl gate set the fixture circuit does not exercise, each built by the generator (ezkl_amd/ezkl_circuit.py), given a
hand-laid witness, checked by the MockProver and PROVED on the CPU-oracle backend + pairing verifier:
  * a static lookup table of a nonlinearity over several table columns (BaseConfig::configure_lookup, chip.rs:452-615, table.rs:161-204):
    two-column (input, output) tuples, theta-compressed, synthetic selector in the index column;
  * dynamic lookups and shuffles (configure_dynamic_lookup / configure_shuffles, chip.rs:619-833): 4-tuple lookup_any arguments whose
    TABLE side is advice x selector."""
import numpy as np
import pytest

from ezkl_amd import ezkl_circuit as EC, ezkl_layout as EL, plonk as P
from oracle import mock_prover as MP

R = P.R


def _prove_and_verify(cs, fixed, adv, copies, golden_srs, inst=()):
    from oracle.cpu_backend import OracleBackend
    from oracle import verifier as V
    from test_plonk import setup
    be = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], cs.k)
    pk, vk = P.keygen(cs, be, [EL.ints_to_mont(f) for f in fixed], copies)
    proof = P.create_proof(pk, be, [EL.ints_to_mont(a) for a in adv], P.Rng(4), instances=inst)
    g1, g2, s_g2 = setup(golden_srs)
    return V.verify(vk, g1, g2, s_g2, proof, instances=inst), vk, proof


def test_static_lookup_table_over_three_columns(golden_srs):
    k, n = 6, 64
    relu6 = lambda x: min(max(x, 0), 6)
    st = EC.GraphSettings(k, 1, 40, total_const_size=0, required_lookups=[("relu6", relu6)], lookup_range=(-64, 63))
    gc = EC.GraphConfig(st)
    cs0 = gc.cs
    table = gc.base.static_tables["relu6"]
    assert len(table.table_inputs) == 3 and table.col_size == 56              # 128 values over 56-row columns
    assert len(cs0.lookups) == 3 and cs0.degree() == 7                        # one argument per table column, input degree 4
    # witness: 20 values through the lookup on rows 0..19 of block 0 (input VarTensor 0, index VarTensor 1, output VarTensor 2)
    xs = [-64, -1, 0, 1, 5, 6, 7, 63, -8, -9, 47, 48, 49, -63, 3, 2, 55, 56, -57, 10]
    adv = [[0] * n for _ in cs0.advice]
    act = [[False] * n for _ in cs0.selectors]
    sel = gc.base.static_selectors[("relu6", 0, 0)]
    a_in, a_idx, a_out = gc.advices[0].inner[0][0].index, gc.advices[1].inner[0][0].index, gc.advices[2].inner[0][0].index
    for r, x in enumerate(xs):
        adv[a_in][r], adv[a_out][r], adv[a_idx][r] = x % R, relu6(x) % R, (x - (-64)) // table.col_size
        act[sel.index][r] = True
    sel_cols = [c.tolist() for c in cs0.compress_selectors(act)]
    cs = cs0.to_plonk(k)
    tabs = gc.table_columns()
    fixed = [tabs.get(c) or [0] * n for c in range(cs.n_fixed - len(sel_cols))] + sel_cols
    assert MP.check(cs, adv, fixed) == []
    bad = [list(c) for c in adv]; bad[a_out][3] = 5                           # relu6(1) != 5
    assert any("lookup" in f for f in MP.check(cs, bad, fixed))
    bad = [list(c) for c in adv]; bad[a_idx][7] = 0                           # 63 lives in the third table column, not the first
    assert MP.check(cs, bad, fixed)
    ok, _, proof = _prove_and_verify(cs, fixed, adv, [], golden_srs)
    assert ok and len(proof) > 0


@pytest.mark.parametrize("kind", ["dynamic_lookup", "shuffle"])
def test_dynamic_lookup_and_shuffle_arguments(golden_srs, kind):
    k, n = 6, 64
    kw = dict(num_dynamic_lookups=1, total_dynamic_col_size=10) if kind == "dynamic_lookup" else dict(num_shuffles=1, total_shuffle_col_size=10)
    st = EC.GraphSettings(k, 1, 40, total_const_size=0, **kw)
    gc = EC.GraphConfig(st)
    cs0 = gc.cs
    assert len(cs0.advice) == 6 and len(cs0.lookups) == 1                      # 3 model VarTensors + 3 single-column table VarTensors
    tab_sel = (gc.base.dynamic_table_selectors if kind == "dynamic_lookup" else gc.base.shuffle_output_selectors)[0]
    in_sel = (gc.base.dynamic_lookup_selectors if kind == "dynamic_lookup" else gc.base.shuffle_input_selectors)[(0, (0, 0))]
    rng = np.random.default_rng(3)
    rows = [tuple(int(v) for v in rng.integers(1, 1000, 3)) for _ in range(10)]           # the table / the shuffled output: 10 triples
    picks = [rows[i] for i in ([3, 3, 0, 9, 5, 5, 5, 1] if kind == "dynamic_lookup" else rng.permutation(10))]
    adv = [[0] * n for _ in cs0.advice]
    act = [[False] * n for _ in cs0.selectors]
    lk_cols = [gc.advices[t].inner[0][0].index for t in range(3)]
    tb_cols = [gc.advices[3 + t].inner[0][0].index for t in range(3)]
    for r, tri in enumerate(rows):
        for t in range(3): adv[tb_cols[t]][r] = tri[t]
        act[tab_sel.index][r] = True
    for r, tri in enumerate(picks):
        for t in range(3): adv[lk_cols[t]][20 + r] = tri[t]
        act[in_sel.index][20 + r] = True
    sel_cols = [c.tolist() for c in cs0.compress_selectors(act)]
    cs = cs0.to_plonk(k)
    fixed = [[0] * n for _ in range(cs.n_fixed - len(sel_cols))] + sel_cols
    assert MP.check(cs, adv, fixed) == []
    bad = [list(c) for c in adv]; bad[lk_cols[1]][21] += 1                    # a triple that is not a row of the table
    assert any("lookup" in f for f in MP.check(cs, bad, fixed))
    ok, _, _ = _prove_and_verify(cs, fixed, adv, [], golden_srs)
    assert ok


@pytest.mark.parametrize("w,length", [(1, 9), (2, 7), (2, 40), (3, 23), (1, 70), (2, 130)])
def test_accumulated_sum_and_cumprod_gates(golden_srs, w, length):
    """SUMINIT / SUM and CUMPRODINIT / CUMPROD (layouts.rs `sum`, `prod`): the running value over rows of `w` inner columns, across
    column boundaries when the vector is longer than a block; MockProver, a wrong running value caught by the gate, then a real proof"""
    from ezkl_amd import ezkl_layout as ELy
    rng = np.random.default_rng(w * 100 + length)
    x = rng.integers(-9, 10, length).tolist()
    c = ELy.SumProdCircuit(6, w, 7 * length + 32)
    cs, fixed, copies, reg = c.keygen_inputs(x)
    adv, inst = c.witness(x)
    want_p = 1
    for v in x:
        want_p = want_p * v % R
    assert inst == [[sum(x) % R, want_p, sum(a * b for a, b in zip(x[:-1], x[1:])) % R]]
    assert MP.check(cs, adv, fixed, inst, copies) == []
    used = {OPN for OPN in (EC.SUMINIT, EC.SUM, EC.CUMPRODINIT, EC.CUMPROD) if any(reg.activations[s.index] is not None and reg.activations[s.index].any()
                                                                                  for (op, blk, col), s in c.gc.base.selectors.items() if op == OPN)}
    assert used == {EC.SUMINIT, EC.SUM, EC.CUMPRODINIT, EC.CUMPROD}
    # tamper with one running value of the output VarTensor: the accumulation gate of that row (or the next) fails
    out_cols = [col.index for blk in c.gc.advices[2].inner for col in blk]
    sel = next(s for (op, blk, col), s in c.gc.base.selectors.items() if op == EC.SUM and reg.activations[s.index] is not None and reg.activations[s.index].any())
    row = int(np.flatnonzero(reg.activations[sel.index])[0])
    bad = [list(a) for a in adv]
    tampered = False
    for ci in out_cols:
        if bad[ci][row]:
            bad[ci][row] = (bad[ci][row] + 1) % R
            tampered = True
            break
    if tampered:
        assert any("gate" in f for f in MP.check(cs, bad, fixed, inst, copies, max_failures=32))
    ok, vk, proof = _prove_and_verify(cs, fixed, adv, copies, golden_srs, inst)
    assert ok
