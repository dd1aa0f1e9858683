"""The ezkl gate-set generator and the halo2 protocol shape, pinned on the reference's own fixture circuit
(tests/assets/{settings.json, vk.key, pk.key, witness.json, proof.json} -> tests/golden/*_k6.*):

  * the generated constraint system has the fixture's 30 advice / 38 fixed / 32 permutation columns / 80 selectors,
    cs.degree() = 7 -> extended_k = 9 (pk.key's l0 has 512 rows), 35 lookup arguments after chunk_lookups;
  * halo2's selector compression, run on the activations stored in vk.key, reproduces the 33 selector-derived fixed columns of
    pk.key BIT FOR BIT, and the four range-check table columns likewise;
  * the reference's proof.json has exactly the layout this prover emits for that constraint system (114 G1 | 231 Fr | 2 G1 =
    14 816 B), and its 38 fixed + 32 sigma evaluations equal pk.key's polynomials at the challenge x (recovered from the identity
    sigma columns, sigma_j(X) = delta^j X) AT THE POSITIONS halo2's query order predicts;
  * the reference's witness (input [2, 1, 1]) laid out on this circuit satisfies every gate, lookup and copy constraint of the
    key (MockProver), and the prover's proof of it is 14 816 bytes and accepted by the pairing verifier.
"""
import json
import os

import numpy as np
import pytest

import fixture_k6 as FX
from ezkl_amd import codecs, plonk as P
from oracle import mock_prover as MP

R = P.R


@pytest.fixture(scope="module")
def fx():
    return FX.load()


def proof_shape(cs):
    """(#leading G1, #scalars, #trailing G1) of a proof for cs: SURVEY.md §3.1's step list"""
    L, Pn = len(cs.lookups), cs.n_chunks
    g1 = cs.n_advice + L + Pn + L + 1 + (cs.degree - 1)
    sc = len(cs.advice_queries) + len(cs.fixed_queries) + 1 + len(cs.perm) + (3 * Pn - 1) + 3 * L
    return g1, sc, 2


def test_fixture_constraint_system_counts(fx):
    gc, cs, pk = fx["gc"], fx["cs"], fx["pk"]
    assert fx["pre"] == dict(n_fixed=5, n_selectors=80, n_lookups=35, degree=7)     # 1 constant + 4 table columns before compression
    assert cs.n_advice == 30 and cs.n_fixed == 38 and cs.n_instance == 1 and len(cs.perm) == 32
    assert cs.n_fixed == pk["vk"]["fixed_commitments"].shape[0] and len(cs.perm) == pk["vk"]["permutation_commitments"].shape[0]
    assert len(gc.cs.selectors) == pk["vk"]["selectors"].shape[0] == 80
    assert cs.degree == 7 and cs.ext_k == 9 == int(np.log2(pk["l0"].shape[0]))
    assert cs.blinding == 5 and cs.usable == 58
    assert len(cs.lookups) == 35 and cs.n_chunks == 7 and cs.chunk == 5
    assert len(cs.advice_queries) == 35 and len(cs.fixed_queries) == 38
    # enable_equality order: 30 advice, the constant column, the instance column
    assert cs.perm == [("adv", c) for c in range(30)] + [("fix", 0), ("inst", 0)]
    # halo2's query order: every advice column at rotation 0 (enable_equality), then the accumulators' previous row
    assert cs.advice_queries[:30] == [(c, 0) for c in range(30)]
    assert cs.advice_queries[30:] == [(20 + 2 * b, -1) for b in range(5)]
    assert cs.fixed_queries == [(c, 0) for c in range(38)]
    assert proof_shape(cs) == (114, 231, 2)


def test_selector_compression_reproduces_pk_fixed_columns(fx):
    fixed, sel_cols = fx["fixed"], fx["selector_columns"]
    assert len(sel_cols) == 33
    for i, colv in enumerate(sel_cols):                       # 20 complex selectors, then 13 combinations of the 60 simple ones
        assert colv == fixed[5 + i], "selector-derived fixed column %d" % (5 + i)
    assert sorted(fx["table_columns"]) == [1, 2, 3, 4]
    for c, colv in fx["table_columns"].items():               # range checks (-1,1) and (0,127) over 56-row columns, scaled by the selector values
        assert colv == fixed[c], "table column %d" % c
    # and the keygen side: the device-independent part of pk.key's fixed_values is reproduced, the polys / cosets are the NTT KATs of test_oracle.py


def test_reference_proof_layout_and_evaluation_order(fx):
    cs, pk = fx["cs"], fx["pk"]
    pr = codecs.read_proof_json(open(os.path.join(FX.G, "proof_k6.json")).read())
    g1, sc, tail = proof_shape(cs)
    assert len(pr["proof"]) == 64 * g1 + 32 * sc + 64 * tail == 14816
    pts, ev, last = codecs.split_evm_proof(pr["proof"], g1, sc)
    q = P.Q
    for x_, y_ in pts + last:
        assert (y_ * y_ - x_ * x_ * x_ - 3) % q == 0
    assert pr["instances"] == [[0, 0, 0, 0]]
    # the evaluation section is [advice queries | fixed queries | random poly | sigma | z | lookups]
    na, nf = len(cs.advice_queries), len(cs.fixed_queries)
    sig_at = na + nf + 1
    perm_polys = [FX.col_ints(p) for p in pk["perm_polys"]]
    fixed_polys = [FX.col_ints(p) for p in pk["fixed_polys"]]
    ident = [j for j, p in enumerate(perm_polys) if p[0] == 0 and all(c == 0 for c in p[2:])]
    assert len(ident) == 18 and all(perm_polys[j][1] == pow(P.DELTA, j, R) for j in ident)
    xs = {ev[sig_at + j] * pow(perm_polys[j][1], -1, R) % R for j in ident}
    assert len(xs) == 1, "the identity sigma evaluations do not agree on x: wrong offset of the sigma block"
    x = xs.pop()
    def horner(p, z):
        acc = 0
        for c in reversed(p): acc = (acc * z + c) % R
        return acc
    w = P.omega(cs.k)
    for j in range(32):
        assert horner(perm_polys[j], x) == ev[sig_at + j]
    for i, (c, r) in enumerate(cs.fixed_queries):
        assert horner(fixed_polys[c], x * pow(w, r % cs.n, R) % R) == ev[na + i], "fixed query %d" % i
    # EvmTranscript: x = keccak(..) mod r, so the digest is one of at most 6 lifts of x (recorded for the curious; the chain before x
    # needs halo2's vk.transcript_repr -- a Blake2b hash of the pinned constraint system's Debug text -- which is not on disk)
    assert 1 <= len([x + t * R for t in range(6) if x + t * R < 1 << 256]) <= 6


def test_reference_witness_satisfies_the_reference_key(fx):
    cs = fx["cs"]
    adv, inst, outputs = FX.witness(fx)
    wj = json.load(open(os.path.join(FX.G, "witness_k6.json")))
    assert [codecs.felt_from_hex_le(h) for h in wj["inputs"][0]] == [2, 1, 1]
    assert [codecs.felt_from_hex_le(h) for h in wj["outputs"][0]] == inst[0] == outputs
    copies = FX.copies_of(FX.copy_cycles(fx["pk"]))
    assert len(copies) > 200
    assert MP.check(cs, adv, fx["fixed"], inst, copies) == []
    # and the gates bite: break one cell of each kind of row
    bad = [list(c) for c in adv]
    bad[20][16] = (bad[20][16] + 1) % R                       # a DOT accumulator
    assert any("gate" in f for f in MP.check(cs, bad, fx["fixed"], inst, copies))
    bad = [list(c) for c in adv]
    bad[1][6] = 128                                           # a decomposition digit outside (0,127)
    assert any("lookup" in f for f in MP.check(cs, bad, fx["fixed"], inst, copies))
    # a different input: the layout is input-independent
    adv2, inst2, out2 = FX.witness(fx, x=(-3, -100, 77))
    assert MP.check(cs, adv2, fx["fixed"], inst2, copies) == [] and out2 == [100, 101, 0, 0]


def _srs():
    from oracle import pairing as E, pyref as pr
    srs = pr.parse_srs(open(os.path.join(FX.G, "kzg_k6.srs"), "rb").read())
    g = np.stack([np.frombuffer(b, np.uint64) for b in srs["g"]])
    gl = np.stack([np.frombuffer(b, np.uint64) for b in srs["g_lagrange"]])
    return g, gl, pr.g1_from_bytes(srs["g"][0]), E.g2_from_bytes(srs["g2"]), E.g2_from_bytes(srs["s_g2"])


def test_prove_reference_circuit_oracle_backend(fx):
    """keygen from pk.key's fixed columns + copy constraints, create_proof of the reference's witness, pairing verifier"""
    from oracle.cpu_backend import OracleBackend
    from oracle import verifier as V
    cs = fx["cs"]
    g, gl, g1, g2, s_g2 = _srs()
    be = OracleBackend(g, gl, FX.K)
    adv, inst, _ = FX.witness(fx)
    copies = FX.copies_of(FX.copy_cycles(fx["pk"]))
    pk, vk = P.keygen(cs, be, FX.mont_cols(fx["fixed"]), copies)
    # keygen reproduces the reference key's permutation polynomials' cycle structure, and the sigma of untouched columns exactly
    ref_sig = fx["pk"]["permutations"]
    for j in (4, 5, 14, 29):
        assert (np.asarray(be.download(pk.sigma_values[j], cs.n)) == ref_sig[j]).all()
    proof = P.create_proof(pk, be, FX.mont_cols(adv), P.Rng(7), instances=inst)
    assert len(proof) == 14816
    assert V.verify(vk, g1, g2, s_g2, proof, instances=inst)
    assert not V.verify(vk, g1, g2, s_g2, proof, instances=[[1, 0, 0, 0]])
    tampered = bytearray(proof); tampered[64 * 114 + 5] ^= 1
    assert not V.verify(vk, g1, g2, s_g2, bytes(tampered), instances=inst)


@pytest.mark.gpu
def test_gpu_native_prover_on_the_reference_pk_file(hip, fx):
    """`ezkl prove` on the reference's artefacts: the reference's OWN pk.key goes through ezkl_prover_pk_read (columns straight
    to HBM), the reference's witness is proved on the GPU by the C++ host, and the proof has the reference proof's layout.  The vk
    commitments in the file were made with the public SRS (not in the tree), so the key is re-committed under the test SRS."""
    from ezkl_amd import backend as B, native as NV
    from oracle import verifier as V
    from oracle.cpu_backend import OracleBackend
    cs = fx["cs"]
    g, gl, g1, g2, s_g2 = _srs()
    bg, bgl = B.Bases(g), B.Bases(gl)
    circ = NV.NativeCircuit(cs)
    assert circ.info()["degree"] == 7 and circ.info()["ext_k"] == 9 and circ.info()["n_advice_queries"] == 35
    pk = NV.NativeProvingKey.from_bytes(circ, open(os.path.join(FX.G, "pk_k6.key"), "rb").read(), recommit=bg)
    adv, inst, _ = FX.witness(fx)
    proof = NV.create_proof(pk, bg, bgl, FX.mont_cols(adv), seed=11, instances=inst)
    assert len(proof) == 14816
    # the one-shot loader (mapped file, n-row sections only, cosets recomputed on the device) yields the same key: same proof bytes
    pk_f = NV.NativeProvingKey.from_file(circ, os.path.join(FX.G, "pk_k6.key"), recommit=bg)
    assert pk_f.to_bytes() == pk.to_bytes()
    assert NV.create_proof(pk_f, bg, bgl, FX.mont_cols(adv), seed=11, instances=inst) == proof
    fc, pc, digest = pk.vk()
    vk = P.VerifyingKey()
    vk.cs = cs
    vk.fixed_commitments = [P.point_to_ints(p) for p in fc]
    vk.sigma_commitments = [P.point_to_ints(p) for p in pc]
    vk.digest = P.vk_digest(vk)
    assert vk.digest == digest
    assert V.verify(vk, g1, g2, s_g2, proof, instances=inst)
    # the same key built by keygen from the fixed values + copy constraints commits to the same fixed columns
    copies = FX.copies_of(FX.copy_cycles(fx["pk"]))
    pk2 = NV.NativeProvingKey(circ, bg, FX.mont_cols(fx["fixed"]), copies)
    assert (pk2.vk()[0] == fc).all()
    # byte-identical to the Python host on the CPU-oracle backend under the same randomness
    cpu = OracleBackend(g, gl, FX.K)
    pk_c, _ = P.keygen(cs, cpu, FX.mont_cols(fx["fixed"]), copies)
    rng_a, rng_b = P.Rng(5), P.Rng(5)
    proof_c = P.create_proof(pk_c, cpu, FX.mont_cols(adv), rng_a, instances=inst)
    proof_n = NV.create_proof(pk2, bg, bgl, FX.mont_cols(adv), rng=rng_b, instances=inst)
    assert proof_n == proof_c


# ---------------------------------------------------------------------------------------------------------------------
# `ezkl setup` parity: the layout engine (ezkl_amd/ezkl_layout.py), run on the fixture MODEL, reproduces the reference's key
# ---------------------------------------------------------------------------------------------------------------------
FIXTURE_W = [[0, -1, 0], [0, -1, 0], [0, 0, -1], [0, 0, 0]]       # round(dense.weight) of tests/assets/network.onnx at scale 0
FIXTURE_B = [0, 1, 0, 0]


def fixture_mlp():
    from ezkl_amd import ezkl_layout as EL
    st = json.load(open(os.path.join(FX.G, "settings_k6.json")))
    ra = st["run_args"]
    return EL.MlpCircuit(ra["logrows"], ra["num_inner_cols"], [FIXTURE_W], [FIXTURE_B], ra["decomp_base"], ra["decomp_legs"],
                         total_assignments=st["total_assignments"])


def test_layout_engine_reproduces_the_reference_key(fx, golden_srs):
    """selector activations == vk.key's 80 x 64 bits; all 38 fixed columns (constants, tables, compressed selectors) and all 32
    permutation columns == pk.key's `fixed_values` / `permutations`, bit for bit; the witness == the one read off the key"""
    from oracle.cpu_backend import OracleBackend
    c = fixture_mlp()
    cs, fixed, copies, reg = c.keygen_inputs([2, 1, 1])
    assert (np.array(reg.selector_rows()) == fx["pk"]["vk"]["selectors"]).all()
    assert reg.linear == 198
    assert fixed == fx["fixed"]
    be = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], FX.K)
    pk, vk = P.keygen(cs, be, FX.mont_cols(fixed), copies)
    for j in range(32):
        assert (np.asarray(be.download(pk.sigma_values[j], cs.n)) == fx["pk"]["permutations"][j]).all(), "sigma column %d" % j
    adv, inst = c.witness([2, 1, 1])
    adv_ref, inst_ref, _ = FX.witness(fx)
    assert adv == adv_ref and inst == inst_ref
    assert MP.check(cs, adv, fixed, inst, copies) == []
    # the serialised constraint system is the fixture's, byte for byte (same gates, lookups, query order)
    assert P.serialize_cs(cs) == P.serialize_cs(fx["cs"])


@pytest.mark.gpu
def test_gpu_setup_reproduces_reference_pk_file(hip, fx, golden_srs):
    """keygen on the GPU from the MODEL (layout engine -> fixed columns + copy constraints), written in halo2's pk.key layout:
    equal to the reference's tests/assets/pk.key byte for byte outside the 70 SRS-dependent commitments (the reference used the
    public powers of tau, which are not in the tree)."""
    from ezkl_amd import backend as B, native as NV
    c = fixture_mlp()
    cs, fixed, copies, reg = c.keygen_inputs([2, 1, 1])
    bg = B.Bases(golden_srs["g"])
    pk = NV.NativeProvingKey(NV.NativeCircuit(cs), bg, FX.mont_cols(fixed), copies)
    pk.set_selectors(reg.selector_rows())
    mine = pk.to_bytes()
    ref = open(os.path.join(FX.G, "pk_k6.key"), "rb").read()
    assert len(mine) == len(ref) == 1489595
    lo, hi = 7, 7 + 64 * (38 + 32)
    assert mine[:lo] == ref[:lo] and mine[hi:] == ref[hi:]


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_layout_engine_random_mlps_satisfy_their_circuits(seed):
    """MlpCircuit over random shapes -- layers, widths, inner columns, logrows chosen so that the VarTensors overflow into 1..4 blocks
    (dot products and decompositions straddling column boundaries, with the duplicated accumulator rows of
    assign_with_duplication) -- every layout satisfies every gate, lookup and copy constraint of its own constraint system"""
    from ezkl_amd import ezkl_layout as EL
    rng = np.random.default_rng(100 + seed)
    layers = int(rng.integers(1, 4))
    width = int(rng.integers(3, 11))
    inner = int(rng.choice([1, 2, 3]))
    Ws = [rng.integers(-4, 5, (width, width)).tolist() for _ in range(layers)]
    bs = [rng.integers(-9, 10, width).tolist() for _ in range(layers)]
    x = rng.integers(-20, 21, width).tolist()
    probe = EL.MlpCircuit(12, inner, Ws, bs, 128, 2)
    cells = probe.settings.total_assignments
    k = 6
    while ((1 << k) - 6) * inner * 4 < cells:                 # the smallest logrows that needs at most ~4 blocks
        k += 1
    c = EL.MlpCircuit(k, inner, Ws, bs, 128, 2)
    cs, fixed, copies, reg = c.keygen_inputs(x)
    adv, inst = c.witness(x)
    assert MP.check(cs, adv, fixed, inst, copies) == []
    assert c.gc.advices[0].num_blocks() >= 1 and reg.linear <= c.settings.total_assignments
    # the model the circuit computes
    v = np.array(x)
    for W, b in zip(Ws, bs):
        v = np.maximum(np.array(W) @ v + np.array(b), 0)
    assert inst == [[int(t) % R for t in v]]


def _conv_small():
    from ezkl_amd import ezkl_layout as EL
    c = EL.ConvMnistCircuit(logrows=10, image=8, kernel=3, out_channels=2, stride=2, classes=3, lookup_range=(-700, 700), denom=4,
                            decomp_base=32, decomp_legs=2)
    rng = np.random.default_rng(1)
    c.kernels = rng.integers(-3, 4, c.kernels.shape)
    return c, rng.integers(0, 4, (8, 8))


def test_conv2d_mnist_layout_small():
    """the conv2d_mnist Config and layout (examples/conv2d_mnist/main.rs) on a small shape whose Div table spans two table columns
    (the index column and the column-selector polynomial are live): satisfied; a wrong public output or a tampered lookup output is not"""
    c, img = _conv_small()
    cs, fixed, copies, reg = c.keygen_inputs(img)
    adv, inst = c.witness(img)
    assert (cs.n_advice, cs.n_instance) == (3, 1) and len(c.gc.base.static_tables["div_4"].table_inputs) == 2
    assert MP.check(cs, adv, fixed, inst, copies) == []
    assert [v if v < R // 2 else v - R for v in inst[0]] == c.model(img)
    bad = [list(inst[0])]
    bad[0][0] = (bad[0][0] + 1) % R
    assert MP.check(cs, adv, fixed, bad, copies)
    # a Div output that is not f(x): find a row with the lookup selector on and bump the output cell
    sel = next(s for key, s in c.gc.base.static_selectors.items())
    row = int(np.flatnonzero(reg.activations[sel.index])[0])
    adv2 = [list(a) for a in adv]
    adv2[2][row] = (adv2[2][row] + 1) % R
    assert any("lookup" in f for f in MP.check(cs, adv2, fixed, inst, copies, max_failures=64))


def test_conv2d_mnist_k17():
    """BASELINE configs[2] at the reference's shapes (28 x 28, 4 x 5 x 5 stride 2, 576 -> 10, Div{32} over (-32768, 32768)): fits k = 17,
    every constraint satisfied, the instance column is the integer model's output"""
    from ezkl_amd import ezkl_layout as EL
    c = EL.ConvMnistCircuit()
    img = np.random.default_rng(3).integers(0, 16, (28, 28))
    cs, fixed, copies, reg = c.keygen_inputs(img)
    adv, inst = c.witness(img)
    assert (cs.k, cs.n_advice, c.length) == (17, 3, 576) and reg.linear < (1 << 17)
    assert MP.check(cs, adv, fixed, inst, copies) == []
    assert [v if v < R // 2 else v - R for v in inst[0]] == c.model(img)


def test_conv2d_mnist_tiny_proof_oracle_backend(golden_srs):
    """a k = 6 instance of the same Config / layout (VarTensors overflowing into several blocks, Div table over four table columns):
    MockProver, then keygen + create_proof on the CPU-oracle backend, accepted by the pairing verifier; a wrong public output is rejected"""
    from ezkl_amd import ezkl_layout as EL
    from test_ezkl_gateset import _prove_and_verify
    from oracle import verifier as V
    from test_plonk import setup
    c = EL.ConvMnistCircuit(logrows=6, image=6, kernel=3, out_channels=1, stride=3, classes=2, lookup_range=(-100, 100), denom=4,
                            decomp_base=16, decomp_legs=2, capacity=220)
    rng = np.random.default_rng(2)
    c.kernels = rng.integers(-3, 4, c.kernels.shape)
    c.fc_w, c.fc_b = rng.integers(-5, 6, c.fc_w.shape), rng.integers(-9, 10, c.fc_b.shape)
    img = rng.integers(0, 4, (6, 6))
    cs, fixed, copies, reg = c.keygen_inputs(img)
    adv, inst = c.witness(img)
    assert c.gc.advices[0].num_blocks() >= 3 and len(c.gc.base.static_tables["div_4"].table_inputs) == 4
    assert MP.check(cs, adv, fixed, inst, copies) == []
    ok, vk, proof = _prove_and_verify(cs, fixed, adv, copies, golden_srs, inst)
    assert ok
    g1, g2, s_g2 = setup(golden_srs)
    assert not V.verify(vk, g1, g2, s_g2, proof, instances=[[(inst[0][0] + 1) % R, inst[0][1]]])


@pytest.mark.gpu
def test_gpu_conv2d_mnist_small_native_prove_and_verify(hip):
    """the conv2d_mnist Config / layout at k = 10 through the product: keygen on the GPU, ezkl_prover_create_proof with CheckMode SAFE
    (the library verifies its own proof with the C++ pairing check), ezkl_prover_verify_proof accepts it and rejects a wrong instance;
    byte-identical to the Python host on the CPU-oracle backend with the same randomness"""
    from ezkl_amd import backend as B, native as NV, ezkl_layout as EL
    from oracle.cpu_backend import OracleBackend
    c, img = _conv_small()
    cs, fixed, copies, reg = c.keygen_inputs(img)
    adv, inst = c.witness(img)
    s = 0x1d5c0ffee1234567
    bg, bgl = B.gen_srs(cs.k, s)
    g2, s_g2 = NV.g2_mul_generator(1), NV.g2_mul_generator(s)
    pk = NV.NativeProvingKey(NV.NativeCircuit(cs), bg, EL.cols_to_mont(fixed, B), copies)
    mont = EL.cols_to_mont(adv, B)
    proof = NV.create_proof(pk, bg, bgl, mont, seed=9, instances=inst, check_mode="SAFE", g2=g2, s_g2=s_g2)
    assert NV.verify_proof(pk, g2, s_g2, proof, inst)
    assert not NV.verify_proof(pk, g2, s_g2, proof, [[(inst[0][0] + 1) % R] + list(inst[0][1:])])
    cpu = OracleBackend(bg.download(), bgl.download(), cs.k)
    pk_c, _ = P.keygen(cs, cpu, [EL.ints_to_mont(f) for f in fixed], copies)
    assert P.create_proof(pk_c, cpu, [EL.ints_to_mont(a) for a in adv], P.Rng(5), instances=inst) == \
        NV.create_proof(pk, bg, bgl, mont, rng=P.Rng(5), instances=inst)
