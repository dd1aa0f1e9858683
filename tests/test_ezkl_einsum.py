"""The reference's einsum bench circuit (/root/reference/benches/accum_einsum_matmul.rs: "ij,jk->ik" through
BaseConfig::configure_einsums, Freivalds' argument with second-phase advice and two post-commitment challenges) built by the
gate-set generator (ezkl_amd/ezkl_circuit.py Einsums) and laid out by ezkl_amd/ezkl_layout.py: MockProver, proof on the CPU-oracle
backend + pairing verifier, and (GPU) the native C++ host through the per-phase advice callback, byte-identical."""
import numpy as np
import pytest

from ezkl_amd import ezkl_layout as EL, plonk as P
from oracle import mock_prover as MP

R = P.R


def circuit(k, L, seed=1):
    c = EL.EinsumMatmulCircuit(k, L)
    rng = np.random.default_rng(seed)
    a, b = rng.integers(-128, 128, (L, L)), rng.integers(-128, 128, (L, L))
    cs, fixed, copies, rows = c.keygen_inputs(a, b)
    return c, a, b, cs, fixed, copies, rows


def test_constraint_system_of_the_bench_circuit():
    c, a, b, cs, fixed, copies, rows = circuit(7, 4)
    # configure_universal: inputs [first, first, second, second] + outputs [first, second], one inner column each
    assert cs.n_advice == 6 and cs.advice_phase == [0, 0, 1, 1, 0, 1] and cs.n_challenges == 2
    assert cs.n_selectors == 25                              # 17 contraction (3 x (MULT + DOTINIT + DOT) + 2 x 4) + 2 x 4 RLC
    assert len(cs.gates) == 25 and cs.degree == 3 and cs.n_instance == 0
    assert rows == 4 * 4 + 4 + 1 + 2 * 4 * 4 + 4 + 1         # output RLCs + scalar, input RLCs, dot, prod
    assert c.reduction_length == 3 * 4 * 4 + 2 * 4           # analyze_single_equation("ij,jk->ik")
    assert cs.perm == [("adv", i) for i in range(6)] + [("fix", 0)]
    # rotation -1 only on the second-phase output column (RLC acc, DOT): 3 queries would still give 5 blinding factors
    assert cs.blinding == 5 and (5, -1) in cs.advice_queries and (4, -1) in cs.advice_queries


def test_mock_prover_accepts_and_rejects():
    c, a, b, cs, fixed, copies, rows = circuit(7, 4)
    chal = [0x1234567890abcdef1234567890abcdef % R, 0xfedcba0987654321 % R]
    fn = c.advice_fn(a, b, cs.n_advice)
    cols = {**fn(0, []), **fn(1, chal)}
    adv = [cols[i] for i in range(cs.n_advice)]
    assert MP.check(cs, adv, fixed, [], copies, challenges=chal) == []
    bad = [list(x) for x in adv]
    bad[0][3] = (bad[0][3] + 1) % R                          # one entry of the claimed product: Freivalds' check catches it
    assert MP.check(cs, bad, fixed, [], copies, challenges=chal)
    other = [chal[0], (chal[1] + 1) % R]                     # second-phase columns made for other challenges do not verify
    assert MP.check(cs, adv, fixed, [], copies, challenges=other)


def _mont_fn(c, a, b, cs):
    fn = c.advice_fn(a, b, cs.n_advice)
    return lambda phase, chal: {i: EL.ints_to_mont(v) for i, v in fn(phase, chal).items()}


def test_prove_on_oracle_backend(golden_srs):
    from oracle.cpu_backend import OracleBackend
    from oracle import verifier as V
    from test_plonk import setup
    c, a, b, cs, fixed, copies, rows = circuit(6, 3)
    assert rows <= cs.usable
    be = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    pk, vk = P.keygen(cs, be, [EL.ints_to_mont(f) for f in fixed], copies)
    proof = P.create_proof(pk, be, _mont_fn(c, a, b, cs), P.Rng(3))
    g1, g2, s_g2 = setup(golden_srs)
    assert V.verify(vk, g1, g2, s_g2, proof)
    # a wrong product: the prover's witness callback lies about one output entry
    def lying(phase, chal):
        cols = _mont_fn(c, a, b, cs)(phase, chal)
        if phase == 0:
            cols[0] = cols[0].copy(); cols[0][1] = P.to_mont(5)
        return cols
    assert not V.verify(vk, g1, g2, s_g2, P.create_proof(pk, be, lying, P.Rng(3)))


@pytest.mark.gpu
def test_gpu_native_einsum_proof_identical(hip, golden_srs):
    from ezkl_amd import backend as B, native as NV
    from oracle.cpu_backend import OracleBackend
    from oracle import verifier as V
    from test_plonk import setup
    c, a, b, cs, fixed, copies, rows = circuit(6, 3, seed=4)
    fm = [EL.ints_to_mont(f) for f in fixed]
    cpu = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    pk_c, vk = P.keygen(cs, cpu, fm, copies)
    bg, bgl = B.Bases(golden_srs["g"]), B.Bases(golden_srs["g_lagrange"])
    pk_n = NV.NativeProvingKey(NV.NativeCircuit(cs), bg, fm, copies)
    assert pk_n.vk()[2] == vk.digest
    p_c = P.create_proof(pk_c, cpu, _mont_fn(c, a, b, cs), P.Rng(9))
    p_n = NV.create_proof(pk_n, bg, bgl, _mont_fn(c, a, b, cs), rng=P.Rng(9))
    assert p_n == p_c
    g1, g2, s_g2 = setup(golden_srs)
    assert V.verify(vk, g1, g2, s_g2, p_n)
