"""The transformer-shaped surrogate of BASELINE configs[4] (ezkl_layout.TransformerSurrogateCircuit): the argument
families a nanoGPT-like circuit switches on -- static lookup tables (softmax / layer norm), a dynamic lookup, a shuffle, Freivalds einsum
with SECOND-PHASE advice and challenges (/root/reference/src/circuit/ops/chip.rs:452-833, chip/einsum/mod.rs:487-783) -- laid out as one
unit and TILED over rows and blocks.  CPU: the MockProver accepts the tiled witness and catches a tampered cell in any tile; a real proof on
the CPU-oracle backend verifies.  GPU: the C++ host prover emits the same bytes as the CPU oracle, resident and with a streamed key."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from ezkl_amd import ezkl_layout as EL, plonk as P
from oracle import mock_prover as MP

sys.path.insert(0, os.path.join(ROOT, "tools"))
CHAL = [12345678901234567890123 % P.R, 98765432109876543210 % P.R]


def _small(k=10, **kw):
    return EL.TransformerSurrogateCircuit(k, blocks=2, d=4, einsum_len=3, decomp_base=16, lookup_max=(1 << k) // 16, **kw)


def test_tiled_witness_satisfies_every_argument_family_and_tampering_is_caught():
    c = _small()
    b = c.build(as_ints=True)
    cs = b["cs"]
    assert cs.degree == 6 and sum(cs.advice_phase) == 3 and cs.n_challenges == 2            # dynamic lookups make degree 6; Freivalds: 3 second-phase columns
    assert len(cs.lookups) == 2 * (3 + 2 + 1 + 1)                                             # per block: 3 static tables, 2 range checks, dynamic lookup, shuffle
    assert b["info"]["tiles"] >= 3 and b["info"]["blocks"] == 2
    a0, a1 = b["advice"](0, []), b["advice"](1, CHAL)
    adv = [(a0 if cs.advice_phase[i] == 0 else a1)[i] for i in range(cs.n_advice)]
    fixed = [list(f) for f in b["fixed"]]
    copies = list(b["copies"])
    assert MP.check(cs, adv, fixed, b["instances"], copies, CHAL) == []
    U, T = b["info"]["unit_rows"], b["info"]["tiles"]
    # a cell of the LAST tile of the SECOND block: its gate, lookup or copy constraint fails
    col = c.gc.advices[2].inner[1][0].index
    row = next(r for r in range((T - 1) * U, T * U) if adv[col][r])
    bad = [list(a) for a in adv]
    bad[col][row] = (bad[col][row] + 1) % P.R
    assert MP.check(cs, bad, fixed, b["instances"], copies, CHAL)
    # a second-phase cell (the Freivalds RLC) of a later tile
    col2 = next(i for i in range(cs.n_advice) if cs.advice_phase[i] == 1 and any(adv[i][U:2 * U]))
    row2 = next(r for r in range(U, 2 * U) if adv[col2][r])
    bad = [list(a) for a in adv]
    bad[col2][row2] = (bad[col2][row2] + 1) % P.R
    assert MP.check(cs, bad, fixed, b["instances"], copies, CHAL)
    # a dynamic-lookup pick that is no row of the table any more: the row is changed in EVERY tile (the argument is a multiset statement over
    # all rows -- changed in one tile only, the picks of that tile still find the row in the others, and the table side is unconstrained)
    tcol = c.gc.advices[5].inner[0][0].index
    trow = next(r for r in range(U) if adv[tcol][r])
    bad = [list(a) for a in adv]
    bad[tcol][trow + U] = (bad[tcol][trow + U] + 1) % P.R
    assert MP.check(cs, bad, fixed, b["instances"], copies, CHAL) == []
    for t in range(T):
        bad[tcol][trow + t * U] = (adv[tcol][trow] + 1) % P.R
    assert any("lookup" in f for f in MP.check(cs, bad, fixed, b["instances"], copies, CHAL))
    # wrong public outputs
    assert MP.check(cs, adv, fixed, [[(v + 1) % P.R for v in b["instances"][0]]], copies, CHAL)


def _python_srs(k, secret):
    """an (insecure) SRS of 2^k points from a known secret, on the host: g[i] = s^i G, g_lagrange[i] = L_i(s) G (what gen_srs builds on the
    device), as Montgomery byte rows"""
    from oracle import pyref as pr
    n, w = 1 << k, pr.omega(k)
    G = (1, 2)
    rows = lambda pts: np.stack([np.frombuffer(pr.g1_to_bytes(p), np.uint64) for p in pts])
    g, acc = [], 1
    for _ in range(n):
        g.append(pr.g1_mul(G, acc)); acc = acc * secret % P.R
    zn = (pow(secret, n, P.R) - 1) * pow(n, -1, P.R) % P.R
    gl, wi = [], 1
    for _ in range(n):
        gl.append(pr.g1_mul(G, zn * wi % P.R * pow(secret - wi, -1, P.R) % P.R)); wi = wi * w % P.R
    return rows(g), rows(gl)


def test_surrogate_proves_on_the_cpu_oracle_backend():
    """a REAL proof of the surrogate (two phases, challenges squeezed from the transcript between them) on the CPU-oracle backend, accepted
    by the pairing verifier -- k = 9, an SRS made on the host"""
    from oracle.cpu_backend import OracleBackend
    from oracle import pairing as E, verifier as V
    from test_bench_parity import G2
    k, s = 9, 0x5eed1234567
    c = _small(k)
    b = c.build()
    cs = b["cs"]
    g, gl = _python_srs(k, s)
    be = OracleBackend(g, gl, k)
    pk, vk = P.keygen(cs, be, EL.cols_to_mont(b["fixed"]), b["copies"])
    proof = P.create_proof(pk, be, b["advice"], P.Rng(4), instances=b["instances"])
    s_g2 = E.g2_mul(G2, s)
    assert V.verify(vk, (1, 2), G2, s_g2, proof, instances=b["instances"])
    assert not V.verify(vk, (1, 2), G2, s_g2, proof, instances=[[(v + 1) % P.R for v in b["instances"][0]]])


@pytest.mark.gpu
@pytest.mark.parametrize("k", [11, 13])
def test_gpu_proof_of_the_surrogate_equals_the_cpu_oracle_proof(hip, k, monkeypatch):
    import bench_circuits as BC
    from ezkl_amd import backend as B, native as NV
    from oracle import pairing as E, verifier as V
    from oracle.cpu_backend import OracleBackend
    from test_bench_parity import G2
    built = BC.build("transformer", k, gpu=B)
    cs, fixed, copies, adv, instances = built["cs"], built["fixed"], built["copies"], built["advice"], built["instances"]
    assert cs.k == k and cs.ext_k == k + 3 and callable(adv)
    s = 0x1234567890abcdef1234567890abcdef % P.R
    gb, glb = B.gen_srs(k, s)
    npk = NV.NativeProvingKey(NV.NativeCircuit(cs), gb, fixed, copies)
    proof_gpu = NV.create_proof(npk, gb, glb, adv, rng=P.Rng(5), instances=instances)
    cpu = OracleBackend(gb.download(), glb.download(), k)
    pk_c, vk_c = P.keygen(cs, cpu, fixed, copies)
    assert npk.vk()[2] == vk_c.digest
    if k == 11:
        proof_cpu = P.create_proof(pk_c, cpu, adv, P.Rng(5), instances=instances)
        assert proof_gpu == proof_cpu
    assert V.verify(vk_c, (1, 2), G2, E.g2_mul(G2, s), proof_gpu, instances=instances)
    # the degraded mode gives the same bytes on this circuit too (E = 8 cosets rebuilt one at a time, second-phase columns included)
    monkeypatch.setenv("EZKL_KEY_COSETS", "recompute")
    npk_s = NV.NativeProvingKey(NV.NativeCircuit(cs), gb, fixed, copies)
    assert npk_s.residency()["streamed"] and NV.create_proof(npk_s, gb, glb, adv, rng=P.Rng(5), instances=instances) == proof_gpu
    gb.free(); glb.free()
