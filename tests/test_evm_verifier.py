"""Protocol parity against the reference's OWN verifier.

halo2 is not in the reference tree, but a verifier is: tests/assets/wasm.code is the compiled Solidity verifier the reference's
tooling generated for a k = 6 key of the fixture's constraint system.  Executed by oracle/mini_evm.py (precompiles = the oracle's
BN254 arithmetic and pairing) it is the zkonduit halo2 verification algorithm for the EVM transcript, bit for bit.  With the
contract's verifying-key constants replaced by this repo's key (tests/evm_fixture.py) and nothing else touched:

  * the eight transcript challenges, x^n, l_0, l_last, l_blind equal this repo's verifier's;
  * its quotient evaluation is this repo's, term by term: all 15 permutation terms, all 105 mv-lookup terms, 70 of the 80 gates; the
    other 10 gates are the same polynomials under four other selector COMBINATIONS (another layout of the model), which are recovered;
  * with the fixture's constraint system compressed under those combinations, **the reference's bytecode ACCEPTS this repo's proof**
    (real pairing) and rejects tampered ones -- which pins what no file fixture could: the rotation-set order and the commitment order
    of SHPLONK (halo2's query order: advice, permutation, lookups, fixed, sigma, h, random), its normalisation by the first set's
    coefficient, the evaluation order of the proof, the challenge derivation."""
import random

import numpy as np
import pytest

import evm_fixture as EV
import fixture_k6 as FX
from ezkl_amd import codecs, plonk as P
from oracle import verifier as OV

R = P.R


@pytest.fixture(scope="module")
def setup():
    from oracle.cpu_backend import OracleBackend
    fx, cs, fixed = EV.contract_constraint_system(FX)
    g, gl, g1, g2, s_g2, srs = EV.srs_k6()
    be = OracleBackend(g, gl, FX.K)
    adv, inst, _ = FX.witness(fx)
    copies = FX.copies_of(FX.copy_cycles(fx["pk"]))
    pk, vk = P.keygen(cs, be, FX.mont_cols(fixed), copies)
    proof = P.create_proof(pk, be, FX.mont_cols(adv), P.Rng(7), instances=inst)
    return dict(fx=fx, cs=cs, fixed=fixed, g=g, gl=gl, g1=g1, g2=g2, s_g2=s_g2, srs=srs, be=be, adv=adv, inst=inst, copies=copies, pk=pk, vk=vk,
                proof=proof)


def test_contract_is_a_verifier_for_the_fixture_constraint_system():
    """the bytecode, untouched, on the reference's own proof.json: it wants exactly this proof length and instance count, absorbs the
    proof in the phases this prover writes (digest + 4 instances + 30 advice | 35 m | . | 7 z + 35 phi + random | 6 h | 231 evals | . | W),
    computes all the way to the pairing -- and rejects, because it was generated for another key"""
    from oracle import mini_evm as V
    pj = codecs.read_proof_json(open(FX.G + "/proof_k6.json").read())
    tr = {}
    evm = V.Evm(EV.runtime(), V.verify_proof_calldata(pj["proof"], [v for col in pj["instances"] for v in col]), tr)
    with pytest.raises(V.Revert):
        evm.run()
    assert [len(k) for k in tr["keccak"]] == EV.KECCAK_SIZES and tr["calls"][-1][0] == 8
    assert tr["keccak"][0][:32] == EV.CONTRACT_DIGEST and tr["keccak"][0][160:] == pj["proof"][:1920]
    with pytest.raises(V.Revert):                       # a shorter proof is refused before anything is hashed
        V.Evm(EV.runtime(), V.verify_proof_calldata(pj["proof"][:-64], [0, 0, 0, 0]), {}).run()


def test_contract_key_is_a_reference_key_of_this_setup():
    """the verifying key inside the bytecode belongs to the reference's own setup: k = 6 (n^-1, omega of the 2^6 domain), G2 and -s*G2 of the
    PUBLIC SRS whose k = 1 file ships in tests/assets (kzg1.srs), and 4 fixed + 17 permutation commitments that are the fixture vk.key's
    (identity-sigma and constant columns do not depend on the layout); the others differ: another layout of the model"""
    import os
    from oracle import mini_evm as V, pairing as E, pyref as pr
    class Rec(V.Evm):
        def mstore(self, off, data):
            if len(data) == 32 and len(self.trace["keccak"]) >= 8:
                self.trace.setdefault("vk", {})[off] = int.from_bytes(data, "big")
            super().mstore(off, data)
    pj = codecs.read_proof_json(open(FX.G + "/proof_k6.json").read())
    tr = {}
    with pytest.raises(V.Revert):
        Rec(EV.runtime(), V.verify_proof_calldata(pj["proof"], [0, 0, 0, 0]), tr).run()
    vkb = tr["vk"]
    assert vkb[EV.VK_MPTR] == int.from_bytes(EV.CONTRACT_DIGEST, "big") and vkb[EV.VK_MPTR + 0x20] == 4 and vkb[EV.VK_MPTR + 0x40] == 6
    assert vkb[EV.VK_MPTR + 0x60] == pow(64, -1, R) and vkb[EV.VK_MPTR + 0x80] == P.omega(6)
    srs = pr.parse_srs(open(os.path.join(FX.G, "kzg_k1_public.srs"), "rb").read())
    g2, s_g2 = E.g2_from_bytes(srs["g2"]), E.g2_from_bytes(srs["s_g2"])
    ns = (s_g2[0], E.f2_neg(s_g2[1]))
    assert [vkb[EV.G2_MPTR + 32 * i] for i in range(8)] == [g2[0][1], g2[0][0], g2[1][1], g2[1][0], ns[0][1], ns[0][0], ns[1][1], ns[1][0]]
    vk = codecs.read_vk(open(os.path.join(FX.G, "vk_k6.key"), "rb").read(), 32)
    pts = [P.point_to_ints(p) or (0, 0) for p in list(vk["fixed_commitments"]) + list(vk["permutation_commitments"])]
    same = [vkb[EV.COMMS_MPTR + 64 * i] == x and vkb[EV.COMMS_MPTR + 64 * i + 32] == y for i, (x, y) in enumerate(pts)]
    assert sum(same[:38]) >= 4 and sum(same[38:]) >= 17 and not all(same)


def test_challenges_and_lagrange_evaluations(setup):
    s = setup
    dbg = {}
    assert OV.verify(s["vk"], s["g1"], s["g2"], s["s_g2"], s["proof"], instances=s["inst"], dbg=dbg)
    ok, tr, evm = EV.run(s["vk"], s["g2"], s["s_g2"], s["proof"], s["inst"])
    names = ["theta", "beta", "gamma", "y", "x", "shplonk_y", "shplonk_v", "shplonk_u"]
    assert [EV.word(evm, EV.CHALLENGE_MPTR + 32 * i) for i in range(8)] == [dbg[k] for k in names]
    assert EV.word(evm, EV.XN_MPTR) == dbg["xn"] and EV.word(evm, EV.L_0_MPTR) == dbg["l0"] and EV.word(evm, EV.L_LAST_MPTR) == dbg["llast"]
    assert (1 - EV.word(evm, EV.L_LAST_MPTR) - EV.word(evm, EV.L_BLIND_MPTR)) % R == dbg["lact"]
    assert EV.word(evm, EV.QUOTIENT_EVAL_MPTR) == dbg["h_eval"]
    # the keccak chain: every later input starts with the previous hash; 0x01 is appended only when nothing was absorbed in between
    from ezkl_amd.transcript import keccak256
    ks = tr["keccak"]
    assert all(ks[i][:32] == keccak256(ks[i - 1]) for i in range(1, 8)) and ks[2][32:] == b"\x01" and ks[6][32:] == b"\x01"


def _selector_value(q, size, root):
    e = q
    for o in range(1, size + 1):
        if o != root:
            e = e * (o - q) % R
    return e


def test_quotient_terms_and_selector_combinations():
    """the FIXTURE's own key (halo2's greedy combinations of its selector activations) with random evaluations in the proof: the contract
    folds 200 terms with y; 190 equal this repo's; the remaining 10 are simple-selector gates whose value equals the same gate polynomial
    times the selector polynomial of ANOTHER (column, combination size, position): the contract's combinations"""
    import json, os
    from ezkl_amd import ezkl_circuit as EC
    from ezkl_amd.halo2_cs import simple_selectors
    from oracle.cpu_backend import OracleBackend
    fx = FX.load()
    cs = fx["cs"]
    g, gl, g1, g2, s_g2, _ = EV.srs_k6()
    be = OracleBackend(g, gl, FX.K)
    adv, inst, _ = FX.witness(fx)
    pk, vk = P.keygen(cs, be, FX.mont_cols(fx["fixed"]), FX.copies_of(FX.copy_cycles(fx["pk"])))
    proof = bytearray(P.create_proof(pk, be, FX.mont_cols(adv), P.Rng(3), instances=inst))
    rnd = random.Random(5)
    for i in range(231):
        proof[114 * 64 + 32 * i:114 * 64 + 32 * i + 32] = rnd.randrange(R).to_bytes(32, "big")
    proof = bytes(proof)
    dbg = {}
    OV.verify(vk, g1, g2, s_g2, proof, instances=inst, dbg=dbg)          # rejects (random evaluations); the terms are what matters
    y, state, terms = dbg["y"], {"pend": None}, []
    def on_mul(a, b, res): state["pend"] = res if (a == y or b == y) else None
    def on_add(a, b, res):
        if state["pend"] is not None and (a == state["pend"] or b == state["pend"]):
            terms.append(b if a == state["pend"] else a)
        state["pend"] = None
    EV.run(vk, g2, s_g2, proof, inst, hooks=(on_mul, on_add))
    mine = [t % R for t in dbg["terms"]]
    assert len(mine) == 200 and len(terms) == 199        # the contract's fold starts AT the first term
    differ = [i for i in range(1, 200) if terms[i - 1] != mine[i]]
    assert len(differ) == 10 and max(differ) < 80         # gates only: every permutation and lookup term is equal
    # which selector does each gate use, and where did halo2's greedy compression put it for the fixture's activations
    st = EC.GraphSettings.from_json(json.load(open(os.path.join(FX.G, "settings_k6.json"))))
    gc0 = EC.GraphConfig(st)
    gate_sel = [next(iter(simple_selectors(p, gc0.cs.selectors, set())), None) for gt in gc0.cs.gates for p in gt.polys]
    smap = fx["gc"].cs.selector_map
    combos = {}
    for s_, c in enumerate(smap):
        combos.setdefault(c, []).append(s_)
    evs = [int.from_bytes(proof[114 * 64 + 32 * i:114 * 64 + 32 * i + 32], "big") for i in range(231)]
    fe = {c: evs[len(cs.advice_queries) + i] for i, (c, r) in enumerate(cs.fixed_queries)}
    found = {}
    for gi in differ:
        s_ = gate_sel[gi]
        c = smap[s_]
        poly = mine[gi] * pow(_selector_value(fe[c], len(combos[c]), combos[c].index(s_) + 1), -1, R) % R     # the gate without its selector
        want = terms[gi - 1] * pow(poly, -1, R) % R
        hits = [(c2, m2, r2) for c2 in fe for m2 in range(1, 7) for r2 in range(1, m2 + 1) if _selector_value(fe[c2], m2, r2) == want]
        assert len(hits) == 1
        found[s_] = hits[0]
    groups = {}
    for s_, (c2, m2, r2) in found.items():
        groups.setdefault(c2, {})[r2] = s_
    for idx, grp in enumerate(EV.CONTRACT_GROUPS):        # every recovered (column, position) is the one CONTRACT_GROUPS lists:
        col = smap[0] + idx                              # combination columns are allocated in order of their first selector
        assert groups[col] == {r2: s_ for r2, s_ in enumerate(grp, start=1) if s_ in found}
    assert sorted(found) == [7, 8, 9, 10, 11, 12, 13, 14, 15, 16]


def test_reference_verifier_accepts_this_provers_proof(setup):
    s = setup
    ok, tr, evm = EV.run(s["vk"], s["g2"], s["s_g2"], s["proof"], s["inst"])
    assert ok and tr["calls"][-1][0] == 8 and tr["calls"][-1][2][-1] == 1       # the pairing precompile said yes
    assert OV.verify(s["vk"], s["g1"], s["g2"], s["s_g2"], s["proof"], instances=s["inst"])
    # another witness, other randomness
    adv2, inst2, out2 = FX.witness(s["fx"], x=(-3, -100, 77))
    proof2 = P.create_proof(s["pk"], s["be"], FX.mont_cols(adv2), P.Rng(99), instances=inst2)
    assert EV.run(s["vk"], s["g2"], s["s_g2"], proof2, inst2)[0]
    assert not EV.run(s["vk"], s["g2"], s["s_g2"], proof2, s["inst"])[0]          # the first proof's instances


@pytest.mark.parametrize("where", ["advice commitment", "evaluation", "h piece", "opening point", "instance"])
def test_reference_verifier_rejects_tampering(setup, where):
    s = setup
    proof, inst = bytearray(s["proof"]), [list(c) for c in s["inst"]]
    if where == "advice commitment":
        proof[0:64] = proof[64:128]
    elif where == "evaluation":
        proof[114 * 64 + 32 * 17 + 31] ^= 1
    elif where == "h piece":
        proof[108 * 64:109 * 64] = proof[109 * 64:110 * 64]
    elif where == "opening point":
        proof[-64:] = proof[-128:-64]
    else:
        inst[0][2] = (inst[0][2] + 1) % R
    assert not EV.run(s["vk"], s["g2"], s["s_g2"], bytes(proof), inst)[0]
    assert not OV.verify(s["vk"], s["g1"], s["g2"], s["s_g2"], bytes(proof), instances=inst)


@pytest.mark.gpu
def test_gpu_native_prover_proof_accepted_by_the_reference_verifier(hip, setup):
    """the product path: keygen + create_proof by libezkl_prover.so on the GPU; the reference's bytecode accepts the proof, the library's
    own C++ verifier accepts it, and the bytes equal the Python host's on the CPU-oracle backend under the same randomness"""
    from ezkl_amd import backend as B, native as NV
    s = setup
    bg, bgl = B.Bases(s["g"]), B.Bases(s["gl"])
    pk = NV.NativeProvingKey(NV.NativeCircuit(s["cs"]), bg, FX.mont_cols(s["fixed"]), s["copies"])
    fc, pc, digest = pk.vk()
    assert digest == s["vk"].digest and [P.point_to_ints(p) for p in fc] == list(s["vk"].fixed_commitments)
    mont = FX.mont_cols(s["adv"])
    proof = NV.create_proof(pk, bg, bgl, mont, seed=21, instances=s["inst"], check_mode="SAFE", g2=s["srs"]["g2"], s_g2=s["srs"]["s_g2"])
    assert len(proof) == 14816
    assert EV.run(s["vk"], s["g2"], s["s_g2"], proof, s["inst"])[0]
    assert NV.verify_proof(pk, s["srs"]["g2"], s["srs"]["s_g2"], proof, s["inst"])
    assert NV.create_proof(pk, bg, bgl, mont, rng=P.Rng(7), instances=s["inst"]) == s["proof"]
    # a fork's own digest through ezkl_prover_pk_set_transcript_repr: with the CONTRACT's digest constant the transcript is the one
    # the untouched digest path of the bytecode reads (only the commitments are still replaced)
    contract_digest = int.from_bytes(EV.CONTRACT_DIGEST, "big")
    pk.set_transcript_repr(contract_digest)
    assert pk.vk()[2] == contract_digest
    proof_d = NV.create_proof(pk, bg, bgl, mont, seed=22, instances=s["inst"])
    vk_d = P.VerifyingKey()
    vk_d.cs, vk_d.fixed_commitments, vk_d.sigma_commitments, vk_d.digest = s["cs"], s["vk"].fixed_commitments, s["vk"].sigma_commitments, contract_digest
    ok, tr, _ = EV.run(vk_d, s["g2"], s["s_g2"], proof_d, s["inst"])
    assert ok and tr["keccak"][0][:32] == EV.CONTRACT_DIGEST
    assert NV.verify_proof(pk, s["srs"]["g2"], s["srs"]["s_g2"], proof_d, s["inst"])
    assert not EV.run(s["vk"], s["g2"], s["s_g2"], proof_d, s["inst"])[0]          # the default digest no longer matches it
