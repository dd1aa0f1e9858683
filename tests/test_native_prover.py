"""The native host prover (libezkl_prover.so, C++ over the C ABI).
CPU: the library loads and exports what include/ezkl_prover.h declares; its Keccak and its constraint-system analysis
agree with the Python restatement; malformed circuit descriptions are rejected.
GPU: keygen and create_proof give the SAME BYTES as the Python prover on the HIP backend and as the CPU oracle backend
(same SRS fixture / witness / randomness), and the independent pairing verifier accepts."""
import os
import re
import struct

import numpy as np
import pytest

from conftest import ROOT
from ezkl_amd import native as N, plonk as P, transcript as TR
from test_plonk import (det_rng, instance_phase_circuit, instance_phase_witness, lookup_circuit, lookup_witness, mul_add_circuit, random_circuit,
                        setup, witness)


def test_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "ezkl_prover.h")).read()
    declared = sorted(set(re.findall(r"\b(ezkl_prover_\w+)\s*\(", hdr)))
    assert declared == sorted(N.SYMBOLS)
    L = N.load()
    for s in declared:
        assert hasattr(L, s), s


def test_keccak_matches_python_restatement():
    rng = np.random.default_rng(5)
    for ln in (0, 1, 31, 32, 64, 135, 136, 137, 272, 1000):
        data = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        assert N.keccak256(data) == TR.keccak256(data)
    assert N.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"     # the well-known empty-input digest


@pytest.mark.parametrize("make", [lambda: mul_add_circuit(6), lambda: lookup_circuit(6), lambda: instance_phase_circuit(6),
                                  lambda: random_circuit(11)[0], lambda: random_circuit(12)[0], lambda: mul_add_circuit(10)])
def test_constraint_system_analysis_matches_python(make):
    cs = make()
    info = N.NativeCircuit(cs).info()
    assert info == dict(degree=cs.degree, ext_k=cs.ext_k, chunk=cs.chunk, n_chunks=cs.n_chunks, usable=cs.usable,
                        n_advice_queries=len(cs.advice_queries), n_fixed_queries=len(cs.fixed_queries), n_instance_queries=len(cs.instance_queries))


@pytest.mark.parametrize("seed", range(40, 60))
def test_constraint_system_analysis_random_circuits(seed):
    """degree / extended_k / permutation chunking / query sets of the C++ parser == the Python ConstraintSystem on random
    gate sets (no GPU needed: the description is parsed and analysed on the host)"""
    cs = random_circuit(seed, k=6 + seed % 5)[0]
    info = N.NativeCircuit(cs).info()
    assert info == dict(degree=cs.degree, ext_k=cs.ext_k, chunk=cs.chunk, n_chunks=cs.n_chunks, usable=cs.usable,
                        n_advice_queries=len(cs.advice_queries), n_fixed_queries=len(cs.fixed_queries), n_instance_queries=len(cs.instance_queries))


@pytest.mark.parametrize("seed", range(70, 82))
def test_constraint_system_blob_round_trips(seed):
    """plonk.deserialize_cs is the inverse of serialize_cs (the laid-out bench circuits are stored as this blob + JSON, nothing a reader
    unpickles): blob -> ConstraintSystem -> the same blob, same derived quantities, sub-expressions shared again; the reference's fixture
    circuit (query order given, unblinded columns, selectors) included"""
    from ezkl_amd import plonk as P
    if seed == 70:
        import fixture_k6 as FX
        cs = FX.load()["cs"]
    elif seed == 71:
        cs = instance_phase_circuit(6)
    elif seed == 72:
        cs = lookup_circuit(7)
    else:
        cs = random_circuit(seed, k=6 + seed % 4)[0]
    blob = P.serialize_cs(cs)
    back = P.deserialize_cs(blob)
    assert P.serialize_cs(back) == blob
    for a in ("k", "n_advice", "n_fixed", "n_instance", "n_challenges", "advice_phase", "degree", "ext_k", "chunk", "n_chunks", "usable", "blinding",
              "advice_queries", "fixed_queries", "instance_queries", "perm", "unblinded", "n_selectors", "queries_given"):
        assert getattr(back, a) == getattr(cs, a), a
    assert N.NativeCircuit(back).info() == N.NativeCircuit(cs).info()
    for bad in (blob[:-1], blob + b"\0", b"\0" * 28 + blob[28:]):
        with pytest.raises(Exception):
            P.deserialize_cs(bad)


def test_malformed_circuit_descriptions_are_rejected():
    import ctypes as C
    good = N.serialize_cs(lookup_circuit(6))

    def parse(blob):
        h = C.c_void_p()
        rc = N.load().ezkl_prover_cs_parse(bytes(blob), C.c_size_t(len(blob)), C.byref(h))
        if rc == 0:
            N.load().ezkl_prover_cs_free(h)
        return rc
    assert parse(good) == 0
    assert parse(good[:-4]) == -3 and parse(good + b"\0\0\0\0") == -3          # truncated / trailing bytes
    bad = bytearray(good); bad[0] ^= 1
    assert parse(bad) == -3                                                    # magic
    bad = bytearray(good); struct.pack_into("<I", bad, 8, 40)
    assert parse(bad) == -3                                                    # k out of range
    # first node referencing a later node (children must precede parents)
    off = 4 * (7 + 4) + 4
    bad = bytearray(good); struct.pack_into("<3I", bad, off, 6, 5, 7)
    assert parse(bad) == -3
    assert b"precede" in N.load().ezkl_prover_last_error() or b"range" in N.load().ezkl_prover_last_error()
    # advice column index out of range
    bad = bytearray(good); struct.pack_into("<3I", bad, off, 1, 99, 0)
    assert parse(bad) == -3


def test_shard_arguments_are_validated():
    """ezkl_prover_cs_set_shard: empty / out-of-range slices and a missing fold callback are refused (no GPU needed)"""
    import ctypes as C
    circ = N.NativeCircuit(mul_add_circuit(6))
    L = N.load()
    fold = N.FOLD_FN(lambda user, pts, count: 0)
    assert L.ezkl_prover_cs_set_shard(circ.h, C.c_uint32(0), C.c_uint32(32), fold, None) == 0
    assert L.ezkl_prover_cs_set_shard(circ.h, C.c_uint32(32), C.c_uint32(64), fold, None) == 0
    assert L.ezkl_prover_cs_set_shard(circ.h, C.c_uint32(0), C.c_uint32(0), C.cast(None, N.FOLD_FN), None) == 0       # off again
    assert L.ezkl_prover_cs_set_shard(circ.h, C.c_uint32(8), C.c_uint32(8), fold, None) == -3                          # empty
    assert L.ezkl_prover_cs_set_shard(circ.h, C.c_uint32(0), C.c_uint32(65), fold, None) == -3                         # beyond 2^k
    assert L.ezkl_prover_cs_set_shard(circ.h, C.c_uint32(0), C.c_uint32(32), C.cast(None, N.FOLD_FN), None) == -3      # no callback
    assert L.ezkl_prover_cs_set_shard(None, C.c_uint32(0), C.c_uint32(32), fold, None) == -3


# ------------------------------------------------------------------ GPU
def _native_setup(golden_srs, cs, fixed, copies):
    from ezkl_amd import backend as B
    g, gl = B.Bases(golden_srs["g"]), B.Bases(golden_srs["g_lagrange"])
    circ = N.NativeCircuit(cs)
    pk = N.NativeProvingKey(circ, g, fixed, copies)
    return g, gl, pk


def _pt(p):
    x, y = (0, 0) if p is None else p
    return np.frombuffer((x * P.MONT % P.Q).to_bytes(32, "little") + (y * P.MONT % P.Q).to_bytes(32, "little"), np.uint64)


def _check_against_python(golden_srs, cs, adv, fixed, copies, seed, instances=()):
    from oracle import verifier as V
    from oracle.cpu_backend import OracleBackend
    g, gl, pk = _native_setup(golden_srs, cs, fixed, copies)
    gpu = P.GpuBackend(golden_srs["g"], golden_srs["g_lagrange"], cs.k)
    pk_g, vk_g = P.keygen(cs, gpu, fixed, copies)
    fc, pc, digest = pk.vk()
    assert digest == vk_g.digest
    assert all((fc[i] == _pt(p)).all() for i, p in enumerate(vk_g.fixed_commitments))
    assert all((pc[i] == _pt(p)).all() for i, p in enumerate(vk_g.sigma_commitments))
    tm = {}
    proof_n = N.create_proof(pk, g, gl, adv, rng=det_rng(seed), instances=instances, timings=tm)
    proof_g = P.create_proof(pk_g, gpu, adv, det_rng(seed), instances=instances)
    assert proof_n == proof_g
    cpu = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], cs.k)
    pk_c, _ = P.keygen(cs, cpu, fixed, copies)
    assert proof_n == P.create_proof(pk_c, cpu, adv, det_rng(seed), instances=instances)
    g1, g2, s_g2 = setup(golden_srs)
    assert V.verify(vk_g, g1, g2, s_g2, proof_n, instances=instances)
    assert tm["total"] > 0 and abs(sum(tm[s] for s in N.STAGES[:-1]) - tm["total"]) < 0.05 * tm["total"] + 1e-3
    return g, gl, pk, vk_g


@pytest.mark.gpu
def test_native_proof_bit_identical_gates_and_permutation(hip, golden_srs):
    cs = mul_add_circuit(6)
    adv, fixed, copies = witness(cs, 2)
    _check_against_python(golden_srs, cs, adv, fixed, copies, 11)


@pytest.mark.gpu
def test_native_proof_bit_identical_lookups(hip, golden_srs):
    cs = lookup_circuit(6)
    adv, fixed, copies = lookup_witness(cs, 4)
    _check_against_python(golden_srs, cs, adv, fixed, copies, 2)


@pytest.mark.gpu
def test_native_proof_bit_identical_instances_and_second_phase(hip, golden_srs):
    cs = instance_phase_circuit(6)
    advice, fixed, copies, inst = instance_phase_witness(cs, 6)
    _check_against_python(golden_srs, cs, advice, fixed, copies, 4, instances=inst)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [21, 22, 23])
def test_native_proof_bit_identical_random_circuits(hip, golden_srs, seed):
    cs, adv, fixed, copies = random_circuit(seed)
    _check_against_python(golden_srs, cs, adv, fixed, copies, seed)


@pytest.mark.gpu
def test_native_sharded_commit_path_single_rank(hip, golden_srs):
    """ezkl_prover_cs_set_shard with one rank owning the whole range: every commitment goes through the slice MSM + fold
    callback and the proof bytes do not change; a failing fold aborts the proof; sharding without shared randomness is refused.
    (Two real ranks: tests/test_plonk.py::test_msm_sharded_prover_two_ranks_same_proof.)"""
    import ctypes as C
    from ezkl_amd import backend as B
    cs = lookup_circuit(6)
    adv, fixed, copies = lookup_witness(cs, 4)
    g, gl, pk = _native_setup(golden_srs, cs, fixed, copies)
    want = N.create_proof(pk, g, gl, adv, rng=det_rng(3))
    circ = N.NativeCircuit(cs)
    assert circ.set_shard(None, None) == (0, cs.n) and circ.shard_slice() == (0, cs.n)
    pk2 = N.NativeProvingKey(circ, g, fixed, copies)
    assert pk2.vk()[2] == pk.vk()[2]
    assert N.create_proof(pk2, g, gl, adv, rng=det_rng(3)) == want
    assert N.create_proof(pk2, g, gl, adv, seed=9) == N.create_proof(pk, g, gl, adv, seed=9)
    with pytest.raises(RuntimeError, match="randomness"):
        N.create_proof(pk2, g, gl, adv)                               # seed 0 = OS entropy: ranks would diverge
    calls = []
    bad = N.FOLD_FN(lambda user, pts, count: calls.append(count) or 1)
    assert N.load().ezkl_prover_cs_set_shard(circ.h, C.c_uint32(0), C.c_uint32(cs.n), bad, None) == 0
    with pytest.raises(RuntimeError, match="fold"):
        N.create_proof(pk2, g, gl, adv, rng=det_rng(3))
    assert calls == [cs.n_advice]                                      # the first commit batch: all phase-0 advice columns
    # a slice needs slice-sized SRS handles
    half = N.FOLD_FN(lambda user, pts, count: 0)
    assert N.load().ezkl_prover_cs_set_shard(circ.h, C.c_uint32(0), C.c_uint32(cs.n // 2), half, None) == 0
    with pytest.raises(RuntimeError, match="slice"):
        N.create_proof(pk2, g, gl, adv, rng=det_rng(3))
    gh, glh = B.Bases(golden_srs["g"][: cs.n // 2]), B.Bases(golden_srs["g_lagrange"][: cs.n // 2])
    assert len(N.create_proof(pk2, gh, glh, adv, rng=det_rng(3))) == len(want)       # runs; the bytes are one rank's partial view


@pytest.mark.gpu
def test_native_library_rng_and_rejections(hip, golden_srs):
    """the library's own generator (det-prove: seeded; default: OS entropy), a failing witness, a short proof buffer"""
    import ctypes as C
    from oracle import verifier as V
    cs = mul_add_circuit(6)
    adv, fixed, copies = witness(cs, 3)
    g, gl, pk = _native_setup(golden_srs, cs, fixed, copies)
    gpu = P.GpuBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    _, vk = P.keygen(cs, gpu, fixed, copies)
    g1, g2, s_g2 = setup(golden_srs)
    p1, p2, p3 = N.create_proof(pk, g, gl, adv, seed=7), N.create_proof(pk, g, gl, adv, seed=7), N.create_proof(pk, g, gl, adv, seed=8)
    assert p1 == p2 and p1 != p3 and len(p1) == len(p3)
    assert V.verify(vk, g1, g2, s_g2, p1) and V.verify(vk, g1, g2, s_g2, p3)
    p4, p5 = N.create_proof(pk, g, gl, adv), N.create_proof(pk, g, gl, adv)            # seed 0: OS entropy
    assert p4 != p5 and V.verify(vk, g1, g2, s_g2, p4)
    bad = [a.copy() for a in adv]
    bad[2][3] = P.to_mont(12345)                                                         # breaks c = a*b on row 3: a proof is produced, the verifier rejects it
    assert not V.verify(vk, g1, g2, s_g2, N.create_proof(pk, g, gl, bad, seed=7))
    # proof buffer too small -> EZKL_ERR_NOMEM and the needed length
    L = N.load()
    keep = [np.ascontiguousarray(a, np.uint64) for a in adv]
    arr = (C.c_void_p * len(keep))(*[a.ctypes.data for a in keep])
    buf, plen = (C.c_uint8 * 16)(), C.c_size_t(0)
    rc = L.ezkl_prover_create_proof(pk.h, g.h, gl.h, arr, C.cast(None, N.ADVICE_FN), None, None, None, C.cast(None, N.RNG_FN), None, C.c_uint64(7),
                                    buf, C.c_size_t(16), C.byref(plen), None)
    assert rc == -4 and plen.value == len(p1)
    # SRS of the wrong size
    from ezkl_amd import backend as B
    half = B.Bases(golden_srs["g_lagrange"][:32])
    with pytest.raises(RuntimeError):
        N.create_proof(pk, g, half, adv, seed=7)


@pytest.mark.gpu
def test_native_prover_k12(hip):
    """a larger domain (k = 12, generated SRS): native proof == Python-on-HIP proof, verifier accepts"""
    from ezkl_amd import backend as B
    from oracle import pairing as E, verifier as V
    k = 12
    s = 0x1234567
    g, gl = B.gen_srs(k, s)
    g_pts, gl_pts = g.download(), gl.download()
    cs = mul_add_circuit(k)
    adv, fixed, copies = witness(cs, 9)
    pk = N.NativeProvingKey(N.NativeCircuit(cs), g, fixed, copies)
    gpu = P.GpuBackend(g_pts, gl_pts, k)
    pk_g, vk_g = P.keygen(cs, gpu, fixed, copies)
    proof = N.create_proof(pk, g, gl, adv, rng=det_rng(5))
    assert proof == P.create_proof(pk_g, gpu, adv, det_rng(5))
    G2 = ((0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed, 0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2),
          (0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa, 0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b))
    assert V.verify(vk_g, (1, 2), G2, E.g2_mul(G2, s), proof)


@pytest.mark.gpu
@pytest.mark.parametrize("k,seed", [(7, 31), (8, 32), (9, 33), (10, 34)])
def test_native_random_circuits_larger_domains(hip, k, seed):
    """random gate sets / lookups / copies on generated SRSs of growing size: native C++ proof == Python-on-HIP proof,
    pairing verifier accepts, a corrupted witness is rejected"""
    from ezkl_amd import backend as B
    from oracle import pairing as E, verifier as V
    s = 0xabcdef12345 + seed
    g, gl = B.gen_srs(k, s)
    g_pts, gl_pts = g.download(), gl.download()
    cs, adv, fixed, copies = random_circuit(seed, k=k)
    pk = N.NativeProvingKey(N.NativeCircuit(cs), g, fixed, copies)
    gpu = P.GpuBackend(g_pts, gl_pts, k)
    pk_g, vk_g = P.keygen(cs, gpu, fixed, copies)
    _, _, digest = pk.vk()
    assert digest == vk_g.digest
    proof = N.create_proof(pk, g, gl, adv, rng=det_rng(seed))
    assert proof == P.create_proof(pk_g, gpu, adv, det_rng(seed))
    G2 = ((0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed, 0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2),
          (0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa, 0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b))
    s_g2 = E.g2_mul(G2, s)
    assert V.verify(vk_g, (1, 2), G2, s_g2, proof)
    bad = [a.copy() for a in adv]
    bad[cs.n_advice - 1][5] = P.to_mont((P.from_mont(bad[cs.n_advice - 1][5]) + 1) % P.R)
    assert not V.verify(vk_g, (1, 2), G2, s_g2, N.create_proof(pk, g, gl, bad, seed=3))


@pytest.mark.gpu
def test_native_key_file_roundtrip(hip, golden_srs):
    """save_pk / load_pk in halo2's raw-bytes layout: the native key serialises to the same bytes as the Python codecs (whose
    layout is pinned on the reference's pk.key), loads back, and the loaded key proves byte-identically"""
    from ezkl_amd import backend as B, codecs
    cs = lookup_circuit(6)
    adv, fixed, copies = lookup_witness(cs, 4)
    g, gl, pk = _native_setup(golden_srs, cs, fixed, copies)
    gpu = P.GpuBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    pk_py, _ = P.keygen(cs, gpu, fixed, copies)
    vkb, pkb = P.export_keys(pk_py, gpu)
    data = pk.to_bytes()
    assert data == pkb and data.startswith(vkb)
    back = codecs.read_pk(data, n_perm=len(cs.perm), n_selectors=0)
    assert back["l0"].shape == (1 << cs.ext_k, 4) and len(back["perm_cosets"]) == len(cs.perm)
    pk2 = N.NativeProvingKey.from_bytes(pk.circuit, data)
    assert pk2.vk()[2] == pk.vk()[2]
    assert N.create_proof(pk2, g, gl, adv, rng=det_rng(9)) == N.create_proof(pk, g, gl, adv, rng=det_rng(9))
    for bad in (data[:-1], data + b"\0", b"\x02" + data[1:], data[:7] + bytes(64) + data[71:]):
        if bad == data[:7] + bytes(64) + data[71:]:
            k2 = N.NativeProvingKey.from_bytes(pk.circuit, bad)          # a different commitment parses, but changes the vk digest
            assert k2.vk()[2] != pk.vk()[2]
        else:
            with pytest.raises(RuntimeError):
                N.NativeProvingKey.from_bytes(pk.circuit, bad)
    # a non-canonical field element is rejected
    off = len(vkb) + 4
    evil = bytearray(data); evil[off:off + 32] = b"\xff" * 32
    with pytest.raises(RuntimeError):
        N.NativeProvingKey.from_bytes(pk.circuit, bytes(evil))
    # the one-shot loader (ezkl_prover_pk_read_file: reader threads -> pinned buffers -> HBM, everything but the n-row sections recomputed)
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "pk.key")
        open(path, "wb").write(data)
        pk3 = N.NativeProvingKey.from_file(pk.circuit, path)
        assert pk3.vk()[2] == pk.vk()[2] and pk3.to_bytes() == data
        assert N.create_proof(pk3, g, gl, adv, rng=det_rng(9)) == N.create_proof(pk, g, gl, adv, rng=det_rng(9))
        # a non-canonical element in an n-row section (the first fixed column: after l0 / l_last / l_active and the vector header)
        n, ne = cs.n, 1 << cs.ext_k
        off = len(vkb) + 3 * (4 + 32 * ne) + 4 + 4 * cs.n_fixed + 4 + 32 * 7
        evil = bytearray(data); evil[off:off + 32] = b"\xff" * 32
        open(path, "wb").write(bytes(evil))
        with pytest.raises(RuntimeError):
            N.NativeProvingKey.from_file(pk.circuit, path)
        open(path, "wb").write(data[:-5])
        with pytest.raises(RuntimeError):
            N.NativeProvingKey.from_file(pk.circuit, path)


@pytest.mark.gpu
def test_native_verifier_agrees_with_the_oracle_verifier(hip, golden_srs):
    """ezkl_prover_verify_proof (C++ host, ate pairing in csrc/prover/pairing.hpp) = verify_proof_circuit / the SAFE self-check: accepts
    what the oracle's verifier (optimal-ate pairing, Python) accepts and rejects what it rejects -- lookups, instances, second phase"""
    import os
    from oracle import verifier as V, pyref as pr, pairing as E
    import test_plonk as TP
    from ezkl_amd import backend as B
    srs_bytes = open(os.path.join(os.path.dirname(__file__), "golden", "kzg_k6.srs"), "rb").read()
    g2b, sg2b = srs_bytes[-256:-128], srs_bytes[-128:]
    assert N.g2_mul_generator(1) == g2b                                     # the SRS's g2 is the generator halo2curves uses
    g1, g2, s_g2 = TP.setup(golden_srs)
    bg, bgl = B.Bases(golden_srs["g"]), B.Bases(golden_srs["g_lagrange"])
    # lookups + permutation
    cs = TP.lookup_circuit(6)
    adv, fixed, copies = TP.lookup_witness(cs, 4)
    pk = N.NativeProvingKey(N.NativeCircuit(cs), bg, fixed, copies)
    proof = N.create_proof(pk, bg, bgl, adv, seed=3, check_mode="SAFE", g2=g2b, s_g2=sg2b)
    _, vk = P.keygen(cs, P.GpuBackend(golden_srs["g"], golden_srs["g_lagrange"], 6), fixed, copies)
    assert V.verify(vk, g1, g2, s_g2, proof) and N.verify_proof(pk, g2b, sg2b, proof)
    for pos in (5, 64 * 3 + 40, len(proof) - 200, len(proof) - 10):
        bad = bytearray(proof); bad[pos] ^= 4
        assert N.verify_proof(pk, g2b, sg2b, bytes(bad)) == V.verify(vk, g1, g2, s_g2, bytes(bad)) == False
    assert not N.verify_proof(pk, g2b, sg2b, proof[:-64]) and not N.verify_proof(pk, g2b, sg2b, proof + b"\0" * 32)
    assert not N.verify_proof(pk, sg2b, g2b, proof)                          # the pairing check really uses s: swapped G2 elements fail
    # instances + second-phase advice with a challenge
    cs2 = TP.instance_phase_circuit(6)
    fn, fixed2, copies2, inst = TP.instance_phase_witness(cs2, 5)
    pk2 = N.NativeProvingKey(N.NativeCircuit(cs2), bg, fixed2, copies2)
    p2 = N.create_proof(pk2, bg, bgl, fn, seed=8, instances=inst, check_mode="SAFE", g2=g2b, s_g2=sg2b)
    assert N.verify_proof(pk2, g2b, sg2b, p2, instances=inst)
    assert not N.verify_proof(pk2, g2b, sg2b, p2, instances=[[(inst[0][0] + 1) % P.R]])
    # a witness that violates a gate: the prover still emits bytes, SAFE refuses to return them
    cs3 = TP.mul_add_circuit(6)
    adv3, fixed3, copies3 = TP.witness(cs3, 2)
    pk3 = N.NativeProvingKey(N.NativeCircuit(cs3), bg, fixed3, copies3)
    assert N.verify_proof(pk3, g2b, sg2b, N.create_proof(pk3, bg, bgl, adv3, seed=3, check_mode="SAFE", g2=g2b, s_g2=sg2b))
    bad_adv = [a.copy() for a in adv3]; bad_adv[2][3] = P.to_mont(12345)
    with pytest.raises(RuntimeError, match="SAFE"):
        N.create_proof(pk3, bg, bgl, bad_adv, seed=3, check_mode="SAFE", g2=g2b, s_g2=sg2b)
    # and a lookup input outside its table is refused by the prover itself, as the reference's mv-lookup prover does
    bad_lk = [a.copy() for a in adv]; bad_lk[0][3] = P.to_mont(12345)
    with pytest.raises(RuntimeError, match="not in table"):
        N.create_proof(pk, bg, bgl, bad_lk, seed=3)


@pytest.mark.gpu
def test_keygen_prepares_the_sweep_kernel(hip, golden_srs, tmp_path, monkeypatch):
    """`setup` pays hiprtc: ezkl_prover_keygen compiles the circuit's quotient-sweep kernel and stores the code object in the on-disk cache;
    create_proof then finds it in memory, and a fresh process (the one-shot `prove`) loads it from disk instead of compiling"""
    import subprocess, sys, json
    from ezkl_amd import backend as B
    monkeypatch.setenv("EZKL_HIP_CACHE_DIR", str(tmp_path))
    from conftest import GOLDEN
    NV = N
    cs = mul_add_circuit(6)
    adv, fixed, copies = witness(cs, 5)
    c0, d0, h0 = B.jit_stats()
    g, gl, pk = _native_setup(golden_srs, cs, fixed, copies)
    c1, d1, h1 = B.jit_stats()
    assert (c1 - c0) + (d1 - d0) + (h1 - h0) == 1                      # keygen asked for exactly one kernel: the circuit's sweep
    files = [f for f in os.listdir(tmp_path) if f.startswith("evalh_") and f.endswith(".co")]
    assert files or (h1 - h0) == 1                                      # on disk now (unless this process already held it in memory)
    NV.create_proof(pk, g, gl, adv, seed=3)
    c2, d2, h2 = B.jit_stats()
    # the proof found the sweep kernel in memory; what it still compiled are the two three-instruction helper programs of the
    # permutation argument (numerator / denominator of one chunk: milliseconds of hiprtc)
    assert c2 - c1 <= 2 and d2 == d1 and h2 > h1
    # a fresh process with the same cache directory: nothing compiled there either
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r); import ezkl_amd; ezkl_amd.init(0)\n"
            "from ezkl_amd import backend as B, native as NV\nfrom test_plonk import mul_add_circuit, witness\n"
            "import numpy as np, os\nbuf = open(os.path.join(%r, 'kzg_k6.srs'), 'rb').read()\n"
            "g = np.frombuffer(buf, np.uint64, count=8 * 64, offset=4).reshape(64, 8).copy()\n"
            "cs = mul_add_circuit(6); adv, fixed, copies = witness(cs, 5)\n"
            "pk = NV.NativeProvingKey(NV.NativeCircuit(cs), B.Bases(g), fixed, copies)\nprint(json.dumps(B.jit_stats()))\n"
            % (ROOT, os.path.join(ROOT, "tests"), GOLDEN))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, EZKL_HIP_CACHE_DIR=str(tmp_path)))
    assert r.returncode == 0, r.stderr[-1500:]
    compiled, from_disk, hits = json.loads(r.stdout.strip().splitlines()[-1])
    if files:
        assert compiled == 0 and from_disk == 1


@pytest.mark.gpu
def test_swap_proof_commitments_cross_check(hip, golden_srs):
    """The reference's one cross-check between two code paths of this boundary (tests/integration_tests.rs:699-707, 1380-1392
    `swap-proof-commitments` after kzg_prove_and_verify_kzg_output; src/pfsys/mod.rs:540): PolyCommitChip::commit of a message
    (gen-witness: the unusable rows hold Blind::default() = 1) must EQUAL the prover's commitment of the advice column that holds the
    message with polycommit visibility (an unblinded column).  Here: the first advice column of a small circuit is unblinded; the first
    64 bytes of the proof (big-endian x | y of the first advice commitment) are polycommit_commit's first point."""
    from ezkl_amd import backend as B
    from conftest import fe_from_int
    k = 6
    a, b, c = P.adv(0), P.adv(1), P.adv(2)
    cs = P.ConstraintSystem(k, 3, 3, [P.fix(0) * (c - a * b), P.fix(1) * (c - P.adv(2, -1) - a * b)], [("adv", 0), ("adv", 1), ("adv", 2), ("fix", 2)], unblinded=(0,))
    adv, fixed, copies = witness(cs, 7)
    g, gl, pk = _native_setup(golden_srs, cs, fixed, copies)
    proof = N.create_proof(pk, g, gl, adv, seed=3)
    u = cs.usable
    params = hip.ParamsKZG.read(golden_srs["buf"])
    pts = B.polycommit_commit(np.ascontiguousarray(adv[0][:u]), cs.n - u, params)          # the message = the column's usable rows
    x, y = P.point_to_ints(pts[0])
    assert proof[:64] == x.to_bytes(32, "big") + y.to_bytes(32, "big")
    # a blinded column commits to something else (random unusable rows): the equality is a property of the unblinded column
    x1, y1 = P.point_to_ints(B.polycommit_commit(np.ascontiguousarray(adv[1][:u]), cs.n - u, params)[0])
    assert proof[64:128] != x1.to_bytes(32, "big") + y1.to_bytes(32, "big")
    params.free()


@pytest.mark.gpu
def test_integer_rep_advice_columns_give_the_same_proof(hip, golden_srs):
    """ezkl_prover_create_proof_fmt: advice handed over as the INTEGERS its cells were made from (ezkl's IntegerRep i128,
    /root/reference/src/fieldutils.rs:6-17) -- int64 columns (8 B per cell across PCIe) and 128-bit columns (16 B) -- is expanded on the
    device to exactly the 32-byte Montgomery columns: same proof bytes, accepted by the verifier.  Then every edge of integer_rep_to_felt
    (negative values, 0, +-1, the int64 and i128 extremes, IntegerRep::MIN) on columns that satisfy nothing: the bytes still agree."""
    from oracle import verifier as V
    from conftest import fe_from_int
    cs = mul_add_circuit(6)
    adv, fixed, copies = witness(cs, 9)
    g, gl, pk = _native_setup(golden_srs, cs, fixed, copies)
    want = N.create_proof(pk, g, gl, adv, rng=det_rng(3))
    ints = [[P.from_mont(r) for r in col] for col in adv]
    assert max(ints[0] + ints[1]) < 1 << 62 and max(ints[2]) < 1 << 127       # a, b fit int64; the accumulator column goes as 128-bit integers
    def i128(vals):
        return np.array([[(v % (1 << 128)) & 0xffffffffffffffff, (v % (1 << 128)) >> 64] for v in vals], np.uint64)
    as_int = [np.array(ints[0], np.int64), np.array(ints[1], np.int64), i128(ints[2])]
    assert N.create_proof(pk, g, gl, as_int, rng=det_rng(3)) == want
    assert N.create_proof(pk, g, gl, [adv[0], as_int[1], i128(ints[2])], rng=det_rng(3)) == want              # formats mix per column
    pk_g, vk_g = P.keygen(cs, P.GpuBackend(golden_srs["g"], golden_srs["g_lagrange"], cs.k), fixed, copies)
    g1, g2, s_g2 = setup(golden_srs)
    assert V.verify(vk_g, g1, g2, s_g2, want)
    n = cs.n
    edge64 = [0, 1, -1, 2, -2, (1 << 63) - 1, -(1 << 63), 12345, -12345, 1 << 62, -(1 << 62)]
    edge128 = edge64 + [(1 << 127) - 1, -(1 << 127), (1 << 64), -(1 << 64), (1 << 100) + 7, -((1 << 100) + 7), (1 << 64) - 1, -((1 << 64) - 1)]
    rng = np.random.default_rng(4)
    c64 = (edge64 + [int(x) for x in rng.integers(-(1 << 62), 1 << 62, n)])[:n]
    c128 = (edge128 + [int(x) * (1 << 40) * (-1 if i & 1 else 1) for i, x in enumerate(rng.integers(0, 1 << 62, n))])[:n]
    mont = lambda vals: np.stack([fe_from_int(v % P.R) for v in vals])
    ref = N.create_proof(pk, g, gl, [mont(c64), mont(c128), mont(c64[::-1])], rng=det_rng(8))
    got = N.create_proof(pk, g, gl, [np.array(c64, np.int64), i128(c128), np.array(c64[::-1], np.int64)], rng=det_rng(8))
    assert got == ref and ref != want


@pytest.mark.gpu
def test_keygen_refuses_up_front_when_the_key_cannot_fit(hip, golden_srs, monkeypatch):
    """a circuit whose key + witness columns cannot fit the device used to fail somewhere inside keygen with a bare
    hipMalloc error.  keygen now estimates the resident bytes first and fails with the sizes (EZKL_ERR_NOMEM = -4); the test hook
    EZKL_PROVER_ASSUME_FREE_GIB stands in for a full device"""
    cs = mul_add_circuit(6)
    adv, fixed, copies = witness(cs, 9)
    from ezkl_amd import backend as B
    g = B.Bases(golden_srs["g"])
    monkeypatch.setenv("EZKL_PROVER_ASSUME_FREE_GIB", "0.00001")
    with pytest.raises(Exception) as e:
        N.NativeProvingKey(N.NativeCircuit(cs), g, fixed, copies)
    msg = str(e.value)
    assert "GiB" in msg and "key columns" in msg and "owner mode" in msg, msg
    # the documented override: the check steps aside and the small key simply builds
    monkeypatch.setenv("EZKL_PROVER_SKIP_FIT_CHECK", "1")
    assert N.NativeProvingKey(N.NativeCircuit(cs), g, fixed, copies).vk()[2]
    monkeypatch.delenv("EZKL_PROVER_SKIP_FIT_CHECK")
    monkeypatch.delenv("EZKL_PROVER_ASSUME_FREE_GIB")
    pk = N.NativeProvingKey(N.NativeCircuit(cs), g, fixed, copies)          # and with the real device it goes through
    assert pk.vk()[2]
    # the key alone fits, key + the estimated witness columns do not: a warning, not a refusal (a keygen-only caller never proves).
    # key = (cols (2 + E) + 4 E) n 32 bytes
    E = 1 << (cs.ext_k - cs.k)
    key_bytes = ((cs.n_fixed + len(cs.perm)) * (2 + E) + 4 * E) * (1 << cs.k) * 32
    monkeypatch.setenv("EZKL_PROVER_ASSUME_FREE_GIB", repr((key_bytes + 64) / 2.0 ** 30))
    assert N.NativeProvingKey(N.NativeCircuit(cs), g, fixed, copies).vk()[2]
    # the read paths check as well (they used to fail deep inside hipMalloc)
    blob = pk.to_bytes()
    monkeypatch.setenv("EZKL_PROVER_ASSUME_FREE_GIB", "0.00001")
    with pytest.raises(Exception) as e:
        N.NativeProvingKey.from_bytes(N.NativeCircuit(cs), blob)
    assert "pk_read" in str(e.value) and "GiB" in str(e.value)
    monkeypatch.delenv("EZKL_PROVER_ASSUME_FREE_GIB")
    assert N.NativeProvingKey.from_bytes(N.NativeCircuit(cs), blob).vk()[2] == pk.vk()[2]


@pytest.mark.gpu
@pytest.mark.parametrize("make", ["lookup", "instance_phase", "random10"])
def test_streamed_key_proves_the_same_bytes(hip, golden_srs, monkeypatch, tmp_path, make):
    """VERDICT r05 item 6, the one-GPU degraded mode: EZKL_KEY_COSETS=recompute holds the key as values + coefficients only (no extended
    column of a fixed / permutation polynomial, none of a witness column either) and the quotient sweep rebuilds coset b of every column
    from its coefficients when it reaches unit b.  Same witness + randomness => the SAME proof bytes as the resident key, the same key file,
    and the three ways to get a key (keygen, pk_read, pk_read_file) all honour the mode.  `auto` picks it when the resident form does not
    fit what the device has left (EZKL_PROVER_ASSUME_FREE_GIB stands in for a full device)."""
    from ezkl_amd import backend as B
    if make == "lookup":
        k, cs = 6, lookup_circuit(6)
        adv, fixed, copies = lookup_witness(cs, 4)
        inst = []
    elif make == "instance_phase":
        k, cs = 6, instance_phase_circuit(6)
        adv, fixed, copies, inst = instance_phase_witness(cs, 3)
    else:
        k = 10
        cs, adv, fixed, copies = random_circuit(34, k=10)
        inst = []
    if k == 6:
        g, gl = B.Bases(golden_srs["g"]), B.Bases(golden_srs["g_lagrange"])
    else:
        g, gl = B.gen_srs(k, 0xabcdef12345)
    kw = dict(instances=inst) if inst else {}
    pk = N.NativeProvingKey(N.NativeCircuit(cs), g, fixed, copies)
    want = N.create_proof(pk, g, gl, adv, rng=det_rng(7), **kw)
    data = pk.to_bytes()
    res = pk.residency()
    assert res["streamed"] is False and res["cosets"] == res["E"]
    monkeypatch.setenv("EZKL_KEY_COSETS", "recompute")
    pk_s = N.NativeProvingKey(N.NativeCircuit(cs), g, fixed, copies)
    rs = pk_s.residency()
    assert rs["streamed"] is True and rs["key_bytes"] < res["key_bytes"]
    n_key_cols = cs.n_fixed + len(cs.perm)
    assert res["key_bytes"] - rs["key_bytes"] == n_key_cols * res["E"] * (1 << k) * 32          # exactly the extended key columns
    assert pk_s.vk()[2] == pk.vk()[2]
    assert N.create_proof(pk_s, g, gl, adv, rng=det_rng(7), **kw) == want
    assert pk_s.to_bytes() == data                                                                   # the file still gets complete extended columns
    pk_b = N.NativeProvingKey.from_bytes(N.NativeCircuit(cs), data)
    assert pk_b.residency()["streamed"] and N.create_proof(pk_b, g, gl, adv, rng=det_rng(7), **kw) == want
    path = tmp_path / "pk.key"
    path.write_bytes(data)
    pk_f = N.NativeProvingKey.from_file(N.NativeCircuit(cs), str(path))
    assert pk_f.residency()["streamed"] and N.create_proof(pk_f, g, gl, adv, rng=det_rng(7), **kw) == want
    # a streamed key is a one-rank mode: a sharded circuit refuses it
    monkeypatch.setenv("EZKL_KEY_COSETS", "bogus")
    with pytest.raises(RuntimeError):
        N.NativeProvingKey(N.NativeCircuit(cs), g, fixed, copies)
    # auto: resident when it fits, streamed when only that fits, refused when not even the streamed KEY fits
    monkeypatch.setenv("EZKL_KEY_COSETS", "auto")
    assert N.NativeProvingKey(N.NativeCircuit(cs), g, fixed, copies).residency()["streamed"] is False
    E_, nn = res["E"], 1 << k
    wcols = cs.n_advice + 3 * len(cs.lookups) + cs.n_chunks + 2
    resident_need = res["key_bytes"] - nn * 32 + wcols * (2 + E_) * nn * 32                            # (key_bytes counts omega_col: one column more)
    streamed_need = (n_key_cols * 2 + 4 * E_) * nn * 32 + (wcols * 2 + n_key_cols + wcols + 2 * E_) * nn * 32
    assert streamed_need < resident_need
    monkeypatch.setenv("EZKL_PROVER_ASSUME_FREE_GIB", repr((streamed_need + resident_need) / 2 / 2.0 ** 30))
    pk_a = N.NativeProvingKey(N.NativeCircuit(cs), g, fixed, copies)
    assert pk_a.residency()["streamed"] is True
    monkeypatch.delenv("EZKL_PROVER_ASSUME_FREE_GIB")
    assert N.create_proof(pk_a, g, gl, adv, rng=det_rng(7), **kw) == want
    monkeypatch.setenv("EZKL_PROVER_ASSUME_FREE_GIB", "0.0000001")
    with pytest.raises(Exception, match="GiB"):
        N.NativeProvingKey(N.NativeCircuit(cs), g, fixed, copies)
