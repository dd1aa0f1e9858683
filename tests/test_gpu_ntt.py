"""GPU parity: HIP NTT (through the C ABI) vs the oracle and the reference's golden pk.key vectors."""
import numpy as np
import pytest
from conftest import R, SEED, fe_from_int, fe_to_int, rand_fr
from oracle import binding as ob, pyref as pr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", list(range(0, 15)) + [16, 17, 18])
def test_forward_matches_oracle(hip, k):
    rng = np.random.default_rng(100 + k)
    a = rand_fr(rng, 1 << k)
    w = ob.omega(k)
    got = hip.ntt(a, k, w)
    assert (got == ob.fft(a, k, w)).all()


@pytest.mark.parametrize("k", [1, 4, 9, 12, 13, 17])
def test_inverse_scaled_matches_oracle(hip, k):
    rng = np.random.default_rng(200 + k)
    a = rand_fr(rng, 1 << k)
    d = hip.EvaluationDomain(2, k)
    got = d.lagrange_to_coeff(a)
    assert (got == ob.lagrange_to_coeff(a, k)).all()
    assert (d.coeff_to_lagrange(got) == a).all()


def test_edge_inputs(hip):
    k = 10
    w = ob.omega(k)
    z = np.zeros((1 << k, 4), np.uint64)
    assert (hip.ntt(z, k, w) == 0).all()
    one = np.zeros((1 << k, 4), np.uint64); one[0] = fe_from_int(1)
    assert (hip.ntt(one, k, w) == np.tile(fe_from_int(1), (1 << k, 1))).all()       # delta -> all ones
    top = np.tile(fe_from_int(R - 1), (1 << k, 1))                                    # maximum residues
    assert (hip.ntt(top, k, w) == ob.fft(top, k, w)).all()


@pytest.mark.parametrize("k", [3, 8, 11, 13, 16])
def test_words_that_are_not_canonical(hip, k):
    """The pass takes its elements as 256-bit words (radix-2^29 lazy limbs inside, a canonical result out): words at and above the modulus --
    what a lazily reduced producer could hand over -- transform like their residues, and the all-ones word does not overflow a limb
    (single-pass sizes, two and three passes)"""
    rng = np.random.default_rng(300 + k)
    n = 1 << k
    a = rand_fr(rng, n)
    words = [int.from_bytes(a[i].tobytes(), "little") for i in range(n)]
    special = [(1 << 256) - 1, R, R + 5, 2 * R - 1, 5 * R + 12345, (1 << 256) - R, 0, R - 1]
    for j, v in enumerate(special):
        words[(j * 7919) % n] = v
    raw = np.array([[(v >> (64 * q)) & 0xFFFFFFFFFFFFFFFF for q in range(4)] for v in words], np.uint64)
    red = np.array([[((v % R) >> (64 * q)) & 0xFFFFFFFFFFFFFFFF for q in range(4)] for v in words], np.uint64)
    w = ob.omega(k)
    assert (hip.ntt(raw, k, w) == ob.fft(red, k, w)).all()
    d = hip.EvaluationDomain(2, k)
    assert (d.lagrange_to_coeff(raw) == ob.lagrange_to_coeff(red, k)).all()


def test_golden_pk_vectors(hip, golden_pk):
    """reference fixture: fixed_polys == iNTT(fixed_values), fixed_cosets == coeff_to_extended(polys)"""
    d = hip.EvaluationDomain(9, 6)      # degree 9 -> ext_k = 9 as in tests/assets/pk.key
    assert d.ext_k == 9
    for nv, npoly, nc in (("fixed_values", "fixed_polys", "fixed_cosets"), ("permutations", "perm_polys", "perm_cosets")):
        for v, p, c in zip(golden_pk[nv], golden_pk[npoly], golden_pk[nc]):
            assert (d.lagrange_to_coeff(v) == p).all()
            assert (d.coeff_to_extended(p) == c).all()
            back = d.extended_to_coeff(c)
            assert (back[:64] == p).all() and (back[64:] == 0).all()
    one = fe_from_int(1)
    e0 = np.zeros((64, 4), np.uint64); e0[0] = one
    assert (d.coeff_to_extended(d.lagrange_to_coeff(e0)) == golden_pk["l0"]).all()


@pytest.mark.parametrize("k,ek", [(3, 5), (8, 10), (10, 13), (12, 14)])
def test_coset_and_vanishing_match_oracle(hip, k, ek):
    rng = np.random.default_rng(300 + k)
    d = hip.EvaluationDomain((1 << (ek - k)) + 1, k)
    assert d.ext_k == ek
    p = rand_fr(rng, 1 << k)
    ext = d.coeff_to_extended(p)
    assert (ext == ob.coeff_to_extended(p, k, ek)).all()
    assert (d.extended_to_coeff(ext) == ob.extended_to_coeff(ext, ek)).all()
    assert (d.divide_by_vanishing_poly(ext) == ob.divide_by_vanishing(ext, k, ek)).all()


def test_batch_device_resident(hip):
    from ezkl_amd import backend as B
    rng = np.random.default_rng(7)
    k, batch = 11, 5
    a = rand_fr(rng, batch << k).reshape(batch, 1 << k, 4)
    w = ob.omega(k)
    buf = B.DeviceBuffer.from_numpy(a)
    B.ntt_dev(buf.ptr, k, w, batch=batch)
    got = buf.to_numpy(shape=a.shape)
    for b in range(batch):
        assert (got[b] == ob.fft(a[b], k, w)).all()


def test_full_size_2_22_properties(hip):
    """BASELINE configs[1] size: round trip, linearity and a spot check against the defining sum"""
    from ezkl_amd import backend as B
    k = 22
    n = 1 << k
    rng = np.random.default_rng(22)
    a, b = rand_fr(rng, n), rand_fr(rng, n)
    d = hip.EvaluationDomain(2, k)
    fa = hip.ntt(a, k, d.omega)
    assert (hip.ntt(fa, k, d.omega_inv, inverse=True) == a).all()              # iNTT(NTT(a)) == a
    fb = hip.ntt(b, k, d.omega)
    bufa, bufb = B.DeviceBuffer.from_numpy(a), B.DeviceBuffer.from_numpy(b)
    B.vec_op("add", bufa.ptr, bufb.ptr, bufa.ptr, n)
    s = bufa.to_numpy(shape=(n, 4))
    fs = hip.ntt(s, k, d.omega)
    bufa2, bufb2 = B.DeviceBuffer.from_numpy(fa), B.DeviceBuffer.from_numpy(fb)
    B.vec_op("add", bufa2.ptr, bufb2.ptr, bufa2.ptr, n)
    assert (fs == bufa2.to_numpy(shape=(n, 4))).all()                           # NTT(a+b) == NTT(a)+NTT(b)
    # X[0] = sum a_i ; X[n/2] = sum (-1)^i a_i : checked with exact integer arithmetic on the host
    ints = [fe_to_int(x) for x in a[: 1 << 12]]   # spot check uses a short-prefix input
    short = np.zeros((n, 4), np.uint64); short[: 1 << 12] = a[: 1 << 12]
    f = hip.ntt(short, k, d.omega)
    assert fe_to_int(f[0]) == sum(ints) % R
    assert fe_to_int(f[n // 2]) == sum(v if i % 2 == 0 else -v for i, v in enumerate(ints)) % R
    w = pr.omega(k)
    j = 123457
    assert fe_to_int(f[j]) == sum(v * pow(w, i * j, R) for i, v in enumerate(ints)) % R


def test_device_resident_pk_columns(hip, golden_pk):
    """pk columns loaded into HBM; the coset NTT of the resident fixed polys reproduces the
    resident fixed cosets of the reference's pk.key"""
    import ctypes as C
    from ezkl_amd import backend as B, codecs, lib as L
    vk = dict(k=6, compress_selectors=True, fixed_commitments=np.zeros((0, 8), np.uint64),
              permutation_commitments=np.zeros((0, 8), np.uint64), selectors=np.zeros((0, 64), bool))
    pk = dict(vk=vk, l0=golden_pk["l0"], l_last=golden_pk["l_last"], l_active_row=golden_pk["l_active_row"],
              fixed_values=list(golden_pk["fixed_values"]), fixed_polys=list(golden_pk["fixed_polys"]),
              fixed_cosets=list(golden_pk["fixed_cosets"]), permutations=list(golden_pk["permutations"]),
              perm_polys=list(golden_pk["perm_polys"]), perm_cosets=list(golden_pk["perm_cosets"]))
    dev = codecs.ProvingKeyDevice(pk)
    assert dev.ext_k == 9 and dev.nbytes() > 0
    out = B.DeviceBuffer(512 * 32)
    for poly, coset in list(zip(dev.fixed_polys, dev.fixed_cosets)) + list(zip(dev.perm_polys, dev.perm_cosets)):
        L.check(L.load().ezkl_hip_coset_ntt_dev(C.c_void_p(poly.ptr), C.c_void_p(out.ptr), C.c_size_t(1), C.c_size_t(64), C.c_size_t(512),
                                                C.c_uint32(6), C.c_uint32(9), C.c_int(0), C.c_void_p(None)), "coset")
        assert (out.to_numpy() == coset.to_numpy()).all()


def test_extended_domain_size_2_25(hip):
    """largest extended domain of the BASELINE configs (k = 22, ext_k = 25: 1 GiB per column): round trip of the
    coset pair and of a plain transform, plus the defining sum at two positions for a sparse input"""
    from ezkl_amd import backend as B
    import ctypes as C
    from ezkl_amd import lib as L
    k, ek = 22, 25
    n, ne = 1 << k, 1 << ek
    rng = np.random.default_rng(25)
    a = rand_fr(rng, n)
    da = B.DeviceBuffer.from_numpy(a)
    dext = B.DeviceBuffer(ne * 32)
    lib = L.load()
    L.check(lib.ezkl_hip_coset_ntt_dev(C.c_void_p(da.ptr), C.c_void_p(dext.ptr), C.c_size_t(1), C.c_size_t(n), C.c_size_t(ne),
                                       C.c_uint32(k), C.c_uint32(ek), C.c_int(0), C.c_void_p(None)), "coset")
    L.check(lib.ezkl_hip_coset_ntt_dev(C.c_void_p(dext.ptr), C.c_void_p(dext.ptr), C.c_size_t(1), C.c_size_t(ne), C.c_size_t(ne),
                                       C.c_uint32(k), C.c_uint32(ek), C.c_int(1), C.c_void_p(None)), "icoset")
    back = dext.to_numpy(shape=(ne, 4))
    assert (back[:n] == a).all() and not back[n:].any()
    d = hip.EvaluationDomain(2, ek)
    sparse = np.zeros((ne, 4), np.uint64)
    idx = [0, 1, 12345, ne - 1]
    vals = rand_fr(rng, 4)
    sparse[idx] = vals
    ds = B.DeviceBuffer.from_numpy(sparse)
    B.ntt_dev(ds.ptr, ek, d.omega)
    f = ds.to_numpy(shape=(ne, 4))
    w = pr.omega(ek)
    for j in (0, 7, ne // 2 + 3):
        want = sum(fe_to_int(v) * pow(w, i * j, R) for i, v in zip(idx, vals)) % R
        assert fe_to_int(f[j]) == want
    B.ntt_dev(ds.ptr, ek, d.omega_inv, inverse=True)
    assert (ds.to_numpy(shape=(ne, 4)) == sparse).all()


def test_benchmark_size_2_22_full_oracle_compare(hip):
    """BASELINE configs[1] size, byte-compared with the C oracle over the WHOLE output (about a second of OpenMP CPU work per
    transform): forward 2^22, scaled inverse 2^22, coeff_to_extended 2^20 -> 2^22 and extended_to_coeff back"""
    k = 22
    n = 1 << k
    rng = np.random.default_rng(2222)
    a = rand_fr(rng, n)
    d = hip.EvaluationDomain(2, k)
    assert (hip.ntt(a, k, d.omega) == ob.fft(a, k, ob.omega(k))).all()
    assert (d.lagrange_to_coeff(a) == ob.lagrange_to_coeff(a, k)).all()
    d4 = hip.EvaluationDomain(5, 20)                       # degree 5 -> 4 n rows: ext_k = 22
    assert d4.ext_k == 22
    p = a[: 1 << 20]
    ext = d4.coeff_to_extended(p)
    assert (ext == ob.coeff_to_extended(p, 20, 22)).all()
    assert (d4.extended_to_coeff(ext) == ob.extended_to_coeff(ext, 22)).all()


def test_extended_2_24_full_oracle_compare(hip):
    """the extended domain of a k = 22 circuit of degree 5 (2^22 -> 2^24), whole-output byte compare with the oracle"""
    rng = np.random.default_rng(2424)
    p = rand_fr(rng, 1 << 22)
    d = hip.EvaluationDomain(5, 22)
    assert d.ext_k == 24
    ext = d.coeff_to_extended(p)
    assert (ext == ob.coeff_to_extended(p, 22, 24)).all()
    back = d.extended_to_coeff(ext)
    assert (back[: 1 << 22] == p).all() and (back[1 << 22:] == 0).all()


def _to_cm(nat, k, ek):
    """natural-order extended column -> coset-major: cm[b 2^k + j] = nat[E j + b]"""
    E = 1 << (ek - k)
    return np.ascontiguousarray(nat.reshape(1 << k, E, 4).transpose(1, 0, 2)).reshape(1 << ek, 4)


@pytest.mark.parametrize("k,ek", [(1, 2), (3, 5), (6, 9), (8, 10), (10, 11), (11, 13), (12, 14), (13, 16), (16, 18), (17, 19)])
def test_coset_major_matches_oracle(hip, golden_pk, k, ek):
    """ezkl_hip_coeff_to_cosets_dev: E transforms of 2^k points = EvaluationDomain::coeff_to_extended with the cosets stored one after
    the other -- byte-compared with the oracle's natural-order transform (and, at k = 6, with the reference's pk.key cosets), batched"""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(900 + k)
    n, ne, batch = 1 << k, 1 << ek, 3
    polys = [rand_fr(rng, n) for _ in range(batch)]
    if (k, ek) == (6, 9):
        polys = [np.ascontiguousarray(golden_pk["fixed_polys"][i]) for i in range(batch)]
    din = B.DeviceBuffer.from_numpy(np.stack(polys))
    dout = B.DeviceBuffer.from_numpy(np.zeros((batch, ne, 4), np.uint64))
    B.coeff_to_cosets_dev(din.ptr, dout.ptr, k, ek, batch=batch)
    got = dout.to_numpy(shape=(batch, ne, 4))
    for b in range(batch):
        want = ob.coeff_to_extended(polys[b], k, ek)
        if (k, ek) == (6, 9):
            assert (want == golden_pk["fixed_cosets"][b]).all()
        assert (got[b] == _to_cm(want, k, ek)).all()
    # a RANGE of cosets (what one rank of a sharded prover keeps of a key column): every aligned power-of-two range equals the slice
    E = 1 << (ek - k)
    cnt = 1
    while cnt <= E:
        for first in range(0, E, cnt):
            dr = B.DeviceBuffer.from_numpy(np.zeros((batch, cnt * n, 4), np.uint64))
            B.coeff_to_cosets_range_dev(din.ptr, dr.ptr, k, ek, first, cnt, batch=batch)
            assert (dr.to_numpy(shape=(batch, cnt * n, 4)) == got[:, first * n:(first + cnt) * n]).all(), (first, cnt)
            if E > 8 and first >= 2 * cnt:
                break
        cnt *= 2
    with pytest.raises(Exception):
        B.coeff_to_cosets_range_dev(din.ptr, dout.ptr, k, ek, E - 1, 2)
    # the transposition both ways
    dnat = B.DeviceBuffer.from_numpy(np.zeros((ne, 4), np.uint64))
    B.cosets_transpose_dev(dout.ptr, dnat.ptr, k, ek, to_natural=True)
    assert (dnat.to_numpy(shape=(ne, 4)) == ob.coeff_to_extended(polys[0], k, ek)).all()
    dcm = B.DeviceBuffer.from_numpy(np.zeros((ne, 4), np.uint64))
    B.cosets_transpose_dev(dnat.ptr, dcm.ptr, k, ek, to_natural=False)
    assert (dcm.to_numpy(shape=(ne, 4)) == got[0]).all()


@pytest.mark.parametrize("k,ek", [(20, 22), (22, 24)])
def test_coset_major_full_oracle_compare(hip, k, ek):
    """the extended domains of the k = 20 and k = 22 circuits (degree 5): the coset-major transform against the oracle's
    coeff_to_extended over the WHOLE output"""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(5000 + k)
    p = rand_fr(rng, 1 << k)
    din = B.DeviceBuffer.from_numpy(p)
    dout = B.DeviceBuffer.from_numpy(np.zeros((1 << ek, 4), np.uint64))
    B.coeff_to_cosets_dev(din.ptr, dout.ptr, k, ek)
    assert (dout.to_numpy(shape=(1 << ek, 4)) == _to_cm(ob.coeff_to_extended(p, k, ek), k, ek)).all()
