"""CPU tests: pin the oracle (C restatement) on the reference's golden fixtures and on the big-int pyref."""
import os
import numpy as np
import pytest
from conftest import GOLDEN, R, Q, MONT, SEED, fe_from_int, fe_to_int, rand_fr
from oracle import binding as ob, pyref as pr


def test_field_against_bigint():
    rng = np.random.default_rng(1)
    import random
    random.seed(2)
    for _ in range(200):
        a, b = random.randrange(R), random.randrange(R)
        assert fe_to_int(ob.fr_mul(fe_from_int(a), fe_from_int(b))) == a * b % R
        a, b = random.randrange(Q), random.randrange(Q)
        assert fe_to_int(ob.fq_mul(fe_from_int(a, Q), fe_from_int(b, Q)), Q) == a * b % Q
    for x in (0, 1, R - 1, 2, (R - 1) // 2):
        got = fe_to_int(ob.fr_inv(fe_from_int(x)))
        assert got == (pow(x, -1, R) if x else 0)
    assert fe_to_int(ob.omega(6)) == pr.omega(6)
    assert pow(pr.omega(28), 1 << 28, R) == 1 and pow(pr.omega(28), 1 << 27, R) != 1


def test_golden_pk_ntt(golden_pk):
    """fixed_polys == iNTT(fixed_values); fixed_cosets == coeff_to_extended(fixed_polys)"""
    for name_v, name_p, name_c in (("fixed_values", "fixed_polys", "fixed_cosets"), ("permutations", "perm_polys", "perm_cosets")):
        for v, p, c in zip(golden_pk[name_v], golden_pk[name_p], golden_pk[name_c]):
            assert (ob.lagrange_to_coeff(v, 6) == p).all()
            assert (ob.coeff_to_lagrange(p, 6) == v).all()
            assert (ob.coeff_to_extended(p, 6, 9) == c).all()
            back = ob.extended_to_coeff(c, 9)
            assert (back[:64] == p).all() and (back[64:] == 0).all()


def test_golden_l0_l_last_l_active(golden_pk):
    one = fe_from_int(1)
    e0 = np.zeros((64, 4), np.uint64); e0[0] = one
    assert (ob.coeff_to_extended(ob.lagrange_to_coeff(e0, 6), 6, 9) == golden_pk["l0"]).all()
    ll = ob.coeff_to_lagrange(ob.extended_to_coeff(golden_pk["l_last"], 9)[:64], 6)
    want = np.zeros((64, 4), np.uint64); want[58] = one       # n - blinding_factors - 1, blinding = 5
    assert (ll == want).all()
    la = ob.coeff_to_lagrange(ob.extended_to_coeff(golden_pk["l_active_row"], 9)[:64], 6)
    want = np.zeros((64, 4), np.uint64); want[:58] = one
    assert (la == want).all()


def test_golden_srs_msm(golden_srs, golden_pk):
    """sum(g_lagrange) == g[0]; MSM(v, g_lagrange) == MSM(iNTT(v), g)"""
    g, gl = golden_srs["g"], golden_srs["g_lagrange"]
    assert all(ob.g1_on_curve(p) for p in g) and all(ob.g1_on_curve(p) for p in gl)
    ones = np.tile(fe_from_int(1), (64, 1))
    assert (ob.msm(ones, gl) == g[0]).all()
    assert pr.g1_from_bytes(g[0].tobytes()) == (1, 2)
    for v, p in zip(golden_pk["fixed_values"], golden_pk["fixed_polys"]):
        a, b = ob.msm(v, gl), ob.msm(p, g)
        assert (a == b).all() and (a == ob.msm(v, gl, naive=True)).all()


def test_golden_vk_sigma_commitments():
    """identity sigma columns: commitment == delta^j * (s G) with the PUBLIC k=1 SRS"""
    vk = open(os.path.join(GOLDEN, "vk_k6.key"), "rb").read()
    srs1 = pr.parse_srs(open(os.path.join(GOLDEN, "kzg_k1_public.srs"), "rb").read())
    assert vk[0] == 3 and vk[1] == 6
    nfixed = int.from_bytes(vk[3:7], "little")
    off = 7 + 64 * nfixed
    sG = np.frombuffer(srs1["g"][1], np.uint64)
    hits = 0
    for j in range(32):
        commit = np.frombuffer(vk[off + 64 * j: off + 64 * j + 64], np.uint64)
        want = ob.g1_mul(sG, fe_from_int(pow(pr.DELTA, j, R)))
        hits += int((commit == want).all())
    assert hits == 18


def test_pippenger_vs_naive_and_edges():
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 5, 31, 32, 33, 200, 1000):
        b = ob.gen_bases(SEED, n)
        s = rand_fr(rng, n)
        assert (ob.msm(s, b) == ob.msm(s, b, naive=True)).all()
    b = ob.gen_bases(SEED, 64)
    s = rand_fr(rng, 64)
    s[:8] = 0; s[8] = fe_from_int(R - 1); s[9] = fe_from_int(1); s[10] = fe_from_int((R - 1) // 2)
    b[20] = b[21]; s[20] = fe_from_int(5); s[21] = fe_from_int(R - 5)       # cancels to identity
    b[30] = 0                                                                # identity base
    assert (ob.msm(s, b) == ob.msm(s, b, naive=True)).all()
    z = ob.msm(np.zeros((64, 4), np.uint64), b)
    assert (z == 0).all()
    # python big-int cross-check on a tiny case
    pts = [pr.g1_from_bytes(x.tobytes()) for x in b[:6]]
    sc = [fe_to_int(x) for x in s[:6]]
    assert pr.g1_to_bytes(pr.msm(sc, pts)) == ob.msm(s[:6], b[:6]).tobytes()


def test_fft_against_pyref():
    rng = np.random.default_rng(4)
    for k in (0, 1, 2, 3, 5, 8):
        a = rand_fr(rng, 1 << k)
        w = pr.omega(k)
        want = pr.ntt([fe_to_int(x) for x in a], w)
        got = ob.fft(a, k, fe_from_int(w))
        assert [fe_to_int(x) for x in got] == want


def test_vanishing_and_batch_invert():
    rng = np.random.default_rng(5)
    k, ek = 4, 6
    a = rand_fr(rng, 1 << ek)
    got = ob.divide_by_vanishing(a, k, ek)
    we = pr.omega(ek)
    for i in range(1 << ek):
        x = pr.ZETA * pow(we, i, R) % R
        t = pow((pow(x, 1 << k, R) - 1) % R, -1, R)
        assert fe_to_int(got[i]) == fe_to_int(a[i]) * t % R
    a[3] = 0
    inv = ob.batch_invert(a)
    for i in range(1 << ek):
        x = fe_to_int(a[i])
        assert fe_to_int(inv[i]) == (pow(x, -1, R) if x else 0)


def test_eval_program_against_python():
    """GraphEvaluator restatement vs a direct big-int evaluation of the same tiny gate program"""
    from ezkl_amd.backend import GraphProgram
    rng = np.random.default_rng(6)
    k, ek = 3, 5
    ne = 1 << ek
    cols = [rand_fr(rng, ne) for _ in range(3)]
    chal = rand_fr(rng, 2)
    prog = GraphProgram(k, ek)
    a, b, c_prev = prog.column(0, 0), prog.column(1, 1), prog.column(2, -1)
    t0 = prog.calc("mul", a, b)
    t1 = prog.calc("sub", t0, c_prev)
    t2 = prog.calc("square", t1)
    t3 = prog.calc("add", t2, prog.constant(fe_from_int(7)))
    t4 = prog.calc("double", t3)
    t5 = prog.calc("negate", t4)
    t6 = prog.horner(prog.previous(), [t5, t1], prog.challenge(1))
    code, consts, rots = prog.arrays()
    prev = rand_fr(rng, ne)
    got = ob.eval_program(code, prog.n_intermediates, consts, rots, cols, chal, k, ek, previous=prev)
    C = [[fe_to_int(x) for x in col] for col in cols]
    y = fe_to_int(chal[1])
    scale = 1 << (ek - k)
    for r in range(ne):
        av, bv, cv = C[0][r], C[1][(r + scale) % ne], C[2][(r - scale) % ne]
        v1 = (av * bv - cv) % R
        v5 = (-(2 * (v1 * v1 + 7))) % R
        want = ((fe_to_int(prev[r]) * y + v5) * y + v1) % R
        assert fe_to_int(got[r]) == want


def test_gen_bases_deterministic_on_curve():
    b = ob.gen_bases(SEED, 300)
    assert all(ob.g1_on_curve(p) for p in b)
    assert (ob.gen_bases(SEED, 100, first=200) == b[200:]).all()


def test_chacha20_rfc8439_vector_and_sampler():
    """the ChaCha20 block function behind ezkl_hip_chacha20_fr_dev, pinned on RFC 8439 section 2.3.2 (key 00..1f,
    counter 1, nonce 00:00:00:09:00:00:00:4a:00:00:00:00 = our counter_hi / stream words), and the rejection sampler"""
    blk = ob.chacha20_block(bytes(range(32)), 0x0900000000000001, 0x000000004a000000)
    want = "e4e7f110 15593bd1 1fdd0f50 c47120a3 c7f4d1c7 0368c033 9aaa2204 4e6cd4c3 466482d2 09aa9f07 05d7c214 a2028bd9 d19c12b5 b94e16de e883d0cb 4e3c50a2"
    assert " ".join("%08x" % x for x in blk) == want
    key = bytes(range(100, 132))
    a = ob.chacha20_fr(key, 3, 2000)
    vals = [fe_to_int_raw(r) for r in a]
    assert all(v < R for v in vals) and len(set(vals)) == 2000
    assert 0.28 < sum(v >> 253 for v in vals) / 2000 < 0.40            # uniform on [0, r): (r - 2^253) / r = 0.34 of the mass above 2^253
    assert (ob.chacha20_fr(key, 3, 50, first=1000) == a[1000:1050]).all()        # element i depends only on (key, stream, i)
    assert not (ob.chacha20_fr(key, 4, 10) == a[:10]).all(axis=1).any()
    # element 0 is the first candidate < r of blocks 0..15
    cands = []
    for b in range(16):
        w = ob.chacha20_block(key, b, 3)
        for h in range(2):
            v = int.from_bytes(w[8 * h:8 * h + 8].tobytes(), "little") & ((1 << 254) - 1)
            cands.append(v)
    assert vals[0] == next(v for v in cands if v < R)


def fe_to_int_raw(a):
    return int.from_bytes(np.ascontiguousarray(a, np.uint64).tobytes(), "little")


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_sweep_schedule_keeps_the_value(seed):
    """the library re-orders a gate program before it runs it (every term right before the Horner step that consumes it: short live
    ranges).  Host-only check of that pass: the re-ordered program is a permutation of the original one, its result is unchanged
    (C oracle on random columns), and the number of values alive at once falls"""
    from ezkl_amd import backend as B
    from test_gpu_misc import _random_program
    rng = np.random.default_rng(seed)
    k, ek, ncols = 6, 8, 10
    ne = 1 << ek
    prog = _random_program(B, rng, k, ek, ncols, 240)
    # a halo2-style tail: many terms first, ONE Horner chain over all of them at the end
    code, consts, rots = prog.arrays()
    sched = prog.scheduled_code(ncols)
    assert sched.shape == code.shape and sorted(map(tuple, sched.tolist())) == sorted(map(tuple, code.tolist()))
    assert (sched[-1] == code[-1]).all()
    cols = [rand_fr(rng, ne) for _ in range(ncols)]
    chal, prev = rand_fr(rng, 3), rand_fr(rng, ne)
    want = ob.eval_program(code, prog.n_intermediates, consts, rots, cols, chal, k, ek, previous=prev)
    got = ob.eval_program(sched, prog.n_intermediates, consts, rots, cols, chal, k, ek, previous=prev)
    assert (got == want).all()
    def max_live(c):
        last = {}
        for i, ins in enumerate(c.tolist()):
            for q in (0, 1):
                if ins[2 + 3 * q] == B.INTERMEDIATE:
                    last[ins[3 + 3 * q]] = i
            last.setdefault(ins[1], i)
        first = {}
        for i, ins in enumerate(c.tolist()):
            first.setdefault(ins[1], i)
        return max(sum(1 for t in first if first[t] <= i <= last[t]) for i in range(len(c)))
    # on a halo2-shaped program -- 60 independent gate terms, then one Horner chain over all of them -- the live set collapses
    gp = B.GraphProgram(k, ek)
    terms = []
    for t in range(60):
        a, b_, c_, s_ = (gp.column(int(rng.integers(0, ncols)), int(rng.integers(-1, 2))) for _ in range(4))
        terms.append(gp.calc("mul", s_, gp.calc("sub", c_, gp.calc("mul", a, b_))))
    gp.horner(gp.previous(), terms, gp.challenge(0))
    gcode, gconsts, grots = gp.arrays()
    gsched = gp.scheduled_code(ncols)
    assert max_live(gcode) >= 60 and max_live(gsched) <= 4
    if gconsts.shape[0] == 0:
        gconsts = np.zeros((1, 4), np.uint64)
    assert (ob.eval_program(gsched, gp.n_intermediates, gconsts, grots, cols, chal, k, ek, previous=prev) ==
            ob.eval_program(gcode, gp.n_intermediates, gconsts, grots, cols, chal, k, ek, previous=prev)).all()


def test_g1_to_lagrange_reproduces_the_reference_srs(golden_srs):
    """the oracle's naive ParamsKZG::downsize / g_to_lagrange (oracle.binding.g1_to_lagrange: one MSM of Lagrange-polynomial
    coefficients per point) pinned on the reference's k = 6 SRS file: from `g` it must rebuild the file's own `g_lagrange` section,
    and the k' = 4 Lagrange basis must commit like the coefficient basis (commit_lagrange(v) == commit(iNTT v))"""
    from oracle import binding as ob
    g, gl = golden_srs["g"], golden_srs["g_lagrange"]
    assert (ob.g1_to_lagrange(g, 6) == gl).all()
    gl4 = ob.g1_to_lagrange(g, 4)
    rng = np.random.default_rng(4)
    v = rand_fr(rng, 16)
    assert (ob.msm(v, gl4) == ob.msm(ob.lagrange_to_coeff(v, 4), g[:16])).all()
    acc = gl4[0]
    for i in range(1, 16):
        acc = ob.g1_add(acc, gl4[i])
    assert (acc == g[0]).all()
