"""GPU parity: vec-ops, batch inversion and the quotient-sweep program interpreter vs the oracle."""
import numpy as np
import pytest
from conftest import R, fe_from_int, rand_fr
from oracle import binding as ob

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 1000, 1 << 16])
def test_vec_ops(hip, n):
    from ezkl_amd import backend as B
    rng = np.random.default_rng(n)
    a, b = rand_fr(rng, n), rand_fr(rng, n)
    da, db, do = B.DeviceBuffer.from_numpy(a), B.DeviceBuffer.from_numpy(b), B.DeviceBuffer(n * 32)
    for op in ("add", "sub", "mul"):
        B.vec_op(op, da.ptr, db.ptr, do.ptr, n)
        assert (do.to_numpy(shape=(n, 4)) == ob.vec_op(op, a, b)).all()


@pytest.mark.parametrize("n", [1, 5, 64, 1000, 4097, 1 << 15, (1 << 20) + 3, 5 * (1 << 20) + 1])
def test_batch_invert(hip, n):
    import ctypes as C
    from ezkl_amd import backend as B, lib as L
    rng = np.random.default_rng(n)
    a = rand_fr(rng, n)
    a[::7] = 0
    d = B.DeviceBuffer.from_numpy(a)
    L.check(L.load().ezkl_hip_batch_invert_dev(C.c_void_p(d.ptr), C.c_size_t(n), C.c_void_p(None)), "batch_invert")
    assert (d.to_numpy(shape=(n, 4)) == ob.batch_invert(a)).all()


@pytest.mark.parametrize("n", [1, 2, 255, 2048, 2049, 5000, 1 << 16, (1 << 22) + 3])
@pytest.mark.parametrize("op", ["add", "mul"])
def test_prefix_scan(hip, n, op):
    """grand sum / grand product building block (mv-lookup commit_grand_sum, permutation commit)"""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(n)
    a = rand_fr(rng, n)
    d = B.DeviceBuffer.from_numpy(a)
    o = B.DeviceBuffer(n * 32)
    for excl in (False, True):
        B.prefix_scan(op, d.ptr, o.ptr, n, exclusive=excl)
        assert (o.to_numpy(shape=(n, 4)) == ob.prefix_scan(a, op, exclusive=excl)).all()
    B.prefix_scan(op, d.ptr, d.ptr, n)                         # in place
    assert (d.to_numpy(shape=(n, 4)) == ob.prefix_scan(a, op)).all()


def _gate_program(B, k, ek):
    """a small ezkl-like gate set: sel*(out - a*b) (chip.rs:344-393), a rotation -1 accumulator
    (chip.rs:352-425), folded with y by Horner exactly as evaluate_h does"""
    prog = B.GraphProgram(k, ek)
    sel, a, b, out = prog.column(0), prog.column(1), prog.column(2), prog.column(3)
    prev_out = prog.column(3, -1)
    nxt = prog.column(1, 1)
    g1 = prog.calc("mul", sel, prog.calc("sub", out, prog.calc("mul", a, b)))
    acc = prog.calc("add", prev_out, prog.calc("mul", a, b))
    g2 = prog.calc("mul", sel, prog.calc("sub", out, acc))
    g3 = prog.calc("mul", prog.calc("square", nxt), prog.constant(fe_from_int(R - 2)))
    g4 = prog.calc("negate", prog.calc("double", prog.calc("add", g3, prog.challenge(0))))
    prog.horner(prog.previous(), [g1, g2, g3, g4], prog.challenge(1))
    return prog


@pytest.mark.parametrize("mode", ["jit", "interp"])
@pytest.mark.parametrize("k,ek", [(3, 5), (8, 10), (12, 15)])
def test_eval_h_program(hip, k, ek, mode, monkeypatch):
    from ezkl_amd import backend as B
    monkeypatch.setenv("EZKL_EVALH_MODE", mode)
    rng = np.random.default_rng(k)
    ne = 1 << ek
    cols = [rand_fr(rng, ne) for _ in range(4)]
    chal = rand_fr(rng, 2)
    prev = rand_fr(rng, ne)
    prog = _gate_program(B, k, ek)
    code, consts, rots = prog.arrays()
    want = ob.eval_program(code, prog.n_intermediates, consts, rots, cols, chal, k, ek, previous=prev)
    dcols = [B.DeviceBuffer.from_numpy(c) for c in cols]
    dout = B.DeviceBuffer.from_numpy(prev)
    prog.evaluate_h([d.ptr for d in dcols], chal, dout.ptr)
    assert (dout.to_numpy(shape=(ne, 4)) == want).all()


def _random_program(B, rng, k, ek, ncols, ninstr):
    """a random straight-line DAG with long- and short-lived values: exercises slot reuse and spilling"""
    prog = B.GraphProgram(k, ek)
    vals = []
    ops2, ops1 = ["add", "sub", "mul"], ["square", "double", "negate"]
    for _ in range(ninstr):
        def src():
            r = rng.random()
            if vals and r < 0.55:
                # bias to recent values, but keep some old ones alive
                return vals[-1 - int(rng.integers(0, min(len(vals), 6)))] if rng.random() < 0.7 else vals[int(rng.integers(0, len(vals)))]
            if r < 0.85:
                return prog.column(int(rng.integers(0, ncols)), int(rng.integers(-2, 3)))
            if r < 0.93:
                return prog.challenge(int(rng.integers(0, 3)))
            return prog.constant(fe_from_int(int(rng.integers(1, 1 << 30))))
        if rng.random() < 0.8:
            vals.append(prog.calc(ops2[int(rng.integers(0, 3))], src(), src()))
        else:
            vals.append(prog.calc(ops1[int(rng.integers(0, 3))], src()))
    prog.horner(prog.previous(), vals[-40:], prog.challenge(0))
    return prog


@pytest.mark.parametrize("mode", ["jit", "interp"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_eval_h_random_dag(hip, seed, mode, monkeypatch):
    from ezkl_amd import backend as B
    monkeypatch.setenv("EZKL_EVALH_MODE", mode)
    rng = np.random.default_rng(seed)
    k, ek, ncols = 9, 11, 12
    ne = 1 << ek
    cols = [rand_fr(rng, ne) for _ in range(ncols)]
    chal = rand_fr(rng, 3)
    prev = rand_fr(rng, ne)
    prog = _random_program(B, rng, k, ek, ncols, 300)
    code, consts, rots = prog.arrays()
    want = ob.eval_program(code, prog.n_intermediates, consts, rots, cols, chal, k, ek, previous=prev)
    dcols = [B.DeviceBuffer.from_numpy(c) for c in cols]
    dout = B.DeviceBuffer.from_numpy(prev)
    prog.evaluate_h([d.ptr for d in dcols], chal, dout.ptr)
    assert (dout.to_numpy(shape=(ne, 4)) == want).all()


def test_eval_h_rejects_bad_program(hip):
    from ezkl_amd import backend as B
    prog = B.GraphProgram(3, 5)
    prog.calc("add", prog.column(5), prog.column(0))      # column index out of range
    d = B.DeviceBuffer(32 << 5)
    with pytest.raises(hip.EzklHipError):
        prog.evaluate_h([d.ptr], np.zeros((1, 4), np.uint64), d.ptr)


def test_permutation_grand_product_and_lookup_sum(hip):
    """A13: z(X) of the permutation argument and phi(X) of mv-lookup, composed from eval/batch-invert/scan
    kernels, against a direct big-int evaluation of the defining recurrences"""
    from ezkl_amd import backend as B
    from conftest import fe_to_int
    from oracle import pyref as pr
    rng = np.random.default_rng(9)
    k, m = 6, 3
    n = 1 << k
    vals = [rand_fr(rng, n) for _ in range(m)]
    sigmas = [rand_fr(rng, n) for _ in range(m)]
    beta, gamma = rand_fr(rng, 1)[0], rand_fr(rng, 1)[0]
    dv = [B.DeviceBuffer.from_numpy(v) for v in vals]
    ds = [B.DeviceBuffer.from_numpy(v) for v in sigmas]
    z = B.permutation_grand_product(k, [d.ptr for d in dv], [d.ptr for d in ds], beta, gamma, first_column_index=2).to_numpy(shape=(n, 4))
    b, g, w = fe_to_int(beta), fe_to_int(gamma), pr.omega(k)
    V = [[fe_to_int(x) for x in v] for v in vals]
    S = [[fe_to_int(x) for x in v] for v in sigmas]
    acc = 1
    for i in range(n):
        assert fe_to_int(z[i]) == acc
        num = den = 1
        for j in range(m):
            num = num * (V[j][i] + b * pow(pr.DELTA, 2 + j, R) * pow(w, i, R) + g) % R
            den = den * (V[j][i] + b * S[j][i] + g) % R
        acc = acc * num * pow(den, -1, R) % R
    # mv-lookup running sum
    f = [rand_fr(rng, n) for _ in range(2)]
    t, mm = rand_fr(rng, n), rand_fr(rng, n)
    df = [B.DeviceBuffer.from_numpy(v) for v in f]
    dt, dm = B.DeviceBuffer.from_numpy(t), B.DeviceBuffer.from_numpy(mm)
    phi = B.lookup_grand_sum(k, [d.ptr for d in df], dt.ptr, dm.ptr, beta).to_numpy(shape=(n, 4))
    F = [[fe_to_int(x) for x in v] for v in f]
    T, M = [fe_to_int(x) for x in t], [fe_to_int(x) for x in mm]
    acc = 0
    for i in range(n):
        assert fe_to_int(phi[i]) == acc
        acc = (acc + sum(pow(F[j][i] + b, -1, R) for j in range(2)) - M[i] * pow(T[i] + b, -1, R)) % R


def test_error_paths_are_status_codes(hip):
    """bad arguments come back as EZKL_ERR_INVALID (-3): nothing throws or unwinds across the C ABI"""
    import ctypes as C
    from ezkl_amd import backend as B, lib as L
    lib = L.load()
    pts = ob.gen_bases(1, 16)
    bases = B.Bases(pts)
    out = np.zeros(8, np.uint64)
    s = rand_fr(np.random.default_rng(0), 32)
    vp = C.c_void_p
    assert lib.ezkl_hip_msm_g1(bases.h, s.ctypes.data_as(vp), C.c_size_t(32), out.ctypes.data_as(vp)) == -3      # more scalars than bases
    assert lib.ezkl_hip_msm_g1(None, s.ctypes.data_as(vp), C.c_size_t(4), out.ctypes.data_as(vp)) == -3
    assert lib.ezkl_hip_msm_g1(bases.h, None, C.c_size_t(4), out.ctypes.data_as(vp)) == -3
    assert lib.ezkl_hip_msm_g1(bases.h, s.ctypes.data_as(vp), C.c_size_t(0), out.ctypes.data_as(vp)) == 0 and not out.any()   # empty input -> identity
    w = ob.omega(4)
    assert lib.ezkl_hip_ntt(s.ctypes.data_as(vp), C.c_uint32(29), w.ctypes.data_as(vp), 0) == -3
    assert lib.ezkl_hip_ntt(None, C.c_uint32(4), w.ctypes.data_as(vp), 0) == -3
    d = B.DeviceBuffer(32 * 16)
    assert lib.ezkl_hip_coset_ntt_dev(vp(d.ptr), vp(d.ptr), C.c_size_t(1), C.c_size_t(16), C.c_size_t(16), C.c_uint32(5), C.c_uint32(4), 0, vp(None)) == -3
    assert lib.ezkl_hip_vec_op_dev(7, vp(d.ptr), vp(d.ptr), vp(d.ptr), C.c_size_t(16), vp(None)) == -3
    assert lib.ezkl_hip_prefix_scan_dev(1, 0, vp(d.ptr), vp(d.ptr), C.c_size_t(16), vp(None)) == -3              # sub is not a scan op
    assert b"invalid" in lib.ezkl_hip_strerror(-3)
    bases.free()


def test_concurrent_host_threads(hip):
    """halo2 calls the backend from rayon worker threads: concurrent MSM / NTT / vec calls must serialise correctly"""
    import threading
    from ezkl_amd import backend as B
    rng = np.random.default_rng(4)
    n, k = 1 << 12, 12
    pts = ob.gen_bases(9, n)
    bases = B.Bases(pts)
    w = ob.omega(k)
    jobs = [(rand_fr(rng, n), rand_fr(rng, n)) for _ in range(6)]
    want = [(ob.msm(a, pts), ob.fft(b, k, w)) for a, b in jobs]
    got = [None] * len(jobs)

    def work(i):
        a, b = jobs[i]
        got[i] = (B.msm_g1(bases, a), hip.ntt(b, k, w))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for g, wv in zip(got, want):
        assert (g[0] == wv[0]).all() and (g[1] == wv[1]).all()
    bases.free()


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 1000, 8192, 8193, 1 << 16, (1 << 20) + 5])
def test_eval_polynomial(hip, n):
    from ezkl_amd import backend as B
    rng = np.random.default_rng(n)
    c, x = rand_fr(rng, n), rand_fr(rng, 1)[0]
    d = B.DeviceBuffer.from_numpy(c)
    assert (B.eval_polynomial(d.ptr, n, x) == ob.eval_poly(c, x)).all()


@pytest.mark.parametrize("k,ninputs", [(4, 1), (10, 3), (16, 2)])
def test_lookup_multiplicity(hip, k, ninputs):
    """m(X) of mv-lookup against a dict-based restatement of mv_lookup::prover::prepare (first table row wins)"""
    from ezkl_amd import backend as B
    from conftest import fe_to_int
    rng = np.random.default_rng(k)
    n = 1 << k
    usable = n - 6
    distinct = max(2, n // 3)
    pool = rand_fr(rng, distinct)
    table = pool[rng.integers(0, distinct, size=n)]            # duplicates in the table on purpose
    inputs = [table[rng.integers(0, usable, size=n)] for _ in range(ninputs)]
    inputs[0][3] = rand_fr(rng, 1)[0]                          # one value that is not in the table
    dt = B.DeviceBuffer.from_numpy(table)
    di = [B.DeviceBuffer.from_numpy(x) for x in inputs]
    m_dev, missing = B.lookup_multiplicity([d.ptr for d in di], dt.ptr, n, usable)
    first = {}
    for i in range(usable):
        first.setdefault(table[i].tobytes(), i)
    want = np.zeros(n, np.int64)
    miss = 0
    for x in inputs:
        for r in range(usable):
            idx = first.get(x[r].tobytes())
            if idx is None:
                miss += 1
            else:
                want[idx] += 1
    got = m_dev.to_numpy(shape=(n, 4))
    assert missing == miss
    assert [fe_to_int(g) for g in got] == [int(w) for w in want]


@pytest.mark.parametrize("k", [5, 12, 17])
def test_lookup_multiplicity_batch(hip, k):
    """every lookup argument of a proof in one call: the columns and the missing count of one call per argument (arguments with 1, 3, 2
    and NO input columns, tables with runs of equal rows)"""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(100 + k)
    n = 1 << k
    usable = n - 6
    tables, inputs = [], []
    for l, nin in enumerate((1, 3, 2, 0)):
        distinct = max(2, n // (2 + l))
        pool = rand_fr(rng, distinct)
        t = np.sort(rng.integers(0, distinct, size=n)) if l == 1 else rng.integers(0, distinct, size=n)        # l = 1: long runs of equal rows
        tables.append(pool[t])
        inputs.append([tables[-1][rng.integers(0, usable, size=n)] for _ in range(nin)])
    inputs[1][2][5] = rand_fr(rng, 1)[0]                       # two values that are in no table
    inputs[2][0][usable - 1] = rand_fr(rng, 1)[0]
    dt = [B.DeviceBuffer.from_numpy(t) for t in tables]
    di = [[B.DeviceBuffer.from_numpy(x) for x in ins] for ins in inputs]
    outs, missing = B.lookup_multiplicity_batch([[d.ptr for d in ins] for ins in di], [d.ptr for d in dt], n, usable)
    total = 0
    for l in range(4):
        m_one, miss_one = B.lookup_multiplicity([d.ptr for d in di[l]], dt[l].ptr, n, usable)
        total += miss_one
        assert (outs[l].to_numpy(shape=(n, 4)) == m_one.to_numpy(shape=(n, 4))).all(), l
    assert missing == total == 2


@pytest.mark.parametrize("n,m", [(1, 1), (33, 3), (1000, 17), (1 << 16, 5), ((1 << 18) + 3, 2)])
def test_eval_polynomial_batch(hip, n, m):
    """the batched form of create_proof's step 10: same values as one call per (polynomial, point)"""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(n + m)
    cols = [rand_fr(rng, n) for _ in range(m)]
    xs = rand_fr(rng, m)
    devs = [B.DeviceBuffer.from_numpy(c) for c in cols]
    got = B.eval_polynomial_batch([d.ptr for d in devs], n, xs)
    for j in range(m):
        assert (got[j] == ob.eval_poly(cols[j], xs[j])).all()
    assert B.eval_polynomial_batch([], n, np.zeros((0, 4), np.uint64)).shape == (0, 4)


@pytest.mark.parametrize("n,m,acc", [(1, 1, False), (1000, 3, True), (4097, 16, False), (1 << 16, 17, True), (1 << 18, 40, False), (513, 0, True)])
def test_lincomb(hip, n, m, acc):
    """out (+)= sum_j c_j * in_j, fused (SHPLONK combinations) == the oracle's scale / add composition"""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(n * 31 + m)
    cols = [rand_fr(rng, n) for _ in range(m)]
    cf = rand_fr(rng, max(m, 1))[:m]
    start = rand_fr(rng, n)
    want = start.copy() if acc else np.zeros((n, 4), np.uint64)
    for j in range(m):
        want = ob.vec_add(want, ob.vec_scale(cols[j], cf[j]))
    devs = [B.DeviceBuffer.from_numpy(c) for c in cols]
    out = B.DeviceBuffer.from_numpy(start)
    B.lincomb([d.ptr for d in devs], cf, out.ptr, n, accumulate=acc)
    assert (out.to_numpy(shape=(n, 4)) == want).all()


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 255, 2047, 2048, 2049, 5000, 1 << 16, (1 << 20) + 3, 1 << 22, (1 << 22) + 2049])
def test_kate_division(hip, n):
    """a(X) / (X - z) without the remainder (halo2's kate_division) == the oracle's synthetic division: sizes around the thread
    (8), chunk (2048) and second-level (2048^2) boundaries, in place and out of place, special points; and the identity
    q(X) (X - z) + a(z) = a(X) through independent evaluations"""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(n)
    a = rand_fr(rng, n)
    d = B.DeviceBuffer.from_numpy(a)
    o = B.DeviceBuffer(32 * n)
    for z in (rand_fr(rng, 1)[0], fe_from_int(0), fe_from_int(1), fe_from_int(R - 1)):
        B.kate_division(d.ptr, z, o.ptr, n)
        want = ob.kate_div(a, z)
        assert (o.to_numpy(shape=(n, 4)) == want).all()
        assert (d.to_numpy(shape=(n, 4)) == a).all()                       # input untouched
    z, x = rand_fr(rng, 1)[0], rand_fr(rng, 1)[0]
    B.kate_division(d.ptr, z, d.ptr, n)                                    # in place
    q = d.to_numpy(shape=(n, 4))
    assert (q == ob.kate_div(a, z)).all()
    one = lambda v: np.asarray(v, np.uint64).reshape(1, 4)
    xz = ob.vec_op("sub", one(x), one(z))[0]
    lhs = ob.vec_op("add", one(ob.fr_mul(ob.eval_poly(q, x), xz)), one(ob.eval_poly(a, z)))[0]
    assert (lhs == ob.eval_poly(a, x)).all()


@pytest.mark.parametrize("n,first", [(1, 0), (1000, 0), (5000, 123456), (1 << 18, 1 << 40)])
def test_chacha20_field_sampler(hip, n, first):
    """device keystream expansion == the C restatement (itself pinned on the RFC 8439 vector), element for element"""
    from ezkl_amd import backend as B
    key = bytes((7 * i + n) & 0xff for i in range(32))
    out = B.DeviceBuffer(n * 32)
    B.chacha20_fr(key, 0xabcdef0123 + n, out.ptr, n, first=first)
    got = out.to_numpy(shape=(n, 4))
    assert (got == ob.chacha20_fr(key, 0xabcdef0123 + n, n, first=first)).all()


def test_radix29_montgomery_product_self_check(hip):
    """the generated carry-free radix-2^29 product (montmul29_gen.hpp, the MSM's field multiplication) against its
    portable restatement on 1M operand pairs, on the device"""
    from ezkl_amd import backend as B
    assert B.ubench("modmul29_check") == 0.0


def test_pinned_host_memory(hip):
    from ezkl_amd import backend as B
    rng = np.random.default_rng(3)
    pa = B.PinnedArray((1 << 12, 4))
    pa.array[:] = rand_fr(rng, 1 << 12)
    d = B.DeviceBuffer.from_numpy(pa.array)
    assert (d.to_numpy(shape=(1 << 12, 4)) == pa.array).all()
    pa.free()


@pytest.mark.gpu
@pytest.mark.parametrize("k,m", [(4, 1), (9, 5), (14, 3)])
def test_permutation_sigma_column(hip, k, m):
    """sigma_c[r] = delta^c' omega^r' for the cycle successor (c', r') of cell (c, r) (halo2 permutation keygen), from the successor map:
    against big-integer arithmetic, with fixed points, successors in other columns and one successor outside the columns (-> 0)"""
    from ezkl_amd import backend as B
    from conftest import fe_to_int
    from oracle import pyref as pr
    n = 1 << k
    rng = np.random.default_rng(100 * k + m)
    w = pr.omega(k)
    wcol = np.stack([fe_from_int(pow(w, r, R)) for r in range(n)])
    dpow = np.stack([fe_from_int(pow(pr.DELTA, c, R)) for c in range(m)])
    d_w, d_d = B.DeviceBuffer.from_numpy(wcol), B.DeviceBuffer.from_numpy(dpow)
    for c in range(m):
        nxt = (c * n + np.arange(n)).astype(np.uint32)                 # fixed points ...
        moved = rng.random(n) < 0.4
        nxt[moved] = rng.integers(0, m * n, size=int(moved.sum()), dtype=np.uint32)   # ... and successors anywhere in the m columns
        nxt[1] = m * n + 3                                             # outside: the kernel must not read there
        d_n = B.DeviceBuffer.from_numpy(nxt)
        out = B.DeviceBuffer(32 * n)
        B.permutation_sigma_dev(d_n.ptr, d_w.ptr, d_d.ptr, m, k, out.ptr)
        got = out.to_numpy(shape=(n, 4))
        for r in list(range(8)) + [int(x) for x in rng.integers(0, n, 40)]:
            t = int(nxt[r])
            want = 0 if t >= m * n else pow(pr.DELTA, t // n, R) * pow(w, t % n, R) % R
            assert fe_to_int(got[r]) == want
    with pytest.raises(Exception):
        B.permutation_sigma_dev(d_n.ptr, d_w.ptr, d_d.ptr, 0, k, out.ptr)


def test_quad_cooperative_addition_on_the_device(hip):
    """ezkl_hip_ubench("coopcheck") (csrc/ubench.hip): the quad-cooperative XYZZ addition of the MSM's reduce trees (curve29.hpp:
    g1x29_add_quad) against the plain addition on the device, every intermediate of one quad replayed on the host -- DPP broadcasts,
    one addition, the butterfly sums of 64 / 32 / 8 lanes and of a 256-thread workgroup, with and without identity operands.
    0 = everything agreed; the bits name what did not (details on stderr)."""
    from ezkl_amd import backend as B
    assert B.ubench("coopcheck") == 0.0
