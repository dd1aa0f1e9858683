"""GPU parity: HIP fixed-base Pippenger MSM (through the C ABI) vs the oracle and the golden SRS."""
import numpy as np
import pytest
from conftest import R, Q, SEED, fe_from_int, fe_to_int, rand_fr, witness_like
from oracle import binding as ob

pytestmark = pytest.mark.gpu


def test_gen_bases_matches_oracle(hip):
    from ezkl_amd import backend as B
    b = B.Bases.generate(SEED, 5000)
    got = b.download()
    assert (got == ob.gen_bases(SEED, 5000)).all()
    b.free()


@pytest.mark.parametrize("n", [1, 2, 3, 7, 64, 100, 1000, 4096, 1 << 14, 1 << 16])
def test_uniform_scalars_match_oracle(hip, n):
    from ezkl_amd import backend as B
    rng = np.random.default_rng(n)
    pts = ob.gen_bases(SEED, n)
    bases = B.Bases(pts)
    s = rand_fr(rng, n)
    assert (B.msm_g1(bases, s) == ob.msm(s, pts)).all()
    bases.free()


def test_golden_srs_commitments(hip, golden_srs, golden_pk):
    """reference fixture (k=6 SRS): commit_lagrange(v) == commit(iNTT(v)); sum(g_lagrange) == g[0]"""
    params = hip.ParamsKZG.read(golden_srs["buf"])
    assert params.k == 6
    ones = np.tile(fe_from_int(1), (64, 1))
    assert (params.commit_lagrange(ones) == golden_srs["g"][0]).all()
    for v, p in zip(golden_pk["fixed_values"], golden_pk["fixed_polys"]):
        a = params.commit_lagrange(v)
        assert (a == params.commit(p)).all()
        assert (a == ob.msm(v, golden_srs["g_lagrange"])).all()
    params.free()


def test_edge_cases(hip):
    from ezkl_amd import backend as B
    rng = np.random.default_rng(11)
    n = 512
    pts = ob.gen_bases(SEED, n)
    pts[30] = 0                                   # identity base
    pts[21] = pts[20]                             # duplicate point: P + P inside a bucket
    pts[41] = pts[40]
    pts[41, 4:] = np.frombuffer(((Q - int.from_bytes(pts[40, 4:].tobytes(), "little")) % Q).to_bytes(32, "little"), np.uint64)
    bases = B.Bases(pts)
    s = rand_fr(rng, n)
    s[:8] = 0
    s[8] = fe_from_int(R - 1); s[9] = fe_from_int(1); s[10] = fe_from_int((R - 1) // 2); s[11] = fe_from_int((R + 1) // 2)
    for j, kbits in enumerate((200, 253, 64, 20, 19, 21, 40)):      # runs of one-bits: a digit of 2^c - 1 plus a carry
        s[50 + j] = fe_from_int((1 << kbits) - 1)                     # makes raw == 2^c (digit 0, carry 1)
        s[60 + j] = fe_from_int(R - ((1 << kbits) - 1))
    s[20] = s[21] = fe_from_int(12345)            # same scalar, same point -> doubling path
    s[40] = s[41] = fe_from_int(777)              # P and -P with the same scalar -> inverse path
    assert (B.msm_g1(bases, s) == ob.msm(s, pts)).all()
    assert (B.msm_g1(bases, np.zeros((n, 4), np.uint64)) == 0).all()             # all-zero scalars -> identity
    same = np.tile(fe_from_int(3), (n, 1))                                        # every point in ONE bucket (heavy path)
    assert (B.msm_g1(bases, same) == ob.msm(same, pts)).all()
    # ONE point repeated with ONE scalar: every lane partial of the bucket is the same multiple of P, so the fixup /
    # reduce trees add EQUAL points (the doubling branch of the general addition) at every level
    n2 = 4096
    rep = np.tile(pts[3], (n2, 1))
    breps = B.Bases(rep)
    for val in (5, R - 5):
        sc = np.tile(fe_from_int(val), (n2, 1))
        assert (B.msm_g1(breps, sc) == ob.msm(sc, rep)).all()
    breps.free()
    short = s[:100]                                                               # ragged: fewer scalars than bases
    assert (B.msm_g1(bases, short) == ob.msm(short, pts[:100])).all()
    bases.free()


def test_witness_like_scalars(hip):
    """skewed distribution of real witnesses (src/fieldutils.rs:9-17): small signed values, zeros"""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(12)
    n = 1 << 15
    pts = ob.gen_bases(SEED, n)
    bases = B.Bases(pts)
    s = witness_like(rng, n)
    assert (B.msm_g1(bases, s) == ob.msm(s, pts)).all()
    bases.free()


def test_full_size_2_20(hip):
    """BASELINE configs[1]: 2^20 points.  Oracle check on the full size (a few CPU-seconds) + linearity."""
    from ezkl_amd import backend as B
    n = 1 << 20
    rng = np.random.default_rng(20)
    bases = B.Bases.generate(SEED, n)
    pts = bases.download()
    assert all(ob.g1_on_curve(p) for p in pts[:: 1 << 12])
    s1, s2 = rand_fr(rng, n), rand_fr(rng, n)
    a = B.msm_g1(bases, s1)
    assert (a == ob.msm(s1, pts)).all()
    b = B.msm_g1(bases, s2)
    # linearity: msm(s1 + s2) == msm(s1) + msm(s2)
    b1, b2 = B.DeviceBuffer.from_numpy(s1), B.DeviceBuffer.from_numpy(s2)
    B.vec_op("add", b1.ptr, b2.ptr, b1.ptr, n)
    c = B.msm_g1_dev(bases, b1.ptr, n)
    assert (c == B.g1_add_affine(a, b)).all()
    # sharded evaluation (the multi-GPU decomposition) gives the same point
    half = n // 2
    p0 = B.msm_g1_dev(bases, b1.ptr, half, offset=0)
    b1hi = B.DeviceBuffer.from_numpy(b1.to_numpy(shape=(n, 4))[half:])
    p1 = B.msm_g1_dev(bases, b1hi.ptr, half, offset=half)
    assert (B.g1_add_affine(p0, p1) == c).all()
    bases.free()


def test_full_size_2_20_witness_like(hip):
    """BASELINE.md §3 distribution (W) at the full size of configs[1]: what the advice columns of a proof look like (one or two non-zero
    digits per scalar, 20 % zeros, every carry of the signed recoding in bucket 0) -- the device-side lane length, the bucket-0 partition
    and the boundary tree all take their skewed branches here.  Oracle check on all 2^20 points + the same column in a fused batch."""
    from ezkl_amd import backend as B
    n = 1 << 20
    rng = np.random.default_rng(21)
    bases = B.Bases.generate(SEED, n)
    pts = bases.download()
    s = witness_like(rng, n)
    want = ob.msm(s, pts)
    assert (B.msm_g1(bases, s) == want).all()
    d = B.DeviceBuffer.from_numpy(s)
    u = B.DeviceBuffer.from_numpy(rand_fr(rng, n))
    got = B.msm_g1_batch_dev(bases, [d.ptr, u.ptr, d.ptr], n)
    assert (got[0] == want).all() and (got[2] == want).all()
    assert (got[1] == B.msm_g1_dev(bases, u.ptr, n)).all()
    bases.free()


@pytest.mark.parametrize("n,batch", [(1000, 7), (1 << 14, 5), (1 << 18, 4)])
def test_batch_pipelined_matches_single(hip, n, batch):
    """one commit phase: `batch` columns against the same bases, pipelined over streams"""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(n + batch)
    pts = ob.gen_bases(SEED, n)
    bases = B.Bases(pts)
    cols = [rand_fr(rng, n) for _ in range(batch)]
    cols[1] = witness_like(rng, n) if n <= (1 << 14) else cols[1]
    dcols = [B.DeviceBuffer.from_numpy(c) for c in cols]
    got = B.msm_g1_batch_dev(bases, [d.ptr for d in dcols], n)
    for j in range(batch):
        assert (got[j] == ob.msm(cols[j], pts)).all()
    # host-pointer batch entry point (what commit of a Vec<Polynomial> binds)
    import ctypes as C
    from ezkl_amd import lib as L
    out = np.zeros((batch, 8), np.uint64)
    arr = (C.c_void_p * batch)(*[c.ctypes.data for c in cols])
    L.check(L.load().ezkl_hip_msm_g1_batch(bases.h, arr, C.c_size_t(batch), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)), "batch")
    assert (out == got).all()
    bases.free()


def test_polycommit_chip_commit(hip, golden_srs):
    """mirror of PolyCommitChip::commit (src/circuit/modules/polycommit.rs:46-81) on the reference's k=6 SRS:
    message split into Lagrange columns, unusable rows = Blind::default(), one commit_lagrange per column"""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(77)
    params = hip.ParamsKZG.read(golden_srs["buf"])
    u = 6                                            # blinding 5 + 1 (src/tensor/var.rs:57-60)
    msg = rand_fr(rng, 100)                          # 100 values over 58 usable rows -> 2 columns
    got = B.polycommit_commit(msg, u, params)
    assert got.shape == (100 // 58 + 1, 8)
    one = fe_from_int(1)
    for j in range(got.shape[0]):
        col = np.zeros((64, 4), np.uint64)
        col[58:] = one
        chunk = msg[j * 58:(j + 1) * 58]
        col[:chunk.shape[0]] = chunk
        assert (got[j] == ob.msm(col, golden_srs["g_lagrange"])).all()
    params.free()


def test_c5_size_2_22_properties(hip):
    """BASELINE configs[4] SRS size (k = 22): linearity and shard-and-fold at 4M points (no oracle at this size)"""
    from ezkl_amd import backend as B
    n = 1 << 22
    rng = np.random.default_rng(22)
    bases = B.Bases.generate(SEED + 1, n)
    s1, s2 = rand_fr(rng, n), witness_like(np.random.default_rng(5), 1 << 12)
    s2 = np.tile(s2, (n >> 12, 1))                   # skewed column: the same 4096 witness-like values repeated
    d1, d2 = B.DeviceBuffer.from_numpy(s1), B.DeviceBuffer.from_numpy(s2)
    a, b = B.msm_g1_dev(bases, d1.ptr, n), B.msm_g1_dev(bases, d2.ptr, n)
    B.vec_op("add", d1.ptr, d2.ptr, d1.ptr, n)
    c = B.msm_g1_dev(bases, d1.ptr, n)
    assert (c == B.g1_add_affine(a, b)).all()
    q = n // 4
    parts = []
    host = d1.to_numpy(shape=(n, 4))
    for r in range(4):
        dq = B.DeviceBuffer.from_numpy(host[r * q:(r + 1) * q])
        parts.append(B.msm_g1_dev(bases, dq.ptr, q, offset=r * q))
    from ezkl_amd import dist as D
    assert (D.fold_points(parts) == c).all()
    # spot check against the oracle on a prefix (the first 2^14 points with the rest zero)
    z = np.zeros((n, 4), np.uint64); z[: 1 << 14] = s1[: 1 << 14]
    pts = bases.download()[: 1 << 14]
    assert (B.msm_g1(bases, z) == ob.msm(s1[: 1 << 14], pts)).all()
    bases.free()


def test_gen_srs_structure(hip):
    """device-side SRS generation (gen_srs): g[i] = s^i G against the oracle's double-and-add, and the KZG
    self-consistency the reference fixture satisfies: sum(g_lagrange) == g[0], MSM(v, g_lagrange) == MSM(iNTT(v), g)"""
    from ezkl_amd import backend as B
    from conftest import fe_to_int
    k, s = 7, 0x1234567890abcdef1234567890abcdef
    n = 1 << k
    g, gl = B.gen_srs(k, s)
    G, GL = g.download(), gl.download()
    gen = G[0]
    for i in (0, 1, 2, 57, n - 1):
        assert (G[i] == ob.g1_mul(gen, fe_from_int(pow(s, i, R)))).all()
    ones = np.tile(fe_from_int(1), (n, 1))
    assert (B.msm_g1(gl, ones) == G[0]).all()
    v = rand_fr(np.random.default_rng(3), n)
    assert (B.msm_g1(gl, v) == B.msm_g1(g, ob.lagrange_to_coeff(v, k))).all()


@pytest.mark.parametrize("n", [5, 300, 1 << 12, 1 << 16, 1 << 18])
def test_small_values_and_bucket_zero(hip, n):
    """Witness-shaped scalars: bucket 0 (digits +-1: ones, minus ones, booleans, the carries of values just above half a window)
    has a sorting partition of its own, and the accumulate kernel sizes its lanes from the pairs that exist.  Parity for columns
    that are all / mostly bucket 0, for values straddling the signed-digit carry of every window width in use, and for mixes."""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(n)
    pts = ob.gen_bases(SEED + 3, n)
    bases = B.Bases(pts)
    one, minus_one = fe_from_int(1), fe_from_int(R - 1)
    cases = {
        "ones": np.tile(one, (n, 1)),
        "minus_ones": np.tile(minus_one, (n, 1)),
        "booleans": np.stack([fe_from_int(int(v)) for v in rng.integers(0, 2, n)]),
        "signs": np.stack([(one, minus_one, fe_from_int(0))[int(v)] for v in rng.integers(0, 3, n)]),
    }
    for bits in (8, 15, 19, 20, 21, 39, 40, 41):                      # around one and two windows of 19 / 20 / smaller-plan bits
        cases["bits%d" % bits] = np.stack([fe_from_int(int(v)) for v in rng.integers(1, 1 << bits, n, dtype=np.uint64)])
    half = [(1 << c) // 2 + d for c in (10, 13, 16, 19, 20) for d in (-1, 0, 1)]      # raw digit == half: the carry boundary itself
    cases["carry_edges"] = np.stack([fe_from_int(half[int(v)]) for v in rng.integers(0, len(half), n)])
    cases["neg_small"] = np.stack([fe_from_int(R - int(v)) for v in rng.integers(1, 1 << 20, n, dtype=np.uint64)])
    mix = rand_fr(rng, n)
    mix[rng.integers(0, n, max(1, n // 2))] = one                     # half ones, half uniform
    cases["half_ones"] = mix
    for name, s in cases.items():
        assert (B.msm_g1(bases, s) == ob.msm(s, pts)).all(), name
    bases.free()


@pytest.mark.parametrize("distinct", [1, 2, 7, 60, 400])
def test_constant_runs_full_size(hip, distinct):
    """2^20 scalars drawn from a handful of values, in long runs (a permutation product over rows without copy constraints, a
    padded column): one heavy bucket per value and window.  1 .. 7 values: every oversized sort partition gets the multi-workgroup
    sort; 60 values: ~780 oversized partitions, more than its 64 slots, so most take the single-workgroup path; 400: ordinary."""
    from ezkl_amd import backend as B
    n = 1 << 20
    rng = np.random.default_rng(distinct)
    pts, bases = _bases_2_20()
    vals = rand_fr(rng, distinct)
    runs = np.sort(rng.integers(0, distinct, n))                      # long runs of each value
    if distinct == 7:
        runs = rng.integers(0, distinct, n)                           # ... or the same values interleaved
    s = vals[runs]
    assert (B.msm_g1(bases, s) == ob.msm(s, pts)).all()


_B20 = []


def _bases_2_20():
    from ezkl_amd import backend as B
    if not _B20:
        bases = B.Bases.generate(SEED, 1 << 20)          # same deterministic function as the oracle's (test_gen_bases_matches_oracle)
        _B20.append((bases.download(), bases))
    return _B20[0]


def test_sparse_column_is_fast_and_correct(hip):
    """a column with a few thousand equal small values plus a handful of full-width blinding rows (the shape of the
    mv-lookup m(X) column): entries are scattered over ~2^19 mostly empty buckets"""
    import time
    from ezkl_amd import backend as B
    n = 1 << 18
    rng = np.random.default_rng(8)
    bases = B.Bases.generate(SEED, n)
    pts = bases.download()
    s = np.zeros((n, 4), np.uint64)
    s[:4096] = fe_from_int(33)
    s[n - 6:] = rand_fr(rng, 6)
    d = B.DeviceBuffer.from_numpy(s)
    B.msm_g1_dev(bases, d.ptr, n)
    t0 = time.perf_counter()
    got = B.msm_g1_dev(bases, d.ptr, n)
    dt = time.perf_counter() - t0
    assert (got == ob.msm(s, pts)).all()
    # a healthy run is ~0.3 ms; walking every empty bucket (the regression this guards: the empty-bucket binary search of the accumulate
    # loop) costs hundreds of ms.  The bound is loose so that a busy box does not turn a parity test red: EZKL_TEST_PERF_STRICT=1 asks for 10 ms
    import os
    print("sparse MSM of 2^%d points: %.2f ms" % (n.bit_length() - 1, dt * 1e3))
    assert dt < (0.01 if os.environ.get("EZKL_TEST_PERF_STRICT") == "1" else 0.1), "sparse column took %.1f ms" % (dt * 1e3)
    bases.free()


def test_incremental_commit_batch(hip):
    """begin / push / finish == the one-shot batch; other MSM calls are refused while a batch is open"""
    import ezkl_amd
    from ezkl_amd import backend as B
    rng = np.random.default_rng(77)
    n = 1 << 12
    pts = ob.gen_bases(5, n)
    bases = B.Bases(pts)
    cols = [rand_fr(rng, n) for _ in range(9)]          # more columns than pipeline slots
    cols[3][:] = 0
    devs = [B.DeviceBuffer.from_numpy(c) for c in cols]
    want = B.msm_g1_batch_dev(bases, [d.ptr for d in devs], n)
    mb = B.MsmBatch(bases, n)
    for d in devs[:4]:
        mb.push(d.ptr)
    with pytest.raises(ezkl_amd.EzklHipError):
        B.msm_g1_dev(bases, devs[0].ptr, n)               # a batch is open
    with pytest.raises(ezkl_amd.EzklHipError):
        B.MsmBatch(bases, n)
    for d in devs[4:]:
        mb.push(d.ptr)
    got = mb.finish()
    assert got.shape == (9, 8) and (got == want).all()
    assert (got[0] == ob.msm(cols[0], pts)).all() and not got[3].any()
    # the pipeline is free again; a short result buffer is an error that still closes the batch
    mb = B.MsmBatch(bases, n)
    mb.push(devs[1].ptr); mb.push(devs[2].ptr)
    with pytest.raises(ezkl_amd.EzklHipError):
        mb.finish(capacity=1)
    assert (B.msm_g1_dev(bases, devs[1].ptr, n) == want[1]).all()
    # a sub-range of the base set and an empty batch
    mb = B.MsmBatch(bases, n // 2, offset=n // 4)
    mb.push(devs[0].ptr)
    assert (mb.finish()[0] == ob.msm(cols[0][: n // 2], pts[n // 4: n // 4 + n // 2])).all()
    assert B.MsmBatch(bases, n).finish().shape == (0, 8)
    # several columns per push (fused groups at this size: 9 columns = groups of 4, 4, 1 over the slots), mixed with single pushes, and
    # queued behind library-stream work that is still running when push returns (asynchronous mode): the MSMs must see its result
    mb = B.MsmBatch(bases, n)
    mb.push_many([d.ptr for d in devs[:5]])
    mb.push(devs[5].ptr)
    mb.push_many([d.ptr for d in devs[6:]])
    mb.push_many([])
    got = mb.finish()
    assert got.shape == (9, 8) and (got == want).all()
    work = B.DeviceBuffer.from_numpy(cols[0])
    prev = B.set_async(True)
    try:
        mb = B.MsmBatch(bases, n)
        for _ in range(20):                                   # a chain on the library stream: work = cols[0] + 20 * cols[1]
            B.vec_op("add", work.ptr, devs[1].ptr, work.ptr, n)
        mb.push_many([work.ptr, devs[2].ptr])
        got = mb.finish()
    finally:
        B.set_async(prev)
    from conftest import fe_to_int, fe_from_int
    acc = np.stack([fe_from_int((fe_to_int(a) + 20 * fe_to_int(b)) % R) for a, b in zip(cols[0][:64], cols[1][:64])])
    assert (work.to_numpy(shape=(n, 4))[:64] == acc).all()
    assert (got[0] == B.msm_g1_dev(bases, work.ptr, n)).all() and (got[1] == want[2]).all()
    bases.free()


def test_upload_phase_in_steps(hip):
    """ezkl_hip_upload_begin / _wait / _commit / _end: work queued on a caller stream behind column j sees that column (with its
    blinding rows); the commits equal the one-call form; a second open phase and calls on a closed handle are refused"""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(92)
    n, t0 = 1 << 13, (1 << 13) - 6
    pts = ob.gen_bases(10, n)
    bases = B.Bases(pts)
    pinned = [B.PinnedArray((n, 4)) for _ in range(5)]
    cols = [rand_fr(rng, n) for _ in range(5)]
    for pa, c in zip(pinned, cols):
        pa.array[:] = c
    tails = [rand_fr(rng, 6) for _ in range(5)]
    st = B.Stream()
    up = B.UploadPhase([pa.array for pa in pinned], tails, t0)
    with pytest.raises(RuntimeError):
        B.UploadPhase(cols[:1])                                   # one phase at a time
    k = 13
    w_inv = ob.fr_inv(ob.omega(k))
    outs = []
    for j in range(5):                                            # per column, behind its copy: clone + iNTT on the caller stream
        up.wait(j, st)
        o = B.DeviceBuffer(32 * n)
        B.vec_scale(up.devs[j].ptr, fe_from_int(1), o.ptr, n, stream=st.ptr)
        B.ntt_dev(o.ptr, k, w_inv, inverse=True, stream=st.ptr)
        outs.append(o)
    commits = up.commit(bases)
    half = up.commit(bases, commit_range=(100, 5000))            # a second commit of the same phase (a slice)
    up.end()
    st.synchronize()
    for j in range(5):
        want = cols[j].copy()
        want[t0:] = tails[j]
        assert (up.devs[j].to_numpy(shape=(n, 4)) == want).all()
        assert (commits[j] == ob.msm(want, pts)).all()
        assert (half[j] == ob.msm(want[100:5000], pts[:4900])).all()
        assert (outs[j].to_numpy(shape=(n, 4)) == ob.lagrange_to_coeff(want, k)).all()
    up.end()                                                      # idempotent on the Python side
    up2 = B.UploadPhase(cols[:2])                                 # the slot is free again; pageable memory works too
    assert (up2.commit(bases)[1] == ob.msm(cols[1], pts)).all()
    up2.end()
    st.free()
    for pa in pinned:
        pa.free()
    bases.free()


def test_upload_commit_batch(hip):
    """the one-call prover phase (async uploads on a copy stream, blinding rows, pipelined commits) == upload + set rows + commit"""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(91)
    n, t0 = 1 << 12, (1 << 12) - 6
    pts = ob.gen_bases(9, n)
    bases = B.Bases(pts)
    cols = [rand_fr(rng, n) for _ in range(8)]
    pinned = [B.PinnedArray((n, 4)) for _ in range(4)]
    for pa, c in zip(pinned, cols):
        pa.array[:] = c
    host = [pa.array for pa in pinned] + cols[4:]            # page-locked and pageable columns mixed
    tails = [rand_fr(rng, 6) for _ in range(8)]
    devs, commits = B.upload_commit_batch(bases, host, tails, t0)
    for j in range(8):
        want = cols[j].copy()
        want[t0:] = tails[j]
        assert (devs[j].to_numpy(shape=(n, 4)) == want).all()
        assert (commits[j] == ob.msm(want, pts)).all()
    devs2, commits2 = B.upload_commit_batch(bases, cols[:2])          # no blinding rows
    assert (commits2[1] == ob.msm(cols[1], pts)).all() and (devs2[0].to_numpy(shape=(n, 4)) == cols[0]).all()
    # a rank's slice of a sharded SRS: the whole column is uploaded and blinded, rows [lo, hi) are committed against bases [0, hi - lo)
    lo, hi = 1000, 3001
    devs3, commits3 = B.upload_commit_batch(bases, cols[:3], tails[:3], t0, commit_range=(lo, hi))
    for j in range(3):
        want = cols[j].copy()
        want[t0:] = tails[j]
        assert (devs3[j].to_numpy(shape=(n, 4)) == want).all()
        assert (commits3[j] == ob.msm(want[lo:hi], pts[: hi - lo])).all()
    _, commits4 = B.upload_commit_batch(bases, cols[:2], commit_range=(5, 5))     # empty slice: the identity
    assert not commits4.any()
    with pytest.raises(RuntimeError):
        B.upload_commit_batch(bases, cols[:1], commit_range=(n - 4, n + 4))
    for pa in pinned:
        pa.free()
    bases.free()


def test_c5_size_2_22_full_oracle_compare(hip):
    """2^22 points (BASELINE configs[4]'s SRS size) against the oracle's Pippenger over ALL points, uniform and witness-shaped"""
    from ezkl_amd import backend as B
    n = 1 << 22
    bases = B.Bases.generate(SEED + 2, n)
    pts = bases.download()
    s = rand_fr(np.random.default_rng(2222), n)
    assert (B.msm_g1(bases, s) == ob.msm(s, pts)).all()
    wl = witness_like(np.random.default_rng(6), n)
    assert (B.msm_g1(bases, wl) == ob.msm(wl, pts)).all()
    bases.free()


def test_concurrent_callers_overlap(hip):
    """single MSM calls from several host threads (halo2's rayon workers): each holds the context lock only while its launches are
    queued; results are the serial ones, and four threads finish four MSMs in less than four serial calls take"""
    import threading, time
    from ezkl_amd import backend as B
    n = 1 << 18
    rng = np.random.default_rng(11)
    pts = ob.gen_bases(SEED + 9, n)
    bases = B.Bases(pts)
    cols = [rand_fr(rng, n) for _ in range(4)]
    devs = [B.DeviceBuffer.from_numpy(c) for c in cols]
    want = [ob.msm(c, pts) for c in cols]
    serial = [B.msm_g1_dev(bases, d.ptr, n) for d in devs]                # warm (tables, slots)
    assert all((a == b).all() for a, b in zip(serial, want))
    t0 = time.perf_counter()
    for _ in range(5):
        for d in devs:
            B.msm_g1_dev(bases, d.ptr, n)
    t_serial = (time.perf_counter() - t0) / 5
    got = [None] * 4
    def work(i, reps):
        for _ in range(reps):
            got[i] = B.msm_g1_dev(bases, devs[i].ptr, n)
    t_par = None
    for reps in (1, 5, 5, 5):                                             # first round warms the call slots; best of three timed rounds
        th = [threading.Thread(target=work, args=(i, reps)) for i in range(4)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = (time.perf_counter() - t0) / reps
        if reps > 1: t_par = dt if t_par is None else min(t_par, dt)
    assert all((a == b).all() for a, b in zip(got, want))
    print("4 MSMs of 2^18: serial %.3f ms, 4 threads %.3f ms" % (t_serial * 1e3, t_par * 1e3))
    # the overlap is reported, not asserted (measured 0.73x; a wall-clock ratio can turn a correct build red on a busy box)
    # concurrent callers must not serialise behind one another's host tails (call slots): loose bound, EZKL_TEST_PERF_STRICT=1 asks for 1.25 x
    import os
    assert t_par < (1.25 if os.environ.get("EZKL_TEST_PERF_STRICT") == "1" else 2.0) * t_serial, (t_par, t_serial)


def test_params_downsize_matches_the_reference_srs_and_the_oracle(hip, golden_srs):
    """ParamsKZG::downsize (halo2; /root/reference/src/execute.rs:1739-1750): g_lagrange of a smaller domain = inverse NTT over G1 of the
    truncated g (ezkl_hip_bases_downsize).  Pinned three ways: (1) on the REFERENCE's k = 6 SRS file, "downsizing" to k = 6 must rebuild
    the file's own g_lagrange section byte for byte; (2) k' = 4, 5: every point equals the oracle's MSM of the Lagrange polynomial's
    coefficients against g, the sum of the points is g[0], and commit_lagrange(v) == commit(iNTT v) on the downsized set; (3) a device-made
    SRS with a known secret at k = 9 downsized to 7 equals the SRS made directly at k = 7 (L_i(s) G from the closed form)."""
    from ezkl_amd import backend as B
    g, gl = golden_srs["g"], golden_srs["g_lagrange"]
    bg = B.Bases(g)
    g6, gl6 = bg.downsize(6)
    assert (g6.download() == g).all() and (gl6.download() == gl).all()             # (1) the reference's own Lagrange basis
    g6.free(); gl6.free()
    rng = np.random.default_rng(77)
    for kk in (0, 1, 4, 5):
        n2 = 1 << kk
        g2_, gl2 = bg.downsize(kk)
        got = gl2.download()
        assert (g2_.download() == g[:n2]).all()
        assert (got == ob.g1_to_lagrange(g, kk)).all(), kk                           # (2) L_i(s) G through the oracle's naive restatement
        acc = got[0]
        for i in range(1, n2):
            acc = B.g1_add_affine(acc, got[i])
        assert (acc == g[0]).all()
        if kk >= 4:
            v = rand_fr(rng, n2)
            assert (B.msm_g1(gl2, v) == B.msm_g1(g2_, ob.lagrange_to_coeff(v, kk))).all()
        g2_.free(); gl2.free()
    with pytest.raises(Exception):
        bg.downsize(7)                                                              # larger than the set
    bg.free()
    s_ = 0x1234567890abcdef1234567890abcdef % R
    g9, gl9 = B.gen_srs(9, s_)
    g7, gl7 = B.gen_srs(7, s_)
    d7, dl7 = g9.downsize(7)
    assert (d7.download() == g7.download()).all() and (dl7.download() == gl7.download()).all()      # (3)
    p = hip.ParamsKZG(9, g9.download(), gl9.download())
    p.downsize(7)
    v = rand_fr(rng, 128)
    assert p.k == 7 and (p.commit_lagrange(v) == B.msm_g1(gl7, v)).all()
    for b in (g9, gl9, g7, gl7, d7, dl7): b.free()
    p.free()


def test_g2_msm_matches_the_oracle_and_the_host_pairing_code(hip):
    """BN254 G2 on the device (ezkl_hip_msm_g2, csrc/g2.hip): sum_i s_i P_i over the twist against the oracle's Python G2 arithmetic
    (oracle/pairing.py) and the product's host C++ (ezkl_prover_g2_mul_generator: gen_srs's s_g2 = [s] g2, /root/reference/src/pfsys/srs.rs:14-16);
    identity points, zero scalars, repeated points (the doubling branch of the fold) and P + (-P)."""
    from ezkl_amd import backend as B, native as NV
    from oracle import pairing as E
    rng = np.random.default_rng(2024)
    G2 = ((0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed, 0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2),
          (0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa, 0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b))
    assert E.g2_on_curve(G2)
    def enc(pt):                                   # oracle point -> 16 x u64 (x.c0, x.c1, y.c0, y.c1 Montgomery Fq; identity = zeros)
        if pt is None:
            return np.zeros(16, np.uint64)
        return np.concatenate([fe_from_int(c, Q) for c in (pt[0][0], pt[0][1], pt[1][0], pt[1][1])])
    def dec(a):
        v = [fe_to_int(a[4 * i:4 * i + 4], Q) for i in range(4)]
        return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))
    # n = 1: [s] g2 as the SRS generator makes it, three ways
    s_ = int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) % R
    got = B.msm_g2(enc(G2)[None], fe_from_int(s_)[None])
    assert dec(got) == E.g2_mul(G2, s_) and got.tobytes() == NV.g2_mul_generator(s_)
    # a batch with every special case in it
    ks = [int(x) for x in rng.integers(1, 1 << 40, 37)]
    pts = [E.g2_mul(G2, k) for k in ks]
    sc = [int(x) * int(y) % R for x, y in zip(rng.integers(1, 1 << 62, 37), rng.integers(1, 1 << 62, 37))]
    pts[3], ks[3] = None, 0                        # identity point
    sc[5] = 0                                      # zero scalar
    pts[8], ks[8], sc[8] = pts[7], ks[7], sc[7]    # the same (point, scalar) twice
    pts[11], ks[11], sc[11] = (pts[10][0], E.f2_neg(pts[10][1])), R - ks[10], sc[10]      # P and -P with the same scalar cancel
    sc[12] = R - 1
    want = E.g2_mul(G2, sum(s * k for s, k in zip(sc, ks)) % R)
    got = B.msm_g2(np.stack([enc(p) for p in pts]), np.stack([fe_from_int(s) for s in sc]))
    assert dec(got) == want
    # all zero scalars / no points: the identity
    assert not B.msm_g2(np.stack([enc(p) for p in pts]), np.zeros((37, 4), np.uint64)).any()
    # more pairs than threads of the partial kernel (1024): the strided sums + the tree
    n = 1500
    ks2 = [int(x) for x in rng.integers(1, 1 << 20, 8)]
    base = [E.g2_mul(G2, k) for k in ks2]
    idx = rng.integers(0, 8, n)
    sc2 = [int(x) for x in rng.integers(0, 1 << 62, n)]
    want2 = E.g2_mul(G2, sum(s * ks2[i] for s, i in zip(sc2, idx)) % R)
    benc = [enc(p) for p in base]
    got2 = B.msm_g2(np.stack([benc[i] for i in idx]), np.stack([fe_from_int(s) for s in sc2]))
    assert dec(got2) == want2


def test_start_finish_tokens_busy_and_spent(hip):
    """ezkl_hip_msm_g1_start_dev / _finish: four call slots per context -- a fifth start from the thread that holds all four is
    EZKL_ERR_BUSY (-6), not an endless spin; a token is spent by its first finish (a second one, or one never started, is EZKL_ERR_INVALID);
    the results are the synchronous call's; and the kernel-time statistics count every region exactly once"""
    from ezkl_amd import backend as B
    from ezkl_amd.lib import EzklHipError
    n = 1 << 14
    rng = np.random.default_rng(21)
    pts = ob.gen_bases(SEED + 3, n)
    bases = B.Bases(pts)
    cols = [rand_fr(rng, n) for _ in range(4)]
    devs = [B.DeviceBuffer.from_numpy(c) for c in cols]
    want = want_pts = [ob.msm(c, pts) for c in cols]
    B.msm_g1_dev(bases, devs[0].ptr, n)                                  # tables
    B.kernel_ms_stats("msm", reset=True)
    toks = [B.msm_g1_start_dev(bases, d.ptr, n) for d in devs]
    assert sorted(toks) == [0, 1, 2, 3]
    with pytest.raises(EzklHipError) as e:
        B.msm_g1_start_dev(bases, devs[0].ptr, n)
    assert e.value.code == -6
    got = {t: B.msm_g1_finish(t) for t in reversed(toks)}                 # any order
    assert all((got[t] == w).all() for t, w in zip(toks, want))
    for bad in (toks[0], 7, -1):
        with pytest.raises(EzklHipError) as e:
            B.msm_g1_finish(bad)
        assert e.value.code == -3
    t = B.msm_g1_start_dev(bases, devs[1].ptr, n)                         # the slots are free again
    assert (B.msm_g1_finish(t) == want[1]).all()
    total_ms, count = B.kernel_ms_stats("msm")
    assert count == 5 and total_ms > 0
    assert abs(B.last_kernel_ms("msm") - total_ms / 5) < total_ms         # the newest pair is readable on its own
    assert B.kernel_ms_stats("msm", reset=True)[1] == 5 and B.kernel_ms_stats("msm") == (0.0, 0)
    assert B.kernel_ms_stats("no such region") == (0.0, 0)
    # EZKL_HIP_TIMING (read at every call): which event pairs a synchronous MSM records
    import os
    for mode, want in (("kernel", (0, 2)), ("none", (0, 0)), ("all", (2, 2))):
        B.kernel_ms_stats("msm", reset=True); B.kernel_ms_stats("msm_accumulate", reset=True)
        os.environ["EZKL_HIP_TIMING"] = mode
        try:
            for _ in range(2):
                assert (B.msm_g1_dev(bases, devs[2].ptr, n) == want_pts[2]).all()
        finally:
            del os.environ["EZKL_HIP_TIMING"]
        assert (B.kernel_ms_stats("msm")[1], B.kernel_ms_stats("msm_accumulate")[1]) == want, mode
    bases.free()
