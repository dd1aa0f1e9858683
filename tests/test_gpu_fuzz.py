"""Property-based parity (hypothesis): random sizes / offsets / batch shapes through the C ABI against the oracle.
The fixed-size tests pin known edge cases; these look for the ones nobody thought of (ragged lengths, odd strides,
window plans of unusual sizes, scalar patterns that stress the signed-digit carries and the lazy radix-2^29 bounds)."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from conftest import R, SEED, fe_from_int, rand_fr
from oracle import binding as ob

pytestmark = pytest.mark.gpu
FUZZ = settings(max_examples=80, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])

_BASES = {}


def _bases(n):
    from ezkl_amd import backend as B
    if n not in _BASES:
        pts = ob.gen_bases(SEED + n, n)
        _BASES[n] = (pts, B.Bases(pts))
    return _BASES[n]


def _pattern_scalars(rng, n, kind):
    """scalar families that stress different parts of the MSM: carries, sign folding, sparsity, skew"""
    if kind == 0:
        return rand_fr(rng, n)
    if kind == 1:                                   # runs of ones around window boundaries (19 / 20-bit windows, others for small n)
        out = np.zeros((n, 4), np.uint64)
        for i in range(n):
            lo, ln = int(rng.integers(0, 250)), int(rng.integers(1, 80))
            out[i] = fe_from_int((((1 << ln) - 1) << lo) % R)
        return out
    if kind == 2:                                   # values just above / below r/2 and near r (sign folding)
        base = [(R - 1) // 2, (R + 1) // 2, R - 1, R - 2, 1, 2]
        return np.stack([fe_from_int((base[int(rng.integers(0, 6))] + int(rng.integers(-3, 4))) % R) for _ in range(n)])
    if kind == 3:                                   # few distinct values: heavy buckets
        vals = rand_fr(rng, 3)
        return vals[rng.integers(0, 3, n)]
    out = np.zeros((n, 4), np.uint64)               # sparse
    idx = rng.integers(0, n, max(1, n // 50))
    out[idx] = rand_fr(rng, len(idx))
    return out


@FUZZ
@given(logn=st.integers(0, 13), frac=st.floats(0.05, 1.0), off_frac=st.floats(0.0, 0.9), kind=st.integers(0, 4), seed=st.integers(0, 1 << 30))
def test_msm_random_shapes(hip, logn, frac, off_frac, kind, seed):
    from ezkl_amd import backend as B
    nb = 1 << logn
    pts, bases = _bases(nb)
    n = max(1, int(nb * frac))
    off = min(nb - n, int((nb - n) * off_frac))
    rng = np.random.default_rng(seed)
    s = _pattern_scalars(rng, n, kind)
    d = B.DeviceBuffer.from_numpy(s)
    got = B.msm_g1_dev(bases, d.ptr, n, offset=off)
    assert (got == ob.msm(s, pts[off:off + n])).all()


@FUZZ
@given(k=st.integers(1, 14), batch=st.integers(1, 5), pad=st.integers(0, 3), inverse=st.booleans(), seed=st.integers(0, 1 << 30))
def test_ntt_random_batches(hip, k, batch, pad, inverse, seed):
    from ezkl_amd import backend as B
    rng = np.random.default_rng(seed)
    n = 1 << k
    stride = n + pad * 8
    buf = rand_fr(rng, batch * stride)
    d = B.DeviceBuffer.from_numpy(buf)
    w = ob.omega(k)
    if inverse:
        w = ob.fr_inv(w)
    B.ntt_dev(d.ptr, k, w, inverse=inverse, batch=batch, stride=stride)
    got = d.to_numpy(shape=(batch * stride, 4))
    for b in range(batch):
        col = buf[b * stride:b * stride + n]
        want = ob.lagrange_to_coeff(col, k) if inverse else ob.fft(col, k, w)
        assert (got[b * stride:b * stride + n] == want).all()
        assert (got[b * stride + n:(b + 1) * stride] == buf[b * stride + n:(b + 1) * stride]).all()       # padding untouched


@FUZZ
@given(k=st.integers(2, 11), extra=st.integers(1, 3), seed=st.integers(0, 1 << 30))
def test_coset_roundtrip_and_vanishing(hip, k, extra, seed):
    """coeff_to_extended == oracle; divide_by_vanishing then extended_to_coeff == oracle"""
    from ezkl_amd import backend as B
    rng = np.random.default_rng(seed)
    ek = k + extra
    a = rand_fr(rng, 1 << k)
    d = B.DeviceBuffer.from_numpy(a)
    e = B.DeviceBuffer(32 << ek)
    B.coset_ntt_dev(d.ptr, e.ptr, k, ek, inverse=False)
    want = ob.coeff_to_extended(a, k, ek)
    assert (e.to_numpy(shape=(1 << ek, 4)) == want).all()
    B.divide_by_vanishing_dev(e.ptr, k, ek)
    want = ob.divide_by_vanishing(want, k, ek)
    assert (e.to_numpy(shape=(1 << ek, 4)) == want).all()
    B.coset_ntt_dev(e.ptr, e.ptr, k, ek, inverse=True)
    assert (e.to_numpy(shape=(1 << ek, 4)) == ob.extended_to_coeff(want, ek)).all()


@FUZZ
@given(n=st.integers(1, 70000), op=st.sampled_from(["add", "mul"]), exclusive=st.booleans(), zeros=st.booleans(), seed=st.integers(0, 1 << 30))
def test_scan_invert_eval_random_lengths(hip, n, op, exclusive, zeros, seed):
    from ezkl_amd import backend as B
    rng = np.random.default_rng(seed)
    a = rand_fr(rng, n)
    if zeros:
        a[rng.integers(0, n, max(1, n // 7))] = 0
    d = B.DeviceBuffer.from_numpy(a)
    o = B.DeviceBuffer(32 * n)
    B.prefix_scan(op, d.ptr, o.ptr, n, exclusive=exclusive)
    assert (o.to_numpy(shape=(n, 4)) == ob.prefix_scan(a, op, exclusive)).all()
    B.batch_invert(d.ptr, n)
    assert (d.to_numpy(shape=(n, 4)) == ob.batch_invert(a)).all()
    x = rand_fr(rng, 1)[0]
    assert (B.eval_polynomial(o.ptr, n, x) == ob.eval_poly(o.to_numpy(shape=(n, 4)), x)).all()
