"""Host-side mirror of the reference interface for the prove hot path, on top of the C ABI.

Names follow halo2 (the reference's proving stack):
  ParamsKZG.commit_lagrange / commit      <- halo2_proofs::poly::kzg::commitment::ParamsKZG (A10;
                                             call site /root/reference/src/circuit/modules/polycommit.rs:71)
  EvaluationDomain.{lagrange_to_coeff, coeff_to_lagrange, coeff_to_extended, extended_to_coeff,
                    divide_by_vanishing_poly}  <- halo2_proofs::poly::EvaluationDomain (A11)
  GraphProgram.evaluate_h                 <- plonk::evaluation::GraphEvaluator / evaluate_h (A12)
All field data are numpy uint64 arrays (..., 4) in Montgomery form = the bytes halo2 holds in memory.
"""
import ctypes as C
import numpy as np
from . import lib as _l

_vp = C.c_void_p


def _p(a):
    return a.ctypes.data_as(_vp)


def _fe(a, shape_last=4):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.shape[-1] != shape_last:
        raise ValueError("expected trailing dimension %d" % shape_last)
    return a


def init(device=-1):
    _l.check(_l.load().ezkl_hip_init(C.c_int(device)), "ezkl_hip_init")


def synchronize():
    _l.check(_l.load().ezkl_hip_synchronize(), "ezkl_hip_synchronize")


def device_count():
    return int(_l.load().ezkl_hip_device_count())


def _stream_ptr(stream):
    return _vp(stream) if stream else _vp(None)


class DeviceBuffer:
    """A library-owned HBM allocation (resident column / scalar vector)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = _vp()
        _l.check(_l.load().ezkl_hip_malloc(C.byref(p), C.c_size_t(self.nbytes)), "ezkl_hip_malloc")
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        _l.check(_l.load().ezkl_hip_memcpy_h2d(_vp(b.ptr), _p(a), C.c_size_t(a.nbytes)), "h2d")
        return b

    def to_numpy(self, dtype=np.uint64, shape=None):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _l.check(_l.load().ezkl_hip_memcpy_d2h(_p(out), _vp(self.ptr), C.c_size_t(self.nbytes)), "d2h")
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if self.ptr:
            _l.load().ezkl_hip_free(_vp(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceView:
    """a window of another resident buffer (no ownership): .ptr / .nbytes like DeviceBuffer, keeps its parent alive"""

    def __init__(self, ptr, nbytes, parent):
        self.ptr, self.nbytes, self._parent = ptr, nbytes, parent


class PinnedArray:
    """a numpy view of page-locked host memory (ezkl_hip_host_malloc): witness columns filled here upload at PCIe speed"""

    def __init__(self, shape, dtype=np.uint64):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = _vp()
        _l.check(_l.load().ezkl_hip_host_malloc(C.byref(self._p), C.c_size_t(self.nbytes)), "ezkl_hip_host_malloc")
        buf = (C.c_uint8 * max(1, self.nbytes)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self._p:
            self.array = None
            _l.load().ezkl_hip_host_free(self._p)
            self._p = None


class _Bases:
    def __init__(self, pts):
        pts = _fe(pts, 8)
        self.n = pts.shape[0]
        h = _vp()
        _l.check(_l.load().ezkl_hip_bases_upload(_p(pts), C.c_size_t(self.n), C.byref(h)), "ezkl_hip_bases_upload")
        self.h = h

    @classmethod
    def generate(cls, seed, n, first=0):
        self = cls.__new__(cls)
        self.n = n
        h = _vp()
        _l.check(_l.load().ezkl_hip_bases_generate(C.c_uint64(seed), C.c_size_t(first), C.c_size_t(n), C.byref(h)),
                 "ezkl_hip_bases_generate")
        self.h = h
        return self

    @classmethod
    def from_scalars(cls, base_point, scalars_ptr, n):
        """bases[i] = scalars[i] * base_point (SRS generation)"""
        self = cls.__new__(cls)
        self.n = n
        h = _vp()
        _l.check(_l.load().ezkl_hip_bases_from_scalars(_p(_fe(base_point, 8)), _vp(scalars_ptr), C.c_size_t(n), C.byref(h)),
                 "ezkl_hip_bases_from_scalars")
        self.h = h
        return self

    def downsize(self, new_k):
        """halo2 ParamsKZG::downsize on a coefficient-basis set: (first 2^new_k points, Lagrange basis of the 2^new_k domain) as new
        resident base sets -- the Lagrange basis is an inverse NTT over G1 on the device (ezkl_hip_bases_downsize)"""
        hg, hl = _vp(), _vp()
        _l.check(_l.load().ezkl_hip_bases_downsize(self.h, C.c_uint32(new_k), C.byref(hg), C.byref(hl)), "ezkl_hip_bases_downsize")
        out = []
        for h in (hg, hl):
            b = type(self).__new__(type(self))
            b.n, b.h = 1 << new_k, h
            out.append(b)
        return tuple(out)

    def download(self):
        out = np.empty((self.n, 8), np.uint64)
        _l.check(_l.load().ezkl_hip_bases_download(self.h, _p(out)), "ezkl_hip_bases_download")
        return out

    def prepare(self):
        """start the window-table precompute now, without waiting (ezkl_hip_bases_prepare)"""
        _l.check(_l.load().ezkl_hip_bases_prepare(self.h), "ezkl_hip_bases_prepare")
        return self

    def free(self):
        if self.h:
            _l.load().ezkl_hip_bases_free(self.h)
            self.h = None


Bases = _Bases


def msm_g1_dev(bases, scalars_ptr, n, offset=0, stream=None):
    out = np.zeros(8, np.uint64)
    _l.check(_l.load().ezkl_hip_msm_g1_dev(bases.h, C.c_size_t(offset), _vp(scalars_ptr), C.c_size_t(n), _p(out),
                                            _stream_ptr(stream)), "ezkl_hip_msm_g1_dev")
    return out


def msm_g1_start_dev(bases, scalars_ptr, n, offset=0):
    """the first half of msm_g1_dev: queues the MSM on a call slot of its own and returns a token (ezkl_hip_msm_g1_start_dev; at most four in
    flight per context -- a fifth from the thread that holds all four raises EzklHipError with code EZKL_ERR_BUSY = -6)"""
    tok = C.c_int(-1)
    _l.check(_l.load().ezkl_hip_msm_g1_start_dev(bases.h, C.c_size_t(offset), _vp(scalars_ptr), C.c_size_t(n), C.byref(tok)), "ezkl_hip_msm_g1_start_dev")
    return int(tok.value)


def msm_g1_finish(token):
    """waits for the MSM behind `token` and returns its affine point; a token is spent by its first finish"""
    out = np.zeros(8, np.uint64)
    _l.check(_l.load().ezkl_hip_msm_g1_finish(C.c_int(token), _p(out)), "ezkl_hip_msm_g1_finish")
    return out


def msm_g1_batch_dev(bases, scalar_ptrs, n, offset=0, stream=None):
    """batch of MSMs over resident scalar vectors (one commit phase); returns (batch, 8) affine points"""
    out = np.zeros((len(scalar_ptrs), 8), np.uint64)
    arr = (C.c_void_p * len(scalar_ptrs))(*scalar_ptrs)
    _l.check(_l.load().ezkl_hip_msm_g1_batch_dev(bases.h, C.c_size_t(offset), arr, C.c_size_t(len(scalar_ptrs)), C.c_size_t(n),
                                                  _p(out), _stream_ptr(stream)), "ezkl_hip_msm_g1_batch_dev")
    return out


class MsmBatch:
    """incremental commit batch: push resident columns one at a time (uploads of the next column overlap the MSMs of
    the previous ones), finish() -> (count, 8) affine points"""

    def __init__(self, bases, n, offset=0):
        self.h = _vp()
        self.count = 0
        _l.check(_l.load().ezkl_hip_msm_batch_begin(bases.h, C.c_size_t(offset), C.c_size_t(n), C.byref(self.h)), "ezkl_hip_msm_batch_begin")

    def push(self, scalars_ptr):
        _l.check(_l.load().ezkl_hip_msm_batch_push_dev(self.h, _vp(scalars_ptr)), "ezkl_hip_msm_batch_push_dev")
        self.count += 1

    def push_many(self, scalar_ptrs):
        """several columns at once (fused into groups like a one-call batch); returns once their MSMs are queued"""
        arr = (C.c_void_p * max(1, len(scalar_ptrs)))(*[_vp(p) for p in scalar_ptrs])
        _l.check(_l.load().ezkl_hip_msm_batch_push_many_dev(self.h, arr, C.c_size_t(len(scalar_ptrs))), "ezkl_hip_msm_batch_push_many_dev")
        self.count += len(scalar_ptrs)

    def finish(self, capacity=None):
        cap = self.count if capacity is None else capacity
        out = np.zeros((max(cap, 1), 8), np.uint64)
        h, self.h = self.h, None
        _l.check(_l.load().ezkl_hip_msm_batch_finish(h, _p(out), C.c_size_t(cap)), "ezkl_hip_msm_batch_finish")
        return out[:self.count]


class Stream:
    """a caller stream for the `stream=` arguments (pass `.ptr`): work queued on it is asynchronous until synchronize()"""

    def __init__(self):
        p = _vp()
        _l.check(_l.load().ezkl_hip_stream_create(C.byref(p)), "ezkl_hip_stream_create")
        self.ptr = p.value

    def synchronize(self):
        _l.check(_l.load().ezkl_hip_stream_synchronize(_vp(self.ptr)), "ezkl_hip_stream_synchronize")

    def free(self):
        if self.ptr:
            _l.check(_l.load().ezkl_hip_stream_destroy(_vp(self.ptr)), "ezkl_hip_stream_destroy")
            self.ptr = None


class UploadPhase:
    """upload_commit_batch in steps (ezkl_hip_upload_begin / _wait / _commit / _end): the copies are queued by the
    constructor; wait(j, stream) orders a caller stream behind column j; commit(bases) returns the (batch, 8) points;
    end() closes the phase.  The host columns are kept alive until end()."""

    def __init__(self, host_cols, tails=None, tail_start=0):
        self.cols = [np.ascontiguousarray(a, np.uint64) for a in host_cols]
        self.n = self.cols[0].shape[0]
        self.devs = [DeviceBuffer(32 * self.n) for _ in self.cols]
        m = len(self.cols)
        hp = (C.c_void_p * m)(*[a.ctypes.data for a in self.cols])
        dp = (C.c_void_p * m)(*[d.ptr for d in self.devs])
        tp, tcount, self._tails = None, 0, []
        if tails is not None:
            self._tails = [np.ascontiguousarray(t, np.uint64) for t in tails]
            tcount = self._tails[0].shape[0]
            tp = (C.c_void_p * m)(*[t.ctypes.data for t in self._tails])
        self.h = _vp()
        _l.check(_l.load().ezkl_hip_upload_begin(hp, dp, C.c_size_t(m), C.c_size_t(self.n), tp, C.c_size_t(tail_start), C.c_size_t(tcount), C.byref(self.h)),
                 "ezkl_hip_upload_begin")

    def wait(self, j, stream):
        _l.check(_l.load().ezkl_hip_upload_wait(self.h, C.c_size_t(j), _vp(stream.ptr if isinstance(stream, Stream) else stream)), "ezkl_hip_upload_wait")

    def commit(self, bases, commit_range=None):
        lo, hi = commit_range if commit_range is not None else (0, self.n)
        out = np.zeros((len(self.cols), 8), np.uint64)
        _l.check(_l.load().ezkl_hip_upload_commit(self.h, bases.h, C.c_size_t(lo), C.c_size_t(hi - lo), _p(out)), "ezkl_hip_upload_commit")
        return out

    def end(self):
        if self.h:
            h, self.h = self.h, None
            _l.check(_l.load().ezkl_hip_upload_end(h), "ezkl_hip_upload_end")


def upload_commit_batch(bases, host_cols, tails=None, tail_start=0, commit_range=None):
    """one prover phase: upload the host columns ((n,4) u64 arrays, ideally PinnedArray views), overwrite rows
    [tail_start, tail_start + t) of column j with tails[j] ((t,4) arrays), commit each (commit_range = (lo, hi): only rows [lo, hi)
    against bases [0, hi - lo), a rank's slice of a sharded SRS).  Returns (device columns, (batch, 8) points)."""
    cols = [np.ascontiguousarray(a, np.uint64) for a in host_cols]
    n = cols[0].shape[0]
    devs = [DeviceBuffer(32 * n) for _ in cols]
    hp = (C.c_void_p * len(cols))(*[a.ctypes.data for a in cols])
    dp = (C.c_void_p * len(cols))(*[d.ptr for d in devs])
    tp, tcount, keep = None, 0, []
    if tails is not None:
        keep = [np.ascontiguousarray(t, np.uint64) for t in tails]
        tcount = keep[0].shape[0]
        tp = (C.c_void_p * len(cols))(*[t.ctypes.data for t in keep])
    out = np.zeros((len(cols), 8), np.uint64)
    lo, hi = commit_range if commit_range is not None else (0, n)
    _l.check(_l.load().ezkl_hip_upload_commit_batch(bases.h, hp, dp, C.c_size_t(len(cols)), C.c_size_t(n), tp, C.c_size_t(tail_start), C.c_size_t(tcount),
                                                    C.c_size_t(lo), C.c_size_t(hi - lo), _p(out)), "ezkl_hip_upload_commit_batch")
    return devs, out


def msm_g1(bases, scalars):
    """sum_i scalars[i]*bases[i]; scalars numpy (n,4) or a DeviceBuffer-resident vector via msm_dev."""
    s = _fe(scalars)
    out = np.zeros(8, np.uint64)
    _l.check(_l.load().ezkl_hip_msm_g1(bases.h, _p(s), C.c_size_t(s.shape[0]), _p(out)), "ezkl_hip_msm_g1")
    return out


class ParamsKZG:
    """KZG SRS with both base sets resident in HBM (g: coefficient basis, g_lagrange: Lagrange basis)."""

    def __init__(self, k, g, g_lagrange):
        self.k, self.n = k, 1 << k
        self._g = _Bases(g)
        self._gl = _Bases(g_lagrange)

    @classmethod
    def read(cls, buf):
        """raw-bytes SRS file: u32 LE k | 2^k G1 g | 2^k G1 g_lagrange | g2 | s_g2 (SURVEY.md §8(c) item 1;
        the reader being mirrored is /root/reference/src/pfsys/srs.rs:40-47 -> ParamsKZG::read)."""
        k = int.from_bytes(buf[0:4], "little")
        n = 1 << k
        if len(buf) != 4 + 2 * 64 * n + 256:
            raise ValueError("bad SRS length")
        g = np.frombuffer(buf, np.uint64, count=8 * n, offset=4).reshape(n, 8)
        gl = np.frombuffer(buf, np.uint64, count=8 * n, offset=4 + 64 * n).reshape(n, 8)
        return cls(k, g, gl)

    def downsize(self, new_k):
        """ParamsKZG::downsize (halo2; /root/reference/src/execute.rs:1739-1750 calls it whenever the SRS file is larger than the circuit):
        truncate g, rebuild g_lagrange for the smaller domain (an inverse NTT over G1, on the device).  In place, like halo2's."""
        if new_k > self.k:
            raise ValueError("cannot downsize a k=%d SRS to k=%d" % (self.k, new_k))
        if new_k == self.k:
            return self
        g, gl = self._g.downsize(new_k)
        self._g.free(); self._gl.free()
        self._g, self._gl, self.k, self.n = g, gl, new_k, 1 << new_k
        return self

    def commit_lagrange(self, poly):
        """MSM against g_lagrange; returns the canonical affine point (8 x u64). Blind is ignored by KZG."""
        return msm_g1(self._gl, poly)

    def commit(self, poly):
        return msm_g1(self._g, poly)

    def commit_dev(self, scalars_dev, n, lagrange=True, offset=0, stream=None):
        out = np.zeros(8, np.uint64)
        b = self._gl if lagrange else self._g
        _l.check(_l.load().ezkl_hip_msm_g1_dev(b.h, C.c_size_t(offset), _vp(scalars_dev), C.c_size_t(n), _p(out),
                                                _stream_ptr(stream)), "ezkl_hip_msm_g1_dev")
        return out

    def free(self):
        self._g.free()
        self._gl.free()


def polycommit_commit(message, num_unusable_rows, params):
    """PolyCommitChip::commit (/root/reference/src/circuit/modules/polycommit.rs:46-81): pad the message into
    ceil-ish(len / (2^k - u)) Lagrange-basis columns (num_poly = len / n + 1 as the reference computes it), leave
    the u unusable rows at Blind::default().0 (= Fr::ONE in halo2, passed here as `blind`), commit each column with
    commit_lagrange and return the normalised affine points.  The batch goes through the pipelined MSM path."""
    message = _fe(message)
    n_rows = 1 << params.k
    n = n_rows - num_unusable_rows
    num_poly = message.shape[0] // n + 1
    polys = np.zeros((num_poly, n_rows, 4), np.uint64)
    polys[:, n:, :] = _to_mont(1)                       # Blind::default() == Blind(Fr::ONE)
    for i in range(message.shape[0]):
        polys[i // n, i % n] = message[i]
    out = np.zeros((num_poly, 8), np.uint64)
    arr = (C.c_void_p * num_poly)(*[polys[j].ctypes.data for j in range(num_poly)])
    _l.check(_l.load().ezkl_hip_msm_g1_batch(params._gl.h, arr, C.c_size_t(num_poly), C.c_size_t(n_rows), _p(out)),
             "ezkl_hip_msm_g1_batch")
    return out


def msm_g2(points, scalars):
    """sum_i scalars[i] * points[i] over G2: points (n, 16) u64 affine (x.c0, x.c1, y.c0, y.c1 Montgomery Fq, the SRS file's layout),
    scalars (n, 4) Montgomery Fr -> (16,) u64 affine (ezkl_hip_msm_g2)"""
    points, scalars = _fe(points, 16), _fe(scalars)
    if points.shape[0] != scalars.shape[0]:
        raise ValueError("one scalar per point")
    out = np.zeros(16, np.uint64)
    _l.check(_l.load().ezkl_hip_msm_g2(_p(points), _p(scalars), C.c_size_t(points.shape[0]), _p(out)), "ezkl_hip_msm_g2")
    return out


def g1_add_affine(a, b):
    a, b = _fe(a, 8), _fe(b, 8)
    out = np.zeros(8, np.uint64)
    _l.check(_l.load().ezkl_hip_g1_add_affine(_p(a), _p(b), _p(out)), "ezkl_hip_g1_add_affine")
    return out


def ntt(a, log_n, omega, inverse=False):
    """best_fft(a, omega, log_n) on a host array (copy in, transform, copy out)."""
    a = _fe(a).copy()
    w = _fe(omega)
    _l.check(_l.load().ezkl_hip_ntt(_p(a), C.c_uint32(log_n), _p(w), C.c_int(1 if inverse else 0)), "ezkl_hip_ntt")
    return a


def ntt_dev(ptr, log_n, omega, inverse=False, batch=1, stride=None, stream=None):
    w = _fe(omega)
    stride = (1 << log_n) if stride is None else stride
    _l.check(_l.load().ezkl_hip_ntt_dev(_vp(ptr), C.c_uint32(log_n), _p(w), C.c_int(1 if inverse else 0),
                                         C.c_size_t(batch), C.c_size_t(stride), _stream_ptr(stream)), "ezkl_hip_ntt_dev")


def set_async(on):
    """ezkl_hip_set_async for the calling thread: library-stream calls return once queued (True) / when done (False, which drains the
    stream).  Returns the previous setting."""
    prev = C.c_int(0)
    _l.check(_l.load().ezkl_hip_set_async(C.c_int(1 if on else 0), C.byref(prev)), "ezkl_hip_set_async")
    return bool(prev.value)


def vec_op(op, a_ptr, b_ptr, out_ptr, n, stream=None):
    code = {"add": 0, "sub": 1, "mul": 2}[op]
    _l.check(_l.load().ezkl_hip_vec_op_dev(C.c_int(code), _vp(a_ptr), _vp(b_ptr), _vp(out_ptr), C.c_size_t(n),
                                            _stream_ptr(stream)), "ezkl_hip_vec_op_dev")


def permutation_sigma_dev(next_ptr, omega_col_ptr, delta_pows_ptr, n_columns, log_n, out_ptr, stream=None):
    """out[r] = delta^(t >> log_n) * omega^(t mod 2^log_n), t = next[r]: one sigma column from the cycle successors of its cells (u32, device)"""
    _l.check(_l.load().ezkl_hip_permutation_sigma_dev(_vp(next_ptr), _vp(omega_col_ptr), _vp(delta_pows_ptr), C.c_uint32(n_columns), C.c_uint32(log_n),
                                                      _vp(out_ptr), _stream_ptr(stream)), "ezkl_hip_permutation_sigma_dev")


def vec_scale(a_ptr, scalar, out_ptr, n, stream=None):
    _l.check(_l.load().ezkl_hip_vec_scale_dev(_vp(a_ptr), _p(_fe(scalar)), _vp(out_ptr), C.c_size_t(n), _stream_ptr(stream)), "ezkl_hip_vec_scale_dev")


def vec_fill(out_ptr, value, n, stream=None):
    _l.check(_l.load().ezkl_hip_vec_fill_dev(_vp(out_ptr), _p(_fe(value)), C.c_size_t(n), _stream_ptr(stream)), "ezkl_hip_vec_fill_dev")


def coset_ntt_dev(in_ptr, out_ptr, k, ext_k, inverse=False, in_len=None, batch=1, stream=None):
    """coeff_to_extended (inverse=False: 2^k coefficients in, 2^ext_k evaluations out) / extended_to_coeff on resident columns"""
    nin = (1 << ext_k) if inverse else (1 << k)
    _l.check(_l.load().ezkl_hip_coset_ntt_dev(_vp(in_ptr), _vp(out_ptr), C.c_size_t(batch), C.c_size_t(nin), C.c_size_t(1 << ext_k),
                                               C.c_uint32(k), C.c_uint32(ext_k), C.c_int(1 if inverse else 0), _stream_ptr(stream)),
             "ezkl_hip_coset_ntt_dev")


def coeff_to_cosets_dev(in_ptr, out_ptr, k, ext_k, batch=1, stream=None):
    """coeff_to_extended with the 2^(ext_k - k) cosets of the extended domain stored one after the other (coset-major): out[b 2^k + j]
    = p(zeta w_ext^b omega^j) = natural-order element E j + b"""
    _l.check(_l.load().ezkl_hip_coeff_to_cosets_dev(_vp(in_ptr), _vp(out_ptr), C.c_size_t(batch), C.c_size_t(1 << k), C.c_size_t(1 << ext_k),
                                                     C.c_uint32(k), C.c_uint32(ext_k), _stream_ptr(stream)), "ezkl_hip_coeff_to_cosets_dev")


def coeff_to_cosets_range_dev(in_ptr, out_ptr, k, ext_k, first_coset, n_cosets, batch=1, stream=None):
    """cosets [first_coset, first_coset + n_cosets) of coeff_to_cosets_dev only (n_cosets a power of two): out[(b - first) 2^k + j]"""
    _l.check(_l.load().ezkl_hip_coeff_to_cosets_range_dev(_vp(in_ptr), _vp(out_ptr), C.c_size_t(batch), C.c_size_t(1 << k), C.c_size_t(n_cosets << k),
                                                           C.c_uint32(k), C.c_uint32(ext_k), C.c_uint32(first_coset), C.c_uint32(n_cosets), _stream_ptr(stream)),
             "ezkl_hip_coeff_to_cosets_range_dev")


def cosets_transpose_dev(in_ptr, out_ptr, k, ext_k, to_natural=True, stream=None):
    _l.check(_l.load().ezkl_hip_cosets_transpose_dev(_vp(in_ptr), _vp(out_ptr), C.c_uint32(k), C.c_uint32(ext_k), C.c_int(1 if to_natural else 0),
                                                      _stream_ptr(stream)), "ezkl_hip_cosets_transpose_dev")


def divide_by_vanishing_dev(ptr, k, ext_k, stream=None):
    _l.check(_l.load().ezkl_hip_divide_by_vanishing_dev(_vp(ptr), C.c_uint32(k), C.c_uint32(ext_k), _stream_ptr(stream)),
             "ezkl_hip_divide_by_vanishing_dev")


def memcpy_h2d(dst_ptr, arr):
    arr = np.ascontiguousarray(arr)
    _l.check(_l.load().ezkl_hip_memcpy_h2d(_vp(dst_ptr), _p(arr), C.c_size_t(arr.nbytes)), "h2d")


def memcpy_d2h(src_ptr, nbytes):
    out = np.empty(nbytes, np.uint8)
    _l.check(_l.load().ezkl_hip_memcpy_d2h(_p(out), _vp(src_ptr), C.c_size_t(nbytes)), "d2h")
    return out


def prefix_scan(op, in_ptr, out_ptr, n, exclusive=False, stream=None):
    code = {"add": 0, "mul": 2}[op]
    _l.check(_l.load().ezkl_hip_prefix_scan_dev(C.c_int(code), C.c_int(1 if exclusive else 0), _vp(in_ptr), _vp(out_ptr),
                                                 C.c_size_t(n), _stream_ptr(stream)), "ezkl_hip_prefix_scan_dev")


def lookup_multiplicity(input_ptrs, table_ptr, n_rows, usable_rows, stream=None):
    """m(X) of mv-lookup: returns (DeviceBuffer with the Fr column, number of input values missing from the table)"""
    out = DeviceBuffer(n_rows * 32)
    miss = C.c_uint32(0)
    arr = (C.c_void_p * max(1, len(input_ptrs)))(*input_ptrs)
    _l.check(_l.load().ezkl_hip_lookup_multiplicity_dev(arr, C.c_uint32(len(input_ptrs)), _vp(table_ptr), C.c_uint32(n_rows),
                                                         C.c_uint32(usable_rows), _vp(out.ptr), C.byref(miss), _stream_ptr(stream)),
             "ezkl_hip_lookup_multiplicity_dev")
    return out, int(miss.value)


def lookup_multiplicity_batch(inputs_per_lookup, table_ptrs, n_rows, usable_rows, stream=None):
    """m(X) of every lookup argument in one call: inputs_per_lookup[l] = the input column pointers of argument l.  Returns
    ([DeviceBuffer per argument], total number of input values missing from their tables)"""
    L = len(table_ptrs)
    outs = [DeviceBuffer(n_rows * 32) for _ in range(L)]
    flat = [p for ins in inputs_per_lookup for p in ins]
    which = [l for l, ins in enumerate(inputs_per_lookup) for _ in ins]
    miss = DeviceBuffer.from_numpy(np.zeros(1, np.uint32))
    a_in = (C.c_void_p * max(1, len(flat)))(*flat)
    a_which = (C.c_uint32 * max(1, len(which)))(*which)
    a_tab = (C.c_void_p * L)(*table_ptrs)
    a_out = (C.c_void_p * L)(*[o.ptr for o in outs])
    _l.check(_l.load().ezkl_hip_lookup_multiplicity_batch_dev(a_in, a_which, C.c_uint32(len(flat)), a_tab, C.c_uint32(L), C.c_uint32(n_rows), C.c_uint32(usable_rows),
                                                               a_out, _vp(miss.ptr), _stream_ptr(stream)), "ezkl_hip_lookup_multiplicity_batch_dev")
    return outs, int(miss.to_numpy(np.uint32, (1,))[0])


def eval_polynomial(coeffs_ptr, n, x, stream=None):
    """halo2 eval_polynomial on a resident coefficient vector"""
    out = np.zeros(4, np.uint64)
    _l.check(_l.load().ezkl_hip_eval_poly_dev(_vp(coeffs_ptr), C.c_size_t(n), _p(_fe(x)), _p(out), _stream_ptr(stream)),
             "ezkl_hip_eval_poly_dev")
    return out


def eval_polynomial_batch(coeff_ptrs, n, xs, stream=None):
    """[eval_polynomial(poly_j, xs[j])] for resident polynomials, one round trip for the whole batch"""
    m = len(coeff_ptrs)
    out = np.zeros((m, 4), np.uint64)
    if m == 0:
        return out
    arr = (C.c_void_p * m)(*coeff_ptrs)
    x = _fe(np.asarray(xs, np.uint64).reshape(m, 4))
    _l.check(_l.load().ezkl_hip_eval_poly_batch_dev(arr, _p(x), C.c_uint32(m), C.c_size_t(n), _p(out), _stream_ptr(stream)), "ezkl_hip_eval_poly_batch_dev")
    return out


def lincomb(input_ptrs, coeffs, out_ptr, n, accumulate=False, stream=None):
    """out = (out if accumulate else 0) + sum_j coeffs[j] * inputs[j], fused"""
    m = len(input_ptrs)
    arr = (C.c_void_p * max(1, m))(*input_ptrs)
    cf = _fe(np.asarray(coeffs, np.uint64).reshape(m, 4)) if m else np.zeros((1, 4), np.uint64)
    _l.check(_l.load().ezkl_hip_lincomb_dev(arr, _p(cf), C.c_uint32(m), _vp(out_ptr), C.c_size_t(n), C.c_int(1 if accumulate else 0), _stream_ptr(stream)),
             "ezkl_hip_lincomb_dev")


def kate_division(a_ptr, z, out_ptr, n, stream=None):
    """out = a(X) / (X - z) without the remainder (halo2's kate_division); out may alias a"""
    _l.check(_l.load().ezkl_hip_kate_division_dev(_vp(a_ptr), _p(_fe(z)), _vp(out_ptr), C.c_size_t(n), _stream_ptr(stream)), "ezkl_hip_kate_division_dev")


def chacha20_fr(key32, stream_id, out_ptr, n, first=0, stream=None):
    """n uniform Fr elements of ChaCha20 stream `stream_id` under the 32-byte key, starting at element `first`"""
    key = np.frombuffer(bytes(key32), np.uint8)
    assert key.size == 32
    _l.check(_l.load().ezkl_hip_chacha20_fr_dev(_p(key), C.c_uint64(stream_id), C.c_size_t(first), _vp(out_ptr), C.c_size_t(n), _stream_ptr(stream)),
             "ezkl_hip_chacha20_fr_dev")


def batch_invert(ptr, n, stream=None):
    _l.check(_l.load().ezkl_hip_batch_invert_dev(_vp(ptr), C.c_size_t(n), _stream_ptr(stream)), "ezkl_hip_batch_invert_dev")


def gen_srs(k, s):
    """test SRS with a KNOWN secret s (insecure, like the reference's gen_srs, src/pfsys/srs.rs:13-16):
    g[i] = s^i G, g_lagrange[i] = L_i(s) G with L_i(s) = (s^n - 1) w^i / (n (s - w^i)); everything on the device.
    Returns (g, g_lagrange) as Bases."""
    n = 1 << k
    w = pow(EvaluationDomain.ROOT, 1 << (28 - k), _R)
    G = np.frombuffer((_MONT % _Q).to_bytes(32, "little") + (2 * _MONT % _Q).to_bytes(32, "little"), np.uint64).copy()
    spow = DeviceBuffer(n * 32)
    vec_fill(spow.ptr, _to_mont(s), n)
    prefix_scan("mul", spow.ptr, spow.ptr, n, exclusive=True)                 # s^i
    g = _Bases.from_scalars(G, spow.ptr, n)
    wp = omega_powers_column(k)                                               # w^i
    den = DeviceBuffer(n * 32)
    vec_fill(den.ptr, _to_mont(s), n)
    vec_op("sub", den.ptr, wp.ptr, den.ptr, n)                                # s - w^i
    batch_invert(den.ptr, n)
    vec_op("mul", den.ptr, wp.ptr, den.ptr, n)
    vec_scale(den.ptr, _to_mont((pow(s, n, _R) - 1) * pow(n, -1, _R) % _R), den.ptr, n)
    gl = _Bases.from_scalars(G, den.ptr, n)
    return g, gl


_Q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47

# Montgomery constants needed host-side (derived, not copied: tools/gen_constants.py)
_R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
_MONT = 1 << 256


def _to_mont(x):
    return np.frombuffer((x * _MONT % _R).to_bytes(32, "little"), np.uint64).copy()


class EvaluationDomain:
    """halo2 EvaluationDomain(j, k): n = 2^k, extended domain 2^ext_k with ext_k = k + ceil(log2(j-1))."""

    ROOT = pow(7, (_R - 1) >> 28, _R)

    def __init__(self, j, k):
        self.k = k
        quotient_poly_degree = j - 1
        self.ext_k = k
        while (1 << self.ext_k) < (1 << k) * quotient_poly_degree:
            self.ext_k += 1
        self.n, self.ne = 1 << k, 1 << self.ext_k
        w = pow(self.ROOT, 1 << (28 - k), _R)
        we = pow(self.ROOT, 1 << (28 - self.ext_k), _R)
        self.omega, self.omega_inv = _to_mont(w), _to_mont(pow(w, -1, _R))
        self.extended_omega, self.extended_omega_inv = _to_mont(we), _to_mont(pow(we, -1, _R))

    def lagrange_to_coeff(self, a):
        return ntt(a, self.k, self.omega_inv, inverse=True)

    def coeff_to_lagrange(self, a):
        return ntt(a, self.k, self.omega, inverse=False)

    def _coset(self, cols, inverse):
        cols = [_fe(c) for c in cols]
        nout = self.ne
        outs = [np.empty((nout, 4), np.uint64) for _ in cols]
        ins = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
        ous = (C.c_void_p * len(cols))(*[o.ctypes.data for o in outs])
        _l.check(_l.load().ezkl_hip_coset_ntt_batch(ins, ous, C.c_size_t(len(cols)), C.c_uint32(self.k),
                                                     C.c_uint32(self.ext_k), C.c_int(1 if inverse else 0)),
                 "ezkl_hip_coset_ntt_batch")
        return outs

    def coeff_to_extended(self, a):
        return self._coset([a], False)[0]

    def extended_to_coeff(self, a):
        return self._coset([a], True)[0]

    def divide_by_vanishing_poly(self, a):
        buf = DeviceBuffer.from_numpy(_fe(a))
        _l.check(_l.load().ezkl_hip_divide_by_vanishing_dev(_vp(buf.ptr), C.c_uint32(self.k), C.c_uint32(self.ext_k),
                                                             _vp(None)), "ezkl_hip_divide_by_vanishing_dev")
        out = buf.to_numpy(shape=(self.ne, 4))
        buf.free()
        return out


class _Prog(C.Structure):
    _fields_ = [("code", _vp), ("n_instr", C.c_uint32), ("n_intermediates", C.c_uint32),
                ("constants", _vp), ("n_constants", C.c_uint32),
                ("rotations", _vp), ("n_rotations", C.c_uint32),
                ("columns", _vp), ("n_columns", C.c_uint32),
                ("challenges", _vp), ("n_challenges", C.c_uint32),
                ("k", C.c_uint32), ("ext_k", C.c_uint32)]


OPS = dict(add=0, sub=1, mul=2, square=3, double=4, negate=5, store=6, horner_step=7)
CONST, INTERMEDIATE, COLUMN, CHALLENGE, PREVIOUS = range(5)


class GraphProgram:
    """Builder for the straight-line program a GraphEvaluator holds (constants, rotations, calculations).

    add_calculation mirrors GraphEvaluator::add_calculation: returns the ValueSource of the result.
    Horner(start, parts, factor) is lowered to store + horner_step, the body of Calculation::Horner."""

    def __init__(self, k, ext_k):
        self.k, self.ext_k = k, ext_k
        self.code, self.constants, self.rotations = [], [], []
        self.n_intermediates = 0

    def constant(self, fe_mont):
        fe_mont = np.asarray(fe_mont, np.uint64)
        for i, c in enumerate(self.constants):
            if (c == fe_mont).all():
                return (CONST, i, 0)
        self.constants.append(fe_mont)
        return (CONST, len(self.constants) - 1, 0)

    def rotation(self, rot):
        if rot not in self.rotations:
            self.rotations.append(rot)
        return self.rotations.index(rot)

    def column(self, idx, rot=0):
        return (COLUMN, idx, self.rotation(rot))

    def challenge(self, idx):
        return (CHALLENGE, idx, 0)

    def previous(self):
        return (PREVIOUS, 0, 0)

    def calc(self, op, s0, s1=(CONST, 0, 0), target=None):
        if target is None:
            target = self.n_intermediates
            self.n_intermediates += 1
        self.code.append([OPS[op], target, *s0, *s1])
        return (INTERMEDIATE, target, 0)

    def horner(self, start, parts, factor):
        t = self.calc("store", start)
        for p in parts:
            self.calc("horner_step", p, factor, target=t[1])
        return t

    def arrays(self):
        code = np.asarray(self.code, np.uint32).reshape(-1, 8)
        consts = np.asarray(self.constants, np.uint64).reshape(-1, 4)
        rots = np.asarray(self.rotations, np.int32)
        return code, consts, rots

    def row_sharded(self, log_world):
        """The same program for ONE row shard of a sweep split over 2^log_world ranks.  Returns
        (program, queries): every distinct (column, rotation) the code reads becomes a column of its own, read at rotation 0,
        and queries[j] = (column, row_shift) says which window of the original extended column the j-th new column is: rows
        [lo + row_shift, hi + row_shift) mod 2^ext_k for the shard [lo, hi) (dist.row_windows / reshard_columns_to_rows).
        The shard is a domain of 2^(ext_k - log_world) rows, so the existing sweep entry point runs it unchanged."""
        assert 0 <= log_world <= self.k
        step = 1 << (self.ext_k - self.k)
        sub = GraphProgram(self.k - log_world, self.ext_k - log_world)
        sub.constants, sub.n_intermediates, sub.rotations = list(self.constants), self.n_intermediates, [0]
        queries, index = [], {}

        def remap(t, idx, rot):
            if t != COLUMN:
                return [t, idx, rot]
            key = (idx, self.rotations[rot] * step)
            if key not in index:
                index[key] = len(queries)
                queries.append(key)
            return [COLUMN, index[key], 0]
        for op, target, t0, i0, r0, t1, i1, r1 in self.code:
            sub.code.append([op, target, *remap(t0, i0, r0), *remap(t1, i1, r1)])
        return sub, queries

    def check_compiles(self, n_columns):
        """host-only: lower the program to HIP source and compile it for gfx950 with hiprtc (no GPU needed)"""
        code, consts, rots = self.arrays()
        cols = (C.c_void_p * max(1, n_columns))()
        ch = np.zeros((1, 4), np.uint64)
        pr = _Prog(_p(code), code.shape[0], self.n_intermediates, _p(consts), consts.shape[0], _p(rots), rots.shape[0],
                   C.cast(cols, _vp), n_columns, _p(ch), 1, self.k, self.ext_k)
        _l.check(_l.load().ezkl_hip_eval_h_check(C.byref(pr)), "ezkl_hip_eval_h_check")

    def scheduled_code(self, n_columns):
        """host-only: the instruction order the library executes this program in (ezkl_hip_eval_h_schedule)"""
        code, consts, rots = self.arrays()
        cols = (C.c_void_p * max(1, n_columns))()
        # every entry point validates the whole program (operand indices included): declare as many challenges as the code reads
        n_chal = 1 + max([int(ins[3 + 3 * q]) for ins in code.tolist() for q in (0, 1) if ins[2 + 3 * q] == CHALLENGE] + [0])
        ch = np.zeros((n_chal, 4), np.uint64)
        pr = _Prog(_p(code), code.shape[0], self.n_intermediates, _p(consts), consts.shape[0], _p(rots), rots.shape[0],
                   C.cast(cols, _vp), n_columns, _p(ch), n_chal, self.k, self.ext_k)
        out = np.zeros_like(code)
        _l.check(_l.load().ezkl_hip_eval_h_schedule(C.byref(pr), out.ctypes.data_as(C.c_void_p)), "ezkl_hip_eval_h_schedule")
        return out

    def evaluate_h(self, column_ptrs, challenges, out_ptr, stream=None):
        """Run on device-resident columns (list of device pointers); out_ptr holds PreviousValue on entry."""
        code, consts, rots = self.arrays()
        ch = _fe(np.asarray(challenges, np.uint64).reshape(-1, 4))
        cols = (C.c_void_p * max(1, len(column_ptrs)))(*column_ptrs)
        pr = _Prog(_p(code), code.shape[0], self.n_intermediates, _p(consts), consts.shape[0], _p(rots), rots.shape[0],
                   C.cast(cols, _vp), len(column_ptrs), _p(ch), ch.shape[0], self.k, self.ext_k)
        _l.check(_l.load().ezkl_hip_eval_h_dev(C.byref(pr), _vp(out_ptr), _stream_ptr(stream)), "ezkl_hip_eval_h_dev")


def jit_stats():
    """(compiled by hiprtc, loaded from the disk cache, found in memory) sweep kernels of this process"""
    a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    _l.check(_l.load().ezkl_hip_eval_h_jit_stats(C.byref(a), C.byref(b), C.byref(c)), "ezkl_hip_eval_h_jit_stats")
    return int(a.value), int(b.value), int(c.value)


def last_kernel_ms(which):
    ms = C.c_float(0)
    _l.check(_l.load().ezkl_hip_last_kernel_ms(which.encode(), C.byref(ms)), "ezkl_hip_last_kernel_ms")
    return float(ms.value)


def kernel_ms_stats(which, reset=False):
    """(sum of device ms, number of regions) of every `which` region recorded since the last reset -- read AFTER a timed loop: the
    regions of back-to-back calls each have an event pair of their own (ezkl_hip_kernel_ms_stats)"""
    s, n = C.c_double(0), C.c_uint64(0)
    _l.check(_l.load().ezkl_hip_kernel_ms_stats(which.encode(), C.byref(s), C.byref(n), C.c_int(1 if reset else 0)), "ezkl_hip_kernel_ms_stats")
    return float(s.value), int(n.value)


def ubench(which):
    out = C.c_double(0)
    _l.check(_l.load().ezkl_hip_ubench(which.encode(), C.byref(out)), "ezkl_hip_ubench")
    return float(out.value)


# ---------------------------------------------------------------------------------------------------
# A13 helpers composed from the kernels above ([UPSTREAM] halo2 permutation::prover::commit
# and mv_lookup::prover::commit_grand_sum).  Everything stays resident; blinding rows / RNG stay with the caller.
_DELTA = pow(7, 1 << 28, _R)


def omega_powers_column(k):
    """device column X[i] = omega_k^i (built with an exclusive product scan of a constant column)"""
    n = 1 << k
    w = pow(EvaluationDomain.ROOT, 1 << (28 - k), _R)
    col = DeviceBuffer(n * 32)
    vec_fill(col.ptr, _to_mont(w), n)
    prefix_scan("mul", col.ptr, col.ptr, n, exclusive=True)
    return col


def permutation_grand_product(k, value_cols, sigma_cols, beta, gamma, omega_col=None, first_column_index=0, z0=None):
    """z[0] = z0 (1), z[i+1] = z[i] * prod_j (v_j[i] + beta*delta^(j0+j)*omega^i + gamma) / (v_j[i] + beta*sigma_j[i] + gamma)
    for one chunk of permutation columns.  value_cols / sigma_cols: lists of device pointers (n rows each).
    Returns a DeviceBuffer with z[0..n) (the caller overwrites the blinding rows)."""
    n = 1 << k
    m = len(value_cols)
    omega_col = omega_col or omega_powers_column(k)
    beta_i, gamma_i = _from_mont_int(beta), _from_mont_int(gamma)
    # denominators and numerators as two straight-line programs over [values..., sigmas..., X]
    cols = list(value_cols) + list(sigma_cols) + [omega_col.ptr]
    chal = [_to_mont(beta_i), _to_mont(gamma_i)] + [_to_mont(beta_i * pow(_DELTA, first_column_index + j, _R) % _R) for j in range(m)]
    den = GraphProgram(k, k)
    acc = None
    for j in range(m):
        t = den.calc("add", den.calc("add", den.calc("mul", den.challenge(0), den.column(m + j)), den.challenge(1)), den.column(j))
        acc = t if acc is None else den.calc("mul", acc, t)
    num = GraphProgram(k, k)
    accn = None
    for j in range(m):
        t = num.calc("add", num.calc("add", num.calc("mul", num.challenge(2 + j), num.column(2 * m)), num.challenge(1)), num.column(j))
        accn = t if accn is None else num.calc("mul", accn, t)
    d_den, d_num = DeviceBuffer(n * 32), DeviceBuffer(n * 32)
    den.evaluate_h(cols, chal, d_den.ptr)
    num.evaluate_h(cols, chal, d_num.ptr)
    batch_invert(d_den.ptr, n)
    vec_op("mul", d_num.ptr, d_den.ptr, d_num.ptr, n)                   # ratio[i]
    prefix_scan("mul", d_num.ptr, d_num.ptr, n, exclusive=True)          # z[i] = prod_{r<i} ratio[r]
    if z0 is not None:
        _l.check(_l.load().ezkl_hip_vec_scale_dev(_vp(d_num.ptr), _p(_fe(z0)), _vp(d_num.ptr), C.c_size_t(n), _vp(None)), "vec_scale")
    d_den.free()
    return d_num


def lookup_grand_sum(k, input_cols, table_col, m_col, beta):
    """phi[0] = 0, phi[i+1] = phi[i] + sum_j 1/(f_j[i] + beta) - m[i]/(t[i] + beta)   (mv-lookup / logUp running sum)"""
    n = 1 << k
    nin = len(input_cols)
    chal = [_fe(beta)]
    acc = DeviceBuffer.from_numpy(np.zeros((n, 4), np.uint64))
    tmp = DeviceBuffer(n * 32)
    shift = GraphProgram(k, k)
    shift.calc("add", shift.column(0), shift.challenge(0))
    for j in range(nin):
        shift.evaluate_h([input_cols[j]], chal, tmp.ptr)
        batch_invert(tmp.ptr, n)
        vec_op("add", acc.ptr, tmp.ptr, acc.ptr, n)
    shift.evaluate_h([table_col], chal, tmp.ptr)
    batch_invert(tmp.ptr, n)
    vec_op("mul", tmp.ptr, m_col, tmp.ptr, n)
    vec_op("sub", acc.ptr, tmp.ptr, acc.ptr, n)
    prefix_scan("add", acc.ptr, acc.ptr, n, exclusive=True)
    tmp.free()
    return acc


def _from_mont_int(a):
    return int.from_bytes(np.ascontiguousarray(a, np.uint64).tobytes(), "little") * pow(_MONT, -1, _R) % _R


# ---- multi-GPU: the library's RCCL communicator (csrc/comm.hip) --------------------------------------------------------------
def comm_unique_id():
    out = (C.c_uint8 * 128)()
    _l.check(_l.load().ezkl_hip_comm_unique_id(out), "ezkl_hip_comm_unique_id")
    return bytes(out)


def comm_init(unique_id, world, rank):
    _l.check(_l.load().ezkl_hip_comm_init(bytes(unique_id), C.c_int(world), C.c_int(rank)), "ezkl_hip_comm_init")


def mem_info():
    """(free, total) bytes of the calling context's device"""
    f, t = C.c_size_t(0), C.c_size_t(0)
    _l.check(_l.load().ezkl_hip_mem_info(C.byref(f), C.byref(t)), "ezkl_hip_mem_info")
    return int(f.value), int(t.value)


def pool_stats():
    """the calling context's column pool: dict(live, live_peak, parked, bound) in bytes (ezkl_hip_pool_stats)"""
    out = (C.c_size_t * 4)()
    _l.check(_l.load().ezkl_hip_pool_stats(out), "ezkl_hip_pool_stats")
    return dict(live=int(out[0]), live_peak=int(out[1]), parked=int(out[2]), bound=int(out[3]))


def pool_trim():
    _l.check(_l.load().ezkl_hip_pool_trim(), "ezkl_hip_pool_trim")


def contexts_configure(devices):
    """the context table of libezkl_hip.so, before its first use: context i on device devices[i] (several may share a device)"""
    arr = (C.c_int * len(devices))(*devices)
    _l.check(_l.load().ezkl_hip_contexts_configure(C.c_int(len(devices)), arr), "ezkl_hip_contexts_configure")


def context_count():
    return int(_l.load().ezkl_hip_context_count())


def set_context(index):
    _l.check(_l.load().ezkl_hip_set_context(C.c_int(index)), "ezkl_hip_set_context")


def comm_init_from_torch(dist, device):
    """rank 0 creates the id, torch.distributed (any backend) broadcasts its 128 bytes, every rank joins"""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == 0:
        t = torch.tensor(list(comm_unique_id()), dtype=torch.uint8, device=device)
    dist.broadcast(t, 0)
    comm_init(bytes(t.cpu().numpy().tobytes()), world, rank)


def comm_info():
    w, r = C.c_int(0), C.c_int(0)
    _l.check(_l.load().ezkl_hip_comm_info(C.byref(w), C.byref(r)), "ezkl_hip_comm_info")
    return w.value, r.value


def comm_destroy():
    _l.check(_l.load().ezkl_hip_comm_destroy(), "ezkl_hip_comm_destroy")


def comm_allgather_dev(ptr, total_bytes):
    _l.check(_l.load().ezkl_hip_comm_allgather_dev(_vp(ptr), C.c_size_t(total_bytes)), "ezkl_hip_comm_allgather_dev")


def comm_fold_points(points):
    """(count, 8) u64 partial sums -> sums over all ranks (in a copy)"""
    a = np.ascontiguousarray(points, np.uint64).copy()
    _l.check(_l.load().ezkl_hip_comm_fold_points(_p(a), C.c_uint32(a.shape[0])), "ezkl_hip_comm_fold_points")
    return a


class _CommSeg(C.Structure):
    _fields_ = [("peer", C.c_int), ("ptr", C.c_void_p), ("bytes", C.c_size_t)]


def comm_alltoallv_dev(sends, recvs):
    """sends / recvs: lists of (peer, device pointer, bytes); per peer the segments are ONE byte stream in list order (totals must agree)"""
    sa = (_CommSeg * max(1, len(sends)))(*[_CommSeg(p, ptr, n) for p, ptr, n in sends])
    ra = (_CommSeg * max(1, len(recvs)))(*[_CommSeg(p, ptr, n) for p, ptr, n in recvs])
    _l.check(_l.load().ezkl_hip_comm_alltoallv_dev(sa, C.c_size_t(len(sends)), ra, C.c_size_t(len(recvs))), "ezkl_hip_comm_alltoallv_dev")


def comm_selftest():
    """ezkl_hip_comm_selftest: the library communicator tries out every collective shape the prover uses (a collective: every rank calls
    it; ezkl_hip_comm_init already ran it unless EZKL_COMM_SELFTEST=0).  Raises EzklHipError with the failing status."""
    _l.check(_l.load().ezkl_hip_comm_selftest(), "ezkl_hip_comm_selftest")


def comm_available():
    """librccl loads and exports every entry point the library communicator binds (no device touched, nothing called)"""
    return _l.load().ezkl_hip_comm_available() == 0


def comm_stats(reset=False):
    """what this process's exchanges have moved (ezkl_hip_comm_stats): calls, bytes to / from other ranks, host seconds inside the calls,
    ncclSend / ncclRecv operations issued, rounds"""
    out = (C.c_uint64 * 8)()
    _l.check(_l.load().ezkl_hip_comm_stats(out, C.c_int(1 if reset else 0)), "ezkl_hip_comm_stats")
    return dict(exchanges=int(out[0]), bytes_sent=int(out[1]), bytes_received=int(out[2]), seconds=out[3] / 1e6, nccl_sends=int(out[4]),
                nccl_recvs=int(out[5]), rounds=int(out[6]))


def comm_allgather_host(arr):
    """in place on a (world, ...) numpy array: row r valid on rank r going in, every row coming out"""
    arr = np.ascontiguousarray(arr)
    world = comm_info()[0]
    _l.check(_l.load().ezkl_hip_comm_allgather_host(_p(arr), C.c_size_t(arr.nbytes // max(1, world))), "ezkl_hip_comm_allgather_host")
    return arr


def comm_alltoall_dev(send_ptr, send_off, send_len, recv_ptr, recv_off, recv_len):
    arr = lambda v: (C.c_size_t * len(v))(*[int(x) for x in v])
    _l.check(_l.load().ezkl_hip_comm_alltoall_dev(_vp(send_ptr), arr(send_off), arr(send_len), _vp(recv_ptr), arr(recv_off), arr(recv_len)),
             "ezkl_hip_comm_alltoall_dev")
