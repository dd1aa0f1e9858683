"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
All field elements are numpy uint64 arrays of shape (..., 4): Montgomery form, little-endian limbs --
byte-identical to halo2's raw-bytes encoding (SURVEY.md §8(b) representation contract)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _load():
    if not os.path.exists(_SO):
        build()
    return C.CDLL(_SO)


lib = _load()


def effective_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a container that sees
    256 hardware threads but has a 16-CPU quota must not run 256 OpenMP threads: they would be throttled)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


if "OMP_NUM_THREADS" not in os.environ:
    lib.oracle_set_threads(C.c_int(effective_cpus()))
_p = C.c_void_p


def _ptr(a):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_p)


def _fe(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.shape[-1] == 4
    return a


def fr_mul(a, b):
    a, b = _fe(a), _fe(b); o = np.empty(4, np.uint64); lib.oracle_fr_mul(_ptr(a), _ptr(b), _ptr(o)); return o


def fq_mul(a, b):
    a, b = _fe(a), _fe(b); o = np.empty(4, np.uint64); lib.oracle_fq_mul(_ptr(a), _ptr(b), _ptr(o)); return o


def fr_inv(a):
    a = _fe(a); o = np.empty(4, np.uint64); lib.oracle_fr_inv(_ptr(a), _ptr(o)); return o


def omega(k):
    o = np.empty(4, np.uint64); lib.oracle_omega(C.c_uint(k), _ptr(o)); return o


def g1_mul(p, s):
    p, s = np.ascontiguousarray(p, np.uint64), _fe(s); o = np.empty(8, np.uint64)
    lib.oracle_g1_mul(_ptr(p), _ptr(s), _ptr(o)); return o


def g1_add(p, q):
    p, q = np.ascontiguousarray(p, np.uint64), np.ascontiguousarray(q, np.uint64); o = np.empty(8, np.uint64)
    lib.oracle_g1_add_affine(_ptr(p), _ptr(q), _ptr(o)); return o


def g1_on_curve(p):
    p = np.ascontiguousarray(p, np.uint64); return bool(lib.oracle_g1_on_curve(_ptr(p)))


def msm(scalars, bases, naive=False):
    """scalars (n,4) u64 Montgomery Fr; bases (n,8) u64 affine Montgomery Fq -> (8,) affine"""
    s, b = _fe(scalars), np.ascontiguousarray(bases, np.uint64)
    n = s.shape[0]; assert b.shape == (n, 8)
    o = np.empty(8, np.uint64)
    (lib.oracle_msm_naive if naive else lib.oracle_msm)(_ptr(s), _ptr(b), C.c_size_t(n), _ptr(o))
    return o


def g1_to_lagrange(g, k):
    """ParamsKZG::downsize's g_to_lagrange, restated naively (halo2: an FFT with omega^-1 over the projective points, scaled by 1 / n;
    called through /root/reference/src/execute.rs:1739-1750): g_lagrange[i] = sum_j c_ij g[j] with c_i the coefficients of the i-th
    Lagrange basis polynomial of the 2^k-point domain -- one oracle MSM per output point, O(n^2): for the small sizes the tests compare
    the device's inverse NTT over G1 against (ezkl_hip_bases_downsize)."""
    n = 1 << k
    g = np.ascontiguousarray(g[:n], np.uint64)
    one = np.frombuffer(((1 << 256) % 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001).to_bytes(32, "little"), np.uint64)
    out = np.empty((n, 8), np.uint64)
    for i in range(n):
        e = np.zeros((n, 4), np.uint64); e[i] = one
        out[i] = msm(lagrange_to_coeff(e, k) if k else e, g)
    return out


def fft(a, log_n, omega_mont):
    a = _fe(a).copy(); assert a.shape == (1 << log_n, 4)
    w = _fe(omega_mont)
    lib.oracle_fft(_ptr(a), C.c_uint(log_n), _ptr(w)); return a


def lagrange_to_coeff(a, k):
    a = _fe(a).copy(); lib.oracle_lagrange_to_coeff(_ptr(a), C.c_uint(k)); return a


def coeff_to_lagrange(a, k):
    a = _fe(a).copy(); lib.oracle_coeff_to_lagrange(_ptr(a), C.c_uint(k)); return a


def coeff_to_extended(a, k, ext_k):
    a = _fe(a); o = np.empty((1 << ext_k, 4), np.uint64)
    lib.oracle_coeff_to_extended(_ptr(a), C.c_uint(k), C.c_uint(ext_k), _ptr(o)); return o


def extended_to_coeff(a, ext_k):
    a = _fe(a).copy(); lib.oracle_extended_to_coeff(_ptr(a), C.c_uint(ext_k)); return a


def divide_by_vanishing(a, k, ext_k):
    a = _fe(a).copy(); lib.oracle_divide_by_vanishing(_ptr(a), C.c_uint(k), C.c_uint(ext_k)); return a


def batch_invert(a):
    a = _fe(a).copy(); lib.oracle_batch_invert(_ptr(a), C.c_size_t(a.shape[0])); return a


def kate_div(a, z):
    a, z = _fe(a).copy(), _fe(z)
    lib.oracle_kate_div(_ptr(a), C.c_size_t(a.shape[0]), _ptr(z)); return a


def vec_add(a, b):
    a, b = _fe(a), _fe(b); o = np.empty_like(a)
    lib.oracle_vec_add(_ptr(a), _ptr(b), _ptr(o), C.c_size_t(a.shape[0])); return o


def vec_scale(a, s):
    a, s = _fe(a), _fe(s); o = np.empty_like(a)
    lib.oracle_vec_scale(_ptr(a), _ptr(s), _ptr(o), C.c_size_t(a.shape[0])); return o


def eval_poly(c, x):
    c, x = _fe(c), _fe(x); o = np.empty(4, np.uint64)
    lib.oracle_eval_poly(_ptr(c), C.c_size_t(c.shape[0]), _ptr(x), _ptr(o)); return o


def prefix_scan(a, op, exclusive=False):
    a = _fe(a); o = np.empty_like(a)
    lib.oracle_prefix_scan(_ptr(a), _ptr(o), C.c_size_t(a.shape[0]), C.c_int({"add": 0, "mul": 2}[op]), C.c_int(1 if exclusive else 0))
    return o


def vec_op(name, a, b):
    a, b = _fe(a), _fe(b); o = np.empty_like(a)
    getattr(lib, "oracle_vec_" + name)(_ptr(a), _ptr(b), _ptr(o), C.c_size_t(a.shape[0])); return o


class _Prog(C.Structure):
    _fields_ = [("code", _p), ("n_instr", C.c_uint32), ("n_intermediates", C.c_uint32),
                ("constants", _p), ("n_constants", C.c_uint32),
                ("rotations", _p), ("n_rotations", C.c_uint32),
                ("columns", _p), ("n_columns", C.c_uint32),
                ("challenges", _p), ("n_challenges", C.c_uint32),
                ("k", C.c_uint32), ("ext_k", C.c_uint32)]


def eval_program(code, n_intermediates, constants, rotations, columns, challenges, k, ext_k, previous=None):
    code = np.ascontiguousarray(code, np.uint32).reshape(-1, 8)
    constants = _fe(np.asarray(constants, np.uint64).reshape(-1, 4))
    challenges = _fe(np.asarray(challenges, np.uint64).reshape(-1, 4))
    rotations = np.ascontiguousarray(rotations, np.int32)
    cols = [_fe(c) for c in columns]
    ne = 1 << ext_k
    for c in cols:
        assert c.shape == (ne, 4)
    ptrs = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data for c in cols])
    out = np.zeros((ne, 4), np.uint64) if previous is None else _fe(previous).copy()
    pr = _Prog(_ptr(code), code.shape[0], n_intermediates, _ptr(constants), constants.shape[0],
               _ptr(rotations), rotations.shape[0], C.cast(ptrs, _p), len(cols),
               _ptr(challenges), challenges.shape[0], k, ext_k)
    lib.oracle_eval_program(C.byref(pr), _ptr(out))
    return out


def num_threads():
    return int(lib.oracle_num_threads())


def chacha20_block(key32, counter, stream):
    key = np.frombuffer(bytes(key32), np.uint32).copy(); out = np.zeros(16, np.uint32)
    lib.oracle_chacha20_block(_ptr(key), C.c_uint64(counter), C.c_uint64(stream), _ptr(out)); return out


def chacha20_fr(key32, stream, n, first=0):
    key = np.frombuffer(bytes(key32), np.uint32).copy(); out = np.zeros((n, 4), np.uint64)
    lib.oracle_chacha20_fr(_ptr(key), C.c_uint64(stream), C.c_size_t(first), C.c_size_t(n), _ptr(out)); return out


def gen_bases(seed, n, first=0):
    o = np.empty((n, 8), np.uint64)
    lib.oracle_gen_bases(C.c_uint64(seed), C.c_size_t(first), C.c_size_t(n), _ptr(o)); return o
