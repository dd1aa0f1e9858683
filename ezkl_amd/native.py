"""ctypes binding of the native host prover (libezkl_prover.so, include/ezkl_prover.h): halo2-shaped keygen and
create_proof written in C++ over the C ABI of libezkl_hip.so.  This module only serialises a `plonk.ConstraintSystem`
into the flat description the library parses and marshals pointers; it computes nothing itself."""
import ctypes as C
import os
import struct

import numpy as np

from . import lib as _l
from . import plonk as _pl

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

SYMBOLS = ["ezkl_prover_cs_parse", "ezkl_prover_cs_free", "ezkl_prover_cs_info", "ezkl_prover_cs_set_shard", "ezkl_prover_cs_set_shard_comm", "ezkl_prover_cs_set_shard_full_bases", "ezkl_prover_cs_set_advice_by_pointer", "ezkl_prover_cs_set_sweep_gather",
           "ezkl_prover_cs_sharded_sweeps", "ezkl_prover_cs_set_shard_exchange", "ezkl_prover_cs_shard_stats", "ezkl_prover_group_create", "ezkl_prover_group_size", "ezkl_prover_group_free",
           "ezkl_prover_group_load_srs", "ezkl_prover_group_keygen", "ezkl_prover_group_pk", "ezkl_prover_group_pk_read_file", "ezkl_prover_pk_residency", "ezkl_prover_group_create_proof", "ezkl_prover_keygen", "ezkl_prover_pk_free", "ezkl_prover_pk_sweep_stats", "ezkl_prover_pk_write", "ezkl_prover_pk_read", "ezkl_prover_pk_read_file", "ezkl_prover_pk_recommit", "ezkl_prover_pk_set_selectors", "ezkl_prover_pk_set_transcript_repr", "ezkl_prover_vk",
           "ezkl_prover_create_proof", "ezkl_prover_create_proof_fmt", "ezkl_prover_verify_proof", "ezkl_prover_verify_proof_vk", "ezkl_prover_g2_mul_generator", "ezkl_prover_keccak256", "ezkl_prover_last_error"]
ADVICE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p))
RNG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t)
FOLD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32)
GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t)
ALLGATHER_HOST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)


class CommSeg(C.Structure):
    _fields_ = [("peer", C.c_int), ("ptr", C.c_void_p), ("bytes", C.c_size_t)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(CommSeg), C.c_size_t, C.POINTER(CommSeg), C.c_size_t)
STAGES = ["advice_commit", "lookup_m", "permutation_z", "lookup_phi", "random_poly", "intt_and_coset_ntt", "quotient_sweep", "h_split_commit",
          "evaluations", "shplonk", "total"]


def lib_path():
    return os.environ.get("EZKL_PROVER_LIB") or os.path.join(_HERE, "libezkl_prover.so")      # override: A/B builds


def load():
    """Fails loudly: there is no Python fallback for the native prover."""
    global _lib
    if _lib is None:
        _l.load()                                    # libezkl_hip.so first (the prover links it)
        if not os.path.exists(lib_path()):
            raise RuntimeError("libezkl_prover.so missing at %s -- build it with `make -C ezkl_amd/csrc`" % lib_path())
        L = C.CDLL(lib_path())
        L.ezkl_prover_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, load().ezkl_prover_last_error().decode(), rc))


def keccak256(data):
    out = (C.c_uint8 * 32)()
    _check(load().ezkl_prover_keccak256(bytes(data), C.c_size_t(len(data)), out), "ezkl_prover_keccak256")
    return bytes(out)


serialize_cs = _pl.serialize_cs


def _ptr_array(arrays):
    arr = (C.c_void_p * max(1, len(arrays)))()
    for i, a in enumerate(arrays):
        arr[i] = a.ctypes.data
    return arr


class NativeCircuit:
    def __init__(self, cs):
        self.cs = cs
        blob = serialize_cs(cs)
        self.h = C.c_void_p()
        _check(load().ezkl_prover_cs_parse(blob, C.c_size_t(len(blob)), C.byref(self.h)), "ezkl_prover_cs_parse")

    def info(self):
        out = (C.c_uint32 * 8)()
        _check(load().ezkl_prover_cs_info(self.h, out), "ezkl_prover_cs_info")
        return dict(zip(["degree", "ext_k", "chunk", "n_chunks", "usable", "n_advice_queries", "n_fixed_queries", "n_instance_queries"], list(out)))

    def set_shard(self, dist, device):
        """Shard the MSMs of keygen / create_proof by points over the ranks of `dist` (torch.distributed, nccl = RCCL on the
        GPU box): afterwards the Bases handles passed to NativeProvingKey / create_proof must hold this rank's slice
        `shard_slice()` of the SRS.  Each commit batch costs one all_gather of 64-byte partials."""
        from . import dist as D
        world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        lo, hi = D.shard_range(self.cs.n, rank, world)

        def _fold(_user, pts, count):
            try:
                a = np.ctypeslib.as_array(C.cast(pts, C.POINTER(C.c_uint64)), shape=(count, 8))
                a[:] = D.fold_columns(D.allgather_points(a.copy(), dist, device))
                return 0
            except Exception:                         # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        def _gather(_user, buf, total, off, nbytes):
            try:
                D.allgather_device_rows(int(buf), int(total), int(off), int(nbytes), dist, device)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._fold = FOLD_FN(_fold)                   # keep the thunks alive as long as the handle
        self._gather = GATHER_FN(_gather)
        self._slice = (lo, hi)
        _check(load().ezkl_prover_cs_set_shard(self.h, C.c_uint32(lo), C.c_uint32(hi), self._fold, None), "ezkl_prover_cs_set_shard")
        if world > 1 and world & (world - 1) == 0 and self.cs.n % world == 0:      # the sweep by rows needs equal power-of-two slices
            self.direct_gather = D.probe_direct_gather(dist, device)               # RCCL on the library's pointers, verified once
            _check(load().ezkl_prover_cs_set_sweep_gather(self.h, self._gather, None), "ezkl_prover_cs_set_sweep_gather")
        return lo, hi

    def set_shard_exchange(self, dist, device):
        """columns and arguments by owner (ezkl_prover_cs_set_shard_exchange) with torch.distributed moving the data (host-staged):
        call after set_shard + set_shard_full_bases(True) on a power-of-two world"""
        from . import dist as D

        def _agh(_user, buf, per):
            try:
                D.allgather_host_bytes(int(buf), int(per), dist, device)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        def _xch(_user, sends, n_sends, recvs, n_recvs):
            try:
                D.exchange_segments([(sends[i].peer, int(sends[i].ptr or 0), int(sends[i].bytes)) for i in range(n_sends)],
                                    [(recvs[i].peer, int(recvs[i].ptr or 0), int(recvs[i].bytes)) for i in range(n_recvs)], dist, device)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._agh, self._xch = ALLGATHER_HOST_FN(_agh), EXCHANGE_FN(_xch)
        _check(load().ezkl_prover_cs_set_shard_exchange(self.h, self._agh, self._xch, None), "ezkl_prover_cs_set_shard_exchange")

    def shard_stats(self):
        """counters of the last create_proof on this rank"""
        out = (C.c_uint64 * 4)()
        _check(load().ezkl_prover_cs_shard_stats(self.h, out), "ezkl_prover_cs_shard_stats")
        return dict(zip(["columns_transformed_here", "witness_columns", "exchange_bytes_received", "arguments_computed_here"], [int(x) for x in out]))

    def set_shard_comm(self):
        """shard over the library's RCCL communicator (backend.comm_init first): returns this rank's slice of the SRS"""
        from . import backend as B, dist as D
        world, rank = B.comm_info()
        _check(load().ezkl_prover_cs_set_shard_comm(self.h), "ezkl_prover_cs_set_shard_comm")
        self._slice = D.shard_range(self.cs.n, rank, world)
        return self._slice

    def set_shard_full_bases(self, on=True):
        """every rank passes COMPLETE base sets: commit batches are divided by columns (point ranges only when columns < ranks)"""
        _check(load().ezkl_prover_cs_set_shard_full_bases(self.h, 1 if on else 0), "ezkl_prover_cs_set_shard_full_bases")

    def sharded_sweeps(self):
        out = C.c_uint64(0)
        _check(load().ezkl_prover_cs_sharded_sweeps(self.h, C.byref(out)), "ezkl_prover_cs_sharded_sweeps")
        return int(out.value)

    def shard_slice(self):
        return getattr(self, "_slice", (0, self.cs.n))

    def free(self):
        if self.h:
            load().ezkl_prover_cs_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _copies_array(copies):
    """copy constraints as the (count, 4) uint32 array ezkl_prover_keygen takes: from pairs ((column position, row), (column position, row)),
    or from anything with such an `.array` already (a circuit read back from disk: converting 10^7 pairs through Python tuples costs more
    than the keygen itself)"""
    if hasattr(copies, "array"):
        return np.ascontiguousarray(copies.array, np.uint32).reshape(-1, 4)
    return np.ascontiguousarray(np.array([[a[0], a[1], b[0], b[1]] for a, b in copies], np.uint32).reshape(-1, 4))


class NativeProvingKey:
    """keygen_vk + keygen_pk on the GPU; `g` is a backend.Bases handle of the coefficient-basis SRS"""

    def __init__(self, circuit, g, fixed_values, copies):
        self.circuit = circuit
        fixed = [np.ascontiguousarray(v, np.uint64) for v in fixed_values]
        cp = _copies_array(copies)
        self.h = C.c_void_p()
        _check(load().ezkl_prover_keygen(circuit.h, g.h, _ptr_array(fixed), cp.ctypes.data_as(C.c_void_p), C.c_size_t(cp.shape[0]), C.byref(self.h)),
               "ezkl_prover_keygen")

    @classmethod
    def from_bytes(cls, circuit, data, recommit=None):
        """load a key in halo2's raw-bytes pk.key layout (load_pk): columns go straight to HBM.  recommit = a Bases handle: commit
        the fixed / permutation polynomials again under that SRS (a key file made under another SRS)"""
        self = cls.__new__(cls)
        self.circuit = circuit
        self.h = C.c_void_p()
        data = bytes(data)
        _check(load().ezkl_prover_pk_read(circuit.h, data, C.c_size_t(len(data)), C.byref(self.h)), "ezkl_prover_pk_read")
        if recommit is not None:
            _check(load().ezkl_prover_pk_recommit(self.h, recommit.h), "ezkl_prover_pk_recommit")
        return self

    def residency(self):
        """dict(first_coset, cosets, E, key_bytes, streamed): what of the extended key columns this key holds in HBM (ezkl_prover_pk_residency);
        streamed (cosets == 0): the degraded mode -- values + coefficients only, the sweep rebuilds each coset (EZKL_KEY_COSETS)"""
        out = (C.c_uint64 * 4)()
        _check(load().ezkl_prover_pk_residency(self.h, out), "ezkl_prover_pk_residency")
        return dict(first_coset=int(out[0]), cosets=int(out[1]), E=int(out[2]), key_bytes=int(out[3]), streamed=int(out[1]) == 0)

    def sweep_stats(self):
        """per extended row of this key's quotient sweep: (instructions, Montgomery products, column slots, kernels)"""
        out = (C.c_uint64 * 4)()
        _check(load().ezkl_prover_pk_sweep_stats(self.h, out), "ezkl_prover_pk_sweep_stats")
        return tuple(int(x) for x in out)

    @classmethod
    def from_file(cls, circuit, path, recommit=None):
        """load_pk of a one-shot prover: the file is mapped and only its n-row sections are uploaded (ezkl_prover_pk_read_file)"""
        self = cls.__new__(cls)
        self.circuit = circuit
        self.h = C.c_void_p()
        _check(load().ezkl_prover_pk_read_file(circuit.h, os.fsencode(path), C.byref(self.h)), "ezkl_prover_pk_read_file")
        if recommit is not None:
            _check(load().ezkl_prover_pk_recommit(self.h, recommit.h), "ezkl_prover_pk_recommit")
        return self

    def set_transcript_repr(self, value):
        """the scalar that heads every transcript of this key (a halo2 fork passes vk.transcript_repr)"""
        _check(load().ezkl_prover_pk_set_transcript_repr(self.h, int(value).to_bytes(32, "little")), "ezkl_prover_pk_set_transcript_repr")

    def set_selectors(self, activations):
        """activations: (n_selectors, n) booleans -> the selector section of the key file"""
        bits = np.packbits(np.asarray(activations, bool), axis=1, bitorder="little").tobytes()
        _check(load().ezkl_prover_pk_set_selectors(self.h, bits, C.c_size_t(len(bits))), "ezkl_prover_pk_set_selectors")

    def to_bytes(self):
        """the key in halo2's raw-bytes pk.key layout (save_pk)"""
        ln = C.c_size_t(0)
        load().ezkl_prover_pk_write(self.h, None, C.c_size_t(0), C.byref(ln))          # first call: the size
        buf = (C.c_uint8 * ln.value)()
        _check(load().ezkl_prover_pk_write(self.h, buf, C.c_size_t(ln.value), C.byref(ln)), "ezkl_prover_pk_write")
        return bytes(buf)

    def vk(self):
        """(fixed commitments (F,8) u64, permutation commitments (P,8) u64, digest as a canonical int)"""
        cs = self.circuit.cs
        fc, pc, dg = np.zeros((cs.n_fixed, 8), np.uint64), np.zeros((len(cs.perm), 8), np.uint64), np.zeros(4, np.uint64)
        _check(load().ezkl_prover_vk(self.h, fc.ctypes.data_as(C.c_void_p), pc.ctypes.data_as(C.c_void_p), dg.ctypes.data_as(C.c_void_p)), "ezkl_prover_vk")
        return fc, pc, _pl.from_mont(dg)

    def free(self):
        if self.h:
            load().ezkl_prover_pk_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _BorrowedKey:
    """the key of one context of a group (owned by the group): enough of NativeProvingKey for vk() and verify_proof"""

    def __init__(self, circuit, h):
        self.circuit, self.h = circuit, h

    vk = NativeProvingKey.vk
    residency = NativeProvingKey.residency


class NativeGroup:
    """One process, several GPUs (include/ezkl_prover.h "the prover group"): the owner-mode prover on n contexts of libezkl_hip.so, one
    host thread per context inside this process.  backend.contexts_configure / ezkl_hip_init(-1) decide what the contexts are."""

    def __init__(self, cs, n_contexts):
        self.cs = cs
        self.circuit = type("C", (), {"cs": cs})()
        blob = serialize_cs(cs)
        self.h = C.c_void_p()
        _check(load().ezkl_prover_group_create(blob, C.c_size_t(len(blob)), C.c_int(n_contexts), C.byref(self.h)), "ezkl_prover_group_create")
        self.world = int(load().ezkl_prover_group_size(self.h))

    def load_srs(self, g, g_lagrange):
        g, gl = np.ascontiguousarray(g, np.uint64), np.ascontiguousarray(g_lagrange, np.uint64)
        assert g.shape == gl.shape
        _check(load().ezkl_prover_group_load_srs(self.h, g.ctypes.data_as(C.c_void_p), gl.ctypes.data_as(C.c_void_p), C.c_size_t(g.shape[0])), "ezkl_prover_group_load_srs")

    def keygen(self, fixed_values, copies):
        fixed = [np.ascontiguousarray(v, np.uint64) for v in fixed_values]
        cp = _copies_array(copies)
        _check(load().ezkl_prover_group_keygen(self.h, _ptr_array(fixed), cp.ctypes.data_as(C.c_void_p), C.c_size_t(cp.shape[0])), "ezkl_prover_group_keygen")

    def pk_read_file(self, path, recommit=False):
        """load_pk for the group: every context reads the key file on its own thread and keeps only the cosets it sweeps"""
        _check(load().ezkl_prover_group_pk_read_file(self.h, os.fsencode(path), C.c_int(1 if recommit else 0)), "ezkl_prover_group_pk_read_file")

    def pk(self, context=0):
        h = C.c_void_p()
        _check(load().ezkl_prover_group_pk(self.h, C.c_int(context), C.byref(h)), "ezkl_prover_group_pk")
        return _BorrowedKey(self.circuit, h)

    def create_proof(self, advice_values, seed=0, instances=(), timings=None, stats=None):
        keep = [np.ascontiguousarray(a, np.uint64) for a in advice_values]
        inst = [np.stack([_pl.to_mont(v) for v in vals]) if len(vals) else np.zeros((0, 4), np.uint64) for vals in instances]
        lens = (C.c_uint32 * max(1, len(inst)))(*[a.shape[0] for a in inst])
        cap = 1 << 20
        buf = (C.c_uint8 * cap)()
        plen = C.c_size_t(0)
        tm = (C.c_double * 12)()
        st = (C.c_uint64 * (4 * self.world))()
        _check(load().ezkl_prover_group_create_proof(self.h, _ptr_array(keep), _ptr_array(inst), lens, C.c_uint64(seed), buf, C.c_size_t(cap), C.byref(plen), tm, st),
               "ezkl_prover_group_create_proof")
        if timings is not None:
            timings.update(dict(zip(STAGES, list(tm))))
        if stats is not None:
            names = ["columns_transformed_here", "witness_columns", "exchange_bytes_received", "arguments_computed_here"]
            stats.extend(dict(zip(names, [int(st[4 * r + i]) for i in range(4)])) for r in range(self.world))
        return bytes(buf[:plen.value])

    def free(self):
        if self.h:
            load().ezkl_prover_group_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def g2_mul_generator(s):
    """[s] G2 as the 128 raw bytes of an SRS file (s = 1: the generator)"""
    out = (C.c_uint8 * 128)()
    _check(load().ezkl_prover_g2_mul_generator(_pl.to_mont(s).ctypes.data_as(C.c_void_p), out), "ezkl_prover_g2_mul_generator")
    return bytes(out)


def verify_proof(pk, g2, s_g2, proof, instances=()):
    """verify_proof_circuit (src/pfsys/mod.rs:557-590) on the host side of the library: g2 / s_g2 = the 128-byte tail elements of the SRS"""
    inst = [np.stack([_pl.to_mont(v) for v in vals]) if len(vals) else np.zeros((0, 4), np.uint64) for vals in instances]
    lens = (C.c_uint32 * max(1, len(inst)))(*[a.shape[0] for a in inst])
    ok = C.c_int(0)
    proof = bytes(proof)
    _check(load().ezkl_prover_verify_proof(pk.h, bytes(g2), bytes(s_g2), proof, C.c_size_t(len(proof)), _ptr_array(inst), lens, C.byref(ok)),
           "ezkl_prover_verify_proof")
    return bool(ok.value)


def verify_proof_vk(circuit, vk_bytes, g2, s_g2, proof, instances=()):
    """verify from vk.key alone (the reference's `verify`: settings + vk, src/execute.rs:1651): host only, no GPU, no proving key"""
    inst = [np.stack([_pl.to_mont(v) for v in vals]) if len(vals) else np.zeros((0, 4), np.uint64) for vals in instances]
    lens = (C.c_uint32 * max(1, len(inst)))(*[a.shape[0] for a in inst])
    ok = C.c_int(0)
    vk_bytes, proof = bytes(vk_bytes), bytes(proof)
    _check(load().ezkl_prover_verify_proof_vk(circuit.h, vk_bytes, C.c_size_t(len(vk_bytes)), bytes(g2), bytes(s_g2), proof, C.c_size_t(len(proof)),
                                              _ptr_array(inst), lens, C.byref(ok)), "ezkl_prover_verify_proof_vk")
    return bool(ok.value)


def create_proof(pk, g, g_lagrange, advice_values, rng=None, seed=0, instances=(), timings=None, check_mode="UNSAFE", g2=None, s_g2=None):
    """advice_values: list of columns, or a callable advice_values(phase, challenges) -> {column: array} (second-phase advice).  A column
    is an (n, 4) uint64 array of Montgomery words (halo2's Fp), or the INTEGERS its cells were made from (ezkl's IntegerRep,
    src/fieldutils.rs:6-17): an (n,) int64 array, or an (n, 2) uint64 array of little-endian two's-complement 128-bit values -- 8 / 16
    bytes per cell across PCIe instead of 32, expanded on the device (ezkl_prover_create_proof_fmt); the proof bytes are the same.  rng: object with .vec(m) -> (m,4) u64 Montgomery residues (None = the library's own
    generator, seeded with `seed`, 0 = OS entropy); instances: list of lists of ints.  Returns the proof bytes."""
    cs = pk.circuit.cs
    n = cs.n
    keep = []
    adv_arr, adv_cb, adv_fmt = None, C.cast(None, ADVICE_FN), None
    if callable(advice_values):
        _check(load().ezkl_prover_cs_set_advice_by_pointer(pk.circuit.h, 1), "ezkl_prover_cs_set_advice_by_pointer")
        def _cb(_user, phase, chal_ptr, n_chal, cols_ptr):
            try:
                ch = np.ctypeslib.as_array(C.cast(chal_ptr, C.POINTER(C.c_uint64)), shape=(n_chal, 4)) if n_chal else np.zeros((0, 4), np.uint64)
                vals = advice_values(int(phase), [_pl.from_mont(c) for c in ch])
                for c, a in vals.items():
                    if cs.advice_phase[c] != phase:
                        continue
                    a = np.ascontiguousarray(a, np.uint64)
                    assert a.size == 4 * n
                    keep.append(a)                    # by pointer: the array must outlive the call
                    cols_ptr[c] = a.ctypes.data
                return 0
            except Exception:                         # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        adv_cb = ADVICE_FN(_cb)
    else:
        fmts = []
        for a in advice_values:
            a = np.asarray(a)
            # the format follows from dtype AND shape; anything else is refused rather than reinterpreted (a mis-shaped Montgomery
            # column taken for integers would prove a different witness)
            if a.dtype == np.int64 and a.ndim == 1:
                fmts.append(1); a = np.ascontiguousarray(a)
            elif a.dtype == np.uint64 and a.ndim == 2 and a.shape[1] == 2:
                fmts.append(2); a = np.ascontiguousarray(a)
            elif a.dtype == np.uint64 and a.size == 4 * n and (a.ndim == 1 or (a.ndim == 2 and a.shape[1] == 4)):
                fmts.append(0); a = np.ascontiguousarray(a).reshape(n, 4)
            else:
                raise ValueError("advice column %d: expected (n, 4) uint64 Montgomery words, (n,) int64 or (n, 2) uint64 IntegerRep values with n = %d; got %s %s"
                                 % (len(fmts), n, a.dtype, a.shape))
            if a.shape[0] != n:
                raise ValueError("advice column %d has %d rows, the circuit has %d" % (len(fmts), a.shape[0], n))
            keep.append(a)
        adv_arr = _ptr_array(keep)
        if any(fmts):
            adv_fmt = (C.c_uint8 * len(fmts))(*fmts)
    rng_cb = C.cast(None, RNG_FN)
    if rng is not None:
        def _rng(_user, out_ptr, m):
            a = np.ascontiguousarray(rng.vec(int(m)), np.uint64)
            C.memmove(out_ptr, a.ctypes.data, 32 * int(m))
        rng_cb = RNG_FN(_rng)
    inst = [np.stack([_pl.to_mont(v) for v in vals]) if len(vals) else np.zeros((0, 4), np.uint64) for vals in instances]
    lens = (C.c_uint32 * max(1, len(inst)))(*[a.shape[0] for a in inst])
    cap = 1 << 20
    buf = (C.c_uint8 * cap)()
    plen = C.c_size_t(0)
    tm = (C.c_double * 12)()
    _check(load().ezkl_prover_create_proof_fmt(pk.h, g.h, g_lagrange.h, adv_arr, adv_fmt, adv_cb, None, _ptr_array(inst), lens, rng_cb, None, C.c_uint64(seed),
                                               buf, C.c_size_t(cap), C.byref(plen), tm), "ezkl_prover_create_proof_fmt")
    if timings is not None:
        timings.update(dict(zip(STAGES, list(tm))))
    proof = bytes(buf[:plen.value])
    if check_mode == "SAFE":                          # create_proof_circuit's CheckMode::SAFE (src/pfsys/mod.rs:470-480): verify what was just proved
        if g2 is None or s_g2 is None:
            raise ValueError("check_mode SAFE needs the SRS's g2 / s_g2")
        if not verify_proof(pk, g2, s_g2, proof, instances):
            raise RuntimeError("SAFE check failed: the proof just made does not verify")
    return proof
