"""The ezkl circuits the end-to-end prove benchmarks run (built by ezkl_amd/ezkl_circuit.py + ezkl_layout.py):

  einsum : the reference's own criterion bench circuit /root/reference/benches/accum_einsum_matmul.rs ("ij,jk->ik" through
           configure_einsums: Freivalds' argument, 3 first-phase + 3 second-phase advice columns, 2 challenges), raised from its
           k = 16 / len 128 to k = 20 / len 512 (BASELINE configs[3]: reduction_length 787 456 of the 1 048 570 usable rows)
  mlp    : `layers` x (Gemm N x N + bias + ReLU) on one input vector, the op family of the reference's fixture model and of
           examples/onnx/large_mlp (gen.py:6-43), private input / parameters, public output, decomposition range checks
           (base 16384, 2 legs = ezkl's defaults, src/lib.rs:257-260): BaseConfig gates + range-check lookups + permutation
  conv   : BASELINE configs[2], /root/reference/examples/conv2d_mnist/main.rs: its Config (3 one-column advice VarTensors, range checks,
           the Div{32} static lookup over (-32768, 32768) that forces k = 17) and its layout (conv as dot products, ReLU by
           decomposition, table lookup, linear layer, outputs on the instance column); synthetic image / parameters
"""
import os
import sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from ezkl_amd import ezkl_layout as EL, plonk as P      # noqa: E402


def sparse_weights(rng, n_out, n_in, nnz=4):
    """mostly-zero +-1 weights: activations stay far inside the 2-leg decomposition range through many layers (ezkl would rescale
    between layers; the zero weights are still private advice cells, so the circuit is as large as with dense weights)"""
    W = np.zeros((n_out, n_in), np.int64)
    for i in range(n_out):
        idx = rng.choice(n_in, min(nnz, n_in), replace=False)
        W[i, idx] = rng.choice([-1, 1], len(idx))
    return W.tolist()


# ---- on-disk cache of laid-out circuits -----------------------------------------------------------------------------
# The Python layout engine is test / bench scaffolding (witness synthesis is ezkl's Rust `Model::layout` in a fork) and takes 35-75 s
# for the k = 20 MLP.  A laid-out circuit is therefore written ONCE -- constraint system, copy constraints, instances and every fixed /
# advice column as canonical values in the narrowest integer type that holds them -- and read back by every later process: the CPU
# prover child of bench.py, the N ranks of `bench.py --gpus N` (rank 0 of a node lays out, the others wait for the file), the
# profiling tools.  EZKL_BENCH_CACHE=<dir> (default <repo>/bench_cache, which travels to the GPU box with the snapshot), `off` disables.
def _cache_dir():
    d = os.environ.get("EZKL_BENCH_CACHE", os.path.join(ROOT, "bench_cache"))
    return None if d in ("off", "0", "") else d


def _pack_col(col):
    """canonical field elements (ints mod r / small signed numpy ints) -> int32 / int64 signed values, or (n, 4) u64 limbs"""
    if isinstance(col, np.ndarray) and col.dtype != object:
        v = col.astype(np.int64)
    else:
        half, R_ = P.R >> 1, P.R
        signed = [x if x <= half else x - R_ for x in col]              # plain Python: ~0.1 us per element, no object arrays
        try:
            v = np.array(signed, dtype=np.int64)
        except OverflowError:
            return EL.ints_to_limbs(col)
        if len(v) and (v.max() >= (1 << 62) or v.min() <= -(1 << 62)):
            return EL.ints_to_limbs(col)
    if len(v) and np.abs(v).max() < (1 << 31):
        return v.astype(np.int32)
    return v


def _unpack_cols(cols, gpu, pinned=False):
    """packed columns -> (n, 4) u64 Montgomery arrays (the conversion runs on the device when a backend is given)"""
    out = []
    for c in cols:
        if c.ndim == 2:                              # canonical limbs
            if gpu is None:
                out.append(EL.ints_to_mont([int.from_bytes(r.tobytes(), "little") for r in c]))
                continue
            limbs = c
        else:
            limbs = EL.ints_to_limbs(c.astype(np.int64))
            if gpu is None:
                out.append(EL.ints_to_mont([int(v) for v in c]))
                continue
        r2 = P.to_mont((1 << 256) % P.R)
        buf = gpu.DeviceBuffer.from_numpy(np.ascontiguousarray(limbs))
        gpu.vec_scale(buf.ptr, r2, buf.ptr, len(c))
        a = buf.to_numpy(shape=(len(c), 4))
        if pinned:
            pa = gpu.PinnedArray((len(c), 4))
            pa.array[:] = a
            EL._PINNED.append(pa)
            a = pa.array
        else:
            a = a.copy()
        out.append(a)
    return out


def _limbs_to_int64(a):
    """a column stored as canonical (n, 4) u64 limbs -> the int64 values it stands for (negatives are r - |x|), or None if a cell needs more
    than 64 bits (such columns were stored as limbs because SOME cell of theirs did at pack time, or held r - x for a tiny x)"""
    small = (a[:, 1:] == 0).all(axis=1) & (a[:, 0] < np.uint64(1 << 63))
    out = np.zeros(a.shape[0], np.int64)
    out[small] = a[small, 0].astype(np.int64)
    rest = np.nonzero(~small)[0]
    if len(rest) > (1 << 18):                        # mostly wide values: not an integer column
        return None
    for i in rest.tolist():
        v = int.from_bytes(a[i].tobytes(), "little")
        v = v if v <= (P.R >> 1) else v - P.R
        if not -(1 << 63) <= v < (1 << 63):
            return None
        out[i] = v
    return out


def _cache_load(path, gpu):
    """plain data only: the constraint system as its EZCS blob (plonk.deserialize_cs), everything else as JSON -- the cache directory
    travels to the GPU box, so nothing in it is unpickled"""
    import json
    with np.load(path, allow_pickle=False) as z:
        meta = json.loads(z["meta"].tobytes().decode())
        cs = P.deserialize_cs(z["cs"].tobytes())
        fixed = _unpack_cols([z["f%d" % i] for i in range(meta["n_fixed"])], gpu)
        raw = [z["a%d" % i] for i in range(meta["n_advice"])]
        advice = _unpack_cols(raw, gpu)
        copies = CopyPairs(z["copies"])
    instances = [[int(v) for v in col] for col in meta["instances"]]
    # the same advice columns as the integers they were laid out as (ezkl's IntegerRep before integer_rep_to_felt), for
    # ezkl_prover_create_proof_fmt: None where a column holds values beyond 64 bits
    advice_int = [c.astype(np.int64) if c.ndim == 1 else _limbs_to_int64(c) for c in raw]
    return dict(cs=cs, fixed=fixed, copies=copies, advice=advice, advice_int=advice_int, instances=instances,
                info=dict(meta["info"], layout="read from " + os.path.basename(path)))


class CopyPairs:
    """the copy constraints of a stored circuit: iterates as ((column position, row), (column position, row)) pairs like the layout engine's
    list, and hands the (count, 4) uint32 array itself to consumers that can take it (ezkl_amd.native.NativeProvingKey)"""

    def __init__(self, array):
        self.array = np.ascontiguousarray(array, np.uint32).reshape(-1, 4)

    def __len__(self):
        return int(self.array.shape[0])

    def __iter__(self):
        for a, b, c, d in self.array.tolist():
            yield ((a, b), (c, d))


def _cache_store(path, cs, fixed_raw, copies, adv_raw, instances, info):
    import json
    meta = dict(instances=[[str(int(v)) for v in col] for col in instances], info=info, n_fixed=len(fixed_raw), n_advice=len(adv_raw))
    arrs = {"meta": np.frombuffer(json.dumps(meta).encode(), np.uint8), "cs": np.frombuffer(P.serialize_cs(cs), np.uint8),
            "copies": copies.array if isinstance(copies, CopyPairs) else np.array([[a[0], a[1], b[0], b[1]] for a, b in copies], np.uint32).reshape(-1, 4)}
    for i, c in enumerate(fixed_raw):
        arrs["f%d" % i] = _pack_col(c)
    for i, c in enumerate(adv_raw):
        arrs["a%d" % i] = _pack_col(c)
    tmp = path + ".tmp%d.npz" % os.getpid()
    np.savez_compressed(tmp, **arrs)
    os.replace(tmp, path)


def build(kind, k, gpu=None, seed=1, **kw):
    """-> dict(cs, fixed (Montgomery arrays), copies, advice (list of arrays, or callable(phase, challenges)), instances, info)"""
    d = _cache_dir()
    if d is None or kind in ("einsum", "transformer"):   # the second-phase witness depends on the proof's challenges: laid out per proof
        return _build(kind, k, gpu, seed, None, **kw)
    os.makedirs(d, exist_ok=True)
    tag = "_".join([kind, "k%d" % k, "s%d" % seed] + ["%s%s" % (a, kw[a]) for a in sorted(kw) if kw[a] is not None])
    path = os.path.join(d, tag + ".npz")
    lock = path + ".lock"
    import time
    while not os.path.exists(path):
        try:
            fd = os.open(lock, os.O_CREAT | os.O_EXCL | os.O_WRONLY)
        except FileExistsError:                      # another process (rank) is laying the circuit out: wait for its file
            try:
                if time.time() - os.path.getmtime(lock) > 900:
                    os.unlink(lock)
            except FileNotFoundError:                # the other process finished in between
                pass
            time.sleep(0.5)
            continue
        try:
            os.close(fd)
            return _build(kind, k, gpu, seed, path, **kw)
        finally:
            os.unlink(lock)
    return _cache_load(path, gpu)


def _build(kind, k, gpu, seed, store, **kw):
    rng = np.random.default_rng(seed)
    if kind == "einsum":
        L = kw.get("length") or {22: 1024, 21: 720, 20: 512, 19: 360, 18: 256, 17: 180, 16: 128, 15: 90, 14: 64, 12: 30, 10: 14, 8: 6, 6: 3}[k]
        c = EL.EinsumMatmulCircuit(k, L)
        a, b = rng.integers(-128, 128, (L, L)), rng.integers(-128, 128, (L, L))
        cs, fixed, copies, rows = c.keygen_inputs(a, b)
        fn = c.advice_fn(a, b, cs.n_advice)
        cache = {}
        def advice(phase, chal):
            key = (phase, tuple(chal))
            if key not in cache:
                cols = fn(phase, chal)
                idx = sorted(cols)
                cache[key] = dict(zip(idx, EL.cols_to_mont([cols[i] for i in idx], gpu, pinned=gpu is not None)))
            return cache[key]
        info = dict(circuit="accum_einsum_matmul (benches/accum_einsum_matmul.rs) ij,jk->ik len %d, Freivalds, k=%d" % (L, k), rows_used=rows)
        return dict(cs=cs, fixed=EL.cols_to_mont(fixed, gpu), copies=copies, advice=advice, instances=[], info=info)
    if kind == "transformer":
        # the transformer-shaped SURROGATE of BASELINE configs[4] (ezkl_layout.TransformerSurrogateCircuit): static lookup tables, a dynamic
        # lookup, a shuffle, Freivalds einsum with second-phase advice -- one unit laid out, tiled with numpy: no cache file, ~25 s at k = 22
        small = k < 16                                 # test sizes: tables and decompositions that fit 2^k rows
        c = EL.TransformerSurrogateCircuit(k, blocks=kw.get("blocks") or 4, d=kw.get("width") or (64 if not small else 4),
                                           einsum_len=kw.get("length") or (48 if not small else 3), decomp_base=kw.get("base") or (16384 if not small else 16),
                                           lookup_max=None if not small else (1 << k) // 16, seed=seed)
        b = c.build(gpu=gpu)
        return dict(cs=b["cs"], fixed=EL.cols_to_mont(b["fixed"], gpu), copies=b["copies"], advice=b["advice"], instances=b["instances"], info=b["info"])
    if kind == "mlp":
        layers, N, blocks, fill = kw.get("layers", 9), kw.get("width"), kw.get("blocks") or 2, kw.get("fill")
        # decomposition base of the range checks (ezkl's default 16384, src/lib.rs:257-260).  A table of `base` rows is split over
        # ceil(base / usable rows) column groups, so at k <= 11 the default makes a circuit with dozens of table columns and lookups whose
        # sweep program takes hiprtc minutes to compile (153 s at k = 10, profiles/r04a_pytest_gpu.log): small test circuits pass base=128,
        # the fixture's own setting (tests/golden/settings.json)
        base = kw.get("base") or 16384
        cap = 2 * ((1 << k) - 6)                       # cells of one block of two inner columns
        if N is None:                                  # fill about (blocks - 0.1) blocks: 3 x 2 x blocks advice columns
            N = int((((blocks - 0.1) * (fill or 100) / 100.0 * cap) / layers) ** 0.5)
        Ws = [sparse_weights(rng, N, N) for _ in range(layers)]
        bs = [rng.integers(-20, 20, N).tolist() for _ in range(layers)]
        x = rng.integers(-60, 60, N).tolist()
        if base < 16384:                               # keep the activations inside the two-leg range of a small base
            bs = [rng.integers(-2, 3, N).tolist() for _ in range(layers)]
            x = rng.integers(-3, 4, N).tolist()
        # fill (percent): lay out only that share of the cells but keep the column allocation of `blocks` full blocks (total_assignments is
        # what gen-settings would report for the full model): the Python layout engine needs ~15 us per cell, and every kernel of the
        # prover except the witness MSMs costs the same whatever the cells hold
        c = EL.MlpCircuit(k, 2, Ws, bs, base, 2)
        if fill and c.settings.total_assignments < int((blocks - 0.1) * cap):
            c = EL.MlpCircuit(k, 2, Ws, bs, base, 2, total_assignments=int((blocks - 0.1) * cap))
        cs, fixed, copies, reg = c.keygen_inputs(x, with_witness=True)      # one synthesis pass for the key and the witness
        adv, inst = c.witness_of(reg)
        info = dict(circuit="MLP %d x (Gemm %dx%d + bias + ReLU), batch 1, ezkl gate set (examples/onnx/large_mlp shape), k=%d" % (layers, N, N, k),
                    cells_used=reg.linear, blocks=c.gc.advices[0].num_blocks(), range_checks=[list(r) for r in c.settings.required_range_checks])
        if store:
            _cache_store(store, cs, fixed, copies, adv, inst, info)
        return dict(cs=cs, fixed=EL.cols_to_mont(fixed, gpu), copies=copies, advice=EL.cols_to_mont(adv, gpu), instances=inst, info=info)
    if kind == "conv":
        c = EL.ConvMnistCircuit(logrows=k, seed=seed)
        img = rng.integers(0, 16, (28, 28))                     # MNIST pixels / 16 (examples/conv2d_mnist/main.rs:326-329)
        cs, fixed, copies, reg = c.keygen_inputs(img, with_witness=True)
        adv, inst = c.witness_of(reg)
        info = dict(circuit="examples/conv2d_mnist: Conv 1->4 5x5 stride 2 on 28x28 + ReLU + Div{32} lookup (65 537-row table) + Linear 576->10, "
                            "3 advice columns, k=%d (synthetic image and parameters of the example's shapes)" % k, cells_used=reg.linear)
        if store:
            _cache_store(store, cs, fixed, copies, adv, inst, info)
        return dict(cs=cs, fixed=EL.cols_to_mont(fixed, gpu), copies=copies, advice=EL.cols_to_mont(adv, gpu), instances=inst, info=info)
    raise ValueError(kind)


def describe(cs):
    return dict(k=cs.k, advice_columns=cs.n_advice, fixed_columns=cs.n_fixed, instance_columns=cs.n_instance, selectors=cs.n_selectors,
                gates=len(cs.gates), lookups=len(cs.lookups), permutation_columns=len(cs.perm), degree=cs.degree, ext_k=cs.ext_k,
                second_phase_advice=sum(cs.advice_phase), challenges=cs.n_challenges, advice_queries=len(cs.advice_queries),
                fixed_queries=len(cs.fixed_queries))
