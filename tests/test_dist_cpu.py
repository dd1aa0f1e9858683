"""world_size-2 gloo test of the multi-GPU decomposition (runs on CPU): point-sharded MSM partials are
all_gathered and folded with the library's host-side group law; the per-shard MSMs themselves are done by
the oracle here (no GPU), so this checks the sharding + exchange + fold logic against the unsharded answer."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from conftest import ROOT, SEED, rand_fr


def free_port():
    """a TCP port nobody listens on right now (bind to 0, read it back): fixed port formulas collide with leftovers of earlier runs"""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as ob
    from ezkl_amd import dist as D
    rng = np.random.default_rng(5)
    s = rand_fr(rng, n)
    pts = ob.gen_bases(SEED, n)
    lo, hi = D.shard_range(n, rank, world)
    partial = ob.msm(s[lo:hi], pts[lo:hi])
    total = D.fold_partials(partial, dist, torch.device("cpu"))
    cols = D.shard_columns(7, rank, world)
    q.put((rank, total.tobytes(), (lo, hi), cols))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_msm_fold_world2():
    from oracle import binding as ob
    n, world = 1001, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rng = np.random.default_rng(5)
    s = rand_fr(rng, n)
    want = ob.msm(s, ob.gen_bases(SEED, n)).tobytes()
    assert res[0][1] == want and res[1][1] == want
    assert res[0][2] == (0, 501) and res[1][2] == (501, 1001)
    assert res[0][3] == [0, 2, 4, 6] and res[1][3] == [1, 3, 5]


def _reshard_check(world, rank, port, n_ext, ncols):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ezkl_amd import dist as D
    rng = np.random.default_rng(9)
    cols = [rand_fr(rng, n_ext) for _ in range(ncols)]                    # every rank can rebuild every column (the ground truth)
    owned = {c: cols[c] for c in D.shard_columns(ncols, rank, world)}
    queries = [(c, s) for c in range(ncols) for s in ((0,) if c % 2 else (0, 4, -4))] + [(0, n_ext - 24), (ncols - 1, 1000)]
    got = D.reshard_columns_to_rows(owned, queries, n_ext, dist, torch.device("cpu"))
    lo, hi = D.shard_range(n_ext, rank, world)
    ok = set(got) == set(queries)
    for (c, s) in queries:
        want = np.stack([cols[c][(r + s) % n_ext] for r in range(lo, hi)])
        ok = ok and got[(c, s)].shape == (hi - lo, 4) and bool((got[(c, s)] == want).all())
    dist.barrier()
    dist.destroy_process_group()
    return ok


def _reshard_worker(rank, port, port2, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    ok = _reshard_check(3, rank, port, 250, 4)                            # ragged row counts: 84 / 83 / 83
    if rank < 2:
        ok = _reshard_check(2, rank, port2, 256, 5) and ok
    q.put((rank, ok))


def test_columns_to_row_windows_world3_and_2():
    """the exchange of the row-sharded quotient sweep: columns owned round-robin -> every rank's row window of every
    (column, rotation) pair, including windows that wrap around the domain; three ranks (ragged row counts), then two"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    port2 = free_port()
    procs = [ctx.Process(target=_reshard_worker, args=(r, port, port2, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _ in res) == [0, 1, 2] and all(ok for _, ok in res)
    # one rank: the identity (every window is the whole, rotated column)
    from ezkl_amd import dist as D
    col = rand_fr(np.random.default_rng(1), 64)
    got = D.reshard_columns_to_rows({0: col}, [(0, 0), (0, 5)], 64, None, None)
    assert (got[(0, 0)] == col).all() and (got[(0, 5)] == np.roll(col, -5, axis=0)).all()


def test_shard_range_covers_everything():
    from ezkl_amd import dist as D
    for n in (0, 1, 7, 8, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def _prove_worker(rank, world, port, q):
    """the column-sharded prover on CPU: MSMs by points, NTTs by columns (owner-computed forms, evaluations, SHPLONK partial sums), the
    sweep by rows fed by the all-to-all -- on the reference's fixture circuit (lookups, 32 permutation columns, instance)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fixture_k6 as FX
    from ezkl_amd import plonk as P
    from oracle.cpu_backend import DistOracleBackend
    from oracle import pyref as pr
    srs = pr.parse_srs(open(os.path.join(FX.G, "kzg_k6.srs"), "rb").read())
    g = np.stack([np.frombuffer(b, np.uint64) for b in srs["g"]])
    gl = np.stack([np.frombuffer(b, np.uint64) for b in srs["g_lagrange"]])
    fx = FX.load()
    be = DistOracleBackend(g, gl, FX.K, dist, torch.device("cpu"))
    adv, inst, _ = FX.witness(fx)
    copies = FX.copies_of(FX.copy_cycles(fx["pk"]))
    pk, vk = P.keygen(fx["cs"], be, FX.mont_cols(fx["fixed"]), copies)
    proof = P.create_proof(pk, be, FX.mont_cols(adv), P.Rng(7), instances=inst)
    q.put((rank, proof, be.sharded_ntt_columns, getattr(be, "sharded_sweeps", 0)))
    dist.barrier()
    dist.destroy_process_group()


def _single_rank_fixture_proof():
    import fixture_k6 as FX
    from ezkl_amd import plonk as P
    from oracle.cpu_backend import OracleBackend
    from oracle import pyref as pr
    srs = pr.parse_srs(open(os.path.join(FX.G, "kzg_k6.srs"), "rb").read())
    g = np.stack([np.frombuffer(b, np.uint64) for b in srs["g"]])
    gl = np.stack([np.frombuffer(b, np.uint64) for b in srs["g_lagrange"]])
    fx = FX.load()
    be = OracleBackend(g, gl, FX.K)
    adv, inst, _ = FX.witness(fx)
    pk, vk = P.keygen(fx["cs"], be, FX.mont_cols(fx["fixed"]), FX.copies_of(FX.copy_cycles(fx["pk"])))
    return P.create_proof(pk, be, FX.mont_cols(adv), P.Rng(7), instances=inst), fx["cs"]


def test_column_sharded_prover_world2_and_4_emit_the_single_rank_proof():
    want, cs = _single_rank_fixture_proof()
    n_cols = cs.n_advice + cs.n_instance + cs.n_chunks + 2 * len(cs.lookups)     # advice, instance, z, m, phi: 30 + 1 + 7 + 70
    for world in (2, 4):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = free_port()
        procs = [ctx.Process(target=_prove_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=600) for _ in range(world))
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert all(r[1] == want for r in res), "world %d: sharded proof differs from the single-rank proof" % world
        assert sum(r[2] for r in res) == n_cols + (70 + 3) * world       # every witness column transformed by exactly ONE rank (+ keygen's columns on all)
        assert all(r[3] == 1 for r in res)


# ---- the exchange callbacks of the owner-mode C++ prover (ezkl_amd/dist.py exchange_segments / allgather_host_bytes) ----
class _FakeDevice:
    """device memory for the CPU test: `pointers` are keys into a dict of byte arrays"""
    mem = {}

    @staticmethod
    def memcpy_d2h(ptr, n):
        return _FakeDevice.mem[ptr][:n].copy()

    @staticmethod
    def memcpy_h2d(ptr, arr):
        _FakeDevice.mem[ptr][:len(arr)] = arr


def _exchange_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes
    from ezkl_amd import dist as D
    D._b = _FakeDevice
    # every rank sends two segments of different sizes to every rank (itself included); the k-th segment sent to p is the k-th p receives from us
    sends, recvs = [], []
    for p in range(world):
        for seg in range(2):
            key = ("s", p, seg)
            _FakeDevice.mem[key] = np.full(100 + 10 * p + seg, (rank * 16 + p) % 251, np.uint8)
            sends.append((p, key, len(_FakeDevice.mem[key])))
            key = ("r", p, seg)
            _FakeDevice.mem[key] = np.zeros(100 + 10 * rank + seg, np.uint8)
            recvs.append((p, key, 100 + 10 * rank + seg))
    D.exchange_segments(sends, recvs, dist, torch.device("cpu"))
    ok = all((_FakeDevice.mem[("r", p, seg)] == (p * 16 + rank) % 251).all() for p in range(world) for seg in range(2))
    # allgather_host_bytes on a real host buffer
    per = 24
    buf = (ctypes.c_uint8 * (world * per))()
    for i in range(per):
        buf[rank * per + i] = rank + 1
    D.allgather_host_bytes(ctypes.addressof(buf), per, dist, torch.device("cpu"))
    ok = ok and all(buf[r * per + i] == r + 1 for r in range(world) for i in range(per))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_callbacks_world3():
    """the host-staged all-to-all (segment lists matched by order per pair of ranks) and the host all_gather the owner-mode prover's
    callbacks use when the library communicator is not available"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True), (2, True)]


# ---- the process-group self-test (ezkl_amd/dist.py selftest: the torch.distributed twin of ezkl_hip_comm_selftest) at the driver's world size ----
class _FakeBuffer:
    count = 0

    def __init__(self, nbytes):
        _FakeBuffer.count += 1
        self.ptr = _FakePtr(("buf", _FakeBuffer.count), 0)
        _FakeDevice.mem[self.ptr.key] = np.zeros(nbytes, np.uint8)


class _FakePtr:
    """a `device pointer` that supports + offset, as the integer pointers of the real backend do"""

    def __init__(self, key, off):
        self.key, self.off = key, off

    def __add__(self, d):
        return _FakePtr(self.key, self.off + int(d))


class _FakeBackend:
    DeviceBuffer = _FakeBuffer

    @staticmethod
    def memcpy_d2h(ptr, n):
        return _FakeDevice.mem[ptr.key][ptr.off:ptr.off + n].copy()

    @staticmethod
    def memcpy_h2d(ptr, arr):
        _FakeDevice.mem[ptr.key][ptr.off:ptr.off + len(arr)] = arr

    @staticmethod
    def g1_add_affine(a, b):
        from ezkl_amd import backend as B
        return B.g1_add_affine(a, b)                      # the library's host-side group law (no device)


def _selftest_worker(rank, world, port, q, sabotage):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ezkl_amd import dist as D
    D._b = _FakeBackend
    if sabotage and rank == 1:                           # one rank sends damaged bytes: every rank that receives from it must notice
        real = _FakeBackend.memcpy_d2h
        def bad(ptr, n):
            a = real(ptr, n)
            if n: a[n // 2] ^= 1
            return a
        _FakeBackend.memcpy_d2h = staticmethod(bad)
    try:
        ok = D.selftest(dist, torch.device("cpu"))
        q.put((rank, bool(ok), ""))
    except RuntimeError as e:
        q.put((rank, False, str(e)))
    dist.barrier()
    dist.destroy_process_group()


def test_process_group_selftest_world8_gloo():
    """eight ranks (the driver's SCALE run) pass the self-test over gloo; with one rank sabotaged the failing step is
    named instead of a wrong proof appearing later"""
    for world, sabotage in ((8, False), (3, True)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = free_port()
        procs = [ctx.Process(target=_selftest_worker, args=(r, world, port, q, sabotage)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=180) for _ in range(world))
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        if not sabotage:
            assert res == [(r, True, "") for r in range(world)]
        else:
            assert all(r[1] is False and "all-to-all of odd-sized segments" in r[2] for r in res), res
