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
    port = 29500 + (os.getpid() % 2000)
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


def test_shard_range_covers_everything():
    from ezkl_amd import dist as D
    for n in (0, 1, 7, 8, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
