"""Row-sharded quotient sweep, the parts that exist: rewriting a gate program for one row shard
(GraphProgram.row_sharded) + the window bookkeeping of ezkl_amd/dist.py.  CPU: the rewritten program evaluated by the oracle on
every shard's windows reproduces the full sweep.  GPU: the same through ezkl_hip_eval_h_dev."""
import numpy as np
import pytest

from conftest import rand_fr
from oracle import binding as ob
from test_gpu_misc import _random_program


def _setup(seed, k, ek, ncols, ninstr):
    from ezkl_amd import backend as B
    rng = np.random.default_rng(seed)
    ne = 1 << ek
    cols = [rand_fr(rng, ne) for _ in range(ncols)]
    chal, prev = rand_fr(rng, 3), rand_fr(rng, ne)
    prog = _random_program(B, rng, k, ek, ncols, ninstr)
    code, consts, rots = prog.arrays()
    want = ob.eval_program(code, prog.n_intermediates, consts, rots, cols, chal, k, ek, previous=prev)
    return prog, cols, chal, prev, want


@pytest.mark.parametrize("log_world", [0, 1, 2, 3])
def test_row_sharded_program_reproduces_the_full_sweep(log_world):
    from ezkl_amd import dist as D
    k, ek = 6, 8
    prog, cols, chal, prev, want = _setup(7 + log_world, k, ek, 9, 120)
    world, ne = 1 << log_world, 1 << ek
    sub, queries = prog.row_sharded(log_world)
    assert sub.rotations == [0] and len(set(queries)) == len(queries)
    code, consts, rots = sub.arrays()
    got = np.zeros_like(want)
    for rank in range(world):
        lo, hi = D.shard_range(ne, rank, world)
        windows = [D._window(cols[c], start, ln) for (c, start, ln) in D.row_windows(queries, ne, rank, world)]
        got[lo:hi] = ob.eval_program(code, sub.n_intermediates, consts, rots, windows, chal, k - log_world, ek - log_world, previous=prev[lo:hi])
    assert (got == want).all()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["jit", "interp"])
def test_row_sharded_program_on_the_device(hip, mode, monkeypatch):
    from ezkl_amd import backend as B, dist as D
    monkeypatch.setenv("EZKL_EVALH_MODE", mode)
    k, ek, log_world = 9, 11, 2
    prog, cols, chal, prev, want = _setup(21, k, ek, 12, 300)
    world, ne = 1 << log_world, 1 << ek
    sub, queries = prog.row_sharded(log_world)
    got = np.zeros_like(want)
    for rank in range(world):
        lo, hi = D.shard_range(ne, rank, world)
        wins = [B.DeviceBuffer.from_numpy(np.ascontiguousarray(D._window(cols[c], start, ln))) for (c, start, ln) in D.row_windows(queries, ne, rank, world)]
        out = B.DeviceBuffer.from_numpy(np.ascontiguousarray(prev[lo:hi]))
        sub.evaluate_h([w.ptr for w in wins], chal, out.ptr)
        got[lo:hi] = out.to_numpy(shape=(hi - lo, 4))
    assert (got == want).all()


_ALIAS_SCRIPT = r"""
import sys, numpy as np, torch
torch.cuda.set_device(0)                      # torch's HIP runtime first, as in every multi-rank entry point (bench.py, tools/prove_bench.py)
sys.path.insert(0, sys.argv[1])
import ezkl_amd
from ezkl_amd import backend as B, dist as D
ezkl_amd.init(0)
rng = np.random.default_rng(3)
a = rng.integers(0, 1 << 60, size=(1000, 4), dtype=np.uint64)
d = B.DeviceBuffer.from_numpy(a)
t = torch.as_tensor(D.DeviceBytes(d.ptr, d.nbytes), device=torch.device("cuda", 0))
assert t.data_ptr() == d.ptr and t.numel() == 32000
assert (t.cpu().numpy().view(np.uint64).reshape(1000, 4) == a).all()
t[32:64] = 7                                                            # torch writes row 1 ...
torch.cuda.synchronize()
got = d.to_numpy(shape=(1000, 4))
assert (got[1] == np.frombuffer(bytes([7]) * 32, np.uint64)).all() and (got[2:] == a[2:]).all() and (got[0] == a[0]).all()
B.vec_fill(d.ptr, a[5], 10)                                             # ... and sees what the library writes
assert (t[:320].cpu().numpy().view(np.uint64).reshape(10, 4) == a[5]).all()
print("alias ok")
"""


@pytest.mark.gpu
def test_torch_sees_library_memory_in_place(hip):
    """the in-place RCCL all_gather of the sharded sweep hands torch the library's device pointer (dist.DeviceBytes through
    __cuda_array_interface__): the tensor must ALIAS the buffer, both ways.  In a child process: torch's own HIP runtime has to
    come up before the library's, as it does in the multi-rank entry points."""
    import subprocess, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, "-c", _ALIAS_SCRIPT, ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "alias ok" in r.stdout, r.stderr[-2000:]
