import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
Q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
MONT = 1 << 256
SEED = 0x657a6b6c  # "ezkl" 


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """end of a GPU run: HBM in use (hipMemGetInfo), the column pool's high-water mark and the host's peak RSS go into the tracked log
    (a suite that can exhaust a box must show that it does not)"""
    import resource
    lines = ["host peak RSS %.2f GiB (children %.2f GiB)" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20,
                                                           resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 2**20)]
    try:
        from ezkl_amd import lib as _lib
        if _lib._LIB is not None and _lib._LIB.ezkl_hip_device_count() > 0:
            from ezkl_amd import backend as B
            free_b, total_b = B.mem_info()
            ps = B.pool_stats()
            lines.append("HBM in use at the end %.2f GiB of %.0f; column pool: live %.2f GiB, live high-water %.2f GiB, parked %.2f GiB, bound %.2f GiB"
                         % ((total_b - free_b) / 2**30, total_b / 2**30, ps["live"] / 2**30, ps["live_peak"] / 2**30, ps["parked"] / 2**30, ps["bound"] / 2**30))
    except Exception as e:                                   # never turn a report into a failure
        lines.append("memory report unavailable: %r" % (e,))
    terminalreporter.write_sep("-", "memory")
    for l in lines:
        terminalreporter.write_line(l)


def fe_from_int(x, mod=R):
    """canonical integer -> Montgomery 4 x u64"""
    return np.frombuffer((x * MONT % mod).to_bytes(32, "little"), np.uint64).copy()


def fe_to_int(a, mod=R):
    return int.from_bytes(np.ascontiguousarray(a, np.uint64).tobytes(), "little") * pow(MONT, -1, mod) % mod


def rand_fr(rng, n):
    """n uniform-ish Montgomery residues: 253 random bits are always < r"""
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 61) - 1)
    return a


def witness_like(rng, n):
    """SURVEY.md §8(d) distribution W: 70% small signed ints via integer_rep_to_felt, 20% zero, 10% uniform
    (vectorised: |x| < 2^15, so x*R mod r is a table lookup of 2^16 Montgomery residues, not a Python loop)"""
    out = rand_fr(rng, n)
    kind = rng.random(n)
    small = rng.integers(-(1 << 15) + 1, 1 << 15, size=n)
    lo = -(1 << 15) + 1
    table = np.frombuffer(b"".join(((v % R) * MONT % R).to_bytes(32, "little") for v in range(lo, 1 << 15)), np.uint64).reshape(-1, 4)
    is_small = kind < 0.7
    out[is_small] = table[small[is_small] - lo]
    out[(kind >= 0.7) & (kind < 0.9)] = 0
    return out


@pytest.fixture(scope="session")
def golden_pk():
    g = np.load(os.path.join(GOLDEN, "pk_k6_subset.npz"))
    out = {k: g[k] for k in g.files}
    for name in ("fixed_values", "fixed_polys", "permutations", "perm_polys"):
        out[name] = out[name].view(np.uint64).reshape(out[name].shape[0], 64, 4)
    for name in ("fixed_cosets", "perm_cosets"):
        out[name] = out[name].view(np.uint64).reshape(out[name].shape[0], 512, 4)
    for name in ("l0", "l_last", "l_active_row"):
        out[name] = out[name].view(np.uint64).reshape(512, 4)
    return out


@pytest.fixture(scope="session")
def golden_srs():
    buf = open(os.path.join(GOLDEN, "kzg_k6.srs"), "rb").read()
    n = 64
    g = np.frombuffer(buf, np.uint64, count=8 * n, offset=4).reshape(n, 8).copy()
    gl = np.frombuffer(buf, np.uint64, count=8 * n, offset=4 + 64 * n).reshape(n, 8).copy()
    return dict(buf=buf, k=6, g=g, g_lagrange=gl)


@pytest.fixture(scope="session")
def hip():
    """the product library on a real GPU; fails (not skips) if it cannot initialise"""
    import ezkl_amd
    ezkl_amd.init()
    return ezkl_amd


def json_lines(stdout):
    """the JSON objects a child printed, one per line -- tolerant of other text in front of the object on the same line: ranks of a
    torchrun world share one stdout pipe with gloo's C-level prints, and a line of theirs without a trailing newline glues itself to the next"""
    import json
    out = []
    for line in stdout.splitlines():
        i = line.find('{"')
        if i < 0:
            continue
        try:
            out.append(json.loads(line[i:]))
        except ValueError:
            pass
    return out
