"""SURVEY.md §8(e) in the C++ host prover: columns and arguments BY OWNER (include/ezkl_prover.h "multi-GPU, the full form").  N ranks
sharing the one GPU of the test box, gloo moving the data through the prover's exchange callbacks, must emit the bytes of the one-rank
proof -- on the reference's fixture circuit (35 lookups, 32 permutation columns, 8 cosets), on an MLP over the ezkl gate set (4 cosets)
and on the reference's einsum bench circuit (second-phase advice; 2 cosets, so 4 ranks sweep ROW RANGES of a coset with halo rows), and on
the transformer-shaped surrogate of configs[4] (static + dynamic lookups, a shuffle, Freivalds einsum with second-phase advice, an instance
column; degree 6: 8 cosets) -- the circuit `bench.py --gpus N` proves at k = 22 with NTTs and MSMs both sharded."""
import json
import os
import subprocess
import sys

import pytest
from conftest import ROOT, json_lines

pytestmark = pytest.mark.gpu
TOOL = os.path.join(ROOT, "tools", "prove_multi.py")


def _run(env, world, port, extra=()):
    env = dict(os.environ, EZKL_BENCH_CACHE="off", REPS="1", **env)
    if world == 1:
        cmd = [sys.executable, TOOL]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), TOOL, "--gloo", "--share-device"] + list(extra)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    objs = json_lines(r.stdout)
    assert objs, r.stderr[-3000:]
    return objs[-1]


@pytest.mark.parametrize("circuit,k", [("fixture", 6), ("mlp", 10), ("einsum", 10), ("transformer", 11)])
def test_owner_sharded_native_prover_same_bytes(hip, circuit, k):
    env = {"CIRCUIT": circuit, "K": str(k), "MLP_BASE": "128"}     # small range-check tables: 16 lookups instead of the default base's 137 at k = 10
    one = _run(env, 1, 0)
    assert one["verifier_accepts"] and one["tampered_rejected"]
    for world, port in ((2, 29571), (4, 29573)) if circuit != "mlp" else ((2, 29571),):
        j = _run(env, world, port)
        assert j["n_gpus"] == world and j["mode"] == "columns and arguments by owner"
        assert j["all_ranks_same_proof"] and j["verifier_accepts"] and j["tampered_rejected"]
        assert j["proof_sha256"] == one["proof_sha256"], (circuit, world)
        total = j["per_rank"][0]["stats"]["witness_columns"]
        done = [r["stats"]["columns_transformed_here"] for r in j["per_rank"]]
        assert sum(done) == total and max(done) < total          # every witness column transformed exactly once, by one rank
        assert all(r["stats"]["exchange_bytes_received"] > 0 for r in j["per_rank"])        # the sweep's rows came through the all-to-all
        assert all(r["sharded_sweeps"] >= 2 for r in j["per_rank"])


def test_shplonk_commitments_are_reduce_scattered(hip):
    """SHPLONK's two commitments of per-rank partial polynomials: every rank receives the other ranks' rows of ITS point range (one segment
    per peer), commits the summed slice and the fold adds the points -- same bytes as whole MSMs of the partials on every rank, and
    exactly 2 x (world - 1) x n / world x 32 more bytes through the exchange"""
    env = {"CIRCUIT": "fixture", "K": "6"}
    n, world = 64, 4
    on = _run(env, world, 29579)
    off = _run(dict(env, EZKL_PROVER_NO_SUM_SCATTER="1"), world, 29581)
    assert on["proof_sha256"] == off["proof_sha256"] and on["all_ranks_same_proof"] and on["verifier_accepts"]
    for a, b in zip(on["per_rank"], off["per_rank"]):
        assert a["stats"]["exchange_bytes_received"] - b["stats"]["exchange_bytes_received"] == 2 * (world - 1) * (n // world) * 32


def test_replicated_mode_still_same_bytes(hip):
    """the round-2 sharding (commit batches by columns, everything else replicated) on the coset-major prover"""
    env = {"CIRCUIT": "mlp", "K": "9", "MLP_BASE": "128"}
    one = _run(env, 1, 0)
    j = _run(env, 2, 29575, extra=["--replicated"])
    assert j["mode"].startswith("replicated") and j["proof_sha256"] == one["proof_sha256"] and j["all_ranks_same_proof"] and j["verifier_accepts"]
    assert all(r["stats"]["exchange_bytes_received"] == 0 for r in j["per_rank"])


def test_lookup_failure_is_collective(hip):
    """a witness value outside its table: every rank returns the error (no rank is left waiting in the next collective)"""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        # every rank leaves its status in a file of its own: the ranks share one stdout pipe with gloo's C-level prints, where lines can run together
        env = dict(os.environ, EZKL_BENCH_CACHE="off", CIRCUIT="fixture", K="6", EZKL_RANK_STATUS_DIR=d)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29577",
               TOOL, "--gloo", "--share-device", "--bad-lookup"]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        for rank in (0, 1):
            path = os.path.join(d, "rank_%d.txt" % rank)
            assert os.path.exists(path), (rank, r.stdout[-1000:], r.stderr[-1000:])
            line = open(path).read()
            assert "ERROR" in line and "lookup input not in table" in line, (rank, line, r.stdout[-1000:])
