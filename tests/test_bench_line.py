"""The contract's ONE JSON line must survive the driver's record (a 20 KB line came back `parsed: null`).
bench.compact_line() is a pure function of the full record, so it is sized here from records of real runs kept under profiles/."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (module level imports nothing that needs a GPU)

CONTRACT_KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"]


def _record(name):
    text = open(os.path.join(ROOT, "profiles", name)).read()
    lines = [l for l in text.splitlines() if l.startswith("{")]
    return json.loads(lines[-1] if lines else text)


@pytest.mark.parametrize("name, n", [("r05av_bench.log", 1), ("r05ax_bench_world8.json", 8)])
def test_line_is_short_and_round_trips(name, n):
    full = _record(name)
    if n == 8:
        full["n_gpus"] = 8                      # the kept record is 8 ranks sharing one device: n_gpus says 1 there
    text = bench.compact_line(full)
    assert "\n" not in text
    assert len(text) < bench.LINE_LIMIT < 8000
    line = json.loads(text)
    for k in CONTRACT_KEYS:
        assert k in line, k
    assert line["config"]["workload"].startswith("configs[1]")
    assert abs(line["value"] / full["value"] - 1) < 1e-4 and abs(line["ms_per_step"] / full["ms_per_step"] - 1) < 1e-4
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic_source", "avg_launch_ms", "ntt", "product_peak", "kernels"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert r["ntt"]["elems_per_s"] > 0 and len(r["kernels"]) >= 2
    if n == 1:
        c = line["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample", "ntt", "prove_seconds_k20_mlp"):
            assert k in c, k
        assert c["prove_seconds_k20_mlp"]["gpu"] == line["prove_seconds_k20_mlp"]["gpu"] > 0
        assert c["prove_seconds_k20_mlp"]["identical"] is True
    else:
        assert "rccl_ranks_seen" in line and line["prove_multi"]["all_ranks_same_proof"] is True
        assert "2^20" in line["msm_strong_scaling"]


def test_line_stays_short_whatever_the_legs_put_into_the_record():
    full = _record("r05av_bench.log")
    blob = "x" * 5000
    full["prove"]["einsum"]["error"] = blob
    full["prove"]["skipped"] = [blob] * 50
    full["roofline"]["traffic_source"] = blob
    full["cpu_baseline"]["sample"] = blob
    full["errors"] = [blob] * 10
    full["extra"]["msm_strong_scaling"] = {"error": blob}
    text = bench.compact_line(full)
    assert len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    assert line["value"] > 0 and "roofline" in line and "cpu_baseline" in line


def test_emit_prints_one_line_and_keeps_the_full_record(tmp_path, capsys, monkeypatch):
    full = _record("r05av_bench.log")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(full)
    out = capsys.readouterr().out
    assert out.count("\n") == 1 and len(out) < bench.LINE_LIMIT
    assert json.load(open(tmp_path / "bench_full.json"))["prove"]["mlp_k20"]["prove_seconds_gpu"] == full["prove"]["mlp_k20"]["prove_seconds_gpu"]


def test_multi_rank_line_carries_the_three_sharded_proofs():
    """N > 1: configs[3] (einsum, MLP k = 20) and configs[4] (the transformer surrogate at k = 22) each with what a SCALE run is checked by --
    seconds, same proof on every rank, verifier, rccl_ranks_seen -- and still a short line"""
    full = _record("r05ax_bench_world8.json")
    full["n_gpus"] = 8
    sub = {"prove_seconds_gpu": 0.31, "all_ranks_same_proof": True, "verifier_accepts": True, "rccl_ranks_seen": 8, "exchange_ms_per_proof_max": 4.2,
           "per_rank": [{"stats": {"x": "y" * 400}}] * 8}
    full["prove"]["rccl_ranks_seen"] = 8
    full["prove"]["mlp_k20"] = dict(sub)
    full["prove"]["transformer_k22"] = dict(sub, prove_seconds_gpu=0.25)
    text = bench.compact_line(full)
    assert len(text) < bench.LINE_LIMIT
    line = json.loads(text)
    assert line["rccl_ranks_seen"] == 8
    pm = line["prove_multi"]
    assert pm["mlp_k20"]["rccl_ranks_seen"] == 8 and pm["transformer_k22"]["prove_seconds_gpu"] == 0.25 and "per_rank" not in pm["transformer_k22"]
    # a leg that failed shows its error, shortened
    full["prove"]["transformer_k22"] = {"error": "z" * 3000}
    line = json.loads(bench.compact_line(full))
    assert len(line["prove_multi"]["transformer_k22"]["error"]) == 200 and len(bench.compact_line(full)) < bench.LINE_LIMIT
