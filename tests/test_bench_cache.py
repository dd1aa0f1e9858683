"""tools/bench_circuits.py keeps laid-out circuits on disk (the N ranks of `bench.py --gpus N` read ONE layout): the cached circuit must be
the circuit -- same constraint system blob, fixed / advice columns, copy constraints, instances -- and concurrent builders must not
clobber each other (lock file + atomic rename)."""
import os
import sys
import threading

import numpy as np
from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_cached_circuit_is_the_circuit(tmp_path, monkeypatch):
    import bench_circuits as BC
    from ezkl_amd import native as NV
    monkeypatch.setenv("EZKL_BENCH_CACHE", str(tmp_path))
    first = BC.build("mlp", 8)                       # lays out and stores
    files = [f for f in os.listdir(tmp_path) if f.endswith(".npz")]
    assert files == ["mlp_k8_s1.npz"] and not [f for f in os.listdir(tmp_path) if f.endswith(".lock")]
    again = BC.build("mlp", 8)                       # reads
    assert "read from" in again["info"].get("layout", "")
    assert NV.serialize_cs(first["cs"]) == NV.serialize_cs(again["cs"])
    assert list(first["copies"]) == list(again["copies"]) and first["instances"] == again["instances"]
    for a, b in zip(first["fixed"] + first["advice"], again["fixed"] + again["advice"]):
        assert (np.asarray(a) == np.asarray(b)).all()
    # other generator options are another key, another file
    other = BC.build("mlp", 8, blocks=3, fill=50)
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz")) == ["mlp_k8_s1.npz", "mlp_k8_s1_blocks3_fill50.npz"]
    assert other["info"]["cells_used"] != first["info"]["cells_used"]


def test_concurrent_builders_share_one_layout(tmp_path, monkeypatch):
    import bench_circuits as BC
    monkeypatch.setenv("EZKL_BENCH_CACHE", str(tmp_path))
    out, errs = [None] * 3, []
    def work(i):
        try:
            out[i] = BC.build("mlp", 7)
        except Exception as e:                       # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in th: t.start()
    for t in th: t.join(300)
    assert not errs and all(o is not None for o in out)
    assert sum("read from" in o["info"].get("layout", "") for o in out) == 2          # one laid it out, two waited and read
    assert [f for f in os.listdir(tmp_path)] == ["mlp_k7_s1.npz"]
