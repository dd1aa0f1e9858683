"""The C++ host mirror (include/ezkl_hip.hpp: ParamsKZG / EvaluationDomain / GraphEvaluator / polycommit_commit with
halo2's names) compiled with plain g++ against libezkl_hip.so and checked against the C oracle + golden fixtures.
CPU: the header and test compile and link (every C-ABI symbol the header uses resolves).  GPU: the binary runs."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_hpp_mirror.cpp")


def _build(out_dir):
    libdir, ordir = os.path.join(ROOT, "ezkl_amd"), os.path.join(ROOT, "oracle")
    if not os.path.exists(os.path.join(ordir, "liboracle.so")):
        subprocess.check_call(["make", "-C", ordir])
    assert os.path.exists(os.path.join(libdir, "libezkl_hip.so")), "libezkl_hip.so missing: run __graft_entry__.build()"
    exe = os.path.join(str(out_dir), "test_hpp_mirror")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", libdir, "-lezkl_hip", "-L", ordir, "-loracle", "-Wl,-rpath," + libdir, "-Wl,-rpath," + ordir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    # no GPU needed to start the binary far enough to print usage
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stdout


@pytest.mark.gpu
def test_cpp_mirror_parity(tmp_path):
    exe = _build(tmp_path)
    z = np.load(os.path.join(ROOT, "tests", "golden", "pk_k6_subset.npz"))
    for name in ("fixed_values", "fixed_polys", "fixed_cosets"):
        np.ascontiguousarray(z[name]).tofile(os.path.join(str(tmp_path), name + ".bin"))
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "kzg_k6.srs"), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
