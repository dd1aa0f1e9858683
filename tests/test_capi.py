"""CPU tests of the C-ABI boundary: the library loads, exports everything include/ezkl_hip.h declares,
and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import numpy as np
import pytest
from conftest import ROOT
import ezkl_amd
from ezkl_amd import lib as L


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ezkl_hip.h")).read()
    declared = set(re.findall(r"\b(ezkl_hip_\w+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = L.load()
    for name in sorted(declared):
        assert hasattr(lib, name), "libezkl_hip.so does not export %s" % name
    assert declared == set(L.SYMBOLS)


def test_rccl_resolves_every_symbol_the_communicator_binds():
    """comm.hip dlopens librccl next to the HIP runtime and dlsyms nine entry points; a missing library or symbol must show HERE (a build
    box without a GPU), not in the first 8-rank run.  Nothing in RCCL is called and no device is touched."""
    assert L.load().ezkl_hip_comm_available() == 0
    maps = open("/proc/self/maps").read()
    assert "librccl" in maps


def test_strerror_and_version():
    lib = L.load()
    assert lib.ezkl_hip_strerror(0) == b"ok"
    assert b"no HIP device" in lib.ezkl_hip_strerror(-1)
    assert b"gfx950" in lib.ezkl_hip_version()
    assert b"calling thread" in lib.ezkl_hip_strerror(-6)


def test_runtime_gate(monkeypatch):
    """ENABLE_HIP_GPU / HIP_SMALL_K: the semantics of the reference's ENABLE_ICICLE_GPU / ICICLE_SMALL_K (README.md:106-122):
    disabled by UNSETTING the variable (any value, even "false", enables); k above the cutoff (default 8) goes to the GPU"""
    import ezkl_amd
    monkeypatch.delenv("ENABLE_HIP_GPU", raising=False)
    monkeypatch.delenv("HIP_SMALL_K", raising=False)
    assert not ezkl_amd.enabled(20)
    monkeypatch.setenv("ENABLE_HIP_GPU", "false")
    assert ezkl_amd.enabled(20) and ezkl_amd.enabled(9) and not ezkl_amd.enabled(8) and not ezkl_amd.enabled(0)
    monkeypatch.setenv("HIP_SMALL_K", "12")
    assert ezkl_amd.enabled(13) and not ezkl_amd.enabled(12)
    monkeypatch.setenv("HIP_SMALL_K", "0")
    assert ezkl_amd.enabled(1) and not ezkl_amd.enabled(0)
    monkeypatch.setenv("HIP_SMALL_K", "junk")
    assert ezkl_amd.enabled(9) and not ezkl_amd.enabled(8)            # malformed: the default


def test_product_path_does_not_import_oracle():
    """the product package must never reach into oracle/ (parity claims depend on it)"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ezkl_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in src and "import oracle" not in src and "from oracle" not in src, f


@pytest.mark.skipif(L.load().ezkl_hip_device_count() > 0, reason="GPU present")
def test_no_gpu_fails_loudly():
    with pytest.raises(ezkl_amd.EzklHipError) as e:
        ezkl_amd.init()
    assert e.value.code == -1
    out = np.zeros(8, np.uint64)
    # every compute entry point reports NO_DEVICE instead of computing on the CPU
    w = np.zeros(4, np.uint64)
    a = np.zeros((4, 4), np.uint64)
    assert L.load().ezkl_hip_ntt(a.ctypes.data_as(C.c_void_p), 2, w.ctypes.data_as(C.c_void_p), 0) == -1


def test_host_side_point_add_matches_oracle():
    """ezkl_hip_g1_add_affine is host code (folds per-GPU partial sums); check it against the oracle"""
    from oracle import binding as ob
    from ezkl_amd.backend import g1_add_affine
    b = ob.gen_bases(7, 8)
    assert (g1_add_affine(b[0], b[1]) == ob.g1_add(b[0], b[1])).all()
    assert (g1_add_affine(b[2], b[2]) == ob.g1_add(b[2], b[2])).all()           # doubling
    neg = b[3].copy()
    Q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
    y = int.from_bytes(neg[4:].tobytes(), "little")
    neg[4:] = np.frombuffer(((Q - y) % Q).to_bytes(32, "little"), np.uint64)
    assert (g1_add_affine(b[3], neg) == 0).all()                                 # P + (-P) = identity
    assert (g1_add_affine(np.zeros(8, np.uint64), b[4]) == b[4]).all()


@pytest.mark.parametrize("radix29", ["2", "1", "0"])
def test_eval_h_program_jit_compiles_offline(radix29, monkeypatch):
    """the quotient-sweep JIT: a gate program lowers to HIP source and hiprtc compiles it for gfx950 here -- the radix-2^29 generator
    (product as a call / inline) and the radix-2^32 one"""
    monkeypatch.setenv("EZKL_EVALH_R29", radix29)
    from ezkl_amd import backend as B
    from conftest import fe_from_int
    prog = B.GraphProgram(4, 6)
    a, b, o = prog.column(0), prog.column(1, 1), prog.column(2, -1)
    g = prog.calc("mul", prog.column(3), prog.calc("sub", o, prog.calc("mul", a, b)))
    h = prog.calc("negate", prog.calc("double", prog.calc("square", prog.calc("add", g, prog.constant(fe_from_int(5))))))
    prog.horner(prog.previous(), [g, h], prog.challenge(0))
    prog.check_compiles(4)


def test_eval_h_check_refuses_malformed_programs():
    """host-only: a program that reads an intermediate no instruction has written, or names an unknown opcode, is refused by the
    code generators' front door (the radix-2^29 generator indexes its bound table by value version)"""
    import numpy as np
    from ezkl_amd import backend as B, lib as L
    prog = B.GraphProgram(4, 6)
    prog.constant(np.zeros(4, np.uint64))
    prog.code.append([B.OPS["add"], 0, B.INTERMEDIATE, 3, 0, B.CONST, 0, 0])
    prog.n_intermediates = 4
    with pytest.raises(L.EzklHipError):
        prog.check_compiles(1)
    bad = B.GraphProgram(4, 6)
    bad.constant(np.zeros(4, np.uint64))
    bad.code.append([99, 0, B.CONST, 0, 0, B.CONST, 0, 0])
    bad.n_intermediates = 1
    with pytest.raises(L.EzklHipError):
        bad.check_compiles(1)
