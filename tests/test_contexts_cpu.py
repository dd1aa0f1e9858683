"""The context table of libezkl_hip.so (ezkl_hip_contexts_configure / ezkl_hip_set_context) on a box WITHOUT a GPU: argument checking and
the loud failure; the multi-context prover itself is tests/test_group.py (GPU)."""
import ctypes as C
import os
import subprocess
import sys

from conftest import ROOT


def test_context_table_argument_checks_without_a_device():
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
from ezkl_amd import lib
L = lib.load()
has_gpu = L.ezkl_hip_device_count() > 0          # (not torch: a second HIP runtime loaded after the library's finds no device)
arr = (C.c_int * 2)(0, 0)
assert L.ezkl_hip_contexts_configure(0, arr) == -3                     # no contexts
assert L.ezkl_hip_contexts_configure(2, None) == -3                    # no device list
assert L.ezkl_hip_contexts_configure(65, arr) == -3                    # more than the table holds
rc = L.ezkl_hip_contexts_configure(2, arr)
if not has_gpu:
    assert rc == -1, rc                                                # EZKL_ERR_NO_DEVICE: loud, no CPU fallback
    assert L.ezkl_hip_context_count() == 0
    assert L.ezkl_hip_set_context(0) == -1
    assert L.ezkl_hip_context_device(0) == -1
    assert L.ezkl_hip_memcpy_peer(None, 0, None, 0, 0) == -3
else:
    got = (rc, L.ezkl_hip_context_count(), L.ezkl_hip_set_context(1), L.ezkl_hip_set_context(2), L.ezkl_hip_set_context(-1), L.ezkl_hip_context_device(1),
           L.ezkl_hip_context_device(2))
    assert got == (0, 2, 0, -3, -3, 0, -1), got
    bad = (C.c_int * 1)(99)
    assert L.ezkl_hip_contexts_configure(1, bad) == -3
print("ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_group_needs_a_device():
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
from ezkl_amd import native as NV, lib
L = NV.load()
has_gpu = lib.load().ezkl_hip_device_count() > 0
h = C.c_void_p()
blob = b"EZCS" + bytes(60)
rc = L.ezkl_prover_group_create(blob, C.c_size_t(len(blob)), C.c_int(2), C.byref(h))
assert rc in ((-1,) if not has_gpu else (-3,)), rc     # no device -> EZKL_ERR_NO_DEVICE; with one: 2 contexts > 1 -> invalid
assert L.ezkl_prover_group_create(None, C.c_size_t(0), C.c_int(2), C.byref(h)) == -3
assert L.ezkl_prover_group_size(None) == 0 and L.ezkl_prover_group_free(None) == 0
print("ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
