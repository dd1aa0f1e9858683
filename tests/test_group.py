"""One process, several GPUs (include/ezkl_prover.h "the prover group"): the owner-mode prover run by N host threads
of ONE process, thread r on context r of libezkl_hip.so, collectives in-process (host memory + peer copies).  The test box has one GPU,
so the context table is configured with several contexts on device 0 (ezkl_hip_contexts_configure) -- separate streams, pools, MSM
tables, NTT plans and JIT modules per context, exactly as on N devices; on a multi-GPU box the same test also runs one context per
device.  Each case runs in a child process: the context table can only be set before the library's first use."""
import json
import os
import subprocess
import sys

import pytest
from conftest import ROOT, json_lines

pytestmark = pytest.mark.gpu

CHILD = r'''
import hashlib, json, os, sys
sys.path.insert(0, os.environ["EZKL_ROOT"]); sys.path.insert(0, os.path.join(os.environ["EZKL_ROOT"], "tools"))
os.environ["EZKL_BENCH_CACHE"] = "off"
import numpy as np
from ezkl_amd import lib as L, backend as B, native as NV, plonk as P
mode, world, circuit, k = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
if mode == "same-device":
    B.contexts_configure([0] * world)
else:                                           # one context per visible device: ezkl_hip_init(-1) without LOCAL_RANK
    os.environ.pop("LOCAL_RANK", None)
    L.check(L.load().ezkl_hip_init(-1), "init")
assert B.context_count() >= world, (B.context_count(), world)
import bench_circuits as BC
s = 0x1234567890abcdef1234567890abcdef % P.R
key_file = None
if circuit == "fixture":                        # the reference's own circuit, witness and pk.key (tests/golden): load_pk, not keygen
    sys.path.insert(0, os.path.join(os.environ["EZKL_ROOT"], "tests"))
    import fixture_k6 as FX
    fx = FX.load()
    k, cs = 6, fx["cs"]
    adv_i, instances, _ = FX.witness(fx)
    adv = FX.mont_cols(adv_i)
    key_file = os.path.join(FX.G, "pk_k6.key")
else:
    built = BC.build(circuit, k, gpu=B, base=128)     # small range-check tables (16-20 lookups): the default base makes a 137-lookup circuit at k = 10 whose sweep program takes hiprtc minutes
    cs, fixed, copies, adv, instances = built["cs"], built["fixed"], built["copies"], built["advice"], built["instances"]
gb, glb = B.gen_srs(k, s)
g, gl = gb.download(), glb.download()
# the one-context prover: the reference bytes
if key_file:
    npk = NV.NativeProvingKey.from_file(NV.NativeCircuit(cs), key_file, recommit=gb)      # the file's commitments were made under the public SRS
else:
    npk = NV.NativeProvingKey(NV.NativeCircuit(cs), gb, fixed, copies)
want = NV.create_proof(npk, gb, glb, adv, seed=7, instances=instances)
grp = NV.NativeGroup(cs, world)
grp.load_srs(g, gl)
if key_file:
    grp.pk_read_file(key_file, recommit=True)
else:
    grp.keygen(fixed, copies)
stats, tm = [], {}
got = grp.create_proof(adv, seed=7, instances=instances, timings=tm, stats=stats)
again = grp.create_proof(adv, seed=7, instances=instances)
ok = NV.verify_proof(grp.pk(0), NV.g2_mul_generator(1), NV.g2_mul_generator(s), got, instances)
fresh = grp.create_proof(adv, seed=0, instances=instances)          # OS entropy: one 256-bit key shared by the threads
ok_fresh = NV.verify_proof(grp.pk(0), NV.g2_mul_generator(1), NV.g2_mul_generator(s), fresh, instances)
# load_pk for the group (ezkl_prover_group_pk_read_file): the key written by the one-context prover, read back by every context --
# same proof bytes, and each context holds only the cosets it sweeps
import tempfile
with tempfile.TemporaryDirectory() as d:
    path = key_file or os.path.join(d, "pk.key")
    if not key_file:
        open(path, "wb").write(npk.to_bytes())
    grp2 = NV.NativeGroup(cs, world)
    grp2.load_srs(g, gl)
    grp2.pk_read_file(path, recommit=bool(key_file))
    from_file = grp2.create_proof(adv, seed=7, instances=instances)
    res = [grp2.pk(r).residency() for r in range(grp2.world)]
    rewritten_equal = None
    if not key_file:                            # a key held by owner writes the same file (the complete extended columns are recomputed)
        pk0 = grp2.pk(0)
        rewritten_equal = NV.NativeProvingKey.to_bytes(pk0) == npk.to_bytes()
    grp2.free()
print(json.dumps({"world": grp.world, "contexts": B.context_count(), "same_bytes": got == want, "repeatable": again == got, "verifier_accepts": bool(ok),
                  "fresh_randomness_differs": fresh != got, "fresh_verifies": bool(ok_fresh), "stats": stats, "total_seconds": tm.get("total"),
                  "from_file_same_bytes": from_file == want, "residency": res, "one_context_residency": npk.residency(), "rewritten_equal": rewritten_equal}))
grp.free()
'''


def _run(mode, world, circuit, k):
    env = dict(os.environ, EZKL_ROOT=ROOT)
    env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, "-c", CHILD, mode, str(world), circuit, str(k)], env=env, capture_output=True, text=True, timeout=900)
    objs = json_lines(r.stdout)
    assert objs, r.stderr[-3000:]
    return objs[-1]


@pytest.mark.parametrize("world,circuit,k", [(2, "mlp", 9), (4, "mlp", 10), (2, "fixture", 6)])
def test_group_of_contexts_on_one_device_same_bytes(hip, world, circuit, k):
    """(k = 20 with 2 and 4 contexts: tools/prove_group.py, NOTEBOOK.md §5.2 -- too slow for the suite)"""
    j = _run("same-device", world, circuit, k)
    assert j["world"] == world and j["contexts"] == world
    assert j["same_bytes"] and j["repeatable"] and j["verifier_accepts"] and j["fresh_randomness_differs"] and j["fresh_verifies"]
    total = j["stats"][0]["witness_columns"]
    done = [s["columns_transformed_here"] for s in j["stats"]]
    assert sum(done) == total and max(done) < total
    assert all(s["exchange_bytes_received"] > 0 for s in j["stats"])
    # load_pk for the group: same bytes from the key FILE, and every context holds only its share of the extended key columns
    assert j["from_file_same_bytes"] and j["rewritten_equal"] in (True, None)
    E = j["one_context_residency"]["E"]
    assert j["one_context_residency"]["cosets"] == E
    per = max(1, E // world)
    assert [r["cosets"] for r in j["residency"]] == [per] * world
    assert sorted({r["first_coset"] for r in j["residency"]}) == sorted({(q * E // world) if world <= E else q // (world // E) for q in range(world)})
    assert all(r["key_bytes"] < j["one_context_residency"]["key_bytes"] for r in j["residency"])


def test_group_one_context_per_device(hip):
    """ezkl_hip_init(-1) = all visible devices: skipped on a one-GPU box"""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs at least two GPUs")
    world = 1
    while world * 2 <= n:
        world *= 2
    j = _run("all-devices", world, "mlp", 12)
    assert j["world"] == world and j["same_bytes"] and j["verifier_accepts"]
