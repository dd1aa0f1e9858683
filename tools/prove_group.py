#!/usr/bin/env python3
"""One process, several GPUs: the prover group (include/ezkl_prover.h ezkl_prover_group_*) on the contexts of libezkl_hip.so.

    CIRCUIT=mlp K=20 python tools/prove_group.py                 one context per visible device (ezkl_hip_init(-1))
    CONTEXTS=0,0 CIRCUIT=mlp K=20 python tools/prove_group.py     explicit context table (here: two contexts on device 0 -- a one-GPU box)
Prints the group's proof time next to the one-context prover's, checks that the bytes are equal and that the product's verifier accepts."""
import hashlib, json, os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.pop("LOCAL_RANK", None)
from ezkl_amd import lib as L, backend as B, native as NV, plonk as P
if os.environ.get("CONTEXTS"):
    B.contexts_configure([int(x) for x in os.environ["CONTEXTS"].split(",")])
else:
    L.check(L.load().ezkl_hip_init(-1), "ezkl_hip_init")
n_ctx = B.context_count()
import bench_circuits as BC
CIRCUIT, k = os.environ.get("CIRCUIT", "mlp"), int(os.environ.get("K", "12"))
kw = {}
if os.environ.get("MLP_BLOCKS"): kw["blocks"] = int(os.environ["MLP_BLOCKS"])
if os.environ.get("MLP_FILL"): kw["fill"] = int(os.environ["MLP_FILL"])
if os.environ.get("MLP_BASE"): kw["base"] = int(os.environ["MLP_BASE"])
built = BC.build(CIRCUIT, k, gpu=B, **kw)
cs, fixed, copies, adv, instances, info = built["cs"], built["fixed"], built["copies"], built["advice"], built["instances"], built["info"]
n = 1 << k
if "--pinned" in sys.argv:
    pinned = [B.PinnedArray((n, 4)) for _ in adv]
    for pa, a in zip(pinned, adv):
        pa.array[:] = a
    adv = [pa.array for pa in pinned]
s = 0x1234567890abcdef1234567890abcdef % P.R
gb, glb = B.gen_srs(k, s)
g, gl = gb.download(), glb.download()
reps = int(os.environ.get("REPS", "3"))
# one context
npk = NV.NativeProvingKey(NV.NativeCircuit(cs), gb, fixed, copies)
NV.create_proof(npk, gb, glb, adv, seed=5, instances=instances)
t_one = []
for _ in range(reps):
    t0 = time.time(); want = NV.create_proof(npk, gb, glb, adv, seed=5, instances=instances); t_one.append(time.time() - t0)
npk.free(); gb.free(); glb.free()
# the group
grp = NV.NativeGroup(cs, n_ctx)
t0 = time.time(); grp.load_srs(g, gl); t_srs = time.time() - t0
t0 = time.time(); grp.keygen(fixed, copies); t_keygen = time.time() - t0
grp.create_proof(adv, seed=5, instances=instances)
runs = []
for _ in range(reps):
    tm, st = {}, []
    t0 = time.time(); got = grp.create_proof(adv, seed=5, instances=instances, timings=tm, stats=st); runs.append((time.time() - t0, tm, st))
t_grp, tm, st = min(runs, key=lambda r: r[0])
ok = NV.verify_proof(grp.pk(0), NV.g2_mul_generator(1), NV.g2_mul_generator(s), got, instances)
print(json.dumps({"what": "prover group: %d context(s) in one process, one host thread each (devices %s)" % (grp.world, [int(L.load().ezkl_hip_context_device(i)) for i in range(grp.world)]),
                  "circuit": dict(info, k=k, advice_columns=cs.n_advice, lookups=len(cs.lookups), permutation_columns=len(cs.perm), ext_k=cs.ext_k),
                  "contexts": grp.world, "same_bytes_as_one_context": got == want, "verifier_accepts": bool(ok), "proof_sha256": hashlib.sha256(got).hexdigest()[:16],
                  "prove_seconds_one_context": round(min(t_one), 4), "prove_seconds_group": round(t_grp, 4), "group_breakdown_seconds_max_over_contexts": {a: round(b, 4) for a, b in tm.items()},
                  "per_context": st, "group_srs_seconds": round(t_srs, 2), "group_keygen_seconds": round(t_keygen, 2)}))
grp.free()
