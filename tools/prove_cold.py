#!/usr/bin/env python3
"""The one-shot `ezkl prove` a user waits for (/root/reference/src/execute.rs:1575-1627): a FRESH process reads the SRS file and the
proving key (halo2 raw-bytes layouts, ezkl_amd/codecs.py) from disk, moves them to HBM, proves once and writes proof.json.  Nothing is
warm: library init, base-set upload, window-table precompute, NTT plans and the gate-program JIT are all inside the numbers.

    python tools/prove_cold.py write DIR     (called by tools/prove_bench.py --artifacts DIR: not timed)
    python tools/prove_cold.py run DIR       -> one JSON line with the stage split

Artefacts in DIR: kzg.srs, pk.key, circuit.ezcs (the constraint-system blob = what re-running `configure` yields in ezkl),
adv_<phase>_<column>.npy (advice columns as the prover's synthesis would produce them), meta.json (instances, seed)."""
import json
import os
import sys
import time

T_START = time.time()
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def run(d):
    t_imp0 = time.time()
    import ctypes as C
    import numpy as np
    import ezkl_amd
    from ezkl_amd import backend as B, codecs, native as NV, plonk as P
    t = {"python_imports": time.time() - t_imp0}
    meta = json.load(open(os.path.join(d, "meta.json")))
    t0 = time.time(); ezkl_amd.init(0); t["device_init"] = time.time() - t0
    t0 = time.time()
    srs = codecs.read_srs(open(os.path.join(d, "kzg.srs"), "rb").read())
    t["srs_read"] = time.time() - t0
    t0 = time.time(); bg, bgl = B.Bases(srs["g"]).prepare(), B.Bases(srs["g_lagrange"]).prepare(); t["srs_to_hbm"] = time.time() - t0   # tables build in the background
    t0 = time.time()
    blob = open(os.path.join(d, "circuit.ezcs"), "rb").read()
    h = C.c_void_p()
    NV._check(NV.load().ezkl_prover_cs_parse(blob, C.c_size_t(len(blob)), C.byref(h)), "ezkl_prover_cs_parse")
    circ = NV.NativeCircuit.__new__(NV.NativeCircuit)
    circ.h = h
    circ.cs = type("CS", (), dict(n=1 << meta["k"], n_fixed=meta["n_fixed"], perm=[None] * meta["n_perm"], advice_phase=meta["advice_phase"]))()
    t["circuit_parse"] = time.time() - t0
    t0 = time.time()
    if os.environ.get("EZKL_COLD_FULL_PK_READ"):     # every section of the file, as halo2's ProvingKey::read does
        pk = NV.NativeProvingKey.from_bytes(circ, open(os.path.join(d, "pk.key"), "rb").read())
    else:
        pk = NV.NativeProvingKey.from_file(circ, os.path.join(d, "pk.key"))
    t["pk_read_to_hbm"] = time.time() - t0
    t0 = time.time()
    files = sorted(f for f in os.listdir(d) if f.startswith("adv_") and f.endswith(".npy"))
    phases = sorted({int(f.split("_")[1]) for f in files})
    cols = {ph: {int(f.split("_")[2][:-4]): np.load(os.path.join(d, f), mmap_mode="r") for f in files if f.startswith("adv_%d_" % ph)} for ph in phases}
    inst = [[int(x) for x in col] for col in meta["instances"]]
    t["witness_read"] = time.time() - t0
    if len(phases) > 1:
        advice = lambda phase, chal: cols[phase]              # the second-phase columns were synthesised for the seed's challenges
    else:
        advice = [cols[0][c] for c in range(len(cols[0]))]
    tm = {}
    t0 = time.time(); proof = NV.create_proof(pk, bg, bgl, advice, seed=meta["seed"], instances=inst, timings=tm); t["create_proof"] = time.time() - t0
    t0 = time.time()
    open(os.path.join(d, "proof.json"), "w").write(codecs.write_proof_json(proof, inst))
    t["proof_write"] = time.time() - t0
    total = time.time() - T_START
    print(json.dumps({"cold_seconds": round(total, 3), "cold_seconds_without_python_imports": round(total - t["python_imports"], 3),
                      "stages": {a: round(b, 4) for a, b in t.items()}, "create_proof_breakdown": {a: round(b, 4) for a, b in tm.items()},
                      "proof_sha256": __import__("hashlib").sha256(proof).hexdigest()[:16], "proof_bytes": len(proof)}))


def write(d, k, g, gl, g2_ints, s_g2_ints, cs, pk_bytes, advice_cols_by_phase, instances, seed):
    """called from prove_bench.py with everything a cold run needs"""
    import numpy as np
    from ezkl_amd import codecs, plonk as P
    os.makedirs(d, exist_ok=True)
    M, Q = 1 << 256, P.Q
    def g2b(pt):
        (x0, x1), (y0, y1) = pt
        return b"".join((v * M % Q).to_bytes(32, "little") for v in (x0, x1, y0, y1))
    open(os.path.join(d, "kzg.srs"), "wb").write(codecs.write_srs(dict(k=k, g=g, g_lagrange=gl, g2=g2b(g2_ints), s_g2=g2b(s_g2_ints))))
    open(os.path.join(d, "pk.key"), "wb").write(pk_bytes)
    open(os.path.join(d, "circuit.ezcs"), "wb").write(P.serialize_cs(cs))
    for ph, cols in advice_cols_by_phase.items():           # the witness columns synthesis hands to create_proof, one mappable file each
        for c, a in cols.items():
            np.save(os.path.join(d, "adv_%d_%d.npy" % (ph, c)), np.ascontiguousarray(a, np.uint64))
    json.dump(dict(k=k, n_fixed=cs.n_fixed, n_perm=len(cs.perm), advice_phase=cs.advice_phase, instances=[[int(v) for v in col] for col in instances], seed=seed),
              open(os.path.join(d, "meta.json"), "w"))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
