#!/usr/bin/env python3
"""A/B of the 2^20-point MSM (the bench's step) across library builds and run-time switches.
    python tools/msm_ab.py            -> one line per (library, switches): device ms of the whole MSM, of the accumulate kernel, wall ms per step,
                                         and whether the result equals the oracle's (2^16 and 2^20 points)
Libraries: ab/libezkl_hip_<variant>.so built by tools/build_variants.sh (compile-time macros), selected through EZKL_HIP_LIB in a child."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
import ezkl_amd
from ezkl_amd import backend as B
from oracle import binding as ob
ezkl_amd.init(0)
rng = np.random.default_rng(7)
def rand(n):
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 61) - 1); return a
res = {}
ok = True
for k in (12, 16):
    n = 1 << k
    s = rand(n); pts = ob.gen_bases(0x657a6b6c, n)
    bases = B.Bases(pts)
    ok = ok and bool((B.msm_g1(bases, s) == ob.msm(s, pts)).all())
    w = s.copy(); w[:, 1:] = 0; w[:, 0] &= np.uint64(0xfffff)      # witness-shaped: 20-bit values
    w[::3] = 0; w[1::7] = s[1::7]
    ok = ok and bool((B.msm_g1(bases, w) == ob.msm(w, pts)).all())
    c = np.repeat(s[:1], n, axis=0)                                 # a constant column: the heavy paths
    ok = ok and bool((B.msm_g1(bases, c) == ob.msm(c, pts)).all())
    bases.free()
n = 1 << 20
bases = B.Bases.generate(0x657a6b6c, n)
s = rand(n)
sc = B.DeviceBuffer.from_numpy(s)
got = B.msm_g1_dev(bases, sc.ptr, n)
if os.environ.get("AB_CHECK20"):
    pts = ob.gen_bases(0x657a6b6c, n)
    ok = ok and bool((np.asarray(got) == ob.msm(s, pts)).all())
for _ in range(200): B.msm_g1_dev(bases, sc.ptr, n)
ms, acc = [], []
B.synchronize(); t0 = time.perf_counter()
K = 50
for _ in range(K):
    B.msm_g1_dev(bases, sc.ptr, n); ms.append(B.last_kernel_ms("msm")); acc.append(B.last_kernel_ms("msm_accumulate"))
B.synchronize(); wall = (time.perf_counter() - t0) / K * 1e3
print(json.dumps({"ok": ok, "msm_ms": float(np.mean(ms)), "msm_ms_min": float(np.min(ms)), "acc_ms": float(np.mean(acc)), "wall_ms": wall, "sha": __import__("hashlib").sha256(np.asarray(got).tobytes()).hexdigest()[:12]}))
"""

def run(label, lib, env):
    e = dict(os.environ, **env)
    if lib: e["EZKL_HIP_LIB"] = lib
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=e, capture_output=True, text=True, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    print("%-44s %s" % (label, line[-1] if line else "FAILED rc=%d %s" % (p.returncode, p.stderr[-400:])), flush=True)

if __name__ == "__main__":
    variants = [("default", None)]
    abdir = os.path.join(ROOT, "ab")
    for f in sorted(os.listdir(abdir)) if os.path.isdir(abdir) else []:
        if f.startswith("libezkl_hip_") and f.endswith(".so"):
            variants.append((f[len("libezkl_hip_"):-3], os.path.join(abdir, f)))
    first = True
    for name, lib in variants:
        for coop in os.environ.get("AB_COOP", "7,0").split(","):
            run("%s COOP=%s" % (name, coop), lib, {"EZKL_MSM_COOP": coop, **({"AB_CHECK20": "1"} if first else {})})
            first = False
    for extra in sys.argv[1:]:            # KEY=VALUE[,KEY=VALUE] settings on the default library
        run(extra, None, dict(kv.split("=", 1) for kv in extra.split(",")))
