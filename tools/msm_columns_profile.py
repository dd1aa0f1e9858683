#!/usr/bin/env python3
"""TRUE (serial, nothing else on the GPU) per-kernel durations of the MSMs a real proof runs: every advice column of a cached ezkl
circuit (tools/bench_circuits.py), one m(X)-like multiplicity column, one permutation-product-like column and a uniform column, each
committed alone on the library stream.  Run under rocprofv3 and reduce the trace with the same script:

    rocprofv3 --kernel-trace -d OUT -- python tools/msm_columns_profile.py run       (on the GPU box)
    python tools/msm_columns_profile.py reduce OUT/<...>.db                           (per-column, per-kernel table)
"""
import os, sys, json
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
REPS = 3
KERNELS = ["msm_hist_kernel", "msm_hist_scan_kernel", "msm_part_scan_kernel", "msm_partition_kernel", "msm_binsort_kernel", "msm_bigsort_count_kernel",
           "msm_bigsort_scatter_kernel", "msm_accumulate_kernel", "msm_fixup_boundary_kernel", "msm_fixup_heavy1_kernel", "msm_fixup_heavy2_kernel",
           "msm_reduce1_kernel", "msm_reduce2_kernel", "msm_planes_kernel"]


def run():
    import ezkl_amd
    from ezkl_amd import backend as B
    import bench_circuits as BC
    ezkl_amd.init(0)
    k = int(os.environ.get("K", "20")); n = 1 << k
    built = BC.build(os.environ.get("CIRCUIT", "mlp"), k, gpu=B)
    adv = built["advice"]
    bases = B.Bases.generate(0x657a6b6c, n)
    rng = np.random.default_rng(1)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import rand_fr
    cols = [("uniform", rand_fr(rng, n))]
    for c, a in enumerate(adv):
        cols.append(("advice%d" % c, np.ascontiguousarray(a)))
    # m(X)-like: zero except a few thousand small counts; z(X)-like: long constant runs of random values
    one = np.asarray(adv[0][:1]).copy()
    m = np.zeros((n, 4), np.uint64)
    idx = rng.choice(n, 20000, replace=False)
    small = rand_fr(rng, 64)
    m[idx] = small[rng.integers(0, 64, 20000)]
    cols.append(("m_like", m))
    z = np.repeat(rand_fr(rng, 64), n // 64, axis=0)
    cols.append(("z_runs", np.ascontiguousarray(z)))
    names = []
    B.msm_g1_dev(bases, B.DeviceBuffer.from_numpy(cols[0][1]).ptr, n)            # tables, first-use allocations
    for name, a in cols:
        d = B.DeviceBuffer.from_numpy(a)
        for _ in range(REPS):
            B.msm_g1_dev(bases, d.ptr, n)
        nz = int((np.asarray(a) != 0).any(axis=1).sum())
        print(json.dumps({"column": name, "device_ms": round(B.last_kernel_ms("msm"), 4), "accumulate_ms": round(B.last_kernel_ms("msm_accumulate"), 4), "nonzero_rows": nz}), flush=True)
        names.append(name)
    print(json.dumps({"order": names, "reps": REPS}))


def reduce(db_path):
    import sqlite3
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    ev = sorted((s, e, nm) for nm, s, e in cur.execute("select %s, start, end from kernels" % name_col))
    msms, curm = [], None
    for s, e, nm in ev:
        short = nm.split("(")[0].replace("ezkl::", "").replace("void ", "").replace("msm_fixup_boundary_tree_kernel", "msm_fixup_boundary_kernel")
        if short == "msm_hist_kernel":
            curm = {}
            msms.append(curm)
        if curm is not None and short in KERNELS:
            curm[short] = curm.get(short, 0.0) + (e - s) / 1e3
    order = None
    if len(sys.argv) > 3:
        for line in open(sys.argv[3]):
            if line.startswith('{"order"'):
                order = json.loads(line)["order"]
    msms = msms[1:]                                   # the warm-up MSM
    print("%-12s" % "column" + "".join("%9s" % k.replace("msm_", "").replace("_kernel", "")[:8] for k in KERNELS) + "%9s%9s" % ("non-acc", "total"))
    for i in range(REPS - 1, len(msms), REPS):        # the last repetition of each column
        m = msms[i]
        nonacc = sum(v for k, v in m.items() if k != "msm_accumulate_kernel")
        label = order[i // REPS] if order and i // REPS < len(order) else "col%d" % (i // REPS)
        print("%-12s" % label + "".join("%9.1f" % m.get(k, 0.0) for k in KERNELS) + "%9.1f%9.1f" % (nonacc, nonacc + m.get("msm_accumulate_kernel", 0.0)))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "reduce":
        reduce(sys.argv[2])
    else:
        run()
