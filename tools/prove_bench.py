#!/usr/bin/env python3
"""End-to-end `prove` wall-seconds, proved on the GPU and checked by the independent pairing verifier.

    CIRCUIT=einsum K=20 python tools/prove_bench.py [--cpu] [--native] [--pinned]     the reference's accum_einsum_matmul bench circuit
    CIRCUIT=mlp K=17 [LAYERS=9 WIDTH=..] python tools/prove_bench.py ...               MLP over the ezkl gate set (tools/bench_circuits.py)
    CIRCUIT=conv K=17 python tools/prove_bench.py ...                                 examples/conv2d_mnist (BASELINE configs[2])
    K=16 BLOCKS=2 python tools/prove_bench.py ...     (CIRCUIT unset) the round-1 hand-written matmul-accumulation + lookup circuit,
                                                      kept for the two-rank sharding tests
(--cpu also times the CPU oracle backend; --native also times the C++ host prover libezkl_prover.so on the same witness /
randomness and requires its proof to be byte-identical)
The SRS is generated here with a known secret (no public SRS without network, src/pfsys/srs.rs:10-11)."""
import json, os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before torch / HIP initialise (the library's default, csrc/capi.hip ctx_init)
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import ezkl_amd
from ezkl_amd import backend as B, plonk as P
from oracle import pairing as E, binding as ob, verifier as V

k = int(os.environ.get("K", "16")); blocks = int(os.environ.get("BLOCKS", "2")); seg = 256
n = 1 << k
R = P.R
# multi-GPU (torchrun): the MSMs of the proof are sharded by points across the ranks (BASELINE configs[3])
world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
dist, ddev = None, None
if world > 1:
    import torch, torch.distributed as dist
    if "--share-device" in sys.argv:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "--gloo" in sys.argv:
        dist.init_process_group(backend="gloo"); ddev = torch.device("cpu")
    else:
        ddev = torch.device("cuda", local_rank); dist.init_process_group(backend="nccl", device_id=ddev)
ezkl_amd.init(local_rank)

# ---- test SRS with a known secret s, generated on the device (insecure, like gen_srs)
s = 0x1234567890abcdef1234567890abcdef % R
t0 = time.time()
gb, glb = B.gen_srs(k, s)
g, gl = gb.download(), glb.download()
gb.free(); glb.free()
g2 = ((0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed, 0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2),
      (0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa, 0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b))
assert E.g2_on_curve(g2)
s_g2 = E.g2_mul(g2, s)
t_srs = time.time() - t0

# ---- circuit
CIRCUIT = os.environ.get("CIRCUIT", "synthetic")
instances, circuit_info = [], {}
if CIRCUIT != "synthetic":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bench_circuits as BC
    t0 = time.time()
    kw = {}
    if os.environ.get("LAYERS"): kw["layers"] = int(os.environ["LAYERS"])
    if os.environ.get("WIDTH"): kw["width"] = int(os.environ["WIDTH"])
    if os.environ.get("LENGTH"): kw["length"] = int(os.environ["LENGTH"])
    if os.environ.get("MLP_BLOCKS"): kw["blocks"] = int(os.environ["MLP_BLOCKS"])
    if os.environ.get("MLP_FILL"): kw["fill"] = int(os.environ["MLP_FILL"])
    if os.environ.get("MLP_BASE"): kw["base"] = int(os.environ["MLP_BASE"])      # 3 x 2 x blocks advice columns (k = 22, 5 blocks: BASELINE configs[4]'s shape)
    built = BC.build(CIRCUIT, k, gpu=B, **kw)
    cs, fixed, copies, adv, instances = built["cs"], built["fixed"], built["copies"], built["advice"], built["instances"]
    circuit_info = dict(built["info"], **BC.describe(cs), layout_seconds_python=round(time.time() - t0, 1))
    lookups, relu, tbits = cs.lookups, False, 0
SYNTH = CIRCUIT == "synthetic"
gates, perm = [], []
for b in range(blocks if SYNTH else 0):
    a_, b_, acc_ = P.adv(3 * b), P.adv(3 * b + 1), P.adv(3 * b + 2)
    gates.append(P.fix(2 * b) * (acc_ - a_ * b_))
    gates.append(P.fix(2 * b + 1) * (acc_ - P.adv(3 * b + 2, -1) - a_ * b_))
    perm += [("adv", 3 * b), ("adv", 3 * b + 1)]
# one ReLU-style nonlinearity as a two-column mv-lookup: (sel*pre, sel*post) in {(x, relu(x))}, x in [-2^(T-1), 2^(T-1))
relu = SYNTH and os.environ.get("RELU", "1") == "1"
tbits = min(15, k - 2)
if SYNTH:
    lookups = []
if relu:
    pre_c, post_c, sel_c, tin_c, tout_c = 3 * blocks, 3 * blocks + 1, 2 * blocks, 2 * blocks + 1, 2 * blocks + 2
    lookups = [([[P.fix(sel_c) * P.adv(pre_c), P.fix(sel_c) * P.adv(post_c)]], [P.fix(tin_c), P.fix(tout_c)])]
    perm += [("adv", post_c)]
if SYNTH:
    cs = P.ConstraintSystem(k, 3 * blocks + (2 if relu else 0), 2 * blocks + (3 if relu else 0), gates, perm, lookups)
u = cs.usable
rng = np.random.default_rng(1)
def canon_col(v64):
    c = np.zeros((n, 4), np.uint64); c[:, 0] = v64; return c
R2 = np.frombuffer(((1 << 512) % R).to_bytes(32, "little"), np.uint64).copy()
def to_mont_dev(v64):
    d = B.DeviceBuffer.from_numpy(canon_col(v64))
    B.vec_scale(d.ptr, R2, d.ptr, n)          # mont_mul(x, R^2) = x*R
    return d.to_numpy(shape=(n, 4))
if SYNTH:
    adv, fixed, copies = [], [], []
rows = np.arange(n)
for b in range(blocks if SYNTH else 0):
    av = rng.integers(1, 1 << 20, size=n).astype(np.uint64); bv = rng.integers(1, 1 << 20, size=n).astype(np.uint64)
    av[u // 2:u] = av[:u - u // 2]                       # second half re-uses the first half's operands (copy constraints)
    bv[u // 2:u] = bv[:u - u // 2]
    prod = av * bv
    segid = rows // seg
    csum = np.cumsum(prod); start = np.zeros(n, np.uint64); first = (rows % seg == 0)
    base = np.where(first, csum - prod, 0).astype(np.uint64)
    base = np.maximum.accumulate(base)
    accv = csum - base
    s_init = (first & (rows < u)).astype(np.uint64); s_acc = ((~first) & (rows < u)).astype(np.uint64)
    adv += [to_mont_dev(av), to_mont_dev(bv), to_mont_dev(accv)]
    fixed += [to_mont_dev(s_init), to_mont_dev(s_acc)]
    ncopy = min(u - u // 2, 1 << 12)
    for r in range(ncopy):
        copies.append(((2 * b, u // 2 + r), (2 * b, r)))
        copies.append(((2 * b + 1, u // 2 + r), (2 * b + 1, r)))

if relu:
    half = 1 << (tbits - 1)
    def signed_to_canon_limbs(v):                       # integer_rep_to_felt (src/fieldutils.rs:9-17): negatives are r - |x|
        out = np.zeros((n, 4), np.uint64)
        neg = v < 0
        out[~neg, 0] = v[~neg].astype(np.uint64)
        rl = np.frombuffer(R.to_bytes(32, "little"), np.uint64)
        mag = (-v[neg]).astype(np.uint64)
        out[neg] = rl
        out[neg, 0] = rl[0] - mag                       # r - |x| for |x| < 2^32 (no borrow: low limb of r is large)
        return out
    def to_mont_dev_limbs(c):
        d = B.DeviceBuffer.from_numpy(c)
        B.vec_scale(d.ptr, R2, d.ptr, n)
        return d.to_numpy(shape=(n, 4))
    pre = rng.integers(-half, half, size=n); pre[u:] = 0
    post = np.maximum(pre, 0)
    sel = (rows < u).astype(np.int64)
    tin = np.zeros(n, np.int64); tin[: 2 * half] = np.arange(-half, half); tin[2 * half:] = 0
    tout = np.maximum(tin, 0)
    assert 2 * half < u
    adv += [to_mont_dev_limbs(signed_to_canon_limbs(pre * sel)), to_mont_dev_limbs(signed_to_canon_limbs(post * sel))]
    fixed += [to_mont_dev_limbs(signed_to_canon_limbs(sel)), to_mont_dev_limbs(signed_to_canon_limbs(tin)), to_mont_dev_limbs(signed_to_canon_limbs(tout))]

def run(backend_name):
    be = (P.DistGpuBackend(g, gl, k, dist, ddev, shard_columns="--shard-columns" in sys.argv) if world > 1 else P.GpuBackend(g, gl, k)) if backend_name == "hip" else __import__("oracle.cpu_backend", fromlist=["OracleBackend"]).OracleBackend(g, gl, k)
    t0 = time.time(); pk, vk = P.keygen(cs, be, fixed, copies); t_keygen = time.time() - t0
    P.create_proof(pk, be, adv, P.Rng(5), instances=instances)      # warm-up (window tables, twiddles, JIT)
    tm = {}
    t0 = time.time(); proof = P.create_proof(pk, be, adv, P.Rng(5), timings=tm, instances=instances); t_prove = time.time() - t0
    run.timings = {a: round(b, 4) for a, b in tm.items()}
    run.sharded_sweeps = getattr(be, "sharded_sweeps", 0)
    run.sharded_ntt_columns = getattr(be, "sharded_ntt_columns", None) if getattr(be, "owner_of", None) else None
    return vk, proof, t_keygen, t_prove

if not SYNTH:
    # ---- ezkl circuits: the C++ host prover (libezkl_prover.so) on the GPU; the Python host only as the CPU baseline's driver
    from ezkl_amd import native as NV
    gb_, glb_ = B.Bases(g), B.Bases(gl)
    nc = NV.NativeCircuit(cs)
    multi = None
    if world > 1:
        # BASELINE configs[3]: the proof's commit batches divided over the ranks (complete base sets on every rank: by columns, point ranges
        # when a batch has fewer columns than ranks), the quotient sweep by rows; collectives = the library's own RCCL communicator when
        # every rank has a GPU of its own, torch.distributed callbacks otherwise (gloo / a shared device)
        import hashlib, torch
        comm_used = False
        if "--gloo" not in sys.argv and "--share-device" not in sys.argv and not os.environ.get("EZKL_NO_LIB_COMM"):
            ok_ = 1
            try:
                B.comm_init_from_torch(dist, ddev)
            except Exception as e:
                print("library communicator unavailable on rank %d: %r" % (rank, e), file=sys.stderr)
                ok_ = 0
            flag = torch.tensor([ok_], dtype=torch.int32, device=ddev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            comm_used = bool(int(flag[0]))
            if ok_ and not comm_used:
                B.comm_destroy()
        if comm_used: nc.set_shard_comm()
        else: nc.set_shard(dist, ddev)
        nc.set_shard_full_bases(True)
        multi = {"collectives": "libezkl_hip.so RCCL communicator (comm.hip)" if comm_used else "torch.distributed callbacks",
                 "commit_sharding": "by columns (whole MSMs; point ranges when a batch has fewer columns than ranks), complete base sets on every rank",
                 "sweep_sharding": "rows across %d ranks, h all_gathered" % world}
    t0 = time.time(); npk = NV.NativeProvingKey(nc, gb_, fixed, copies); t_keygen = time.time() - t0
    fc, pc, digest = npk.vk()
    vk = P.VerifyingKey(); vk.cs = cs
    vk.fixed_commitments = [P.point_to_ints(p) for p in fc]; vk.sigma_commitments = [P.point_to_ints(p) for p in pc]
    vk.digest = P.vk_digest(vk)
    assert vk.digest == digest
    if "--pinned" in sys.argv and not callable(adv):
        pinned = [B.PinnedArray((n, 4)) for _ in adv]
        for pa, a in zip(pinned, adv):
            pa.array[:] = a
        adv = [pa.array for pa in pinned]
    t0 = time.time(); NV.create_proof(npk, gb_, glb_, adv, seed=5, instances=instances); t_first = time.time() - t0   # first proof of the process: tables, plans, JIT
    reps = int(os.environ.get("REPS", "3"))
    runs = []
    for _ in range(max(1, reps)):                  # the proof is deterministic (seed): the fastest of a few runs is reported, all are listed
        tm_ = {}
        t0 = time.time(); proof = NV.create_proof(npk, gb_, glb_, adv, seed=5, instances=instances, timings=tm_); runs.append((time.time() - t0, tm_))
    t_prove, tm = min(runs, key=lambda r: r[0])
    # --integer-rep: the same witness handed over as int64 columns (8 B per cell across PCIe instead of 32; ezkl_prover_create_proof_fmt) --
    # same seed, so the proof must be the same bytes
    int_rep = None
    if "--integer-rep" in sys.argv and multi is None and not callable(adv) and any(a is not None for a in (built.get("advice_int") or [None])):
        ints, n_int = [], 0
        for a, full in zip(built["advice_int"], adv):
            if a is None:                         # a column with values beyond 64 bits (inverses, ...): stays 32-byte Montgomery words
                class _K:                          # (already pinned above)
                    array = full
                ints.append(_K)
                continue
            pa = B.PinnedArray((n,), dtype=np.int64)
            pa.array[:] = a
            ints.append(pa)
            n_int += 1
        iruns = []
        for _ in range(max(1, reps)):
            tm_ = {}
            t0 = time.time(); iproof = NV.create_proof(npk, gb_, glb_, [p_.array for p_ in ints], seed=5, instances=instances, timings=tm_); iruns.append((time.time() - t0, tm_))
        int_rep = {"prove_seconds_gpu": round(min(r[0] for r in iruns), 4), "runs": [round(r[0], 4) for r in iruns], "same_proof_bytes": iproof == proof,
                   "breakdown_seconds": {a: round(b, 4) for a, b in min(iruns, key=lambda r: r[0])[1].items()},
                   "int64_columns": n_int, "columns": len(ints),
                   "what": "advice columns as int64 IntegerRep values where they fit (8 bytes per cell across PCIe), expanded to Montgomery form on the device"}
    if multi is not None:
        hs_ = torch.tensor(list(hashlib.sha256(proof).digest()), dtype=torch.uint8, device=ddev)
        all_h = [torch.empty_like(hs_) for _ in range(world)]
        dist.all_gather(all_h, hs_)
        tt = torch.tensor([t_prove], dtype=torch.float64, device=ddev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); t_prove = float(tt[0])
        multi.update(all_ranks_same_proof=all(bool((h == all_h[0]).all()) for h in all_h), sharded_sweeps=nc.sharded_sweeps())
        if rank != 0:
            dist.barrier(); dist.destroy_process_group(); sys.exit(0)
    t0 = time.time(); ok = V.verify(vk, (1, 2), g2, s_g2, proof, instances=instances); t_verify = time.time() - t0
    out = {"what": "create_proof (KZG / SHPLONK, Keccak EVM transcript) of an ezkl circuit by libezkl_prover.so over the C ABI", "circuit": circuit_info,
           "proof_bytes": len(proof), "prove_seconds_gpu": round(t_prove, 4), "prove_seconds_gpu_runs": [round(r[0], 4) for r in runs], "first_prove_seconds_gpu": round(t_first, 4), "keygen_seconds_gpu": round(t_keygen, 3),
           "prove_breakdown_seconds": {a: round(b, 4) for a, b in tm.items()}, "verifier_accepts": bool(ok), "verify_seconds_python": round(t_verify, 2),
           "srs_setup_seconds": round(t_srs, 1), "n_gpus": world, "proof_sha256": __import__("hashlib").sha256(proof).hexdigest()[:16]}
    if int_rep is not None:
        out["integer_rep_advice"] = int_rep
    if multi is not None:
        out["multi_gpu"] = multi
        print(json.dumps(out))
        dist.barrier(); dist.destroy_process_group(); sys.exit(0)
    # the quotient sweep of this key, for the roofline: HIP events around the last eval_h launch of the proof = the last coset's sweep
    n_sweep_cols = (cs.n_advice + len({c for c, _ in cs.fixed_queries}) + cs.n_instance + (4 if (cs.perm or cs.lookups) else 0) + cs.n_chunks + len(cs.perm)
                    + 2 * len(cs.lookups))
    free_b, total_b = B.mem_info()
    out["hbm_in_use_gib_after_prove"] = round((total_b - free_b) / 2**30, 2)          # keys, SRS + window tables, and what the column pool keeps parked
    out["hbm_pool_high_water_gib"] = round(B.pool_stats()["live_peak"] / 2**30, 2)     # most column bytes alive at once during these proofs
    import resource
    out["host_peak_rss_gib"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20, 2)
    out["sweep_kernel"] = {"kernel": "evalh_jit", "avg_launch_ms": round(B.last_kernel_ms("eval_h"), 4), "rows_per_launch": n, "launches_per_proof": 1 << (cs.ext_k - k),
                           "columns": n_sweep_cols, "algorithmic_bytes_per_launch": 32 * (n_sweep_cols + 1) * n}
    try:
        ins, prods, slots, kernels = npk.sweep_stats()
        out["sweep_kernel"].update({"instructions_per_row": ins, "products_per_row": prods, "products_per_launch": prods * n, "kernels": kernels})
    except Exception as e:
        out["sweep_kernel"]["stats_error"] = repr(e)[:100]
    if "--cold" in sys.argv:
        # the one-shot `ezkl prove`: artefact files on disk, a FRESH process (tools/prove_cold.py) reads SRS + pk into HBM and proves once
        import subprocess, tempfile, shutil
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import prove_cold
        d = tempfile.mkdtemp(prefix="ezkl_cold_", dir=os.environ.get("EZKL_COLD_DIR", "/tmp"))
        try:
            if callable(adv):
                seen = {}
                def recording(phase, chal):
                    seen[phase] = adv(phase, chal)
                    return seen[phase]
                NV.create_proof(npk, gb_, glb_, recording, seed=5, instances=instances)
                by_phase = {ph: {c: np.ascontiguousarray(a) for c, a in cols.items()} for ph, cols in seen.items()}
            else:
                by_phase = {0: {c: np.ascontiguousarray(a) for c, a in enumerate(adv)}}
            t0 = time.time()
            prove_cold.write(d, k, g, gl, g2, s_g2, cs, npk.to_bytes(), by_phase, instances, 5)
            t_write = time.time() - t0
            size = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
            B.pool_trim()                                               # the children share this device: give the parked columns back first
            gap = float(os.environ.get("EZKL_COLD_GAP_S", "0"))          # let the driver finish releasing the previous process's VRAM first
            def cold_child(extra_env):
                time.sleep(gap)
                r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "prove_cold.py"), "run", d],
                                   capture_output=True, text=True, timeout=900, env=dict(os.environ, **extra_env))
                return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]) if r.returncode == 0 else {"error": r.stderr[-400:]}
            first_ever = cold_child({"EZKL_HIP_CACHE_DIR": "off"})       # no code-object cache: the very first prove of this circuit on a machine
            cj = cold_child({})                                          # every later one: the gate programs' code objects come from disk
            cj["first_ever_cold_seconds"] = first_ever.get("cold_seconds")
            cj["first_ever_create_proof_seconds"] = (first_ever.get("stages") or {}).get("create_proof")
            if "proof_sha256" in cj:
                cj["same_proof_as_warm"] = cj["proof_sha256"] == __import__("hashlib").sha256(proof).hexdigest()[:16]
            cj["artifact_bytes"] = size; cj["artifact_write_seconds"] = round(t_write, 2)
            if os.environ.get("EZKL_COLD_AB"):                           # A/B of an environment switch on the SAME artefacts and box: VAR=VALUE
                var, val = os.environ["EZKL_COLD_AB"].split("=", 1)
                ab = {"default": [], os.environ["EZKL_COLD_AB"]: []}
                for _ in range(int(os.environ.get("EZKL_COLD_AB_REPS", "3"))):
                    for name, env_ in (("default", {}), (os.environ["EZKL_COLD_AB"], {var: val})):
                        c_ = cold_child(env_)
                        ab[name].append({"cold_seconds": c_.get("cold_seconds"), "pk_read_to_hbm": (c_.get("stages") or {}).get("pk_read_to_hbm"),
                                         "create_proof": (c_.get("stages") or {}).get("create_proof"), "device_init": (c_.get("stages") or {}).get("device_init")})
                cj["ab"] = ab
            out["cold"] = cj
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if "--cpu" in sys.argv:
        # the SAME create_proof on the host cores: Python host + the C oracle's OpenMP kernels (oracle/cpu_backend.py), same witness and
        # randomness as a GPU proof made through the rng callback, bytes compared
        from oracle.cpu_backend import OracleBackend
        proof_g = NV.create_proof(npk, gb_, glb_, adv, rng=P.Rng(5), instances=instances)
        cpu = OracleBackend(g, gl, k)
        t0 = time.time(); pk_c, _ = P.keygen(cs, cpu, fixed, copies); t_ck = time.time() - t0
        ctm = {}
        t0 = time.time(); proof_c = P.create_proof(pk_c, cpu, adv, P.Rng(5), timings=ctm, instances=instances); t_cpu = time.time() - t0
        out.update({"prove_seconds_cpu": round(t_cpu, 3), "cpu_threads": ob.num_threads(), "cpu_keygen_seconds": round(t_ck, 2),
                    "cpu_prover": "the same create_proof, Python host + C oracle kernels (OpenMP): a CPU restatement, not halo2",
                    "cpu_breakdown_seconds": {a: round(b, 3) for a, b in ctm.items()}, "proofs_identical": proof_c == proof_g})
    print(json.dumps(out))
    sys.exit(0)

vk, proof, t_keygen, t_prove = run("hip")
native_multi = None
if world > 1 and "--native" in sys.argv:
    # the C++ host prover with its MSMs sharded the same way (ezkl_prover_cs_set_shard): every rank holds its slice of the SRS
    from ezkl_amd import native as NV
    import hashlib, torch
    if "--pinned" in sys.argv and not callable(adv):
        pinned = [B.PinnedArray((n, 4)) for _ in adv]
        for pa, a in zip(pinned, adv):
            pa.array[:] = a
        adv = [pa.array for pa in pinned]
    nc = NV.NativeCircuit(cs)
    # one GPU per rank over RCCL: shard through the library's OWN communicator (csrc/comm.hip: partial points folded and h all_gathered
    # on device pointers, no Python in the data path); the ranks agree on whether it came up, otherwise the torch.distributed callbacks
    comm_used = False
    if "--gloo" not in sys.argv and "--share-device" not in sys.argv and not os.environ.get("EZKL_NO_LIB_COMM"):
        import torch
        ok = 1
        try:
            B.comm_init_from_torch(dist, ddev)
        except Exception as e:
            print("library communicator unavailable on rank %d: %r" % (rank, e), file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=ddev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        comm_used = bool(int(flag[0]))
        if ok and not comm_used:
            B.comm_destroy()
    lo, hi = nc.set_shard_comm() if comm_used else nc.set_shard(dist, ddev)
    full_bases = "--slice-bases" not in sys.argv
    if full_bases:       # every rank holds the whole SRS (and its window tables): commit batches divide by columns, see ezkl_prover.h
        nc.set_shard_full_bases(True)
        gb_, glb_ = B.Bases(np.ascontiguousarray(g)), B.Bases(np.ascontiguousarray(gl))
    else:
        gb_, glb_ = B.Bases(np.ascontiguousarray(g[lo:hi])), B.Bases(np.ascontiguousarray(gl[lo:hi]))
    npk = NV.NativeProvingKey(nc, gb_, fixed, copies)
    nproof = NV.create_proof(npk, gb_, glb_, adv, rng=P.Rng(5), instances=instances)   # warm-up; same randomness as the Python host above
    ltm = {}
    dist.barrier()
    t0 = time.time(); lproof = NV.create_proof(npk, gb_, glb_, adv, seed=5, timings=ltm, instances=instances); t_native = time.time() - t0
    hs_ = torch.tensor(list(hashlib.sha256(lproof).digest()), dtype=torch.uint8, device=ddev)
    all_h = [torch.empty_like(hs_) for _ in range(world)]
    dist.all_gather(all_h, hs_)
    tt = torch.tensor([t_native], dtype=torch.float64, device=ddev); dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    native_multi = {"prove_seconds_library_rng": round(float(tt[0]), 4),
                    "commit_sharding": ("by columns (whole MSMs; point ranges when a batch has fewer columns than ranks), complete base sets on every rank"
                                        if full_bases else "by points, 1/%d of the SRS per rank" % world), "proof_identical_to_python_prover": nproof == proof,
                    "sharded_sweeps": nc.sharded_sweeps(), "gather_on_device_pointers": bool(getattr(nc, "direct_gather", False)) or comm_used,
                    "collectives": "libezkl_hip.so RCCL communicator (comm.hip)" if comm_used else "torch.distributed callbacks",
                    "all_ranks_same_proof": all(bool((h == all_h[0]).all()) for h in all_h), "library_rng_proof": lproof,
                    "breakdown_seconds_library_rng": {a: round(b, 4) for a, b in ltm.items()}}
if world > 1:
    import hashlib, torch
    hs_ = torch.tensor(list(hashlib.sha256(proof).digest()), dtype=torch.uint8, device=ddev)
    all_h = [torch.empty_like(hs_) for _ in range(world)]
    dist.all_gather(all_h, hs_)
    assert all(bool((h == all_h[0]).all()) for h in all_h), "ranks produced different proofs"
    tt = torch.tensor([t_prove], dtype=torch.float64, device=ddev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); t_prove = float(tt[0])
    if rank != 0:
        dist.barrier(); dist.destroy_process_group(); sys.exit(0)
t0 = time.time(); ok = V.verify(vk, (1, 2), g2, s_g2, proof, instances=instances); t_verify = time.time() - t0
out = {"what": "ezkl_amd.plonk prove (gates + permutation + mv-lookup, KZG/SHPLONK, Keccak EVM transcript): " +
               (circuit_info.get("circuit") or "matmul-accumulation blocks + ReLU lookup"), "circuit": circuit_info,
       "lookups": len(lookups), "lookup_table_rows": (1 << tbits) if relu else 0,
       "n_gpus": world, "msm_sharding": "points across %d rank(s), all_gather of 64-B partials per commit batch" % world,
       "sweep_sharding": "rows across %d rank(s) (Python host), all_gather of h: %d sharded sweep(s)" % (world, run.sharded_sweeps),
       "ntt_sharding": ("columns round-robin across %d ranks: this rank transformed %s columns; sweep windows by all-to-all" % (world, run.sharded_ntt_columns)) if run.sharded_ntt_columns else "replicated",
       "proof_sha256": __import__("hashlib").sha256(proof).hexdigest()[:16],
       "k": k, "advice_columns": cs.n_advice, "fixed_columns": cs.n_fixed, "degree": cs.degree, "ext_k": cs.ext_k, "copies": len(copies),
       "proof_bytes": len(proof), "prove_seconds_gpu": round(t_prove, 4), "keygen_seconds_gpu": round(t_keygen, 3),
       "prove_breakdown_seconds": run.timings, "verifier_accepts": bool(ok), "verify_seconds_python": round(t_verify, 2), "srs_setup_seconds": round(t_srs, 1)}
if native_multi is not None:
    lproof = native_multi.pop("library_rng_proof")
    native_multi["library_rng_proof_verifies"] = bool(V.verify(vk, (1, 2), g2, s_g2, lproof, instances=instances))
    out["native_prover"] = native_multi
if "--native" in sys.argv and world == 1:
    from ezkl_amd import native as NV
    if "--pinned" in sys.argv and not callable(adv):     # witness in page-locked memory (ezkl_hip_host_malloc): uploads run under the commits
        pinned = [B.PinnedArray((n, 4)) for _ in adv]
        for pa, a in zip(pinned, adv):
            pa.array[:] = a
        adv = [pa.array for pa in pinned]
    gb_, glb_ = B.Bases(g), B.Bases(gl)
    t0 = time.time(); npk = NV.NativeProvingKey(NV.NativeCircuit(cs), gb_, fixed, copies); t_nkeygen = time.time() - t0
    NV.create_proof(npk, gb_, glb_, adv, rng=P.Rng(5), instances=instances)            # warm-up
    ntm = {}
    t0 = time.time(); nproof = NV.create_proof(npk, gb_, glb_, adv, rng=P.Rng(5), timings=ntm, instances=instances); t_native = time.time() - t0
    ltm = {}
    t0 = time.time(); lproof = NV.create_proof(npk, gb_, glb_, adv, seed=5, timings=ltm, instances=instances); t_native_librng = time.time() - t0
    out["native_prover"] = {"prove_seconds": round(t_native, 4), "prove_seconds_library_rng": round(t_native_librng, 4), "keygen_seconds": round(t_nkeygen, 3),
                            "proof_identical_to_python_prover": nproof == proof, "library_rng_proof_verifies": bool(V.verify(vk, (1, 2), g2, s_g2, lproof, instances=instances)),
                            "breakdown_seconds": {a: round(b, 4) for a, b in ntm.items()},
                            "breakdown_seconds_library_rng": {a: round(b, 4) for a, b in ltm.items()}}
if "--cpu-kernels" in sys.argv:
    # the C oracle (OpenMP, all host cores) timed on ONE instance of each kernel class at this size, scaled by the call
    # counts the prover actually issued: a bounded CPU sample, not a CPU prover
    nz, nl = cs.n_chunks, len(cs.lookups)
    counts = {"msm": cs.n_advice + nl + nz + nl + 1 + (cs.degree - 1) + 2,
              "intt": cs.n_advice + nz + 2 * nl, "coset_ntt": cs.n_advice + nz + 2 * nl + 1}
    rs = np.random.default_rng(3)
    col = rs.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); col[:, 3] &= np.uint64((1 << 61) - 1)
    ob.msm(col[:4096], gl[:4096]); ob.lagrange_to_coeff(col[: 1 << 12], 12)      # warm the OpenMP pool
    cf = ob.lagrange_to_coeff(col, k); ob.coeff_to_extended(cf, k, cs.ext_k)        # and the page cache of the work buffers
    t0 = time.time(); ob.msm(col, gl); t_msm = time.time() - t0
    t0 = time.time(); cf = ob.lagrange_to_coeff(col, k); t_intt = time.time() - t0
    t0 = time.time(); ob.coeff_to_extended(cf, k, cs.ext_k); t_coset = time.time() - t0
    est = counts["msm"] * t_msm + counts["intt"] * t_intt + counts["coset_ntt"] * t_coset
    out["cpu_kernel_sample"] = {"threads": ob.num_threads(), "one_msm_s": round(t_msm, 3), "one_intt_s": round(t_intt, 3),
                                "one_coset_ntt_s": round(t_coset, 3), "call_counts": counts,
                                "msm_plus_ntt_seconds_at_call_counts": round(est, 2),
                                "note": "C restatement kernels only (no sweep, no scans, no host work): a lower bound for a CPU prove of this circuit"}
if "--cpu" in sys.argv:
    vk2, proof2, _, t_cpu = run("oracle")
    out["prove_seconds_cpu_oracle"] = round(t_cpu, 3); out["cpu_threads"] = ob.num_threads(); out["proofs_identical"] = proof2 == proof
print(json.dumps(out))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
