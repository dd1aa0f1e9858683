#!/usr/bin/env python3
"""The C++ host prover (libezkl_prover.so) with columns and arguments BY OWNER across the ranks of one node (SURVEY.md §8(e);
include/ezkl_prover.h "multi-GPU, the full form"): every witness column is transformed (iNTT + cosets) and committed by one rank only,
lookup / permutation arguments are computed by their owner, the quotient sweep is divided into row units fed by ONE all-to-all, h is
all_gathered, evaluations travel as scalars and SHPLONK as two 64-byte folds.  Every rank must emit the bytes of the one-GPU proof.

    CIRCUIT=mlp K=20 python tools/prove_multi.py                                             one rank: the reference bytes
    python -m torch.distributed.run --nproc-per-node N ... tools/prove_multi.py              N GPUs: RCCL through the library communicator
    ... tools/prove_multi.py --gloo --share-device                                           tests: N ranks on one GPU, gloo moving the data
    --replicated: the round-2 mode (commit batches by columns, everything else replicated) for comparison
CIRCUIT: fixture (the reference's k = 6 circuit, tests/fixture_k6.py), mlp, einsum, conv (tools/bench_circuits.py)."""
import hashlib, json, os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ezkl_amd
from ezkl_amd import backend as B, plonk as P, native as NV

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
CIRCUIT, k = os.environ.get("CIRCUIT", "mlp"), int(os.environ.get("K", "10"))
if CIRCUIT == "fixture":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixture_k6 as FX
    fx = FX.load()
    k, cs = 6, fx["cs"]
    adv_i, inst, _ = FX.witness(fx)
    fixed, adv, instances = FX.mont_cols(fx["fixed"]), FX.mont_cols(adv_i), inst
    copies = FX.copies_of(FX.copy_cycles(fx["pk"]))
    info = {"circuit": "the reference's fixture circuit (tests/assets: Gemm 3->4 + ReLU, k=6, 35 lookups, 32 permutation columns)"}
else:
    import bench_circuits as BC
    kw = {a.lower(): int(os.environ[a]) for a in ("LAYERS", "WIDTH", "LENGTH") if os.environ.get(a)}
    if os.environ.get("MLP_BLOCKS"): kw["blocks"] = int(os.environ["MLP_BLOCKS"])
    if os.environ.get("MLP_FILL"): kw["fill"] = int(os.environ["MLP_FILL"])
    if os.environ.get("MLP_BASE"): kw["base"] = int(os.environ["MLP_BASE"])
    built = BC.build(CIRCUIT, k, gpu=B, **kw)
    cs, fixed, copies, adv, instances, info = built["cs"], built["fixed"], built["copies"], built["advice"], built["instances"], built["info"]
n = 1 << k
s = 0x1234567890abcdef1234567890abcdef % P.R
gb, glb = B.gen_srs(k, s)                      # every rank holds the complete SRS (288 GB of HBM: base sets + window tables are < 4 GB at k = 22)
nc = NV.NativeCircuit(cs)
mode, collectives = "one rank", None
if world > 1:
    import torch
    comm_used = False
    if "--gloo" not in sys.argv and "--share-device" not in sys.argv and not os.environ.get("EZKL_NO_LIB_COMM"):
        ok_ = 1
        try:
            B.comm_init_from_torch(dist, ddev)
        except Exception as e:
            print("library communicator unavailable on rank %d: %r" % (rank, e), file=sys.stderr); ok_ = 0
        flag = torch.tensor([ok_], dtype=torch.int32, device=ddev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        comm_used = bool(int(flag[0]))
        if ok_ and not comm_used:
            B.comm_destroy()
    if comm_used:
        nc.set_shard_comm()                    # fold, gather, allgather_host and the all-to-all: the library's RCCL communicator
    else:
        # the callbacks' process group tries itself out first (ezkl_amd/dist.py selftest: the twin of ezkl_hip_comm_selftest, which
        # ezkl_hip_comm_init ran on the other branch): a wrong exchange must end the job here, with the step named, not as a rejected proof
        from ezkl_amd import dist as D_
        D_.selftest(dist, ddev)
        nc.set_shard(dist, ddev)
        if "--replicated" not in sys.argv:
            nc.set_shard_exchange(dist, ddev)
    if "--replicated" in sys.argv and comm_used:
        NV._check(NV.load().ezkl_prover_cs_set_shard_exchange(nc.h, None, None, None), "ezkl_prover_cs_set_shard_exchange")
    nc.set_shard_full_bases(True)
    mode = "replicated columns, commit batches by columns" if "--replicated" in sys.argv else "columns and arguments by owner"
    collectives = "libezkl_hip.so RCCL communicator (comm.hip)" if comm_used else "torch.distributed callbacks (%s)" % ("gloo" if "--gloo" in sys.argv else "nccl")
t0 = time.time(); npk = NV.NativeProvingKey(nc, gb, fixed, copies); t_keygen = time.time() - t0
if "--pinned" in sys.argv and not callable(adv):
    pinned = [B.PinnedArray((n, 4)) for _ in adv]
    for pa, a in zip(pinned, adv):
        pa.array[:] = a
    adv = [pa.array for pa in pinned]
if "--bad-lookup" in sys.argv:
    # a witness with values outside their lookup tables (the most common ezkl prover failure): EVERY rank must come back with the error --
    # only the owner of the offending argument sees the counter, the others learn it through the collective status
    adv = [np.array(a, copy=True) for a in adv]
    for a in adv:
        a[:32] = P.to_mont(123456789)
    try:
        NV.create_proof(npk, gb, glb, adv, seed=5, instances=instances)
        status = "RANK %d NO_ERROR" % rank
    except Exception as e:
        status = "RANK %d ERROR %s" % (rank, str(e).replace("\n", " ")[:200])
    print(status, flush=True)
    if os.environ.get("EZKL_RANK_STATUS_DIR"):          # the ranks share one stdout pipe with gloo's own C-level prints: a file per rank cannot interleave
        with open(os.path.join(os.environ["EZKL_RANK_STATUS_DIR"], "rank_%d.txt" % rank), "w") as f:
            f.write(status + "\n")
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()      # a rank left behind in a collective would never get here
    sys.exit(0)
t0 = time.time(); NV.create_proof(npk, gb, glb, adv, seed=5, instances=instances); t_first = time.time() - t0
runs = []
for _ in range(int(os.environ.get("REPS", "3"))):
    if dist is not None:
        dist.barrier()
    tm = {}
    t0 = time.time(); proof = NV.create_proof(npk, gb, glb, adv, seed=5, instances=instances, timings=tm); runs.append((time.time() - t0, tm))
t_prove, tm = min(runs, key=lambda r: r[0])
stats = nc.shard_stats()
# what the library's RCCL communicator saw and moved for the LAST proof's worth of exchanges (ezkl_hip_comm_info / ezkl_hip_comm_stats):
# `rccl_ranks_seen` is the world size of the ncclComm the exchanges ran on (0 = torch.distributed callbacks), so a SCALE run can be checked
cw, cr = B.comm_info()
cst = B.comm_stats()
nproofs = 1 + len(runs)
stats = dict(stats, rccl_ranks_seen=cw, rccl_rank=cr, exchange_calls_per_proof=cst["exchanges"] / nproofs, exchange_bytes_sent_per_proof=cst["bytes_sent"] / nproofs,
             exchange_ms_per_proof=round(1e3 * cst["seconds"] / nproofs, 3), nccl_sends_per_proof=cst["nccl_sends"] / nproofs, nccl_recvs_per_proof=cst["nccl_recvs"] / nproofs,
             exchange_rounds_per_proof=cst["rounds"] / nproofs)
res = npk.residency()
free_b, total_b = B.mem_info()
stats = dict(stats, key_cosets_held="%d of %d" % (res["cosets"], res["E"]), key_gib=round(res["key_bytes"] / 2**30, 3), hbm_in_use_gib=round((total_b - free_b) / 2**30, 2),
             hbm_pool_high_water_gib=round(B.pool_stats()["live_peak"] / 2**30, 2))
sha = hashlib.sha256(proof).hexdigest()
if world > 1:
    import torch
    objs = [None] * world
    dist.all_gather_object(objs, {"rank": rank, "sha": sha, "seconds": t_prove, "stats": stats, "sharded_sweeps": nc.sharded_sweeps()})
    if rank != 0:
        dist.barrier(); dist.destroy_process_group(); sys.exit(0)
    t_prove = max(o["seconds"] for o in objs)
else:
    objs = [{"rank": 0, "sha": sha, "seconds": t_prove, "stats": stats, "sharded_sweeps": 0}]
# rank 0: the product's own verifier (C++ pairing) on the proof
g2b = NV.g2_mul_generator(1); s_g2b = NV.g2_mul_generator(s)
ok = NV.verify_proof(npk, g2b, s_g2b, proof, instances)
bad = bytearray(proof); bad[len(bad) // 2] ^= 1
print(json.dumps({"what": "create_proof by libezkl_prover.so across %d rank(s): %s" % (world, mode), "circuit": dict(info, k=k, advice_columns=cs.n_advice, lookups=len(cs.lookups),
                  permutation_columns=len(cs.perm), degree=cs.degree, ext_k=cs.ext_k), "n_gpus": world, "mode": mode, "collectives": collectives,
                  "proof_bytes": len(proof), "proof_sha256": sha[:16], "all_ranks_same_proof": all(o["sha"] == sha for o in objs),
                  "verifier_accepts": bool(ok), "tampered_rejected": not NV.verify_proof(npk, g2b, s_g2b, bytes(bad), instances),
                  "prove_seconds_gpu": round(t_prove, 4), "prove_seconds_gpu_runs": [round(r[0], 4) for r in runs], "first_prove_seconds_gpu": round(t_first, 4),
                  "keygen_seconds_gpu": round(t_keygen, 3), "prove_breakdown_seconds": {a: round(b, 4) for a, b in tm.items()},
                  "per_rank": [{"rank": o["rank"], "stats": o["stats"], "sharded_sweeps": o["sharded_sweeps"], "seconds": round(o["seconds"], 4)} for o in objs]}))
if world > 1:
    dist.barrier(); dist.destroy_process_group()
