#!/usr/bin/env python3
"""Workload for the PMC passes over the PROVE workload (run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`, separately): the
calibration kernels with known byte counts, keygen + one warm proof of the k = 20 MLP circuit, a marker kernel, then ONE measured proof
(C++ host prover).  tools/pmc_prove_reduce.py turns the two counter files into per-kernel HBM traffic of that last proof."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ezkl_amd
from ezkl_amd import backend as B, native as NV, plonk as P
import bench_circuits as BC
ezkl_amd.init(0)
print("copy", B.ubench("copy"), "gather64", B.ubench("gather64"))
k = int(os.environ.get("K", "20"))
kw = {}
if os.environ.get("MLP_BLOCKS"): kw["blocks"] = int(os.environ["MLP_BLOCKS"])
if os.environ.get("MLP_FILL"): kw["fill"] = int(os.environ["MLP_FILL"])
built = BC.build(os.environ.get("CIRCUIT", "mlp"), k, gpu=B, **kw)
cs, fixed, copies, adv, instances = built["cs"], built["fixed"], built["copies"], built["advice"], built["instances"]
gb, glb = B.gen_srs(k, 0x1234567890abcdef1234567890abcdef % P.R)
npk = NV.NativeProvingKey(NV.NativeCircuit(cs), gb, fixed, copies)
NV.create_proof(npk, gb, glb, adv, seed=5, instances=instances)
B.ubench("gather64")                      # marker: everything after the last ub_gather_kernel launch is the measured proof
proof = NV.create_proof(npk, gb, glb, adv, seed=5, instances=instances)
print("proof bytes", len(proof))
