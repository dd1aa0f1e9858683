"""Proof parity at the sizes of BASELINE.json's configs, inside the `-m gpu` suite (until round 5 only bench.py compared
GPU and CPU proofs at k >= 17).  For each laid-out bench circuit (bench_cache/*.npz, tools/bench_circuits.py) the C++ host prover on the GPU
(libezkl_prover.so over the C ABI: MSM, NTT, coset forms, quotient sweep, SHPLONK on the device) and the SAME create_proof on the host cores
(Python host + the C oracle's kernels, oracle/cpu_backend.py) must emit the same proof bytes from the same witness and randomness, and the
oracle's pairing verifier must accept them.  The quotient sweep of these keys runs over 2^17 x 4 (conv: 2^17 x 8) extended rows -- the
evaluate_h sizes the small-circuit tests do not reach.

  configs[2]  examples/conv2d_mnist at k = 17 (its own Config and layout, /root/reference/examples/conv2d_mnist/main.rs)
  configs[2]' the MLP over the ezkl gate set at k = 17 (the k = 20 circuit of the metric, three doublings smaller)
  the metric  the MLP at k = 20 ("ezkl prove wall-seconds (k = 20 MLP)"): a k = 20 regression turns the suite red, not a bench field
  configs[3]  the reference's accum_einsum_matmul bench circuit raised to k = 20: second-phase advice + two challenges (Freivalds)
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))

G2 = ((0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed, 0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2),
      (0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa, 0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b))


@pytest.mark.gpu
@pytest.mark.parametrize("kind,k", [("conv", 17), ("mlp", 17), ("mlp", 20), ("einsum", 20)])
def test_gpu_proof_equals_cpu_oracle_proof_at_bench_size(hip, kind, k):
    import bench_circuits as BC
    from ezkl_amd import backend as B, native as NV, plonk as P
    from oracle import pairing as E, verifier as V
    from oracle.cpu_backend import OracleBackend
    if kind != "einsum":                                        # (the einsum witness depends on the proof's challenges: laid out per proof, 1.3 s)
        assert os.path.exists(os.path.join(ROOT, "bench_cache", "%s_k%d_s1.npz" % (kind, k))), "the laid-out circuit ships with the repo"
    built = BC.build(kind, k, gpu=B)
    if kind != "einsum":
        assert "read from" in built["info"].get("layout", "")
    else:
        assert sum(built["cs"].advice_phase) == 3 and built["cs"].n_challenges == 2
    cs, fixed, copies, adv, instances = built["cs"], built["fixed"], built["copies"], built["advice"], built["instances"]
    assert cs.k == k
    s = 0x1234567890abcdef1234567890abcdef % P.R
    gb, glb = B.gen_srs(k, s)                                   # test SRS with a known secret (insecure, like `ezkl gen-srs`)
    g, gl = gb.download(), glb.download()
    # GPU: the product path
    npk = NV.NativeProvingKey(NV.NativeCircuit(cs), gb, fixed, copies)
    proof_gpu = NV.create_proof(npk, gb, glb, adv, rng=P.Rng(5), instances=instances)
    # CPU: the checker
    cpu = OracleBackend(g, gl, k)
    pk_c, vk_c = P.keygen(cs, cpu, fixed, copies)
    proof_cpu = P.create_proof(pk_c, cpu, adv, P.Rng(5), instances=instances)
    assert len(proof_gpu) == len(proof_cpu)
    assert proof_gpu == proof_cpu, "first differing byte at %d" % next(i for i, (a, b) in enumerate(zip(proof_gpu, proof_cpu)) if a != b)
    # the keys agree as well (fixed / sigma commitments, digest), and the verifier accepts
    fc, pc, digest = npk.vk()
    assert digest == vk_c.digest
    assert V.verify(vk_c, (1, 2), G2, E.g2_mul(G2, s), proof_gpu, instances=instances)
    # a second GPU proof with the library's own randomness: other bytes, accepted as well; a corrupted proof is rejected
    proof_lib = NV.create_proof(npk, gb, glb, adv, seed=9, instances=instances)
    assert proof_lib != proof_gpu and V.verify(vk_c, (1, 2), G2, E.g2_mul(G2, s), proof_lib, instances=instances)
    bad = bytearray(proof_gpu)
    bad[len(bad) // 2] ^= 1
    assert not V.verify(vk_c, (1, 2), G2, E.g2_mul(G2, s), bytes(bad), instances=instances)
    gb.free(); glb.free()
