#!/usr/bin/env python3
"""Extract small golden fixtures from the reference's own test assets.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):
    python tests/golden/make_golden.py
Outputs (committed): tests/golden/kzg_k6.srs, kzg_k1_public.srs, vk_k6.key, pk_k6_subset.npz, pk_k6.key, proof_k6.json,
                     settings_k6.json, witness_k6.json, input_k6.json, verifier_k6.code, model_k6.compiled

Sources (read-only, data not code): /root/reference/tests/assets/{kzg,kzg1.srs,vk.key,pk.key}.
What they pin:
  * kzg (k=6 test SRS)        : sum(g_lagrange)==g[0]; MSM(v,g_lagrange)==MSM(iNTT(v),g)
  * kzg1.srs (public k=1 SRS) : vk permutation commitments of identity sigma columns == delta^j * sG
  * pk.key                    : fixed_polys==iNTT_64(fixed_values); fixed_cosets==coeff_to_extended(polys);
                                l0/l_last/l_active_row extended forms; permutation polys/cosets likewise
pk_k6_subset.npz keeps a few columns for the kernel KATs; pk_k6.key is the whole key (1.4 MB): the fixture CIRCUIT tests
(tests/test_ezkl_circuit.py) load it as `ezkl prove` would and need every fixed / permutation column.
  * proof.json / settings.json / witness.json : the proof the reference made for this key (layout 114 G1 | 231 Fr | 2 G1;
    its fixed / sigma evaluations equal pk.key's polynomials at the challenge x recovered from the identity sigma columns)
  * wasm.code -> verifier_k6.code : the deployment bytecode (solc 0.8.20, hex text) of the Solidity verifier the reference's tooling
    generated for a k = 6 key of the SAME constraint system (proof length 14 816, 4 instances; other selector combinations, other
    commitments): compiled DATA, executed by oracle/mini_evm.py -- the reference's verifier for the EVM transcript, the only
    executable piece of the zkonduit halo2 protocol in the tree (tests/test_evm_verifier.py)
"""
import os, shutil, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
from pyref import parse_pk  # noqa: E402

A = "/root/reference/tests/assets/"
shutil.copyfile(A + "kzg", os.path.join(HERE, "kzg_k6.srs"))
shutil.copyfile(A + "kzg1.srs", os.path.join(HERE, "kzg_k1_public.srs"))
shutil.copyfile(A + "vk.key", os.path.join(HERE, "vk_k6.key"))
shutil.copyfile(A + "pk.key", os.path.join(HERE, "pk_k6.key"))
shutil.copyfile(A + "settings.json", os.path.join(HERE, "settings_k6.json"))
shutil.copyfile(A + "witness.json", os.path.join(HERE, "witness_k6.json"))
shutil.copyfile(A + "input.json", os.path.join(HERE, "input_k6.json"))           # the GraphData `gen-witness` made witness.json from
shutil.copyfile(A + "wasm.code", os.path.join(HERE, "verifier_k6.code"))
shutil.copyfile(A + "model.compiled", os.path.join(HERE, "model_k6.compiled"))   # bincode of GraphCircuit: codecs.read_compiled_circuit
import json
_p = json.load(open(A + "proof.json"))
_p["proof"] = []          # the byte list duplicates hex_proof
json.dump(_p, open(os.path.join(HERE, "proof_k6.json"), "w"))
pk = parse_pk(open(A + "pk.key", "rb").read(), n_perm=32, n_sel=80)
u8 = lambda b: np.frombuffer(b, dtype=np.uint8)
FIXED = [0, 1, 5, 17, 37]
PERM = [0, 3, 20, 31]
out = dict(k=np.array([pk["k"]]), ext_k=np.array([9]), fixed_idx=np.array(FIXED), perm_idx=np.array(PERM),
           l0=u8(pk["l0"]), l_last=u8(pk["l_last"]), l_active_row=u8(pk["l_active_row"]))
for name, idx in (("fixed_values", FIXED), ("fixed_polys", FIXED), ("fixed_cosets", FIXED),
                  ("permutations", PERM), ("perm_polys", PERM), ("perm_cosets", PERM)):
    out[name] = np.stack([u8(pk[name][i]) for i in idx])
np.savez_compressed(os.path.join(HERE, "pk_k6_subset.npz"), **out)
print({k: v.shape for k, v in out.items()})
