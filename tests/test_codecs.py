"""CPU tests of the artefact codecs on the reference's golden files."""
import os
import numpy as np
import pytest
from conftest import GOLDEN, fe_from_int
from ezkl_amd import codecs

REF_ASSETS = "/root/reference/tests/assets"


def test_srs_roundtrip_and_fields():
    buf = open(os.path.join(GOLDEN, "kzg_k6.srs"), "rb").read()
    srs = codecs.read_srs(buf)
    assert srs["k"] == 6 and srs["g"].shape == (64, 8) and srs["g_lagrange"].shape == (64, 8)
    assert codecs.write_srs(srs) == buf
    small = codecs.read_srs(open(os.path.join(GOLDEN, "kzg_k1_public.srs"), "rb").read())
    assert small["k"] == 1 and len(small["g2"]) == 128
    with pytest.raises(ValueError):
        codecs.read_srs(buf[:-1])


def test_vk_golden():
    buf = open(os.path.join(GOLDEN, "vk_k6.key"), "rb").read()
    vk = codecs.read_vk(buf, n_perm=32)
    assert vk["k"] == 6 and vk["compress_selectors"]
    assert vk["fixed_commitments"].shape == (38, 8) and vk["permutation_commitments"].shape == (32, 8)
    assert vk["selectors"].shape == (80, 64) and vk["end"] == len(buf) == 5127
    assert codecs.write_vk(vk) == buf


def test_pk_roundtrip_synthetic():
    rng = np.random.default_rng(1)
    k, ek, nf, npm, ns = 3, 5, 2, 3, 4
    n, ne = 1 << k, 1 << ek
    col = lambda m: rng.integers(0, 1 << 60, size=(m, 4), dtype=np.uint64)
    vk = dict(k=k, compress_selectors=True, fixed_commitments=rng.integers(0, 1 << 60, size=(nf, 8), dtype=np.uint64),
              permutation_commitments=rng.integers(0, 1 << 60, size=(npm, 8), dtype=np.uint64), selectors=rng.random((ns, n)) < 0.5)
    pk = dict(vk=vk, l0=col(ne), l_last=col(ne), l_active_row=col(ne),
              fixed_values=[col(n) for _ in range(nf)], fixed_polys=[col(n) for _ in range(nf)], fixed_cosets=[col(ne) for _ in range(nf)],
              permutations=[col(n) for _ in range(npm)], perm_polys=[col(n) for _ in range(npm)], perm_cosets=[col(ne) for _ in range(npm)])
    buf = codecs.write_pk(pk)
    back = codecs.read_pk(buf, npm, ns)
    assert codecs.write_pk(back) == buf
    assert (back["fixed_cosets"][1] == pk["fixed_cosets"][1]).all() and (back["vk"]["selectors"] == vk["selectors"]).all()
    with pytest.raises(ValueError):
        codecs.read_pk(buf, npm, ns + 1)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_ASSETS, "pk.key")), reason="reference assets only exist in the build container")
def test_reference_pk_and_proof_files():
    """the real artefacts of the reference parse to exactly EOF and agree with the golden subset"""
    buf = open(os.path.join(REF_ASSETS, "pk.key"), "rb").read()
    pk = codecs.read_pk(buf, n_perm=32, n_selectors=80)
    assert len(pk["fixed_values"]) == 38 and pk["l0"].shape == (512, 4) and len(pk["perm_cosets"]) == 32
    assert codecs.write_pk(pk) == buf
    g = np.load(os.path.join(GOLDEN, "pk_k6_subset.npz"))
    for j, idx in enumerate(g["fixed_idx"]):
        assert pk["fixed_cosets"][idx].tobytes() == g["fixed_cosets"][j].tobytes()
    pr = codecs.read_proof_json(open(os.path.join(REF_ASSETS, "proof.json")).read())
    assert len(pr["proof"]) == 14816 and pr["transcript_type"] == "EVM"
    pts, ev, tail = codecs.split_evm_proof(pr["proof"], 114, 231)
    q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
    assert all((y * y - x * x * x - 3) % q == 0 for x, y in pts + tail) and len(tail) == 2
    assert all(e < 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001 for e in ev)


def test_felt_hex():
    assert codecs.felt_from_hex_le("02" + "00" * 31) == 2
    assert codecs.felt_to_hex_le(2) == "02" + "00" * 31


def test_snark_json_round_trip_and_witness_reader():
    """Snark JSON writer against the reference's proof.json (same instances encoding, hex_proof == 0x + hex(proof)); witness.json reader"""
    import json, os
    from ezkl_amd import codecs
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ref = codecs.read_proof_json(open(os.path.join(G, "proof_k6.json")).read())
    text = codecs.write_proof_json(ref["proof"], ref["instances"], pretty_public_inputs=ref["raw"]["pretty_public_inputs"], timestamp_ms=ref["raw"]["timestamp"])
    j = json.loads(text)
    assert list(j) == ["protocol", "instances", "proof", "hex_proof", "split", "pretty_public_inputs", "timestamp", "version"]   # Snark's field order
    assert j["hex_proof"] == ref["raw"]["hex_proof"] and j["instances"] == ref["raw"]["instances"] and j["protocol"] is None
    assert bytes(j["proof"]) == ref["proof"] and ", " not in text
    back = codecs.read_proof_json(text)
    assert back["proof"] == ref["proof"] and back["instances"] == ref["instances"]
    w = codecs.read_witness_json(open(os.path.join(G, "witness_k6.json")).read())
    assert w["inputs"] == [[2, 1, 1]] and w["outputs"] == [[0, 0, 0, 0]] and w["max_range_size"] == 127


def test_compiled_circuit_reader_on_the_reference_model():
    """tests/assets/model.compiled (bincode of GraphCircuit): the node graph is the fixture model (Gemm 3 -> 4 as Einsum "mk,nk->mn" + Add +
    LeakyReLU slope 0, weights / bias of network.onnx at scale 0) and the GraphSettings inside equal the reference's settings.json"""
    import json
    from ezkl_amd import codecs, execute as X
    from test_ezkl_circuit import FIXTURE_B, FIXTURE_W
    c = codecs.read_compiled_circuit(open(os.path.join(GOLDEN, "model_k6.compiled"), "rb").read())
    ref = json.load(open(os.path.join(GOLDEN, "settings_k6.json")))
    st = json.loads(json.dumps(c["settings"]))
    for key, val in st.items():
        if key == "run_args":
            for k2, v2 in val.items():
                if k2 in ref["run_args"]:
                    assert v2 == ref["run_args"][k2], k2
        elif key in ref:
            assert val == ref[key], key
    assert st["total_assignments"] == 472 and st["required_range_checks"] == [[-1, 1], [0, 127]] and st["timestamp"] == ref["timestamp"]
    kinds = {k: (n["opkind"]["kind"], n["opkind"].get("op")) for k, n in c["model"]["nodes"].items()}
    assert kinds == {0: ("Input", None), 1: ("Constant", None), 2: ("Linear", "Einsum"), 3: ("Constant", None), 4: ("Linear", "Add"), 6: ("Linear", "LeakyReLU")}
    assert c["model"]["nodes"][2]["opkind"]["equation"] == "mk,nk->mn" and c["model"]["outputs"] == [(6, 0)]
    assert X._mlp_of_graph(c["model"]) == ([FIXTURE_W], [FIXTURE_B], True, False, 3)
    with pytest.raises(ValueError):
        codecs.read_compiled_circuit(open(os.path.join(GOLDEN, "model_k6.compiled"), "rb").read()[:700])


def test_compiled_circuit_reader_refuses_damaged_files():
    """truncated or corrupted model.compiled files end in ValueError (never an index / decode error from inside the reader)"""
    from ezkl_amd import codecs
    b = open(os.path.join(GOLDEN, "model_k6.compiled"), "rb").read()
    rng = np.random.default_rng(1)
    for t in range(300):
        if t % 2 == 0:
            data = b[:int(rng.integers(0, len(b)))]
        else:
            data = bytearray(b)
            for _ in range(3):
                data[int(rng.integers(0, 1300))] = int(rng.integers(0, 256))
            data = bytes(data)
        try:
            codecs.read_compiled_circuit(data)
        except ValueError:
            pass
