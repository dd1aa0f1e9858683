"""`ezkl gen-witness` (/root/reference/src/execute.rs:577-660 -> GraphCircuit::forward, src/graph/mod.rs:1734-1849):
ezkl_amd.execute.gen_witness on the reference's own artefacts -- its compiled model (tests/assets/model.compiled) and its input.json
must give its witness.json BYTE FOR BYTE (integer work: bit-exact; serde_json field order, Rust's f64 formatting, hex felts).
Host arithmetic only: no GPU needed for Private / Public visibilities (the KZGCommit path is in tests/test_execute.py)."""
import json
import os

import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_gen_witness_reproduces_the_reference_witness_file(tmp_path):
    from ezkl_amd import execute as X
    out = tmp_path / "witness.json"
    w = X.gen_witness(os.path.join(G, "model_k6.compiled"), os.path.join(G, "input_k6.json"), output=str(out))
    assert out.read_bytes() == open(os.path.join(G, "witness_k6.json"), "rb").read()
    assert w["max_range_size"] == 127 and w["max_lookup_inputs"] == 0 and w["pretty_elements"]["rescaled_inputs"] == [["2", "1", "1"]]
    # the same from a JSON string and from a dict; output_data in the file is ignored, as in the reference
    data = json.load(open(os.path.join(G, "input_k6.json")))
    assert X.gen_witness(os.path.join(G, "model_k6.compiled"), json.dumps(data)) == w
    assert X.gen_witness(os.path.join(G, "model_k6.compiled"), {"input_data": data["input_data"]}) == w


def test_quantization_is_the_reference_s():
    """quantize_float (src/graph/utilities.rs:53-69) after InputType::roundtrip (src/circuit/ops/mod.rs:112-141): f32 round trip, times
    2^scale, f64::round = half AWAY from zero (Python's round is half to even)"""
    from ezkl_amd import execute as X
    assert [X._quantize(v, 0, "F32") for v in (0.5, 1.5, 2.5, -0.5, -2.5, 1.5417295, 0.49999997)] == [1, 2, 3, -1, -3, 2, 0]
    assert X._quantize(0.1, 7, "F32") == 13 and X._quantize(0.1, 7, "F64") == 13 and X._quantize(3.9, 0, "Int") == 3
    assert X._quantize(0.30000001192092896, 24, "F32") == 5033165 and X._quantize(0.3, 24, "F64") == 5033165     # f32(0.3) * 2^24 = 5033165.0 exactly
    assert X._quantize(True, 3, "Bool") == 1
    with pytest.raises(ValueError, match="SigBitTruncation"):
        X._quantize(1e40, 7, "F64")


def test_rust_float_formatting():
    from ezkl_amd import codecs
    f = codecs.rust_f64_to_string
    assert [f(x) for x in (0.0, 2.0, -3.0, 0.5, 0.28125, 1e-7, 123456789012345680000.0, -0.0)] == \
        ["0", "2", "-3", "0.5", "0.28125", "0.0000001", "123456789012345680000", "-0"]


def test_forward_pass_values_and_errors(tmp_path):
    from ezkl_amd import execute as X
    model = tmp_path / "m.json"
    ra = dict(logrows=8, num_inner_cols=2, decomp_base=128, decomp_legs=2)
    model.write_text(json.dumps({"model": "mlp", "run_args": ra, "weights": [[[1, -2, 3], [0, 5, -1]], [[2, 1]]], "biases": [[1, -40], [-3]], "relu_last": False}))
    w = X.gen_witness(str(model), {"input_data": [[3.2, -1.6, 7.0]]})             # x = [3, -2, 7]
    # layer 0: [3 + 4 + 21 + 1, -10 - 7 - 40] = [29, -57] -> relu [29, 0]; layer 1: 58 + 0 - 3 = 55 (no relu on the last layer)
    assert w["pretty_elements"]["rescaled_outputs"] == [["55"]] and w["outputs"] == [["37" + "00" * 31]]
    assert w["inputs"][0][1] == (X.EL.R - 2).to_bytes(32, "little").hex()         # negative values are field elements r - |v|
    with pytest.raises(ValueError, match="decomposition range"):
        X.gen_witness(str(model), {"input_data": [[20000.0, 0.0, 0.0]]})         # 128^2 = 16384: the input range check fails
    with pytest.raises(ValueError, match="input shape"):
        X.gen_witness(str(model), {"input_data": [[1.0, 2.0]]})
    # KZGCommit visibility without an SRS: the module result stays empty, as the reference warns ("will be ignored")
    ra2 = dict(ra, input_visibility="KZGCommit")
    model.write_text(json.dumps({"model": "mlp", "run_args": ra2, "weights": [[[1, 1, 1]]], "biases": [[0]]}))
    w = X.gen_witness(str(model), {"input_data": [[1.0, 2.0, 3.0]]})
    assert w["processed_inputs"] == {"poseidon_hash": None, "polycommit": None} and w["processed_outputs"] is None
    model.write_text(json.dumps({"model": "mlp", "run_args": dict(ra, output_visibility={"Hashed": {"hash_is_public": True, "outlets": []}}), "weights": [[[1, 1, 1]]], "biases": [[0]]}))
    with pytest.raises(ValueError, match="Hashed"):
        X.gen_witness(str(model), {"input_data": [[1.0, 2.0, 3.0]]})


RELU_1L_INPUT = [-0.40077725052833557, 2.493845224380493, 0.5796360969543457]      # /root/reference/examples/onnx/1l_relu/input.json (input_data)


def test_1l_relu_by_name_forward_pass_and_graph_walker(tmp_path):
    """BASELINE configs[0] under its own name: examples/onnx/1l_relu is nn.ReLU on a vector of 3 (gen.py) -- Input -> LeakyReLU slope 0 ->
    output, NO Gemm.  VERDICT r05 missing #4: the op-family reader used to require a Gemm and refused this graph.  The forward pass at
    ezkl's default input scale 7 on the example's own input.json: round(x * 128) half away from zero, then max(., 0)"""
    from ezkl_amd import execute as X
    model = tmp_path / "1l_relu.json"
    ra = dict(logrows=8, num_inner_cols=2, decomp_base=128, decomp_legs=2, input_scale=7)
    model.write_text(json.dumps({"model": "mlp", "run_args": ra, "weights": [], "biases": [], "n_inputs": 3, "relu_first": True}))
    w = X.gen_witness(str(model), {"input_data": [RELU_1L_INPUT]})
    assert w["pretty_elements"]["rescaled_inputs"] == [["-0.3984375", "2.4921875", "0.578125"]]            # -51, 319, 74 over 2^7
    assert w["pretty_elements"]["rescaled_outputs"] == [["0", "2.4921875", "0.578125"]]
    assert w["inputs"][0][0] == (X.EL.R - 51).to_bytes(32, "little").hex() and w["outputs"][0][0] == "00" * 32
    assert w["outputs"][0][1] == (319).to_bytes(32, "little").hex() and w["max_range_size"] == 127
    # the node graph a compiled 1l_relu holds (as codecs.read_compiled_circuit returns it): one scale everywhere, no Linear node but the ReLU
    vis = dict(input="Private", params="Private", output="Public")
    nodes = {0: dict(opkind=dict(kind="Input", datum_type="F32", decomp=True), out_scale=7, inputs=[], out_dims=[1, 3], idx=0),
             1: dict(opkind=dict(kind="Linear", op="LeakyReLU", slope=0.0), out_scale=7, inputs=[(0, 0)], out_dims=[1, 3], idx=1)}
    graph = dict(nodes=nodes, inputs=[0], outputs=[(1, 0)], visibility=vis)
    assert X._mlp_of_graph(graph) == ([], [], False, True, 3)
    nodes[1]["opkind"]["slope"] = 0.1
    with pytest.raises(ValueError, match="slope"):
        X._mlp_of_graph(graph)
    nodes[1]["opkind"]["slope"], nodes[1]["out_scale"] = 0.0, 14                    # two scales in a graph: something rescales, not this family
    with pytest.raises(ValueError, match="scales"):
        X._mlp_of_graph(graph)
    # the layout: input decomposition + ReLU by sign decomposition + output range check, no dot product anywhere
    c = X.EL.MlpCircuit(8, 2, [], [], 128, 2, n_inputs=3, relu_first=True)
    reg = c.synthesize([-51, 319, 74])
    assert c.outputs == [0, 319, 74] and c.settings.model_instance_shapes == [[1, 3]]
    with pytest.raises(ValueError):
        X.EL.MlpCircuit(8, 2, [], [], 128, 2)                                         # no weights and no input length
