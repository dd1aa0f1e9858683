"""The command bodies around the prove hot path, with the reference's names and argument meaning
(/root/reference/src/execute.rs): artefact FILES in, artefact files out, everything O(n) on the GPU through libezkl_prover.so.

    gen_srs(srs_path, logrows)                                                  execute.rs gen_srs -> pfsys::srs::gen_srs (srs.rs:13-16)
    setup(compiled_circuit, srs_path, vk_path, pk_path)                         execute.rs:1543-1572 -> pfsys::create_keys (mod.rs:376-400)
    gen_witness(compiled_circuit, data, output, vk_path, srs_path)               execute.rs:577-660 -> GraphCircuit::forward (graph/mod.rs:1734-1849)
    prove(witness, compiled_circuit, pk_path, proof_path, srs_path, check_mode)  execute.rs:1575-1627 -> create_proof_circuit (mod.rs:404-489)
    verify(proof_path, compiled_circuit, pk_path, srs_path)                      execute.rs:1651-1722 -> verify_proof_circuit (mod.rs:557-590)

File formats are the reference's: kzg*.srs / pk.key / vk.key in halo2 raw bytes, witness.json (GraphWitness), proof.json (Snark) and
`model.compiled` -- the bincode of ezkl's GraphCircuit (node graph + GraphSettings), read by codecs.read_compiled_circuit: on the
reference's fixture `prove` runs from the reference's artefact files alone (tests/test_execute.py).  What a compiled circuit may
CONTAIN here is the op family ezkl_layout.MlpCircuit lays out (Input -> Gemm as Einsum "mk,nk->mn" + bias + LeakyReLU slope 0, private
parameters, public output): ezkl's general layout lives in 6.8k lines of Rust (out of scope) and any other graph
is refused by name.  On the fixture model that layout reproduces the reference's pk.key bit for bit (tests/test_ezkl_circuit.py).  A JSON
description of the same family ({"model": "mlp", "run_args": {...}, "weights": [...], "biases": [...]}) is accepted as well.
Errors surface as exceptions with the reference's wording where it has one."""
import json
import os

import numpy as np

from . import backend as B
from . import codecs, ezkl_layout as EL, native as NV


class CheckMode:
    SAFE, UNSAFE = "SAFE", "UNSAFE"


def _load_circuit(path):
    """the reference's own `model.compiled` (bincode of GraphCircuit, codecs.read_compiled_circuit) or this package's JSON description"""
    raw = open(path, "rb").read()
    if raw[:1] == b"{":
        j = json.loads(raw)
        if j.get("model") != "mlp":
            raise ValueError("unsupported compiled circuit: %r" % j.get("model"))
        ra = j["run_args"]
        return EL.MlpCircuit(ra["logrows"], ra["num_inner_cols"], j["weights"], j["biases"], ra["decomp_base"], ra["decomp_legs"],
                             total_assignments=j.get("total_assignments"), relu_last=j.get("relu_last", True),
                             n_inputs=j.get("n_inputs"), relu_first=j.get("relu_first", False)), j
    c = codecs.read_compiled_circuit(raw)
    weights, biases, relu_last, relu_first, n_inputs = _mlp_of_graph(c["model"])
    st, ra = c["settings"], c["settings"]["run_args"]
    want = [(-1, 1), (0, ra["decomp_base"] - 1)]
    if [tuple(x) for x in st["required_range_checks"]] != want or st["required_lookups"] or st["num_dynamic_lookups"] or st["num_shuffles"] \
            or st["einsum_params"]["equations"]:
        raise ValueError("unsupported compiled circuit: its settings ask for arguments outside the MLP family")
    return EL.MlpCircuit(ra["logrows"], ra["num_inner_cols"], weights, biases, ra["decomp_base"], ra["decomp_legs"],
                         total_assignments=st["total_assignments"], relu_last=relu_last, n_inputs=n_inputs, relu_first=relu_first), c


def _mlp_of_graph(model, any_visibility=False):
    """the op family ezkl_layout.MlpCircuit lays out, read off ezkl's node graph: Input -> (Einsum "mk,nk->mn" with a constant [n, k]
    -> Add of a constant [1, n] -> LeakyReLU slope 0)*, private input and parameters at scale 0, public output.  A LeakyReLU slope 0 may
    sit straight on the input, and the layer list may be EMPTY: Input -> LeakyReLU -> output is examples/onnx/1l_relu (BASELINE
    configs[0]; with no Gemm nothing is rescaled, so that graph may carry any ONE scale -- ezkl's default input scale is 7).
    any_visibility (the forward pass of gen_witness, which lays nothing out): the visibilities are the caller's business.
    -> (weights, biases, relu_last, relu_first, n_inputs)"""
    nodes, vis = model["nodes"], model["visibility"]
    if len(model["inputs"]) != 1 or len(model["outputs"]) != 1 or \
            (not any_visibility and (vis["input"], vis["params"], vis["output"]) != ("Private", "Private", "Public")):
        raise ValueError("unsupported compiled circuit: visibility / arity")
    signed = lambda v: v if v < EL.R // 2 else v - EL.R
    def const(idx, dims_ok):
        op = nodes[idx]["opkind"]
        if op["kind"] != "Constant" or not dims_ok(op["quantized_values"]["dims"]):
            raise ValueError("unsupported compiled circuit: node %d is not the expected constant" % idx)
        q = op["quantized_values"]
        return [signed(v) for v in q["inner"]], q["dims"]
    cur = model["inputs"][0]
    order = sorted(k for k in nodes if nodes[k]["opkind"]["kind"] == "Linear")
    gemm_free = all(nodes[k]["opkind"]["op"] == "LeakyReLU" for k in order)
    scales = {n["out_scale"] for n in nodes.values()}
    if nodes[cur]["opkind"]["kind"] != "Input" or (scales != {0} and not (gemm_free and len(scales) == 1)):
        raise ValueError("unsupported compiled circuit: input node / non-zero scales")
    n_inputs = 1
    for d in nodes[cur]["out_dims"]:
        n_inputs *= d
    weights, biases, relu_last, relu_first, i = [], [], False, False, 0
    if order and nodes[order[0]]["opkind"]["op"] == "LeakyReLU":
        n = nodes[order[0]]
        if n["opkind"]["slope"] != 0.0 or n["inputs"] != [(cur, 0)]:
            raise ValueError("unsupported compiled circuit: LeakyReLU with a slope")
        cur, i, relu_first = order[0], 1, True
    while i < len(order):
        n = nodes[order[i]]
        if n["opkind"]["op"] != "Einsum" or n["opkind"]["equation"] != "mk,nk->mn" or n["inputs"][0] != (cur, 0):
            raise ValueError("unsupported compiled circuit: node %d" % order[i])
        w, dims = const(n["inputs"][1][0], lambda d: len(d) == 2)
        weights.append([w[r * dims[1]:(r + 1) * dims[1]] for r in range(dims[0])])
        cur, i = order[i], i + 1
        n = nodes[order[i]] if i < len(order) else None
        if n is None or n["opkind"]["op"] != "Add" or n["inputs"][0] != (cur, 0):
            raise ValueError("unsupported compiled circuit: a Gemm without its bias")
        biases.append(const(n["inputs"][1][0], lambda d: d[-1] == dims[0])[0])
        cur, i = order[i], i + 1
        relu_last = False
        if i < len(order) and nodes[order[i]]["opkind"]["op"] == "LeakyReLU":
            n = nodes[order[i]]
            if n["opkind"]["slope"] != 0.0 or n["inputs"] != [(cur, 0)]:
                raise ValueError("unsupported compiled circuit: LeakyReLU with a slope")
            cur, i, relu_last = order[i], i + 1, True
    if model["outputs"][0] != (cur, 0) or not (weights or relu_first):
        raise ValueError("unsupported compiled circuit: output node")
    return weights, biases, relu_last, relu_first, n_inputs


G2_GENERATOR = ((0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed, 0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2),
                (0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa, 0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b))


def gen_srs(srs_path, logrows, secret=None):
    """`ezkl gen-srs` (/root/reference/src/execute.rs gen_srs -> pfsys::srs::gen_srs, src/pfsys/srs.rs:13-16 -> ParamsKZG::setup): an INSECURE
    test SRS from a secret s -- g[i] = s^i G, g_lagrange[i] = L_i(s) G and s_g2 = [s] g2, all on the device (the G1 sets by
    backend.gen_srs, the G2 point by ezkl_hip_msm_g2) -- written in halo2's raw-bytes layout.  secret=None draws s from OS entropy."""
    R_ = EL.R
    if secret is None:
        secret = int.from_bytes(os.urandom(40), "little")
    secret %= R_
    mont_q = lambda v: np.frombuffer((v * (1 << 256) % B._Q).to_bytes(32, "little"), np.uint64)
    mont_r = lambda v: np.frombuffer((v * (1 << 256) % R_).to_bytes(32, "little"), np.uint64)
    g2 = np.concatenate([mont_q(c) for c in (G2_GENERATOR[0][0], G2_GENERATOR[0][1], G2_GENERATOR[1][0], G2_GENERATOR[1][1])])
    s_g2 = B.msm_g2(g2[None], mont_r(secret)[None])
    g, gl = B.gen_srs(logrows, secret)
    try:
        data = codecs.write_srs(dict(k=logrows, g=g.download(), g_lagrange=gl.download(), g2=g2.tobytes(), s_g2=s_g2.tobytes()))
    finally:
        g.free(); gl.free()
    open(srs_path, "wb").write(data)
    return len(data)


def load_params_prover(srs_path, logrows):
    """execute.rs:1739-1750: read the SRS and, if the file is larger than the circuit (the normal case with a shared kzg22.srs),
    `params.downsize(logrows)`: the coefficient basis is truncated, the Lagrange basis of the smaller domain is rebuilt on the device by an
    inverse NTT over G1 (ezkl_hip_bases_downsize = halo2's g_to_lagrange)"""
    srs = codecs.read_srs(open(srs_path, "rb").read())
    if srs["k"] < logrows:
        raise ValueError("SRS too small: k=%d < logrows=%d" % (srs["k"], logrows))
    if srs["k"] > logrows:
        n = 1 << logrows
        big = B.Bases(np.ascontiguousarray(srs["g"][:n]))
        try:
            g, gl = big.downsize(logrows)
            try:
                srs = dict(k=logrows, g=g.download(), g_lagrange=gl.download(), g2=srs["g2"], s_g2=srs["s_g2"])
            finally:
                g.free(); gl.free()
        finally:
            big.free()
    return srs


def setup(compiled_circuit, srs_path, vk_path, pk_path, sample_input=None):
    """keygen_vk + keygen_pk on the GPU, keys written in halo2's raw-bytes layout"""
    circuit, j = _load_circuit(compiled_circuit)
    srs = load_params_prover(srs_path, circuit.k)
    x = sample_input if sample_input is not None else [0] * circuit.n_inputs
    cs, fixed, copies, reg = circuit.keygen_inputs(x)          # synthesis without witness values: selectors + copy constraints
    bg = B.Bases(srs["g"])
    try:
        pk = NV.NativeProvingKey(NV.NativeCircuit(cs), bg, EL.cols_to_mont(fixed, B), copies)
        pk.set_selectors(reg.selector_rows())
        data = pk.to_bytes()
    finally:
        bg.free()
    open(pk_path, "wb").write(data)
    vk_len = 7 + 64 * (cs.n_fixed + len(cs.perm)) + cs.n_selectors * ((cs.n + 7) // 8)
    open(vk_path, "wb").write(data[:vk_len])                   # vk.key is the prefix of pk.key
    return dict(n_advice=cs.n_advice, n_fixed=cs.n_fixed, n_lookups=len(cs.lookups), degree=cs.degree, pk_bytes=len(data))


def _rust_round(x):
    """f64::round: half away from zero (Python's round() is half to even)"""
    import math
    return int(math.floor(x + 0.5)) if x >= 0 else -int(math.floor(-x + 0.5))


def _quantize(v, scale, datum_type):
    """FileSourceInner::{as_type, to_field} (src/graph/input.rs:85-111) -> quantize_float (src/graph/utilities.rs:53-69): the value as the
    input's datum type would hold it (InputType::roundtrip, src/circuit/ops/mod.rs:112-141), times 2^scale, rounded half away from zero;
    values beyond what an i128 holds at that scale are the reference's SigBitTruncationError"""
    import struct
    if isinstance(v, bool):
        return 1 if v else 0
    if isinstance(v, str):                                   # a field element, as hex (FileSourceInner::Field)
        return codecs.felt_from_hex_le(v) % EL.R
    x = float(v)
    if datum_type == "F32":
        x = struct.unpack("f", struct.pack("f", x))[0]
    elif datum_type == "F16":                                # through half precision, as InputType::roundtrip does (f16::from_f64)
        x = float(np.float16(x))
    elif datum_type in ("Int", "TDim"):
        x = float(int(x))
    elif datum_type == "Bool":
        if int(x) not in (0, 1):
            raise ValueError("a Bool input must be 0 or 1")
        x = float(int(x))
    mult = 2.0 ** scale
    if abs(x) > round((2.0 ** 127 - 1) / mult):
        raise ValueError("SigBitTruncationError: the input does not fit the fixed-point representation")
    return _rust_round(mult * x)


def gen_witness(compiled_circuit, data, output=None, vk_path=None, srs_path=None):
    """`ezkl gen-witness` (/root/reference/src/execute.rs:577-660 -> GraphCircuit::forward, src/graph/mod.rs:1734-1849; Python binding
    src/bindings/python.rs:914-941): the compiled circuit + the input data (GraphData JSON: a path, a JSON string or a dict with
    "input_data") -> the GraphWitness, written to `output` as the reference's witness.json and returned as a dict.

    * inputs are quantized as load_graph_input does (datum-type round trip, times 2^scale, half away from zero);
    * the forward pass is the integer arithmetic of the op family this package lays out (_mlp_of_graph: Einsum "mk,nk->mn", Add,
      LeakyReLU slope 0 at scale 0) with the range checks the layout would make: a value outside (-base^legs, base^legs) where the
      circuit decomposes it is the reference's decomposition error, here a ValueError; max_range_size = decomp_base - 1;
    * KZGCommit visibility of the input / parameters / output ("polycommit"): PolyCommitChip::commit on the GPU (backend.polycommit_commit:
      one commit_lagrange MSM per column of 2^k - (blinding factors + 1) values) -- needs srs_path, and like the reference a vk to know the
      blinding factors (here: any file at vk_path; the count is read off the constraint system);  without an SRS the processed value stays
      None, as in the reference ("SRS for poly commit does not exist (will be ignored)").  Hashed (Poseidon) visibility is refused.
    The statistics fields are what the reference's dummy layout reports for this family: no lookups (0, 0), max_range_size."""
    raw = open(compiled_circuit, "rb").read()
    if raw[:1] == b"{":
        j = json.loads(raw)
        if j.get("model") != "mlp":
            raise ValueError("unsupported compiled circuit: %r" % j.get("model"))
        ra = j["run_args"]
        weights, biases, relu_last = j["weights"], j["biases"], j.get("relu_last", True)
        relu_first, n_inputs = j.get("relu_first", False), j.get("n_inputs", len(weights[0][0]) if weights else None)
        vis = dict(input=ra.get("input_visibility", "Private"), params=ra.get("param_visibility", "Private"), output=ra.get("output_visibility", "Public"))
        in_scale = out_scale = ra.get("input_scale", 0) if not weights else 0       # a Gemm-free graph keeps its input scale (ezkl's default: 7)
        datum_type, input_decomp = "F32", True
    else:
        c = codecs.read_compiled_circuit(raw)
        weights, biases, relu_last, relu_first, n_inputs = _mlp_of_graph(c["model"], any_visibility=True)
        ra, vis = c["settings"]["run_args"], c["model"]["visibility"]
        in_node = c["model"]["nodes"][c["model"]["inputs"][0]]
        in_scale, out_scale = c["settings"]["model_input_scales"][0], c["settings"]["model_output_scales"][0]
        datum_type, input_decomp = in_node["opkind"].get("datum_type", "F32"), in_node["opkind"].get("decomp", True)
    for what, v in vis.items():
        if isinstance(v, dict):
            raise ValueError("unsupported visibility: %s is Hashed (Poseidon modules are outside this package's scope)" % what)
    base, legs = ra["decomp_base"], ra["decomp_legs"]
    if isinstance(data, (bytes, str)) and os.path.exists(data):
        data = open(data).read()
    if isinstance(data, (bytes, str)):
        data = json.loads(data)
    cols = data["input_data"]
    if len(cols) != 1 or len(cols[0]) != n_inputs:
        raise ValueError("input data does not match the circuit's input shape")
    x = [_quantize(v, in_scale, datum_type) for v in cols[0]]
    signed = lambda v: v if v < EL.R // 2 else v - EL.R
    x = [signed(v % EL.R) for v in x]
    limit = base ** legs
    ranged = False
    def decomposed(vals, where):
        nonlocal ranged
        ranged = True
        for v in vals:
            if abs(v) >= limit:
                raise ValueError("%s: %d exceeds the decomposition range (-%d^%d, %d^%d)" % (where, v, base, legs, base, legs))
    inputs = list(x)
    if input_decomp:
        decomposed(x, "input")
    if relu_first:
        decomposed(x, "LeakyReLU input")
        x = [v if v > 0 else 0 for v in x]
    for li, (W, bvec) in enumerate(zip(weights, biases)):
        x = [sum(a * w for a, w in zip(x, row)) + bvec[o] for o, row in enumerate(W)]
        if li + 1 < len(weights) or relu_last:
            decomposed(x, "LeakyReLU input of layer %d" % li)
            x = [v if v > 0 else 0 for v in x]
    outputs = x
    decomposed(outputs, "output")                        # `output` range-checks the outputs and the instance cells (layouts.rs:6740-6779)
    processed = dict(input=None, params=None, output=None)
    if any(v == "KZGCommit" for v in vis.values()):
        have_srs = srs_path is not None and os.path.exists(srs_path)
        if vk_path is not None and have_srs:
            blinding = 5                 # vk.cs().blinding_factors() of this gate family: no advice column is queried at more than 3 rotations
            srs = load_params_prover(srs_path, ra["logrows"])
            params = B.ParamsKZG(ra["logrows"], srs["g"], srs["g_lagrange"])
            try:
                def commit(vals):
                    msg = np.stack([np.frombuffer(((v % EL.R) * (1 << 256) % EL.R).to_bytes(32, "little"), np.uint64) for v in vals])
                    pts = B.polycommit_commit(msg, blinding + 1, params)
                    out = []
                    for p in pts:
                        xi, yi = (int.from_bytes(p[4 * h:4 * h + 4].tobytes(), "little") * pow(1 << 256, -1, B._Q) % B._Q for h in (0, 1))
                        out.append({"x": codecs.felt_to_hex_le(xi), "y": codecs.felt_to_hex_le(yi)})
                    return {"poseidon_hash": None, "polycommit": [out]}
                if vis["input"] == "KZGCommit": processed["input"] = commit(inputs)
                if vis["params"] == "KZGCommit": processed["params"] = commit([w for W in weights for row in W for w in row] + [b for bv in biases for b in bv])
                if vis["output"] == "KZGCommit": processed["output"] = commit(outputs)
            finally:
                params.free()
        else:
            for what, v in vis.items():                      # the reference: a module result without its commitments
                if v == "KZGCommit": processed[what] = {"poseidon_hash": None, "polycommit": None}
    text = codecs.write_witness_json([[v % EL.R for v in inputs]], [[v % EL.R for v in outputs]], [in_scale], [out_scale],
                                     processed_inputs=processed["input"], processed_params=processed["params"], processed_outputs=processed["output"],
                                     max_lookup_inputs=0, min_lookup_inputs=0, max_range_size=(base - 1) if ranged else 0)
    if output is not None:
        open(output, "w").write(text)
    return json.loads(text)


def prove(witness_path, compiled_circuit, pk_path, proof_path, srs_path, check_mode=CheckMode.UNSAFE, seed=0, recommit=False):
    """GraphWitness + compiled circuit + pk + SRS files -> proof.json (Snark).  seed = 0: OS entropy (OsRng); otherwise the
    reference's det-prove.  CheckMode.SAFE verifies the proof before returning it, as create_proof_circuit does."""
    w = codecs.read_witness_json(open(witness_path).read())
    circuit, j = _load_circuit(compiled_circuit)
    if len(w["inputs"]) != 1 or len(w["inputs"][0]) != circuit.n_inputs:
        raise ValueError("witness does not match the circuit's input shape")
    signed = lambda v: v if v < EL.R // 2 else v - EL.R
    cs = circuit.gc.cs
    adv, inst = circuit.witness([signed(v) for v in w["inputs"][0]])            # GraphCircuit::synthesize
    if w["outputs"] and inst != w["outputs"]:
        raise ValueError("the witness file's outputs do not match the circuit's outputs")
    ncs = _plonk_cs(circuit)
    srs = load_params_prover(srs_path, circuit.k)                               # downsizes an SRS file larger than the circuit
    bg, bgl = B.Bases(srs["g"]), B.Bases(srs["g_lagrange"])
    try:
        # recommit: the key file's commitments were made under ANOTHER SRS (the reference's fixture key: the public powers of tau)
        pk = NV.NativeProvingKey.from_bytes(NV.NativeCircuit(ncs), open(pk_path, "rb").read(), recommit=bg if recommit else None)
        proof = NV.create_proof(pk, bg, bgl, EL.cols_to_mont(adv, B), seed=seed, instances=inst, check_mode=check_mode,
                                g2=srs["g2"], s_g2=srs["s_g2"])
    finally:
        bg.free(); bgl.free()
    open(proof_path, "w").write(codecs.write_proof_json(proof, inst))
    return proof


def verify(proof_path, compiled_circuit, vk_path=None, srs_path=None, recommit=False, pk_path=None):
    """-> True / False, from the proof, the compiled circuit (settings), vk.key and the SRS's g2 / s_g2 -- what the reference's `verify`
    reads (src/execute.rs:1651).  Host only: no GPU, no proving key, none of the prover's private weights.  A pk.key path works too (vk.key
    is its prefix).  recommit=True (a key whose commitments were made under another SRS, re-committed by `prove`) needs the proving key and
    a device: the fixed / permutation polynomials are committed again under this SRS first."""
    if vk_path is None:
        vk_path = pk_path                                                       # the parameter's former name: keyword callers keep working
    if vk_path is None or srs_path is None:
        raise TypeError("verify needs a vk.key (or pk.key) path and an SRS path")
    circuit, j = _load_circuit(compiled_circuit)
    pr = codecs.read_proof_json(open(proof_path).read())
    srs = codecs.read_srs(open(srs_path, "rb").read())
    if recommit:
        cs_ = _plonk_cs(circuit)
        # the exact length of a vk.key of this constraint system (codecs.py): header, the fixed + permutation commitments AND the bit-packed
        # selector rows (80 selectors at k = 20 are 10 MB: ADVICE r04 -- a bound without them let a vk-only file through to a later,
        # misleading "proving key truncated")
        vk_len = 7 + 64 * (cs_.n_fixed + len(cs_.perm)) + cs_.n_selectors * (((1 << cs_.k) + 7) // 8)
        if os.path.getsize(vk_path) <= vk_len:                                  # a vk-only file has no polynomials to commit again
            raise ValueError("recommit=True needs the PROVING key file (its polynomials are committed again under this SRS); %s is a verifying key" % vk_path)
    if not recommit:
        return NV.verify_proof_vk(NV.NativeCircuit(_plonk_cs(circuit)), open(vk_path, "rb").read(), srs["g2"], srs["s_g2"], pr["proof"], pr["instances"])
    bg = B.Bases(srs["g"])
    try:
        pk = NV.NativeProvingKey.from_bytes(NV.NativeCircuit(_plonk_cs(circuit)), open(vk_path, "rb").read(), recommit=bg)
        return NV.verify_proof(pk, srs["g2"], srs["s_g2"], pr["proof"], pr["instances"])
    finally:
        bg.free()


def _plonk_cs(circuit):
    """the constraint system as keygen saw it: re-running `configure` + selector compression (halo2 does the same on load_pk,
    src/pfsys/mod.rs:627) -- a fresh synthesis pass without witness values gives the selector activations"""
    fresh = EL.MlpCircuit(circuit.k, circuit.w, circuit.weights, circuit.biases, circuit.base, circuit.legs,
                          total_assignments=circuit.settings.total_assignments, relu_last=circuit.relu_last, n_inputs=circuit.n_inputs,
                          relu_first=circuit.relu_first)
    return fresh.keygen_inputs([0] * circuit.n_inputs)[0]
