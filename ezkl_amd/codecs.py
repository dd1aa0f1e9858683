"""File codecs for the artefacts that bracket the prove hot path, and device-resident loaders for them
(SURVEY.md §8(f) item 1).  Formats are the raw-bytes layouts of the reference, verified on its fixtures
(SURVEY.md §8(c) items 1, 3, 7):

  SRS  (/root/reference/src/pfsys/srs.rs:30-47 -> ParamsKZG::{read,write})
        u32 LE k | 2^k x G1 g | 2^k x G1 g_lagrange | G2 g2 (128 B) | G2 s_g2 (128 B)
  VK   (/root/reference/src/pfsys/mod.rs:593-613 load_vk -> VerifyingKey::read, SerdeFormat::RawBytes)
        [version=3, k, compress_selectors] | u32 LE #fixed | #fixed x G1 | #perm x G1 | selectors bit-packed (n/8 B each)
  PK   (src/pfsys/mod.rs:615-636 load_pk -> ProvingKey::read)
        VK | poly l0 | poly l_last | poly l_active_row | vec fixed_values | vec fixed_polys | vec fixed_cosets
           | vec permutations | vec perm_polys | vec perm_cosets
        poly = u32 BE len | len x 32 B ;  vec = u32 BE count | count x u32 BE len | count x poly
  proof.json  (src/pfsys/mod.rs:198-315 Snark): "hex_proof" = 0x + hex(proof bytes); instances = 32-byte LE hex felts

  model.compiled  (src/graph/mod.rs:1250-1268 GraphCircuit::save / load: bincode, fixed-width little-endian integers, u64 lengths, u32
        enum tags) -- read_compiled_circuit: the Model (nodes of src/graph/node.rs:522-536 with SupportedOp / PolyOp / Input / Constant,
        field elements as 32 canonical LE bytes) and the GraphSettings 19-tuple (src/graph/mod.rs:545-568), checked against the
        reference's fixture (its settings.json says the same, its weights are the fixture model's)

All field / curve bytes stay in their on-disk Montgomery form: they are handed to the GPU unchanged.
The number of permutation columns and selectors is not stored in the file (halo2 re-derives it by re-running
`configure`, src/pfsys/mod.rs:627); callers pass them."""
import json
import struct
import numpy as np
from . import backend as _b


# ------------------------------------------------------------------ SRS
def read_srs(buf):
    k = struct.unpack_from("<I", buf, 0)[0]
    n = 1 << k
    if len(buf) != 4 + 128 * n + 256:
        raise ValueError("SRS length %d does not match k=%d" % (len(buf), k))
    g = np.frombuffer(buf, np.uint64, 8 * n, 4).reshape(n, 8)
    gl = np.frombuffer(buf, np.uint64, 8 * n, 4 + 64 * n).reshape(n, 8)
    off = 4 + 128 * n
    return dict(k=k, g=g, g_lagrange=gl, g2=bytes(buf[off:off + 128]), s_g2=bytes(buf[off + 128:off + 256]))


def write_srs(srs):
    n = 1 << srs["k"]
    g, gl = np.ascontiguousarray(srs["g"], np.uint64), np.ascontiguousarray(srs["g_lagrange"], np.uint64)
    assert g.shape == (n, 8) and gl.shape == (n, 8)
    return struct.pack("<I", srs["k"]) + g.tobytes() + gl.tobytes() + srs["g2"] + srs["s_g2"]


def downsize_srs(srs, k):
    """ParamsKZG::downsize for the coefficient basis only (src/execute.rs:1739-1750).  g_lagrange of the smaller domain is NOT a prefix
    of the larger one: halo2 recomputes it with an inverse FFT over G1 -- on the device: ezkl_amd.backend.ParamsKZG.downsize /
    execute.load_params_prover (ezkl_hip_bases_downsize)."""
    if k > srs["k"]:
        raise ValueError("cannot upsize")
    return dict(k=k, g=srs["g"][: 1 << k], g_lagrange=None, g2=srs["g2"], s_g2=srs["s_g2"])


# ------------------------------------------------------------------ polys / vecs
def _read_poly(buf, off):
    ln = struct.unpack_from(">I", buf, off)[0]
    off += 4
    a = np.frombuffer(buf, np.uint64, 4 * ln, off).reshape(ln, 4)
    return a, off + 32 * ln


def _write_poly(a):
    a = np.ascontiguousarray(a, np.uint64)
    return struct.pack(">I", a.shape[0]) + a.tobytes()


def _read_vec(buf, off):
    cnt = struct.unpack_from(">I", buf, off)[0]
    off += 4 + 4 * cnt
    out = []
    for _ in range(cnt):
        a, off = _read_poly(buf, off)
        out.append(a)
    return out, off


def _write_vec(polys):
    return struct.pack(">I", len(polys)) + b"".join(struct.pack(">I", p.shape[0]) for p in polys) + b"".join(_write_poly(p) for p in polys)


# ------------------------------------------------------------------ VK / PK
def read_vk(buf, n_perm, n_selectors=None):
    if buf[0] != 3:
        raise ValueError("unsupported key version %d" % buf[0])
    k, compress = buf[1], bool(buf[2])
    n = 1 << k
    nfixed = struct.unpack_from("<I", buf, 3)[0]
    off = 7
    fixed = np.frombuffer(buf, np.uint64, 8 * nfixed, off).reshape(nfixed, 8)
    off += 64 * nfixed
    perm = np.frombuffer(buf, np.uint64, 8 * n_perm, off).reshape(n_perm, 8)
    off += 64 * n_perm
    row_bytes = (n + 7) // 8
    if n_selectors is None:
        n_selectors = (len(buf) - off) // row_bytes
    sel = np.frombuffer(buf, np.uint8, n_selectors * row_bytes, off).reshape(n_selectors, row_bytes)
    off += n_selectors * row_bytes
    return dict(k=k, compress_selectors=compress, fixed_commitments=fixed, permutation_commitments=perm,
                selectors=np.unpackbits(sel, axis=1, bitorder="little")[:, :n].astype(bool), end=off)


def write_vk(vk):
    sel = np.packbits(np.asarray(vk["selectors"], bool), axis=1, bitorder="little")
    return (bytes([3, vk["k"], 1 if vk["compress_selectors"] else 0]) + struct.pack("<I", vk["fixed_commitments"].shape[0])
            + np.ascontiguousarray(vk["fixed_commitments"], np.uint64).tobytes()
            + np.ascontiguousarray(vk["permutation_commitments"], np.uint64).tobytes() + sel.tobytes())


_PK_POLYS = ("l0", "l_last", "l_active_row")
_PK_VECS = ("fixed_values", "fixed_polys", "fixed_cosets", "permutations", "perm_polys", "perm_cosets")


def read_pk(buf, n_perm, n_selectors):
    vk = read_vk(buf, n_perm, n_selectors)
    off = vk["end"]
    pk = dict(vk=vk)
    for name in _PK_POLYS:
        pk[name], off = _read_poly(buf, off)
    for name in _PK_VECS:
        pk[name], off = _read_vec(buf, off)
    if off != len(buf):
        raise ValueError("trailing bytes in pk (%d of %d consumed): wrong n_perm / n_selectors?" % (off, len(buf)))
    return pk


def write_pk(pk):
    return write_vk(pk["vk"]) + b"".join(_write_poly(pk[n]) for n in _PK_POLYS) + b"".join(_write_vec(pk[n]) for n in _PK_VECS)


class ProvingKeyDevice:
    """Device-resident image of the pk columns evaluate_h reads: l0 / l_last / l_active_row, fixed and permutation
    polynomials and their extended cosets, one HBM column each (field-SoA, NOTEBOOK.md §3)."""

    def __init__(self, pk):
        self.k = pk["vk"]["k"]
        up = _b.DeviceBuffer.from_numpy
        self.l0, self.l_last, self.l_active_row = up(pk["l0"]), up(pk["l_last"]), up(pk["l_active_row"])
        self.fixed_polys = [up(p) for p in pk["fixed_polys"]]
        self.fixed_cosets = [up(p) for p in pk["fixed_cosets"]]
        self.perm_polys = [up(p) for p in pk["perm_polys"]]
        self.perm_cosets = [up(p) for p in pk["perm_cosets"]]
        self.ext_k = int(np.log2(pk["l0"].shape[0]))

    def nbytes(self):
        bufs = [self.l0, self.l_last, self.l_active_row] + self.fixed_polys + self.fixed_cosets + self.perm_polys + self.perm_cosets
        return sum(b.nbytes for b in bufs)


# ------------------------------------------------------------------ proof / felts
def felt_from_hex_le(h):
    """instances / witness felts: 32-byte little-endian canonical hex (src/pfsys/mod.rs:161-175)"""
    return int.from_bytes(bytes.fromhex(h), "little")


def felt_to_hex_le(x):
    return int(x).to_bytes(32, "little").hex()


def read_proof_json(text):
    j = json.loads(text)
    proof = bytes.fromhex(j["hex_proof"][2:]) if "hex_proof" in j and j["hex_proof"] else bytes(j.get("proof", []))
    inst = [[felt_from_hex_le(h) for h in col] for col in j.get("instances", [])]
    return dict(proof=proof, instances=inst, transcript_type=j.get("transcript_type"), raw=j)


def split_evm_proof(proof, n_commitments, n_evals):
    """EvmTranscript layout: n x G1 as 32-B BE x || y, then scalars 32-B BE, then the 2 SHPLONK points"""
    pts = [(int.from_bytes(proof[64 * i:64 * i + 32], "big"), int.from_bytes(proof[64 * i + 32:64 * i + 64], "big")) for i in range(n_commitments)]
    off = 64 * n_commitments
    ev = [int.from_bytes(proof[off + 32 * i: off + 32 * i + 32], "big") for i in range(n_evals)]
    off += 32 * n_evals
    tail = [(int.from_bytes(proof[off + 64 * i: off + 64 * i + 32], "big"), int.from_bytes(proof[off + 64 * i + 32: off + 64 * i + 64], "big"))
            for i in range((len(proof) - off) // 64)]
    return pts, ev, tail


def write_proof_json(proof, instances, pretty_public_inputs=None, timestamp_ms=None, version="ezkl_amd"):
    """`Snark::save` (src/pfsys/mod.rs:198-230, 291-298): serde_json of the Snark struct, compact, fields in declaration order --
    protocol (None for a plain proof), instances as 32-byte little-endian hex felts, the proof as a byte list AND as "0x.." hex
    (create_hex_proof, :285-289), split, pretty_public_inputs, timestamp (ms), version."""
    import time
    j = {"protocol": None,
         "instances": [[felt_to_hex_le(v) for v in col] for col in instances],
         "proof": list(proof),
         "hex_proof": "0x" + bytes(proof).hex(),
         "split": None,
         "pretty_public_inputs": pretty_public_inputs,
         "timestamp": int(time.time() * 1000) if timestamp_ms is None else timestamp_ms,
         "version": version}
    return json.dumps(j, separators=(",", ":"))


def rust_f64_to_string(x):
    """`f64::to_string()` (Display): shortest digits that round-trip, never an exponent, no trailing ".0" -- what
    GraphWitness::generate_rescaled_elements prints (src/graph/mod.rs:188, 204)"""
    x = float(x)
    if x != x: return "NaN"
    if x in (float("inf"), float("-inf")): return "inf" if x > 0 else "-inf"
    r = repr(x)
    if "e" in r or "E" in r:
        from decimal import Decimal
        r = format(Decimal(r), "f")
    if r.endswith(".0"): r = r[:-2]
    return "-0" if r == "-0" else r


def write_witness_json(inputs, outputs, input_scales, output_scales, processed_inputs=None, processed_params=None, processed_outputs=None,
                       max_lookup_inputs=0, min_lookup_inputs=0, max_range_size=0, version="source - no compatibility guaranteed"):
    """`GraphWitness::save` (src/graph/mod.rs:120-141, 174-244, as_json): serde_json, compact, fields in declaration order -- inputs /
    outputs as 32-byte little-endian hex felts, pretty_elements (dequantized values as Rust prints an f64, felts as `{:?}` = "0x" + 64
    big-endian hex digits), the processed_* module results ({"poseidon_hash": ..., "polycommit": [[{"x","y"}]]} or null), the lookup
    statistics, the version string.  inputs / outputs: lists of lists of ints mod r."""
    from . import ezkl_layout as _EL
    R_ = _EL.R
    def signed(v): return v if v < R_ // 2 else v - R_
    def dbg(v): return "0x%064x" % (v % R_)
    def resc(cols, scales): return [[rust_f64_to_string(signed(v % R_) / float(2.0 ** scales[i])) for v in col] for i, col in enumerate(cols)]
    pretty = {"rescaled_inputs": resc(inputs, input_scales),
              "inputs": [[dbg(v) for v in col] for col in inputs],
              "processed_inputs": [], "processed_params": [], "processed_outputs": [],
              "rescaled_outputs": resc(outputs, output_scales),
              "outputs": [[dbg(v) for v in col] for col in outputs]}
    j = {"inputs": [[felt_to_hex_le(v % R_) for v in col] for col in inputs],
         "pretty_elements": pretty,
         "outputs": [[felt_to_hex_le(v % R_) for v in col] for col in outputs],
         "processed_inputs": processed_inputs, "processed_params": processed_params, "processed_outputs": processed_outputs,
         "max_lookup_inputs": max_lookup_inputs, "min_lookup_inputs": min_lookup_inputs, "max_range_size": max_range_size,
         "version": version}
    return json.dumps(j, separators=(",", ":"))


def read_witness_json(text):
    """GraphWitness (src/graph/mod.rs:120-141, loaded by `prove` at src/execute.rs:1584): inputs / outputs as 32-byte little-endian
    hex felts, the optional processed_* module results, and the lookup statistics -> ints"""
    j = json.loads(text)
    def felts(t): return [[felt_from_hex_le(h) for h in col] for col in (t or [])]
    out = dict(inputs=felts(j.get("inputs")), outputs=felts(j.get("outputs")), raw=j)
    for key in ("processed_inputs", "processed_params", "processed_outputs"):
        out[key] = j.get(key)
    for key in ("max_lookup_inputs", "min_lookup_inputs", "max_range_size"):
        out[key] = j.get(key)
    return out


# ------------------------------------------------------------------ model.compiled (bincode of ezkl's GraphCircuit)
_FR = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
_INPUT_TYPES = ["Bool", "F16", "F32", "F64", "Int", "TDim", "Unknown"]                       # src/circuit/ops/mod.rs:88-103
# src/circuit/ops/lookup.rs:16-38: (name, number of f32 fields)
_LOOKUP_OPS = [("Div", 1), ("IsOdd", 0), ("PowersOfTwo", 1), ("Ln", 1), ("Sigmoid", 1), ("Exp", 2), ("Cos", 1), ("ACos", 1), ("Cosh", 1), ("ACosh", 1),
               ("Sin", 1), ("ASin", 1), ("Sinh", 1), ("ASinh", 1), ("Tan", 1), ("ATan", 1), ("Tanh", 1), ("ATanh", 1), ("Erf", 1), ("Pow", 2), ("HardSwish", 1)]


class _Bincode:
    """bincode 1.x default options: little-endian fixed-width integers, u64 sequence lengths, u32 enum variant tags, u8 Option tags"""

    def __init__(self, buf):
        self.b, self.o = bytes(buf), 0

    def take(self, n):
        v = self.b[self.o:self.o + n]
        if len(v) != n:
            raise ValueError("compiled circuit truncated at byte %d" % self.o)
        self.o += n
        return v

    def u8(self): return self.take(1)[0]
    def u32(self): return struct.unpack("<I", self.take(4))[0]
    def i32(self): return struct.unpack("<i", self.take(4))[0]
    def u64(self): return struct.unpack("<Q", self.take(8))[0]
    def f32(self): return struct.unpack("<f", self.take(4))[0]
    def f64(self): return struct.unpack("<d", self.take(8))[0]
    def i128(self): return int.from_bytes(self.take(16), "little", signed=True)
    def u128(self): return int.from_bytes(self.take(16), "little")

    def boolean(self):
        v = self.u8()
        if v > 1:
            raise ValueError("bad bool %d at byte %d" % (v, self.o - 1))
        return bool(v)

    def vec(self, f):
        n = self.u64()
        if n > len(self.b):
            raise ValueError("bad sequence length %d at byte %d" % (n, self.o - 8))
        return [f() for _ in range(n)]

    def opt(self, f):
        t = self.u8()
        if t > 1:
            raise ValueError("bad Option tag %d at byte %d" % (t, self.o - 1))
        return f() if t else None

    def string(self): return bytes(self.take(self.u64())).decode()

    def char(self):
        b0 = self.u8()
        n = 1 if b0 < 0x80 else 2 if b0 < 0xe0 else 3 if b0 < 0xf0 else 4
        return (bytes([b0]) + bytes(self.take(n - 1))).decode()

    def tag(self, names, what):
        """a u32 enum variant tag, checked against the enum's variant list"""
        t = self.u32()
        if t >= len(names):
            raise ValueError("bad %s tag %d at byte %d" % (what, t, self.o - 4))
        return names[t]

    def felt(self):
        v = int.from_bytes(self.take(32), "little")
        if v >= _FR:
            raise ValueError("non-canonical field element at byte %d" % (self.o - 32))
        return v


def _visibility(r):
    """src/graph/vars.rs:22-41"""
    t = r.u32()
    if t == 2:
        return {"Hashed": {"hash_is_public": r.boolean(), "outlets": r.vec(r.u64)}}
    if t > 4:
        raise ValueError("bad Visibility tag %d" % t)
    return ["Private", "Public", None, "KZGCommit", "Fixed"][t]


def _tensor(r, elem):
    """src/tensor/mod.rs:176-182"""
    return dict(inner=r.vec(elem), dims=r.vec(r.u64), scale=r.opt(r.i32), visibility=r.opt(lambda: _visibility(r)))


def _poly_op(r):
    """src/circuit/ops/poly.rs:15-108 (the variants an MLP-family graph holds; the others are refused by name)"""
    names = ["Abs", "Sign", "LeakyReLU", "GatherElements", "GatherND", "ScatterElements", "ScatterND", "MultiBroadcastTo", "Einsum", "Conv", "Downsample",
             "DeConv", "Add", "Sub", "Neg", "Mult", "Identity", "Reshape", "MoveAxis", "Flatten", "Pad", "Sum", "MeanOfSquares", "Prod", "Pow", "Concat",
             "Slice", "Iff", "Resize", "Not", "And", "Or", "Xor", "Trilu"]
    t = r.u32()
    if t >= len(names):
        raise ValueError("bad PolyOp tag %d" % t)
    name = names[t]
    if name in ("Abs", "Sign", "Add", "Sub", "Neg", "Mult", "Iff", "Not", "And", "Or", "Xor"): return {"op": name}
    if name == "LeakyReLU": return {"op": name, "slope": r.f32(), "scale": r.i32()}
    if name == "MultiBroadcastTo": return {"op": name, "shape": r.vec(r.u64)}
    if name == "Einsum": return {"op": name, "equation": r.string()}
    if name == "Identity": return {"op": name, "out_scale": r.opt(r.i32)}
    if name in ("Reshape", "Flatten"): return {"op": name, "shape": r.vec(r.u64)}
    if name in ("Sum", "MeanOfSquares"): return {"op": name, "axes": r.vec(r.u64)}
    if name == "Pow": return {"op": name, "power": r.u32()}
    if name == "Concat": return {"op": name, "axis": r.u64()}
    if name == "Slice": return {"op": name, "axis": r.u64(), "start": r.u64(), "end": r.u64()}
    if name == "MoveAxis": return {"op": name, "source": r.u64(), "destination": r.u64()}
    if name == "Pad": return {"op": name, "padding": r.vec(lambda: (r.u64(), r.u64()))}
    raise ValueError("PolyOp::%s in a compiled circuit is not supported by this reader" % name)


def _supported_op(r, depth=0):
    """src/graph/node.rs:295-312"""
    if depth > 4:                                     # the reference never nests Rescaled; a damaged file must not become a RecursionError
        raise ValueError("SupportedOp::Rescaled nested too deeply")
    t = r.u32()
    if t == 0: return {"kind": "Linear", **_poly_op(r)}
    if t == 1:
        name, nf = r.tag(_LOOKUP_OPS, "LookupOp")
        return {"kind": "Nonlinear", "op": name, "params": [r.f32() for _ in range(nf)]}
    if t == 3: return {"kind": "Input", "scale": r.i32(), "datum_type": r.tag(_INPUT_TYPES, "InputType"), "decomp": r.boolean()}     # ops/mod.rs:186-193
    if t == 4:                                                                                                           # ops/mod.rs:295-305
        return {"kind": "Constant", "quantized_values": _tensor(r, r.felt), "raw_values": _tensor(r, r.f32), "decomp": r.boolean()}
    if t == 5: return {"kind": "Unknown"}
    if t == 6: return {"kind": "Rescaled", "inner": _supported_op(r, depth + 1), "scale": r.vec(lambda: (r.u64(), r.u128()))}     # node.rs:87-92
    raise ValueError("SupportedOp tag %d (Hybrid / RebaseScale) is not supported by this reader" % t)


def _run_args(r):
    """src/lib.rs:198-285.  The reference's fixture was written by a build whose RunArgs still had `commitment: Option<Commitments>`
    after check_mode (its settings.json shows the key); both layouts are accepted and the caller
    checks the result (run_args.check_mode must equal the settings' own check_mode, the version string must be text)."""
    a = dict(input_scale=r.i32(), param_scale=r.i32(), rebase_scale=r.opt(r.i32), scale_rebase_multiplier=r.u32(), lookup_range=(r.i128(), r.i128()),
             logrows=r.u32(), num_inner_cols=r.u64(), variables=r.vec(lambda: (r.string(), r.u64())), input_visibility=_visibility(r),
             output_visibility=_visibility(r), param_visibility=_visibility(r), rebase_frac_zero_constants=r.boolean())
    t = r.u32()
    if t > 1:
        raise ValueError("bad CheckMode tag %d" % t)
    a["check_mode"] = ["SAFE", "UNSAFE"][t]
    return a


def _settings(r, legacy):
    a = _run_args(r)
    if legacy:
        a["commitment"] = r.opt(lambda: r.tag(["KZG", "IPA"], "Commitments"))
    a["decomp_base"], a["decomp_legs"] = r.u64(), r.u64()
    a.update(bounded_log_lookup=r.boolean(), ignore_range_check_inputs_outputs=r.boolean(), epsilon=r.opt(r.f64), disable_freivalds=r.boolean())
    s = dict(run_args=a, num_rows=r.u64(), total_assignments=r.u64(), total_const_size=r.u64())                          # graph/mod.rs:545-568
    s.update(total_dynamic_col_size=r.u64(), max_dynamic_input_len=r.u64(), num_dynamic_lookups=r.u64(), num_shuffles=r.u64(), total_shuffle_col_size=r.u64())
    s["einsum_params"] = dict(equations=r.vec(lambda: (r.string(), dict(r.vec(lambda: (r.char(), r.u64()))))), total_einsum_col_size=r.u64())
    s.update(model_instance_shapes=r.vec(lambda: r.vec(r.u64)), model_output_scales=r.vec(r.i32), model_input_scales=r.vec(r.i32))
    s["module_sizes"] = dict(polycommit=r.vec(r.u64), poseidon=[r.u64(), r.vec(r.u64)])
    def lookup():
        name, nf = r.tag(_LOOKUP_OPS, "LookupOp")
        return {"op": name, "params": [r.f32() for _ in range(nf)]}
    s.update(required_lookups=r.vec(lookup), required_range_checks=r.vec(lambda: (r.i128(), r.i128())))
    t = r.u32()
    if t > 1:
        raise ValueError("bad CheckMode tag %d" % t)
    s.update(check_mode=["SAFE", "UNSAFE"][t], version=r.string(), num_blinding_factors=r.opt(r.u64), timestamp=r.opt(r.u128),
             input_types=r.opt(lambda: r.vec(lambda: r.tag(_INPUT_TYPES, "InputType"))), output_types=r.opt(lambda: r.vec(lambda: r.tag(_INPUT_TYPES, "InputType"))))
    if s["check_mode"] != a["check_mode"] or not s["version"].isprintable() or not (1 <= a["logrows"] <= 28) or a["decomp_base"] < 2:
        raise ValueError("settings do not parse under this layout")
    return s


def read_compiled_circuit(buf):
    """bincode of GraphCircuit { core: CoreCircuit { model, settings }, graph_witness } -> {"model": {...}, "settings": {...}}; the witness
    that trails the settings is not read (prove takes it from witness.json).  Nodes keep the reference's names."""
    r = _Bincode(buf)
    nodes = {}
    for _ in range(r.u64()):                                   # ParsedNodes.nodes: BTreeMap<usize, NodeType> (graph/model.rs:378-384)
        key = r.u64()
        if r.u32() != 0:
            raise ValueError("NodeType::SubGraph is not supported by this reader")
        nodes[key] = dict(opkind=_supported_op(r), out_scale=r.i32(), inputs=r.vec(lambda: (r.u64(), r.u64())), out_dims=r.vec(r.u64), idx=r.u64(),
                          num_uses=r.u64())
    model = dict(nodes=nodes, inputs=r.vec(r.u64), outputs=r.vec(lambda: (r.u64(), r.u64())), output_types=r.vec(lambda: r.tag(_INPUT_TYPES, "InputType")),
                 visibility=dict(input=_visibility(r), params=_visibility(r), output=_visibility(r)))
    start, err = r.o, None
    for legacy in (False, True):
        r.o = start
        try:
            return dict(model=model, settings=_settings(r, legacy), settings_layout="legacy (commitment field)" if legacy else "current")
        except (ValueError, IndexError, UnicodeDecodeError) as e:
            err = e
    raise ValueError("the GraphSettings of this compiled circuit do not parse: %s" % err)
