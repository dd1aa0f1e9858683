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
    """ParamsKZG::downsize for the coefficient basis (src/execute.rs:1739-1750).  g_lagrange of the smaller domain is
    NOT a prefix of the larger one; halo2 recomputes it with an inverse FFT over G1, which is out of scope here."""
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
    polynomials and their extended cosets, one HBM column each (field-SoA)."""

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
