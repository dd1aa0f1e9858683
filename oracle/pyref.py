"""Pure-Python big-int restatement of the BN254 / halo2 KZG primitives on the ezkl prove path.

TEST INFRASTRUCTURE ONLY (oracle). Nothing in the product path (ezkl_amd/) may import this.
It is the slowest, most obviously-correct layer: it pins the C oracle (oracle/oracle.c), which in
turn pins the HIP kernels.  Semantics follow SURVEY.md §8(c) / Appendix A, which were verified
against the reference fixtures /root/reference/tests/assets/{kzg,kzg1.srs,pk.key,vk.key}.

Reference call sites restated here (the arithmetic itself lives in the un-vendored crates
halo2curves 0.7.0 @ b753a83 and zkonduit/halo2 @ 01c8884):
  * ParamsKZG::commit_lagrange / commit  -> msm()              (src/circuit/modules/polycommit.rs:71)
  * EvaluationDomain::lagrange_to_coeff  -> intt()             (src/circuit/modules/polycommit.rs:52)
  * EvaluationDomain::coeff_to_extended  -> coeff_to_extended()
  * integer_rep_to_felt                  -> int_to_felt()      (src/fieldutils.rs:9-17)
"""
Q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47  # base field
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001  # scalar field
MONT = 1 << 256
RINV_Q = pow(MONT, -1, Q)
RINV_R = pow(MONT, -1, R)
S = 28                                   # Fr two-adicity
ROOT = pow(7, (R - 1) >> S, R)           # 2^28-th primitive root used by halo2curves bn256::Fr
ZETA = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23  # Fr::ZETA (cube root of 1)
DELTA = pow(7, 1 << S, R)
assert ROOT == 0x03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c
assert pow(ZETA, 3, R) == 1 and ZETA != 1
assert DELTA == 0x09226b6e22c6f0ca64ec26aad4c86e715b5f898e5e963f25870e56bbe533e9a2


def omega(k):
    return pow(ROOT, 1 << (S - k), R)


# ---------------- byte codecs (raw-bytes format: Montgomery, little-endian u64x4) -----------------
def fr_from_mont_bytes(b):
    return int.from_bytes(b[:32], "little") * RINV_R % R


def fr_to_mont_bytes(x):
    return (x * MONT % R).to_bytes(32, "little")


def fq_from_mont_bytes(b):
    return int.from_bytes(b[:32], "little") * RINV_Q % Q


def fq_to_mont_bytes(x):
    return (x * MONT % Q).to_bytes(32, "little")


def g1_from_bytes(b):
    """64 B affine, (0,0) = identity -> None"""
    x, y = fq_from_mont_bytes(b[0:32]), fq_from_mont_bytes(b[32:64])
    return None if (x == 0 and y == 0) else (x, y)


def g1_to_bytes(p):
    if p is None:
        return bytes(64)
    return fq_to_mont_bytes(p[0]) + fq_to_mont_bytes(p[1])


def int_to_felt(x):
    """src/fieldutils.rs:9-17"""
    return x % R


# ---------------- G1: y^2 = x^3 + 3 over Fq, affine with None = identity ---------------------------
def g1_on_curve(p):
    return p is None or (p[1] * p[1] - p[0] ** 3 - 3) % Q == 0


def g1_neg(p):
    return None if p is None else (p[0], (-p[1]) % Q)


def g1_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % Q == 0:
            return None
        lam = 3 * p[0] * p[0] * pow(2 * p[1], -1, Q) % Q
    else:
        lam = (q[1] - p[1]) * pow(q[0] - p[0], -1, Q) % Q
    x = (lam * lam - p[0] - q[0]) % Q
    return (x, (lam * (p[0] - x) - p[1]) % Q)


def g1_mul(p, k):
    k %= R
    acc = None
    while k:
        if k & 1:
            acc = g1_add(acc, p)
        p = g1_add(p, p)
        k >>= 1
    return acc


def msm(scalars, points):
    acc = None
    for s, p in zip(scalars, points):
        acc = g1_add(acc, g1_mul(p, s))
    return acc


# ---------------- NTT (natural order in/out), halo2 EvaluationDomain semantics --------------------
def ntt(a, w):
    """a'[j] = sum_i a[i] w^(ij); iterative radix-2, O(n log n)."""
    n = len(a)
    a = list(a)
    logn = n.bit_length() - 1
    for i in range(n):
        j = int(format(i, "0%db" % logn)[::-1], 2) if logn else 0
        if i < j:
            a[i], a[j] = a[j], a[i]
    m = 1
    while m < n:
        wm = pow(w, n // (2 * m), R)
        for s in range(0, n, 2 * m):
            t = 1
            for j in range(m):
                u, v = a[s + j], a[s + j + m] * t % R
                a[s + j], a[s + j + m] = (u + v) % R, (u - v) % R
                t = t * wm % R
        m *= 2
    return a


def intt(a, w):
    n = len(a)
    ninv = pow(n, -1, R)
    return [x * ninv % R for x in ntt(a, pow(w, -1, R))]


def coeff_to_extended(coeffs, k, ext_k):
    """EvaluationDomain::coeff_to_extended: distribute powers of zeta (g_coset = ZETA, so only
    i mod 3 matters), zero-pad to 2^ext_k, forward NTT with omega_ext."""
    n, ne = 1 << k, 1 << ext_k
    assert len(coeffs) == n
    z = [1, ZETA, ZETA * ZETA % R]
    a = [c * z[i % 3] % R for i, c in enumerate(coeffs)] + [0] * (ne - n)
    return ntt(a, omega(ext_k))


def extended_to_coeff(ext, k, ext_k):
    """inverse of the above without truncation (EvaluationDomain::extended_to_coeff): iNTT with
    omega_ext^-1, scale 1/2^ext_k, then multiply by zeta^-i (g_coset_inv = ZETA^2)."""
    a = intt(ext, omega(ext_k))
    zi = [1, ZETA * ZETA % R, ZETA]
    return [c * zi[i % 3] % R for i, c in enumerate(a)]


# ---------------- file parsers (formats: SURVEY.md §8(c) items 1,3) -------------------------------
def parse_srs(buf):
    k = int.from_bytes(buf[0:4], "little")
    n = 1 << k
    off = 4
    g = [buf[off + 64 * i: off + 64 * i + 64] for i in range(n)]
    off += 64 * n
    gl = [buf[off + 64 * i: off + 64 * i + 64] for i in range(n)]
    off += 64 * n
    g2, s_g2 = buf[off:off + 128], buf[off + 128:off + 256]
    assert off + 256 == len(buf)
    return dict(k=k, g=g, g_lagrange=gl, g2=g2, s_g2=s_g2)


def parse_pk(buf, n_perm, n_sel):
    """vk || l0 || l_last || l_active_row || 6 x vec(poly); returns raw 32-B-element byte blobs."""
    assert buf[0] == 3
    k = buf[1]
    n = 1 << k
    nfixed = int.from_bytes(buf[3:7], "little")
    off = 7
    fixed_commit = [buf[off + 64 * i: off + 64 * i + 64] for i in range(nfixed)]
    off += 64 * nfixed
    perm_commit = [buf[off + 64 * i: off + 64 * i + 64] for i in range(n_perm)]
    off += 64 * n_perm
    off += n_sel * (n // 8)
    vk_len = off

    def poly():
        nonlocal off
        ln = int.from_bytes(buf[off:off + 4], "big")
        off += 4
        d = buf[off:off + 32 * ln]
        off += 32 * ln
        return d

    def vec():
        nonlocal off
        cnt = int.from_bytes(buf[off:off + 4], "big")
        off += 4 + 4 * cnt
        return [poly() for _ in range(cnt)]

    out = dict(k=k, vk_len=vk_len, fixed_commit=fixed_commit, perm_commit=perm_commit)
    out["l0"], out["l_last"], out["l_active_row"] = poly(), poly(), poly()
    for name in ("fixed_values", "fixed_polys", "fixed_cosets", "permutations", "perm_polys", "perm_cosets"):
        out[name] = vec()
    assert off == len(buf), (off, len(buf))
    return out


def felts(blob):
    return [fr_from_mont_bytes(blob[i:i + 32]) for i in range(0, len(blob), 32)]
