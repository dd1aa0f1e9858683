"""BN254 optimal-ate pairing in plain Python big-ints (TEST INFRASTRUCTURE ONLY: the verifier side of the oracle).

The reference's acceptance criterion for a proof is "the verifier accepts" (/root/reference/src/pfsys/mod.rs:557-590
verify_proof_circuit -> halo2 verify_proof + pairing), so the oracle needs a pairing.  This follows the
published construction: Fq2 = Fq[u]/(u^2+1), Fq12 = Fq2[w]/(w^6 - xi) with xi = 9 + u, the D-type sextic twist
E'(Fq2): y^2 = x^3 + 3/xi, Miller loop over 6x+2 with x = 4965661367192848881, two Frobenius correction lines, and a
plain final exponentiation f^((q^12-1)/r).  Pinned on the reference's own SRS fixture (tests/golden/kzg_k6.srs):
e(g[1], g2) == e(g[0], s_g2)."""
Q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
BN_X = 4965661367192848881
ATE = 6 * BN_X + 2
MONT = 1 << 256
RINV_Q = pow(MONT, -1, Q)


# ---------------- Fq2 ----------------
def f2_add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def f2_sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def f2_neg(a): return ((-a[0]) % Q, (-a[1]) % Q)
def f2_mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
def f2_muls(a, s): return (a[0] * s % Q, a[1] * s % Q)
def f2_conj(a): return (a[0], (-a[1]) % Q)
def f2_inv(a):
    d = pow((a[0] * a[0] + a[1] * a[1]) % Q, -1, Q)
    return (a[0] * d % Q, (-a[1]) * d % Q)
def f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1: r = f2_mul(r, a)
        a = f2_mul(a, a); e >>= 1
    return r

F2_ZERO, F2_ONE = (0, 0), (1, 0)
XI = (9, 1)
B2 = f2_mul((3, 0), f2_inv(XI))               # twist curve constant 3/xi


# ---------------- Fq12 = Fq2[w]/(w^6 - xi): list of 6 Fq2 coefficients ----------------
F12_ONE = [F2_ONE] + [F2_ZERO] * 5

def f12_mul(a, b):
    t = [F2_ZERO] * 11
    for i in range(6):
        if a[i] == F2_ZERO: continue
        for j in range(6):
            if b[j] == F2_ZERO: continue
            t[i + j] = f2_add(t[i + j], f2_mul(a[i], b[j]))
    return [f2_add(t[i], f2_mul(t[i + 6], XI)) if i < 5 else t[i] for i in range(6)]

def f12_pow(a, e):
    r = F12_ONE
    while e:
        if e & 1: r = f12_mul(r, a)
        a = f12_mul(a, a); e >>= 1
    return r


# ---------------- G1 (affine, None = identity) and G2 on the twist ----------------
def g1_add(p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % Q == 0: return None
        lam = 3 * p[0] * p[0] * pow(2 * p[1], -1, Q) % Q
    else:
        lam = (q[1] - p[1]) * pow(q[0] - p[0], -1, Q) % Q
    x = (lam * lam - p[0] - q[0]) % Q
    return (x, (lam * (p[0] - x) - p[1]) % Q)

def g1_neg(p): return None if p is None else (p[0], (-p[1]) % Q)

def g1_mul(p, k):
    k %= R
    acc = None
    while k:
        if k & 1: acc = g1_add(acc, p)
        p = g1_add(p, p); k >>= 1
    return acc

def g2_on_curve(p): return f2_sub(f2_mul(p[1], p[1]), f2_add(f2_mul(f2_mul(p[0], p[0]), p[0]), B2)) == F2_ZERO

def g2_add(p, q):
    if p is None: return q
    if q is None: return p
    if p[0] == q[0]:
        if f2_add(p[1], q[1]) == F2_ZERO: return None
        lam = f2_mul(f2_muls(f2_mul(p[0], p[0]), 3), f2_inv(f2_muls(p[1], 2)))
    else:
        lam = f2_mul(f2_sub(q[1], p[1]), f2_inv(f2_sub(q[0], p[0])))
    x = f2_sub(f2_sub(f2_mul(lam, lam), p[0]), q[0])
    return (x, f2_sub(f2_mul(lam, f2_sub(p[0], x)), p[1]))

def g2_mul(p, k):
    acc = None
    while k:
        if k & 1: acc = g2_add(acc, p)
        p = g2_add(p, p); k >>= 1
    return acc

def g2_from_bytes(b):
    """halo2curves G2Affine raw bytes: x.c0 | x.c1 | y.c0 | y.c1, each 32 B little-endian Montgomery"""
    v = [int.from_bytes(b[32 * i:32 * i + 32], "little") * RINV_Q % Q for i in range(4)]
    return ((v[0], v[1]), (v[2], v[3]))


# ---------------- Miller loop ----------------
def _line(r, q, p):
    """line through r and q (tangent if equal) on the twist, evaluated at P in G1 after untwisting:
    yP + (-lam*xP) w + (lam*x_r - y_r) w^3; returns (Fq12 element, r + q)"""
    if r[0] == q[0] and r[1] == q[1]:
        lam = f2_mul(f2_muls(f2_mul(r[0], r[0]), 3), f2_inv(f2_muls(r[1], 2)))
    else:
        lam = f2_mul(f2_sub(q[1], r[1]), f2_inv(f2_sub(q[0], r[0])))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), r[0]), q[0])
    y3 = f2_sub(f2_mul(lam, f2_sub(r[0], x3)), r[1])
    l = [(p[1], 0), f2_neg(f2_muls(lam, p[0])), F2_ZERO, f2_sub(f2_mul(lam, r[0]), r[1]), F2_ZERO, F2_ZERO]
    return l, (x3, y3)

_GAMMA_X = f2_pow(XI, (Q - 1) // 3)
_GAMMA_Y = f2_pow(XI, (Q - 1) // 2)

def _frob(q):
    return (f2_mul(f2_conj(q[0]), _GAMMA_X), f2_mul(f2_conj(q[1]), _GAMMA_Y))

def miller(q, p):
    """Miller function f_{6x+2,Q}(P) with the two Frobenius lines; q in G2 (twist coordinates), p in G1"""
    if q is None or p is None:
        return F12_ONE
    f, r = F12_ONE, q
    for i in range(ATE.bit_length() - 2, -1, -1):
        l, r2 = _line(r, r, p)
        f = f12_mul(f12_mul(f, f), l)
        r = r2
        if (ATE >> i) & 1:
            l, r = _line(r, q, p)
            f = f12_mul(f, l)
    q1 = _frob(q)
    q2 = _frob(q1)
    q2 = (q2[0], f2_neg(q2[1]))
    l, r = _line(r, q1, p)
    f = f12_mul(f, l)
    l, r = _line(r, q2, p)
    return f12_mul(f, l)

_FINAL = (Q ** 12 - 1) // R

def final_exp(f): return f12_pow(f, _FINAL)

def pairing(q, p): return final_exp(miller(q, p))

def pairing_check(pairs):
    """prod e(P_i, Q_i) == 1 for pairs [(P in G1, Q in G2)]"""
    f = F12_ONE
    for p, q in pairs:
        f = f12_mul(f, miller(q, p))
    return final_exp(f) == F12_ONE
