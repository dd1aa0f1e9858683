"""CPU model of the quad-cooperative XYZZ addition (ezkl_amd/csrc/curve29.hpp: g1x29_add_quad): the four lanes of a quad each run one
product per dependency level and pass results around by quad broadcasts.  The model replays the lane assignment and the broadcasts of
the device code with plain integers mod p and compares the result with add-2008-s and with the affine sum -- it pins the DATAFLOW (which
lane multiplies what, which lane's result is read where); the limb bounds are those of g1x29_add (same formulas, same operand ranges)."""
import random

P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47


def aff_add(a, b):
    if a is None: return b
    if b is None: return a
    (x1, y1), (x2, y2) = a, b
    if x1 == x2:
        if (y1 + y2) % P == 0: return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return x3, (lam * (x1 - x3) - y1) % P


def rand_point(rng):
    while True:
        x = rng.randrange(P)
        y2 = (x * x * x + 3) % P
        y = pow(y2, (P + 1) // 4, P)
        if y * y % P == y2:
            return x, y


def to_xyzz(pt, rng):                      # a random XYZZ representation of an affine point
    z = rng.randrange(1, P)
    zz, zzz = z * z % P, z * z * z % P
    return dict(x=pt[0] * zz % P, y=pt[1] * zzz % P, zz=zz, zzz=zzz)


def to_affine(q):
    return q["x"] * pow(q["zz"], -1, P) % P, q["y"] * pow(q["zzz"], -1, P) % P


def sel4(q, x0, x1, x2, x3):
    return (x0, x1, x2, x3)[q]


def add_quad(a, b):
    """the four lanes hold the same a and b; lane q computes column q of every level (curve29.hpp)"""
    lanes = range(4)
    m1 = [sel4(q, a["x"], b["x"], a["y"], b["y"]) * sel4(q, b["zz"], a["zz"], b["zzz"], a["zzz"]) % P for q in lanes]
    u1, s1 = m1[0], m1[2]                                   # quad_bcast<0>, <2>
    p, r = (m1[1] - u1) % P, (m1[3] - s1) % P
    assert p != 0
    m2 = [sel4(q, a["zz"], a["zzz"], p, r) * sel4(q, b["zz"], b["zzz"], p, r) % P for q in lanes]
    pp, rr = m2[2], m2[3]
    m3 = [(p if q == 1 else (u1 if q == 2 else m2[q])) * pp % P for q in lanes]
    ppp, qq = m3[1], m3[2]
    x3 = (rr - ppp - 2 * qq) % P
    d = (qq - x3) % P
    m4 = [(r if q == 2 else (s1 if q == 3 else m2[q])) * (d if q == 2 else ppp) % P for q in lanes]
    return dict(x=x3, y=(m4[2] - m4[3]) % P, zz=m3[0], zzz=m4[1])


def add_2008_s(a, b):
    u1, u2 = a["x"] * b["zz"] % P, b["x"] * a["zz"] % P
    s1, s2 = a["y"] * b["zzz"] % P, b["y"] * a["zzz"] % P
    p, r = (u2 - u1) % P, (s2 - s1) % P
    pp = p * p % P
    ppp, q = p * pp % P, u1 * pp % P
    x3 = (r * r - ppp - 2 * q) % P
    return dict(x=x3, y=(r * (q - x3) - s1 * ppp) % P, zz=a["zz"] * b["zz"] * pp % P, zzz=a["zzz"] * b["zzz"] * ppp % P)


def test_quad_addition_dataflow_matches_add_2008_s():
    rng = random.Random(5)
    for _ in range(50):
        A, B = rand_point(rng), rand_point(rng)
        a, b = to_xyzz(A, rng), to_xyzz(B, rng)
        got = add_quad(a, b)
        assert got == add_2008_s(a, b)
        assert got["zz"] ** 3 % P == got["zzz"] ** 2 % P
        assert to_affine(got) == aff_add(A, B)


def test_quad_sum4_and_butterfly_levels():
    """g1x29_quad_sum4 + the levels of g1x29_group_sum_coop on 16 lanes: every lane ends with the total"""
    rng = random.Random(7)
    pts = [rand_point(rng) for _ in range(16)]
    acc = [to_xyzz(p, rng) for p in pts]
    # quad_sum4: (x0 + x1) + (x2 + x3), the same three additions in the four lanes
    for base in range(0, 16, 4):
        x = acc[base:base + 4]
        s = add_quad(add_quad(x[0], x[1]), add_quad(x[2], x[3]))
        for l in range(4): acc[base + l] = dict(s)
    s = 4
    while s < 16:
        acc = [add_quad(acc[l], acc[l ^ s]) for l in range(16)]
        for base in range(0, 16, 4):                       # the copies of a quad stay identical
            assert all(acc[base + l] == acc[base] for l in range(4))
        s <<= 1
    want = None
    for p in pts: want = aff_add(want, p)
    assert all(to_affine(v) == want for v in acc)
