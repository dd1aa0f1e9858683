// BN254 pairing on the host, for the SAFE self-check of create_proof_circuit (/root/reference/src/pfsys/mod.rs:470-480:
// CheckMode::SAFE verifies every proof it has just made) and for verify_proof_circuit (:557-590).  O(1) work per proof.
//
// Construction (published; written here from the definitions, no code taken from elsewhere):
//   Fq2 = Fq[u]/(u^2 + 1),  Fq12 = Fq2[w]/(w^6 - xi) with xi = 9 + u  (a degree-6 polynomial basis: schoolbook products),
//   G2 = the r-torsion of the D-type sextic twist E'(Fq2): y^2 = x^3 + 3/xi, untwisted by (x', y') -> (x' w^2, y' w^3);
//   the ATE pairing  a(Q, P) = f_{T,Q}(P)^((q^12 - 1)/r)  with T = t - 1 = 6 x^2, x = 4965661367192848881 (T = q mod r, so the
//   Miller function over T needs none of the Frobenius correction lines of the optimal-ate variant; 127 doubling steps);
//   affine line functions  l(P) = y_P + (-lambda x_P) w + (lambda x_R - y_R) w^3  (vertical lines vanish in the final exponentiation);
//   the final exponentiation as ONE square-and-multiply with the 2790-bit exponent (q^12 - 1)/r.
// Any non-degenerate bilinear pairing decides the KZG / SHPLONK check  e(A, [s]_2) * e(B, [1]_2) == 1  identically, so this one is
// interchangeable with the optimal-ate pairing of the test oracle (tests/test_native_prover.py compares the verdicts).
// G2 is also what the SRS carries (g2, s_g2: /root/reference/src/pfsys/srs.rs:14-16): scalar multiplication on the twist is here for gen_srs.
#pragma once
#include <vector>
#include "hostfield.hpp"

namespace ezkl_prover {
namespace bn {

struct Fq {
    U256 v{};
    static Fq zero() { return Fq{}; }
    static Fq one() { return Fq{FQ.one}; }
    static Fq from_u64(uint64_t x) { return Fq{mont_mul(U256{x, 0, 0, 0}, FQ.r2, FQ)}; }
    bool is_zero() const { return v == U256{0, 0, 0, 0}; }
    bool operator==(const Fq& o) const { return v == o.v; }
    Fq operator*(const Fq& o) const { return Fq{mont_mul(v, o.v, FQ)}; }
    Fq operator+(const Fq& o) const {
        Fq r;
        uint64_t c = add_raw(r.v, v, o.v);
        if (c || cmp(r.v, FQ.p) >= 0) sub_raw(r.v, r.v, FQ.p);
        return r;
    }
    Fq operator-(const Fq& o) const {
        Fq r;
        if (sub_raw(r.v, v, o.v)) add_raw(r.v, r.v, FQ.p);
        return r;
    }
    Fq operator-() const { return zero() - *this; }
    Fq inv() const {
        U256 e = FQ.p;
        e[0] -= 2;
        Fq acc = one(), b = *this;
        for (int i = 0; i < 256; i++) {
            if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * b;
            b = b * b;
        }
        return acc;
    }
};

struct Fq2 {
    Fq a, b;                                   // a + b u
    static Fq2 zero() { return Fq2{}; }
    static Fq2 one() { return Fq2{Fq::one(), Fq::zero()}; }
    bool is_zero() const { return a.is_zero() && b.is_zero(); }
    bool operator==(const Fq2& o) const { return a == o.a && b == o.b; }
    Fq2 operator+(const Fq2& o) const { return {a + o.a, b + o.b}; }
    Fq2 operator-(const Fq2& o) const { return {a - o.a, b - o.b}; }
    Fq2 operator-() const { return {-a, -b}; }
    Fq2 operator*(const Fq2& o) const {
        const Fq aa = a * o.a, bb = b * o.b;
        return {aa - bb, (a + b) * (o.a + o.b) - aa - bb};
    }
    Fq2 scale(const Fq& s) const { return {a * s, b * s}; }
    Fq2 mul_xi() const {                       // (a + b u)(9 + u) = (9a - b) + (a + 9b) u
        const Fq n = Fq::from_u64(9);
        return {a * n - b, a + b * n};
    }
    Fq2 inv() const {
        const Fq d = (a * a + b * b).inv();
        return {a * d, -(b * d)};
    }
};

struct Fq12 {
    Fq2 c[6];                                  // sum c[i] w^i, w^6 = xi
    static Fq12 one() {
        Fq12 r{};
        r.c[0] = Fq2::one();
        return r;
    }
    bool is_one() const {
        if (!(c[0] == Fq2::one())) return false;
        for (int i = 1; i < 6; i++)
            if (!c[i].is_zero()) return false;
        return true;
    }
    Fq12 operator*(const Fq12& o) const {
        Fq2 t[11] = {};
        for (int i = 0; i < 6; i++) {
            if (c[i].is_zero()) continue;
            for (int j = 0; j < 6; j++) {
                if (o.c[j].is_zero()) continue;
                t[i + j] = t[i + j] + c[i] * o.c[j];
            }
        }
        Fq12 r;
        for (int i = 0; i < 6; i++) r.c[i] = i < 5 ? t[i] + t[i + 6].mul_xi() : t[i];
        return r;
    }
};

// (q^12 - 1) / r, little-endian 64-bit limbs
static const uint64_t FINAL_EXP[] = {
    0x86964b64ca86f120ull, 0x40a4efb7e54523a4ull, 0x837fa97896e84abbull, 0x361102b6b9b2b918ull,
    0xc0de81def35692daull, 0xbe04c7e8a6c3c760ull, 0xd766f9c9d570bb7full, 0xc230974d83561841ull,
    0x5bba1668c3be69a3ull, 0x7f3811c410526294ull, 0x29baee7ddadda71cull, 0xbf813b8d145da900ull,
    0x641bbadf423f9a2cull, 0xa80bb4ea44eacc5eull, 0xcd65664814fde37cull, 0x4a0364b9580291d2ull,
    0xee93dfb10826f0ddull, 0x6b42db8dc5514724ull, 0xbb10cf430b0f3785ull, 0x40494e406f804216ull,
    0x55cfe107acf3aafbull, 0x2088ec80e0ebae87ull, 0x846a3ed011a337a0ull, 0x48a45a4a1e3a5195ull,
    0xe5664568dfc50e16ull, 0xab6a41294c0cc4ebull, 0x82d0d602d268c7daull, 0x6668449aed3cc48aull,
    0x5062cd0fb2015dfcull, 0x7f2940a8b1ddb3d1ull, 0x77f5b63a2a226448ull, 0xfef0781361e443aeull,
    0xf977870e88d5c6c8ull, 0x790364a61f676baaull, 0x5887e72eceaddea3ull, 0x1377e563a09a1b70ull,
    0x0c54efee1bd8c3b2ull, 0x3ec3d15ad524d8f7ull, 0xdaf15466b2383a5dull, 0xe1e30a73bb94fec0ull,
    0x6a1c71015f3f7be2ull, 0x842d43bf6369b1ffull, 0x20fddadf107d20bcull, 0x0000002f4b6dc970ull,
};
constexpr int FINAL_EXP_BITS = 2790;
// T = t - 1 = 6 x^2
constexpr uint64_t ATE_T[2] = {0xf83e9682e87cfd46ull, 0x6f4d8248eeb859fbull};
constexpr int ATE_T_BITS = 127;

struct G2 {
    Fq2 x, y;
    bool inf = true;
};
inline Fq2 twist_b() {                          // 3 / xi
    Fq2 xi{Fq::from_u64(9), Fq::one()};
    return xi.inv().scale(Fq::from_u64(3));
}
inline bool g2_on_curve(const G2& p) { return p.inf || p.y * p.y == p.x * p.x * p.x + twist_b(); }
inline G2 g2_add(const G2& p, const G2& q) {
    if (p.inf) return q;
    if (q.inf) return p;
    Fq2 lam;
    if (p.x == q.x) {
        if ((p.y + q.y).is_zero()) return G2{};
        lam = (p.x * p.x).scale(Fq::from_u64(3)) * (p.y + p.y).inv();
    } else {
        lam = (q.y - p.y) * (q.x - p.x).inv();
    }
    G2 r;
    r.inf = false;
    r.x = lam * lam - p.x - q.x;
    r.y = lam * (p.x - r.x) - p.y;
    return r;
}
inline G2 g2_mul(const G2& p, const U256& k) {   // k canonical
    G2 acc, b = p;
    for (int i = 0; i < 256; i++) {
        if ((k[i >> 6] >> (i & 63)) & 1) acc = g2_add(acc, b);
        b = g2_add(b, b);
    }
    return acc;
}
// halo2curves G2Affine raw bytes: x.c0 | x.c1 | y.c0 | y.c1, 32-byte little-endian Montgomery each; all-zero = identity
inline G2 g2_from_bytes(const uint8_t* b) {
    G2 p;
    U256 l[4];
    std::memcpy(l, b, 128);
    p.x = {Fq{l[0]}, Fq{l[1]}};
    p.y = {Fq{l[2]}, Fq{l[3]}};
    p.inf = p.x.is_zero() && p.y.is_zero();
    return p;
}
inline void g2_to_bytes(const G2& p, uint8_t* b) {
    U256 l[4] = {p.x.a.v, p.x.b.v, p.y.a.v, p.y.b.v};
    if (p.inf) std::memset(l, 0, sizeof l);
    std::memcpy(b, l, 128);
}
// the generator of G2 halo2curves uses (the standard alt_bn128 one)
inline G2 g2_generator() {
    auto fq = [](uint64_t a, uint64_t b, uint64_t c, uint64_t d) { return Fq{mont_mul(U256{d, c, b, a}, FQ.r2, FQ)}; };
    G2 g;
    g.inf = false;
    g.x = {fq(0x1800deef121f1e76ull, 0x426a00665e5c4479ull, 0x674322d4f75edaddull, 0x46debd5cd992f6edull),
           fq(0x198e9393920d483aull, 0x7260bfb731fb5d25ull, 0xf1aa493335a9e712ull, 0x97e485b7aef312c2ull)};
    g.y = {fq(0x12c85ea5db8c6debull, 0x4aab71808dcb408full, 0xe3d1e7690c43d37bull, 0x4ce6cc0166fa7daaull),
           fq(0x090689d0585ff075ull, 0xec9e99ad690c3395ull, 0xbc4b313370b38ef3ull, 0x55acdadcd122975bull)};
    return g;
}

// ---- G1 on the host (affine, Montgomery Fq): the verifier's linear combinations of commitments
struct P1 {
    Fq x, y;
    bool inf = true;
};
inline P1 p1_from(const G1& g) {
    P1 p;
    p.x = Fq{g.x}; p.y = Fq{g.y};
    p.inf = p.x.is_zero() && p.y.is_zero();
    return p;
}
struct J1 {                                     // Jacobian accumulator (Z = 0: identity)
    Fq X, Y, Z;
};
inline J1 j1_double(const J1& p) {
    if (p.Z.is_zero()) return p;
    const Fq A = p.X * p.X, B = p.Y * p.Y, C = B * B;
    const Fq t = p.X + B;
    Fq D = t * t - A - C;
    D = D + D;
    const Fq E = A + A + A, F = E * E;
    J1 r;
    r.X = F - D - D;
    Fq C8 = C + C; C8 = C8 + C8; C8 = C8 + C8;
    r.Y = E * (D - r.X) - C8;
    r.Z = (p.Y * p.Z); r.Z = r.Z + r.Z;
    return r;
}
inline J1 j1_add_affine(const J1& p, const P1& q) {
    if (q.inf) return p;
    if (p.Z.is_zero()) return J1{q.x, q.y, Fq::one()};
    const Fq Z2 = p.Z * p.Z, U2 = q.x * Z2, S2 = q.y * Z2 * p.Z;
    const Fq H = U2 - p.X, Rr = S2 - p.Y;
    if (H.is_zero()) {
        if (Rr.is_zero()) return j1_double(p);
        return J1{};
    }
    const Fq H2 = H * H, H3 = H2 * H, V = p.X * H2;
    J1 r;
    r.X = Rr * Rr - H3 - V - V;
    r.Y = Rr * (V - r.X) - p.Y * H3;
    r.Z = p.Z * H;
    return r;
}
inline P1 j1_affine(const J1& p) {
    if (p.Z.is_zero()) return P1{};
    const Fq zi = p.Z.inv(), z2 = zi * zi;
    P1 r;
    r.inf = false;
    r.x = p.X * z2;
    r.y = p.Y * z2 * zi;
    return r;
}
inline P1 p1_neg(const P1& p) {
    P1 r = p;
    if (!r.inf) r.y = -r.y;
    return r;
}
inline P1 p1_mul(const P1& p, const Fe& k) {     // double-and-add from the top bit
    const U256 e = k.canonical();
    J1 acc{};
    for (int i = 255; i >= 0; i--) {
        acc = j1_double(acc);
        if ((e[i >> 6] >> (i & 63)) & 1) acc = j1_add_affine(acc, p);
    }
    return j1_affine(acc);
}
inline P1 p1_add(const P1& a, const P1& b) {
    if (a.inf) return b;
    return j1_affine(j1_add_affine(J1{a.x, a.y, Fq::one()}, b));
}

// ---- Miller loop (ate, affine) and the pairing-product check
inline Fq12 line(const G2& r, const Fq2& lam, const P1& p) {
    Fq12 l{};
    l.c[0] = {p.y, Fq::zero()};
    l.c[1] = -(lam.scale(p.x));
    l.c[3] = lam * r.x - r.y;
    return l;
}
inline Fq12 miller(const P1& p, const G2& q) {
    Fq12 f = Fq12::one();
    if (p.inf || q.inf) return f;
    G2 r = q;
    for (int i = ATE_T_BITS - 2; i >= 0; i--) {
        const Fq2 lam = (r.x * r.x).scale(Fq::from_u64(3)) * (r.y + r.y).inv();
        f = f * f * line(r, lam, p);
        r = g2_add(r, r);
        if ((ATE_T[i >> 6] >> (i & 63)) & 1) {
            if (r.x == q.x) {                   // r = +-q cannot happen for points of prime order r inside the loop
                r = g2_add(r, q);
                continue;
            }
            const Fq2 l2 = (q.y - r.y) * (q.x - r.x).inv();
            f = f * line(r, l2, p);
            r = g2_add(r, q);
        }
    }
    return f;
}
inline Fq12 final_exponentiation(const Fq12& f) {
    Fq12 acc = Fq12::one();
    for (int i = FINAL_EXP_BITS - 1; i >= 0; i--) {
        acc = acc * acc;
        if ((FINAL_EXP[i >> 6] >> (i & 63)) & 1) acc = acc * f;
    }
    return acc;
}
// prod e(p_i, q_i) == 1
inline bool pairing_check(const std::vector<std::pair<P1, G2>>& pairs) {
    Fq12 f = Fq12::one();
    for (auto& pq : pairs) f = f * miller(pq.first, pq.second);
    return final_exponentiation(f).is_one();
}

}  // namespace bn
}  // namespace ezkl_prover
