// host64.hpp -- small, fast HOST-side Fq / G1 arithmetic (4 x 64-bit limbs, unsigned __int128) used for the
// last few dozen group operations of an MSM (plane Horner + to-affine) and for folding per-GPU partial
// points.  On the GPU a dependent chain of point additions costs ~5 us per link (one wave, one lane);
// on a host core the same link is ~0.5 us, so the strictly serial tail runs here.  Same Montgomery
// representation (R = 2^256) as the device code and as halo2curves, so bytes can be memcpy'd across.
#pragma once
#include <stdint.h>
#include <string.h>
#include "bn254_constants.h"

namespace ezkl {
namespace h64 {

typedef unsigned __int128 u128;
struct fe {
    uint64_t v[4];
};
struct xyzz {
    fe x, y, zz, zzz;
};
struct aff {
    fe x, y;
};

static inline fe from32(const uint32_t (&c)[8]) {
    fe r;
    for (int i = 0; i < 4; i++) r.v[i] = (uint64_t)c[2 * i] | ((uint64_t)c[2 * i + 1] << 32);
    return r;
}
static const uint32_t Q32[8] = BN32_FQ_MOD_INIT, QONE32[8] = BN32_FQ_R_INIT;
static const fe Q = from32(Q32), ONE = from32(QONE32);

static inline uint64_t inv64() {    // -q^-1 mod 2^64 by Newton iteration from the 32-bit value
    uint64_t q0 = Q.v[0], x = (uint64_t)0 - (uint64_t)BN32_FQ_INV;   // x = q^-1 mod 2^32
    x *= 2 - q0 * x;                 // now mod 2^64
    return (uint64_t)0 - x;
}
static const uint64_t NINV = inv64();

static inline bool is_zero(const fe& a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
static inline bool eq(const fe& a, const fe& b) { return memcmp(&a, &b, 32) == 0; }
static inline bool geq(const fe& a, const fe& b) {
    for (int i = 3; i >= 0; i--) {
        if (a.v[i] > b.v[i]) return true;
        if (a.v[i] < b.v[i]) return false;
    }
    return true;
}
static inline uint64_t sub4(fe& o, const fe& a, const fe& b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.v[i] - b.v[i] - br;
        o.v[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
static inline void add4(fe& o, const fe& a, const fe& b) {
    uint64_t c = 0;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a.v[i] + b.v[i] + c;
        o.v[i] = (uint64_t)s;
        c = (uint64_t)(s >> 64);
    }
}
static inline fe add(const fe& a, const fe& b) {
    fe t;
    add4(t, a, b);
    if (geq(t, Q)) sub4(t, t, Q);
    return t;
}
static inline fe sub(const fe& a, const fe& b) {
    fe t;
    if (sub4(t, a, b)) add4(t, t, Q);
    return t;
}
static inline fe dbl(const fe& a) { return add(a, a); }
static inline fe neg(const fe& a) {
    if (is_zero(a)) return a;
    fe t;
    sub4(t, Q, a);
    return t;
}
static inline fe mul(const fe& a, const fe& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a.v[j] * b.v[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * NINV;
        c = (u128)m * Q.v[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * Q.v[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq(r, Q)) sub4(r, r, Q);
    return r;
}
static inline fe sqr(const fe& a) { return mul(a, a); }
static inline fe inv_pow(const fe& a) {     // a^(q-2): 256 squarings + ~127 products, ~13 us (kept as the reference of tests/test_host64.py)
    fe e = Q, two = {{2, 0, 0, 0}};
    sub4(e, e, two);
    fe acc = ONE, base = a;
    for (int i = 0; i < 256; i++) {
        if ((e.v[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base);
        base = sqr(base);
    }
    return acc;
}
static inline fe r_cubed() {                 // R^3 mod q: R^2 by 256 modular doublings of R, then one Montgomery product
    fe r2 = ONE;
    for (int i = 0; i < 256; i++) r2 = dbl(r2);
    return mul(r2, r2);
}
static const fe R3 = r_cubed();
// a^-1 (Montgomery form in and out) by the binary extended Euclid on the integers: for A = a R it finds A^-1 = a^-1 R^-1, and one
// Montgomery product with R^3 turns that into a^-1 R.  ~4 us instead of the 13 us of the exponentiation: every synchronous MSM ends
// with one inversion on the host (to_affine), 1 % of a 2^20-point call.
static inline fe inv(const fe& a) {
    if (is_zero(a)) return a;
    fe u = a, v = Q, x1 = {{1, 0, 0, 0}}, x2 = {{0, 0, 0, 0}};
    auto is_one = [](const fe& t) { return t.v[0] == 1 && (t.v[1] | t.v[2] | t.v[3]) == 0; };
    auto shr1 = [](fe& t) {
        t.v[0] = (t.v[0] >> 1) | (t.v[1] << 63); t.v[1] = (t.v[1] >> 1) | (t.v[2] << 63);
        t.v[2] = (t.v[2] >> 1) | (t.v[3] << 63); t.v[3] >>= 1;
    };
    auto half = [&](fe& x) {                 // x / 2 mod q for x < q (x + q < 2^255: no carry out)
        if (x.v[0] & 1) add4(x, x, Q);
        shr1(x);
    };
    while (!is_one(u) && !is_one(v)) {
        while (!(u.v[0] & 1)) { shr1(u); half(x1); }
        while (!(v.v[0] & 1)) { shr1(v); half(x2); }
        if (geq(u, v)) { sub4(u, u, v); x1 = sub(x1, x2); }
        else { sub4(v, v, u); x2 = sub(x2, x1); }
    }
    return mul(is_one(u) ? x1 : x2, R3);
}

static inline bool is_id(const xyzz& p) { return is_zero(p.zz); }
static inline xyzz identity() {
    xyzz r;
    memset(&r, 0, sizeof r);
    return r;
}
static inline xyzz from_affine(const aff& p) {
    if (is_zero(p.x) && is_zero(p.y)) return identity();
    xyzz r;
    r.x = p.x; r.y = p.y; r.zz = ONE; r.zzz = ONE;
    return r;
}
static inline xyzz dbl(const xyzz& p) {
    if (is_id(p)) return p;
    xyzz r;
    fe u = dbl(p.y), v = sqr(u), w = mul(u, v), s = mul(p.x, v), xx = sqr(p.x);
    fe m = add(dbl(xx), xx);
    r.x = sub(sqr(m), dbl(s));
    r.y = sub(mul(m, sub(s, r.x)), mul(w, p.y));
    r.zz = mul(v, p.zz);
    r.zzz = mul(w, p.zzz);
    return r;
}
// a coordinate as the MSM kernels leave it (field29.hpp: 9 limbs of 29 bits, lazily reduced, Montgomery R' = 2^261)
// -> canonical Montgomery R = 2^256: rebuild the integer, subtract q while >= q (values are < 32 q), multiply by 2^-5
static inline fe from_limbs29(const uint32_t v[9]) {
    uint64_t w[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, k = bit >> 6, sh = bit & 63;
        u128 t = (u128)v[i] << sh;
        u128 c = (u128)w[k] + (uint64_t)t;
        w[k] = (uint64_t)c;
        c = (c >> 64) + (uint64_t)(t >> 64);
        for (int j = k + 1; j < 5 && c; j++) {
            c += w[j];
            w[j] = (uint64_t)c;
            c >>= 64;
        }
    }
    for (;;) {
        fe lo = {{w[0], w[1], w[2], w[3]}};
        if (w[4] == 0 && !geq(lo, Q)) break;
        fe d;
        uint64_t br = sub4(d, lo, Q);
        w[0] = d.v[0]; w[1] = d.v[1]; w[2] = d.v[2]; w[3] = d.v[3];
        w[4] -= br;
    }
    const fe x = {{w[0], w[1], w[2], w[3]}}, c251 = {{0, 0, 0, (uint64_t)1 << 59}};
    return mul(x, c251);                 // x * 2^251 / 2^256 = x * 2^-5
}
static inline xyzz from_limbs29_point(const uint32_t* p36) {
    xyzz r;
    r.x = from_limbs29(p36);
    r.y = from_limbs29(p36 + 9);
    r.zz = from_limbs29(p36 + 18);
    r.zzz = from_limbs29(p36 + 27);
    return r;
}

static inline xyzz add(const xyzz& a, const xyzz& b) {
    if (is_id(a)) return b;
    if (is_id(b)) return a;
    fe u1 = mul(a.x, b.zz), u2 = mul(b.x, a.zz), s1 = mul(a.y, b.zzz), s2 = mul(b.y, a.zzz);
    fe p = sub(u2, u1), r = sub(s2, s1);
    if (is_zero(p)) return is_zero(r) ? dbl(a) : identity();
    fe pp = sqr(p), ppp = mul(p, pp), q = mul(u1, pp);
    xyzz o;
    o.x = sub(sub(sqr(r), ppp), dbl(q));
    o.y = sub(mul(r, sub(q, o.x)), mul(s1, ppp));
    o.zz = mul(mul(a.zz, b.zz), pp);
    o.zzz = mul(mul(a.zzz, b.zzz), ppp);
    return o;
}
static inline aff to_affine(const xyzz& p) {
    aff r;
    if (is_id(p)) {
        memset(&r, 0, sizeof r);
        return r;
    }
    fe t = inv(mul(p.zz, p.zzz));
    r.x = mul(p.x, mul(t, p.zzz));
    r.y = mul(p.y, mul(t, p.zz));
    return r;
}

}  // namespace h64
}  // namespace ezkl
