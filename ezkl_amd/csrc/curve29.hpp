// curve29.hpp -- BN254 G1 group law on lazily reduced radix-2^29 coordinates (field29.hpp) for the MSM kernels.
// Same formulas as curve.hpp (XYZZ: madd-2008-s, add-2008-s, dbl-2008-s-1), restated with the bound bookkeeping the
// lazy representation needs.  Invariant of every stored point: X < 12p, Y < 8p, ZZ, ZZZ < 4p, all limbs normalized;
// identity = ZZ with all limbs zero.  Per formula the comment gives (value bound in units of p | limb looseness).
#pragma once
#include "curve.hpp"
#include "field29.hpp"

namespace ezkl {

struct alignas(16) g1x29_t {
    f29_t x, y, zz, zzz;          // 36 dwords = 144 bytes
};
struct g1a29_t {
    f29_t x, y;                   // affine, canonical (< p), normalized
};

EZ_D bool g1x29_is_id(const g1x29_t& p) { return Fq29::limbs_zero(p.zz); }
EZ_D g1x29_t g1x29_identity() {
    g1x29_t r;
    r.x = Fq29::zero(); r.y = Fq29::zero(); r.zz = Fq29::zero(); r.zzz = Fq29::zero();
    return r;
}
// table record (64 bytes, canonical x * 2^261 mod p and y * 2^261 mod p; (0,0) = identity) -> limbs; `negate` flips y
EZ_D g1a29_t g1a29_unpack(const g1a_t& p) {
    g1a29_t r;
    r.x = Fq29::unpack(p.x);
    r.y = Fq29::unpack(p.y);
    return r;
}
EZ_D bool g1a29_is_id(const g1a29_t& p) { return Fq29::limbs_zero(p.x) && Fq29::limbs_zero(p.y); }

// shared tail of the addition formulas: given U1 (= X1 scaled), S1, P = U2 - U1, R = S2 - S1 (both normalized), ZZ and ZZZ
// factors already multiplied together
//   PP = P^2, PPP = P PP, Q = U1 PP, X3 = R^2 - PPP - 2Q, Y3 = R (Q - X3) - S1 PPP, ZZ3 = zz PP, ZZZ3 = zzz PPP
// bounds: P, R < 20p; U1 < 12p; S1 < 8p; zz, zzz < 4p.
EZ_D g1x29_t g1x29_add_tail(const f29_t& u1, const f29_t& s1, const f29_t& p, const f29_t& r, const f29_t& zz, const f29_t& zzz) {
    // ordered so that every input dies as early as possible (the accumulate kernel lives at 128 VGPRs)
    g1x29_t o;
    const f29_t pp = Fq29::sqr(p);                       // < 400/169 + 1 < 4p
    o.zz = Fq29::mul(zz, pp);                            // < 16/169 + 1 < 2p
    const f29_t ppp = Fq29::mul(p, pp);                  // < 80/169 + 1  < 2p
    o.zzz = Fq29::mul(zzz, ppp);                         // < 2p
    const f29_t q = Fq29::mul(u1, pp);                   // < 60/169 + 1  < 2p
    const f29_t e = Fq29::mul(s1, ppp);                  // < 30/169 + 1  < 2p
    const f29_t rr = Fq29::sqr(r);                       // < 324/169 + 1 < 3p
    // X3 = rr + (4p - ppp) + 2 (4p - q) < 3 + 4 + 8 = 15p; limbs < 2^29 + 2^30 + 2^31, normalized right away
    const f29_t nq = Fq29::neg<1>(q);
    o.x = Fq29::normalize(Fq29::add(Fq29::add(rr, Fq29::neg<1>(ppp)), Fq29::add(nq, nq)));
    const f29_t d = Fq29::sub<3>(q, o.x);                // q + 16p - X3 < 18p, loose(3); needs X3 < 15p normalized
    const f29_t rd = Fq29::mul(r, d);                    // normalized x loose(3): < 18*18/169 + 1 < 3p
    o.y = Fq29::normalize(Fq29::sub<1>(rd, e));          // < 3 + 4 = 7p
    return o;
}
// 2 (x, y) for an affine point != identity (mdbl-2008-s-1)
EZ_D g1x29_t g1x29_double_affine(const g1a29_t& p) {
    const f29_t u = Fq29::add(p.y, p.y);                 // < 2p, loose(2)
    const f29_t v = Fq29::sqr_cold(u);                        // 2 x 2 = 4 < 6.1 ok; < 2p
    const f29_t w = Fq29::mul_cold(u, v);                     // < 2p
    const f29_t s = Fq29::mul_cold(p.x, v);                   // < 2p
    const f29_t xx = Fq29::sqr_cold(p.x);                     // < 2p
    const f29_t m = Fq29::normalize(Fq29::add(Fq29::add(xx, xx), xx));    // < 6p
    const f29_t ns = Fq29::neg<1>(s);
    g1x29_t r;
    r.x = Fq29::normalize(Fq29::add(Fq29::sqr_cold(m), Fq29::add(ns, ns)));    // m^2 < 36/169+1 < 2p ; < 2 + 8 = 10p
    const f29_t d = Fq29::sub<3>(s, r.x);                // < 18p loose(3)
    r.y = Fq29::normalize(Fq29::sub<1>(Fq29::mul_cold(m, d), Fq29::mul_cold(w, p.y)));   // (6*18/169+1 < 2p) + 4p < 6p
    r.zz = v;
    r.zzz = w;
    return r;
}
// 2 P (dbl-2008-s-1)
EZ_D g1x29_t g1x29_double(const g1x29_t& p) {
    if (g1x29_is_id(p)) return p;
    const f29_t u = Fq29::add(p.y, p.y);                 // < 16p, loose(2)
    const f29_t v = Fq29::sqr_cold(u);                        // < 256/169 + 1 < 4p
    const f29_t w = Fq29::mul_cold(u, v);                     // < 64/169 + 1 < 2p
    const f29_t s = Fq29::mul_cold(p.x, v);                   // < 48/169 + 1 < 2p
    const f29_t xx = Fq29::sqr_cold(p.x);                     // < 144/169 + 1 < 2p
    const f29_t m = Fq29::normalize(Fq29::add(Fq29::add(xx, xx), xx));    // < 6p
    const f29_t ns = Fq29::neg<1>(s);
    g1x29_t r;
    r.x = Fq29::normalize(Fq29::add(Fq29::sqr_cold(m), Fq29::add(ns, ns)));    // < 10p
    const f29_t d = Fq29::sub<3>(s, r.x);                // < 18p loose(3)
    r.y = Fq29::normalize(Fq29::sub<1>(Fq29::mul_cold(m, d), Fq29::mul_cold(w, p.y)));   // w*y < 16/169+1 < 2p ; < 6p
    r.zz = Fq29::mul_cold(v, p.zz);                           // < 2p
    r.zzz = Fq29::mul_cold(w, p.zzz);                         // < 2p
    return r;
}
// acc + q, q affine (madd-2008-s); `negate` adds -q
EZ_D g1x29_t g1x29_add_mixed(const g1x29_t& a, const g1a29_t& q, bool negate) {
    if (g1a29_is_id(q)) return a;
    const f29_t qy = negate ? Fq29::neg<0>(q.y) : q.y;   // 2p - y, loose(2) (y canonical < p)
    if (g1x29_is_id(a)) {
        g1x29_t r;
        r.x = q.x;
        r.y = negate ? Fq29::normalize(qy) : q.y;
        r.zz = Fq29::one();
        r.zzz = Fq29::one();
        return r;
    }
    const f29_t u2 = Fq29::mul(q.x, a.zz);               // < 4/169 + 1 < 2p
    const f29_t s2 = Fq29::mul(qy, a.zzz);               // loose(2) x normalized; < 8/169 + 1 < 2p
    const f29_t p = Fq29::normalize(Fq29::sub<3>(u2, a.x));      // u2 + 16p - X1 < 18p   (X1 < 15p)
    const f29_t r = Fq29::normalize(Fq29::sub<3>(s2, a.y));      // < 18p                 (Y1 < 15p)
    if (Fq29::is_zero_mod_p(p)) {
        if (Fq29::is_zero_mod_p(r)) {
            g1a29_t qq = q;
            if (negate) qq.y = Fq29::normalize(qy);
            return g1x29_double_affine(qq);
        }
        return g1x29_identity();
    }
    return g1x29_add_tail(a.x, a.y, p, r, a.zz, a.zzz);
}
// a + b (add-2008-s)
EZ_D g1x29_t g1x29_add(const g1x29_t& a, const g1x29_t& b) {
    if (g1x29_is_id(a)) return b;
    if (g1x29_is_id(b)) return a;
    const f29_t u1 = Fq29::mul(a.x, b.zz), u2 = Fq29::mul(b.x, a.zz);          // < 48/169 + 1 < 2p
    const f29_t s1 = Fq29::mul(a.y, b.zzz), s2 = Fq29::mul(b.y, a.zzz);        // < 2p
    const f29_t p = Fq29::normalize(Fq29::sub<1>(u2, u1));                     // < 6p
    const f29_t r = Fq29::normalize(Fq29::sub<1>(s2, s1));                     // < 6p
    if (Fq29::is_zero_mod_p(p)) {
        if (Fq29::is_zero_mod_p(r)) return g1x29_double(a);
        return g1x29_identity();
    }
    return g1x29_add_tail(u1, s1, p, r, Fq29::mul(a.zz, b.zz), Fq29::mul(a.zzz, b.zzz));
}

EZ_D g1x29_t ld_g1x29(const g1x29_t* p) {
    g1x29_t r;
    const uint4* s = reinterpret_cast<const uint4*>(p);
    uint32_t* d = r.x.v;                                 // x, y, zz, zzz are 36 contiguous dwords
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const uint4 t = s[k];
        d[4 * k] = t.x; d[4 * k + 1] = t.y; d[4 * k + 2] = t.z; d[4 * k + 3] = t.w;
    }
    return r;
}
EZ_D void st_g1x29(g1x29_t* p, const g1x29_t& v) {
    uint4* d = reinterpret_cast<uint4*>(p);
    const uint32_t* s = v.x.v;
#pragma unroll
    for (int k = 0; k < 9; k++) d[k] = make_uint4(s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]);
}
EZ_D g1x29_t g1x29_shfl_xor(const g1x29_t& p, uint32_t mask) {
    g1x29_t r;
    const uint32_t* s = p.x.v;
    uint32_t* d = r.x.v;
#pragma unroll
    for (int k = 0; k < 36; k++) d[k] = __shfl_xor(s[k], mask);
    return r;
}
// butterfly sums: see g1x_group_sum in curve.hpp for why every lane adds at every level
EZ_D g1x29_t g1x29_group_sum(g1x29_t acc, uint32_t width) {
#pragma unroll 1
    for (uint32_t s = width >> 1; s > 0; s >>= 1) acc = g1x29_add(acc, g1x29_shfl_xor(acc, s));
    return acc;
}
// sum over the 256 threads of a workgroup, valid in every thread; sh: 9 * 4 uint4 (plane layout, one slot per wave)
EZ_D g1x29_t g1x29_block256_sum(g1x29_t acc, uint4* sh) {
    acc = g1x29_group_sum(acc, 64);
    if ((threadIdx.x & 63) == 0) {
        const uint32_t* s = acc.x.v;
#pragma unroll
        for (int k = 0; k < 9; k++) sh[k * 4 + (threadIdx.x >> 6)] = make_uint4(s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]);
    }
    __syncthreads();
    {
        uint32_t* d = acc.x.v;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const uint4 t = sh[k * 4 + (threadIdx.x & 3)];
            d[4 * k] = t.x; d[4 * k + 1] = t.y; d[4 * k + 2] = t.z; d[4 * k + 3] = t.w;
        }
    }
    __syncthreads();
    return g1x29_group_sum(acc, 4);
}

}  // namespace ezkl
