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
// (one OR over the 18 limbs: the short-circuit form `zero(x) && zero(y)` became two divergent mini-blocks with eight register copies each in
// the accumulate loop)
EZ_D bool g1a29_is_id(const g1a29_t& p) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= p.x.v[i] | p.y.v[i];
    return o == 0;
}

// shared tail of the addition formulas: given U1 (= X1 scaled), S1, P = U2 - U1, R = S2 - S1 (both normalized), ZZ and ZZZ
// factors already multiplied together
//   PP = P^2, PPP = P PP, Q = U1 PP, X3 = R^2 - PPP - 2Q, Y3 = R (Q - X3) - S1 PPP, ZZ3 = zz PP, ZZZ3 = zzz PPP
// bounds: P, R < 20p; U1 < 12p; S1 < 8p; zz, zzz < 4p.
// S1 < 15p normalized (the accumulator's Y in the mixed addition; a product < 2p in the full one).
EZ_D g1x29_t g1x29_add_tail(const f29_t& u1, const f29_t& s1, const f29_t& p, const f29_t& r, const f29_t& zz, const f29_t& zzz) {
    // ordered so that every input dies as early as possible (the accumulate kernel lives at 128 VGPRs)
    g1x29_t o;
    const f29_t pp = Fq29::sqr(p);                       // < 400/169 + 1 < 4p
    o.zz = Fq29::mul(zz, pp);                            // < 16/169 + 1 < 2p
    const f29_t ppp = Fq29::mul(p, pp);                  // < 80/169 + 1  < 2p
    o.zzz = Fq29::mul(zzz, ppp);                         // < 2p
    const f29_t q = Fq29::mul(u1, pp);                   // < 60/169 + 1  < 2p
#if !EZKL_FUSED_Y
    const f29_t e = Fq29::mul(s1, ppp);                  // < 30/169 + 1  < 2p
#endif
    const f29_t rr = Fq29::sqr(r);                       // < 324/169 + 1 < 3p
    // X3 = rr + (4p - ppp) + 2 (4p - q) < 3 + 4 + 8 = 15p; limbs < 2^29 + 2^30 + 2^31, normalized right away
    const f29_t nq = Fq29::neg<1>(q);
    o.x = Fq29::normalize(Fq29::add(Fq29::add(rr, Fq29::neg<1>(ppp)), Fq29::add(nq, nq)));
    const f29_t d = Fq29::sub<3>(q, o.x);                // q + 16p - X3 < 18p, loose(3); needs X3 < 15p normalized
#if EZKL_FUSED_Y
    // Y3 = r d + (16p - s1) ppp in one reduction: limbs 1 x 3 + 2 x 1 = 5 < 6.1; value < (20*18 + 16*2) / 169 + 1 < 4p, normalized
    // (tests/test_montmul29_asm.py interprets the generated instructions on these operand shapes)
    o.y = Fq29::mul2add(r, d, Fq29::neg<3>(s1), ppp);
#else
    const f29_t rd = Fq29::mul(r, d);                    // normalized x loose(3): < 18*18/169 + 1 < 3p
    o.y = Fq29::normalize(Fq29::sub<1>(rd, e));          // < 3 + 4 = 7p
#endif
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

// ---- the quad-cooperative addition (round 5) ------------------------------------------------------------------------------------------
// The reduce trees of the MSM are chains of DEPENDENT additions, each 14 products one after the other in one wave (9.3 us per level at one
// wave per SIMD), and in a butterfly the lanes of the upper levels all hold the same operands anyway.  The 14 products of add-2008-s are
// four dependency levels of 4, 4, 3, 3:
//     1: U1 = X1 ZZ2    U2 = X2 ZZ1    S1 = Y1 ZZZ2    S2 = Y2 ZZZ1          -> P = U2 - U1, R = S2 - S1
//     2: zz = ZZ1 ZZ2   zzz = ZZZ1 ZZZ2  PP = P P      RR = R R
//     3: ZZ3 = zz PP    PPP = P PP     Q = U1 PP       (idle)                -> X3 = RR - PPP - 2Q
//     4: (idle)         ZZZ3 = zzz PPP RD = R (Q - X3) E = S1 PPP            -> Y3 = RD - E
// so the four lanes of a QUAD that hold bitwise the same a and b each run one product per level (lane q = the q-th column above) and pass
// the results around with DPP quad permutes (a VALU move, no LDS): 4 products + ~260 moves / selects per addition instead of 14 products.
// Bounds: exactly those of g1x29_add / g1x29_add_tail (same formulas, same operand ranges; the idle lanes multiply normalized values).
// Special cases (an identity operand, P = 0: doubling or inverse) are decided for the whole WAVE: every lane then runs the plain addition,
// which keeps the four copies of a quad identical.
// The empty asm after every move keeps the compiler from FOLDING the move into the arithmetic that uses it: measured on gfx950
// (ezkl_hip_ubench("coopcheck"), profiles/r05f_coopcheck.log) the folded form `v_subrev_u32_dpp d, m, t quad_perm:[3,3,3,3]` that
// LLVM's DPP combiner builds for t - bcast<3>(m) returned t - m of the lane ITSELF in every lane (right only in lane 3) -- the plain
// v_mov_b32_dpp is what every other broadcast of the addition compiled to, and those were all correct.
template <int S>
EZ_D f29_t f29_quad_bcast(const f29_t& x) {               // lane S of every quad to its four lanes
    // a DPP read needs two wait states after the VALU write of its source; the compiler inserts them for code it can see, but the
    // products are inline asm whose last instructions write the limbs -- the s_nop below sits between any such block and the moves
    f29_t t = x, r;
    asm volatile("s_nop 1" : "+v"(t.v[0]), "+v"(t.v[1]), "+v"(t.v[2]), "+v"(t.v[3]), "+v"(t.v[4]), "+v"(t.v[5]), "+v"(t.v[6]), "+v"(t.v[7]), "+v"(t.v[8]));
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)t.v[i], S * 0x55, 0xf, 0xf, true);
        asm volatile("" : "+v"(r.v[i]));
    }
    return r;
}
EZ_D f29_t f29_sel(bool c, const f29_t& a, const f29_t& b) {
    f29_t r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}
EZ_D f29_t f29_sel4(uint32_t q, const f29_t& x0, const f29_t& x1, const f29_t& x2, const f29_t& x3) {
    return f29_sel(q < 2, f29_sel(q == 0, x0, x1), f29_sel(q == 2, x2, x3));
}
template <int S>
EZ_D g1x29_t g1x29_quad_bcast(const g1x29_t& p) {
    g1x29_t r;
    r.x = f29_quad_bcast<S>(p.x); r.y = f29_quad_bcast<S>(p.y); r.zz = f29_quad_bcast<S>(p.zz); r.zzz = f29_quad_bcast<S>(p.zzz);
    return r;
}
// a + b where the four lanes of every quad hold bitwise the same a and the same b; so does the result.  `special` (wave-uniform): the
// operands need the plain addition (an identity, or P = 0) and the returned value is meaningless -- the caller runs g1x29_add.
EZ_D g1x29_t g1x29_add_quad(const g1x29_t& a, const g1x29_t& b, bool& special, uint32_t* dbg = nullptr) {
    const uint32_t q = threadIdx.x & 3u;
    g1x29_t o = a;
    special = __any(g1x29_is_id(a) || g1x29_is_id(b)) != 0;
    if (special) return o;
    // level 1
    const f29_t m1 = Fq29::mul(f29_sel4(q, a.x, b.x, a.y, b.y), f29_sel4(q, b.zz, a.zz, b.zzz, a.zzz));      // < 48/169 + 1 < 2p
    const f29_t u1 = f29_quad_bcast<0>(m1), s1 = f29_quad_bcast<2>(m1);
    const f29_t p = Fq29::normalize(Fq29::sub<1>(f29_quad_bcast<1>(m1), u1));                                // < 6p
    const f29_t r = Fq29::normalize(Fq29::sub<1>(f29_quad_bcast<3>(m1), s1));                                // < 6p
    special = __any(Fq29::is_zero_mod_p(p)) != 0;                                  // rare (doubling / inverse)
    if (special) return o;
    // level 2: zz (q0), zzz (q1), PP (q2), RR (q3)
    const f29_t m2 = Fq29::mul(f29_sel4(q, a.zz, a.zzz, p, r), f29_sel4(q, b.zz, b.zzz, p, r));              // < 36/169 + 1 < 2p
    const f29_t pp = f29_quad_bcast<2>(m2), rr = f29_quad_bcast<3>(m2);
    // level 3: ZZ3 = zz PP (q0), PPP = P PP (q1), Q = U1 PP (q2), q3 idle (RR PP, unused)
    const f29_t m3 = Fq29::mul(f29_sel(q == 1, p, f29_sel(q == 2, u1, m2)), pp);                             // < 2p
    const f29_t ppp = f29_quad_bcast<1>(m3), qq = f29_quad_bcast<2>(m3);
    const f29_t nq = Fq29::neg<1>(qq);
    o.x = Fq29::normalize(Fq29::add(Fq29::add(rr, Fq29::neg<1>(ppp)), Fq29::add(nq, nq)));                   // < 2 + 4 + 8 = 14p
    const f29_t d = Fq29::sub<3>(qq, o.x);                                                                   // < 18p, loose(3)
    // level 4: q0 idle (zz PPP, unused), ZZZ3 = zzz PPP (q1), RD = R D (q2), E = S1 PPP (q3)
    const f29_t m4 = Fq29::mul(f29_sel(q == 2, r, f29_sel(q == 3, s1, m2)), f29_sel(q == 2, d, ppp));        // < 6*18/169 + 1 < 2p
    o.zz = f29_quad_bcast<0>(m3);
    o.zzz = f29_quad_bcast<1>(m4);
    o.y = Fq29::normalize(Fq29::sub<1>(f29_quad_bcast<2>(m4), f29_quad_bcast<3>(m4)));                       // < 2 + 4 = 6p
    if (dbg) {                                   // ezkl_hip_ubench("coopcheck"): every intermediate, 9 limbs each
        const f29_t* t[14] = {&m1, &u1, &s1, &p, &r, &m2, &pp, &rr, &m3, &ppp, &qq, &o.x, &d, &m4};
        for (int k = 0; k < 14; k++)
            for (int i = 0; i < 9; i++) dbg[9 * k + i] = t[k]->v[i];
    }
    return o;
}
// Butterfly sums with cooperative additions, ONE loop around one copy of the addition (it is ~1300 instructions, its plain fall-back with
// the doubling ~6000): the group total of `width` lanes (a power of two in 4..64, aligned) in every lane.  Steps: the four DIFFERENT
// values of a quad are summed by three additions the quad runs together -- (x0 + x1), (x2 + x3), their sum -- then one addition per
// butterfly level 4, 8, ....  Measured in the MSM's reduce2 (1 plain + 7 cooperative levels): 73 -> 50 us (profiles/r05n_*).
EZ_D g1x29_t g1x29_group_sum_coop(g1x29_t acc, uint32_t width) {
    const uint32_t nsteps = 3u + (31u - (uint32_t)__clz(width)) - 2u;      // 3 inside the quad + the butterfly levels above it
    g1x29_t t0 = acc;
#pragma unroll 1
    for (uint32_t k = 0; k < nsteps; k++) {
        g1x29_t a, b;
        if (k == 0) { a = g1x29_quad_bcast<0>(acc); b = g1x29_quad_bcast<1>(acc); }
        else if (k == 1) { a = g1x29_quad_bcast<2>(acc); b = g1x29_quad_bcast<3>(acc); }
        else if (k == 2) { a = t0; b = acc; }
        else { a = acc; b = g1x29_shfl_xor(acc, 4u << (k - 3u)); }
        bool special;
        g1x29_t r = g1x29_add_quad(a, b, special);
        if (special) r = g1x29_add(a, b);                            // wave-uniform
        if (k == 0) t0 = r; else acc = r;
    }
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
// block256_sum with the in-wave levels cooperative (3 + 4 additions instead of 6 plain levels); the four wave totals go through LDS and
// the last two levels are plain butterflies.  (A fully cooperative form -- the wave totals summed by another quad_sum4 inside the same
// loop -- was built and is exact in a stand-alone kernel, but produced wrong plane sums inside msm_planes_kernel / msm_fixup_heavy*
// (every plane, ezkl_hip_ubench("coopcheck") + EZKL_MSM_DEBUG_PLANES, round 5); it would save ~8 us per MSM and was dropped.)
EZ_D g1x29_t g1x29_block256_sum_coop(const g1x29_t& acc0, uint4* sh) {
    g1x29_t acc = g1x29_group_sum_coop(acc0, 64);
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

// ... and with the four wave totals summed cooperatively as well (three additions inside every quad instead of two plain butterfly levels:
// 11 us instead of 19 at one wave per SIMD).  EZKL_MSM_COOP bit 3 selects it in msm_planes_kernel: exact (all 20 planes equal the plain tree's,
// results equal the oracle's) and worth 2 us of a 1.27 ms chain (profiles/r05aj_coop15.log), so it is not the default.
EZ_D g1x29_t g1x29_block256_sum_coop_full(const g1x29_t& acc0, uint4* sh) {
    g1x29_t acc = g1x29_group_sum_coop(acc0, 64);
    if ((threadIdx.x & 63) == 0) {
        const uint32_t* s = acc.x.v;
#pragma unroll
        for (int k = 0; k < 9; k++) sh[k * 4 + (threadIdx.x >> 6)] = make_uint4(s[4 * k], s[4 * k + 1], s[4 * k + 2], s[4 * k + 3]);
    }
    __syncthreads();
    g1x29_t w;
    {
        uint32_t* d = w.x.v;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const uint4 t = sh[k * 4 + (threadIdx.x & 3)];
            d[4 * k] = t.x; d[4 * k + 1] = t.y; d[4 * k + 2] = t.z; d[4 * k + 3] = t.w;
        }
    }
    __syncthreads();
    return g1x29_group_sum_coop(w, 4);
}

}  // namespace ezkl
