// curve.hpp -- BN254 G1 (y^2 = x^3 + 3 over Fq) group law for the MSM kernels.
//
// Bucket accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// a mixed add (affine base into a bucket) is 8M + 2S with no inversion, the cheapest complete-enough
// formula for Pippenger's inner loop.  Identity is ZZ = 0.  Bases are affine, 64 B, (0,0) = identity,
// byte-identical to the raw SRS layout.  The doubling / inverse-point branches are
// data dependent but rare (random bases), so divergence cost is negligible.
#pragma once
#include "field.hpp"

namespace ezkl {

struct alignas(16) g1a_t {
    fe_t x, y;
};
struct alignas(16) g1x_t {
    fe_t x, y, zz, zzz;
};

EZ_HD bool g1a_is_id(const g1a_t& p) { return Fq::is_zero(p.x) && Fq::is_zero(p.y); }
EZ_HD bool g1x_is_id(const g1x_t& p) { return Fq::is_zero(p.zz); }
EZ_HD g1x_t g1x_identity() {
    g1x_t r;
    r.x = Fq::zero(); r.y = Fq::zero(); r.zz = Fq::zero(); r.zzz = Fq::zero();
    return r;
}
EZ_HD g1x_t g1x_from_affine(const g1a_t& p) {
    g1x_t r;
    if (g1a_is_id(p)) return g1x_identity();
    r.x = p.x; r.y = p.y; r.zz = Fq::one(); r.zzz = Fq::one();
    return r;
}

// 2*P for affine P (mdbl-2008-s-1), P != identity
EZ_HD g1x_t g1x_double_affine(const g1a_t& p) {
    g1x_t r;
    fe_t u = Fq::dbl(p.y);
    fe_t v = Fq::sqr(u);
    fe_t w = Fq::mul(u, v);
    fe_t s = Fq::mul(p.x, v);
    fe_t xx = Fq::sqr(p.x);
    fe_t m = Fq::add(Fq::dbl(xx), xx);
    r.x = Fq::sub(Fq::sqr(m), Fq::dbl(s));
    r.y = Fq::sub(Fq::mul(m, Fq::sub(s, r.x)), Fq::mul(w, p.y));
    r.zz = v;
    r.zzz = w;
    return r;
}
// 2*P (dbl-2008-s-1)
EZ_HD g1x_t g1x_double(const g1x_t& p) {
    if (g1x_is_id(p)) return p;
    g1x_t r;
    fe_t u = Fq::dbl(p.y);
    fe_t v = Fq::sqr(u);
    fe_t w = Fq::mul(u, v);
    fe_t s = Fq::mul(p.x, v);
    fe_t xx = Fq::sqr(p.x);
    fe_t m = Fq::add(Fq::dbl(xx), xx);
    r.x = Fq::sub(Fq::sqr(m), Fq::dbl(s));
    r.y = Fq::sub(Fq::mul(m, Fq::sub(s, r.x)), Fq::mul(w, p.y));
    r.zz = Fq::mul(v, p.zz);
    r.zzz = Fq::mul(w, p.zzz);
    return r;
}
// acc + q, q affine (madd-2008-s) with the identity / doubling / inverse cases handled
EZ_HD g1x_t g1x_add_mixed(const g1x_t& a, const g1a_t& q) {
    if (g1a_is_id(q)) return a;
    if (g1x_is_id(a)) return g1x_from_affine(q);
    fe_t u2 = Fq::mul(q.x, a.zz);
    fe_t s2 = Fq::mul(q.y, a.zzz);
    fe_t p = Fq::sub(u2, a.x);
    fe_t r = Fq::sub(s2, a.y);
    if (Fq::is_zero(p)) {
        if (Fq::is_zero(r)) return g1x_double_affine(q);
        return g1x_identity();
    }
    fe_t pp = Fq::sqr(p);
    fe_t ppp = Fq::mul(p, pp);
    fe_t qq = Fq::mul(a.x, pp);
    g1x_t o;
    o.x = Fq::sub(Fq::sub(Fq::sqr(r), ppp), Fq::dbl(qq));
    o.y = Fq::sub(Fq::mul(r, Fq::sub(qq, o.x)), Fq::mul(a.y, ppp));
    o.zz = Fq::mul(a.zz, pp);
    o.zzz = Fq::mul(a.zzz, ppp);
    return o;
}
// a + b (add-2008-s)
EZ_HD g1x_t g1x_add(const g1x_t& a, const g1x_t& b) {
    if (g1x_is_id(a)) return b;
    if (g1x_is_id(b)) return a;
    fe_t u1 = Fq::mul(a.x, b.zz), u2 = Fq::mul(b.x, a.zz);
    fe_t s1 = Fq::mul(a.y, b.zzz), s2 = Fq::mul(b.y, a.zzz);
    fe_t p = Fq::sub(u2, u1), r = Fq::sub(s2, s1);
    if (Fq::is_zero(p)) {
        if (Fq::is_zero(r)) return g1x_double(a);
        return g1x_identity();
    }
    fe_t pp = Fq::sqr(p);
    fe_t ppp = Fq::mul(p, pp);
    fe_t qq = Fq::mul(u1, pp);
    g1x_t o;
    o.x = Fq::sub(Fq::sub(Fq::sqr(r), ppp), Fq::dbl(qq));
    o.y = Fq::sub(Fq::mul(r, Fq::sub(qq, o.x)), Fq::mul(s1, ppp));
    o.zz = Fq::mul(Fq::mul(a.zz, b.zz), pp);
    o.zzz = Fq::mul(Fq::mul(a.zzz, b.zzz), ppp);
    return o;
}
EZ_HD g1a_t g1a_neg(const g1a_t& p) {
    g1a_t r;
    r.x = p.x;
    r.y = Fq::neg(p.y);
    return r;
}
// canonical affine: x = X/ZZ, y = Y/ZZZ ; identity -> (0,0)
EZ_HD g1a_t g1x_to_affine(const g1x_t& p) {
    g1a_t r;
    if (g1x_is_id(p)) {
        r.x = Fq::zero();
        r.y = Fq::zero();
        return r;
    }
    // 1/ZZZ = i ; 1/ZZ = i^2 * ZZ^2 ... cheaper: invert ZZ*ZZZ once
    fe_t t = Fq::inv(Fq::mul(p.zz, p.zzz));
    fe_t izz = Fq::mul(t, p.zzz), izzz = Fq::mul(t, p.zz);
    r.x = Fq::mul(p.x, izz);
    r.y = Fq::mul(p.y, izzz);
    return r;
}

EZ_D g1a_t ld_g1a(const g1a_t* p) {
    g1a_t r;
    r.x = ld_fe(&p->x);
    r.y = ld_fe(&p->y);
    return r;
}
EZ_D void st_g1a(g1a_t* p, const g1a_t& v) {
    st_fe(&p->x, v.x);
    st_fe(&p->y, v.y);
}
EZ_D g1x_t ld_g1x(const g1x_t* p) {
    g1x_t r;
    r.x = ld_fe(&p->x); r.y = ld_fe(&p->y); r.zz = ld_fe(&p->zz); r.zzz = ld_fe(&p->zzz);
    return r;
}
EZ_D void st_g1x(g1x_t* p, const g1x_t& v) {
    st_fe(&p->x, v.x); st_fe(&p->y, v.y); st_fe(&p->zz, v.zz); st_fe(&p->zzz, v.zzz);
}

// ---- moving points between lanes -------------------------------------------------------------------
// A g1x_t is 128 bytes.  An LDS array of g1x_t makes lane i touch dwords 32i..32i+31: every lane of a wave lands in the
// same banks and each 16-byte access serialises 64 ways -- measured: a 6-level LDS tree of additions took 4x the time of
// the additions themselves.  So: inside a wave points move by lane shuffles; across waves they go through LDS in
// "plane" layout, uint4 plane k of slot t at base[k * nslots + t] (consecutive lanes, consecutive 16-byte words).
EZ_D g1x_t g1x_shfl_xor(const g1x_t& p, uint32_t mask) {
    g1x_t r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        r.x.v[k] = __shfl_xor(p.x.v[k], mask);
        r.y.v[k] = __shfl_xor(p.y.v[k], mask);
        r.zz.v[k] = __shfl_xor(p.zz.v[k], mask);
        r.zzz.v[k] = __shfl_xor(p.zzz.v[k], mask);
    }
    return r;
}
EZ_D void g1x_lds_store(uint4* base, uint32_t slot, uint32_t nslots, const g1x_t& p) {
    const fe_t* f = &p.x;               // x, y, zz, zzz are contiguous fe_t
#pragma unroll
    for (int k = 0; k < 4; k++) {
        base[(2 * k) * nslots + slot] = make_uint4(f[k].v[0], f[k].v[1], f[k].v[2], f[k].v[3]);
        base[(2 * k + 1) * nslots + slot] = make_uint4(f[k].v[4], f[k].v[5], f[k].v[6], f[k].v[7]);
    }
}
EZ_D g1x_t g1x_lds_load(const uint4* base, uint32_t slot, uint32_t nslots) {
    g1x_t r;
    fe_t* f = &r.x;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint4 lo = base[(2 * k) * nslots + slot], hi = base[(2 * k + 1) * nslots + slot];
        f[k].v[0] = lo.x; f[k].v[1] = lo.y; f[k].v[2] = lo.z; f[k].v[3] = lo.w;
        f[k].v[4] = hi.x; f[k].v[5] = hi.y; f[k].v[6] = hi.z; f[k].v[7] = hi.w;
    }
    return r;
}
// Sum over each aligned group of `width` lanes (a power of two <= 64) as a butterfly: EVERY lane adds at every level
// and every lane ends with the group total.  Deliberately not the usual halving tree: measured on MI355X, the same
// dependent chain of additions runs 2.3x slower when most lanes are masked off than when all 64 lanes execute it
// (tools/ec_probe.py, ecaddv vs ecaddx) -- the redundant lanes are free and keep the chain at full speed.
EZ_D g1x_t g1x_group_sum(g1x_t acc, uint32_t width) {
#pragma unroll 1
    for (uint32_t s = width >> 1; s > 0; s >>= 1) acc = g1x_add(acc, g1x_shfl_xor(acc, s));
    return acc;
}
// sum over the 256 threads of a workgroup, valid in every thread; sh: 8 * 4 uint4 (one slot per wave)
EZ_D g1x_t g1x_block256_sum(g1x_t acc, uint4* sh) {
    acc = g1x_group_sum(acc, 64);
    if ((threadIdx.x & 63) == 0) g1x_lds_store(sh, threadIdx.x >> 6, 4, acc);
    __syncthreads();
    acc = g1x_lds_load(sh, threadIdx.x & 3, 4);
    __syncthreads();
    return g1x_group_sum(acc, 4);
}

}  // namespace ezkl
