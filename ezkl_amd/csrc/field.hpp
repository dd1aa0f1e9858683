// field.hpp -- BN254 Fr / Fq arithmetic for gfx950 (CDNA4), 8 x 32-bit limbs, Montgomery R = 2^256.
//
// Memory representation is exactly halo2curves' (4 x u64 little-endian Montgomery limbs == 8 x u32
// little-endian), i.e. the raw-bytes SRS / pk layout (SURVEY.md §8(b) "Representation contract"), so
// device buffers are byte-identical to what the Rust host holds and no conversion kernel exists.
//
// The hot primitive is a 32x32+64 -> 64 multiply-add, which gfx950 has as one VALU instruction
// (v_mad_u64_u32).  A Montgomery product is 8 rows x (8 + 8) of those plus carry adds; all loops are
// fully unrolled so the 8 limbs of every operand live in VGPRs and the modulus limbs become SGPR/literal
// operands.  No MFMA: this is modular integer arithmetic (BASELINE.json north_star).
#pragma once
#ifndef __HIPCC_RTC__            // hiprtc (the eval_h JIT) supplies the HIP builtins itself
#include <stdint.h>
#include <hip/hip_runtime.h>
#else
typedef unsigned int uint32_t;
typedef unsigned long long uint64_t;
typedef unsigned long size_t;
#endif
#include "bn254_constants.h"

#define EZ_HD __host__ __device__ __forceinline__
#define EZ_D __device__ __forceinline__

namespace ezkl {

struct alignas(16) fe_t {
    uint32_t v[8];
};
#define EZKL_FE_DEFINED 1

struct FqP {
    static constexpr uint32_t MOD[8] = BN32_FQ_MOD_INIT;
    static constexpr uint32_t ONE[8] = BN32_FQ_R_INIT;
    static constexpr uint32_t R2[8] = BN32_FQ_R2_INIT;
    static constexpr uint32_t INV = BN32_FQ_INV;
};
struct FrP {
    static constexpr uint32_t MOD[8] = BN32_FR_MOD_INIT;
    static constexpr uint32_t ONE[8] = BN32_FR_R_INIT;
    static constexpr uint32_t R2[8] = BN32_FR_R2_INIT;
    static constexpr uint32_t INV = BN32_FR_INV;
};

EZ_HD uint64_t mad_wide(uint32_t a, uint32_t b, uint64_t c) { return (uint64_t)a * b + c; }
// carry chains: lower to v_add_co_u32 / v_addc_co_u32 / v_subb_co_u32 on gfx950
EZ_HD uint32_t addc32(uint32_t a, uint32_t b, uint32_t& c) {
    uint32_t co, r = __builtin_addc(a, b, c, &co);
    c = co;
    return r;
}
EZ_HD uint32_t subb32(uint32_t a, uint32_t b, uint32_t& br) {
    uint32_t bo, r = __builtin_subc(a, b, br, &bo);
    br = bo;
    return r;
}
template <class P> __device__ __forceinline__ fe_t mont_mul_asm(const fe_t& a, const fe_t& b);
template <class P> __device__ __forceinline__ void mont_mul_single(const uint32_t (&a)[8], const uint32_t (&b)[8], uint32_t (&r)[8]);

template <class P>
struct Field {
    EZ_HD static fe_t zero() {
        fe_t r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = 0;
        return r;
    }
    EZ_HD static fe_t one() {
        fe_t r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = P::ONE[i];
        return r;
    }
    EZ_HD static fe_t r2() {
        fe_t r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = P::R2[i];
        return r;
    }
    EZ_HD static bool is_zero(const fe_t& a) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= a.v[i];
        return o == 0;
    }
    EZ_HD static bool eq(const fe_t& a, const fe_t& b) {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
        return o == 0;
    }
    // r = a - MOD if a >= MOD else a  (a < 2*MOD)
    EZ_HD static fe_t reduce_once(const fe_t& a) {
        fe_t d;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d.v[i] = subb32(a.v[i], P::MOD[i], br);
        fe_t r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = br ? a.v[i] : d.v[i];
        return r;
    }
    EZ_HD static fe_t add(const fe_t& a, const fe_t& b) {
        fe_t s;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s.v[i] = addc32(a.v[i], b.v[i], c);
        return reduce_once(s);   // a + b < 2p < 2^255, no carry out of limb 7
    }
    EZ_HD static fe_t sub(const fe_t& a, const fe_t& b) {
        fe_t d;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d.v[i] = subb32(a.v[i], b.v[i], br);
        const uint32_t mask = (uint32_t)0 - br;
        uint32_t c = 0;
        fe_t r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = addc32(d.v[i], P::MOD[i] & mask, c);
        return r;
    }
    EZ_HD static fe_t neg(const fe_t& a) {
        fe_t d;
        uint32_t br = 0, nz = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            d.v[i] = subb32(P::MOD[i], a.v[i], br);
            nz |= a.v[i];
        }
        fe_t r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = nz ? d.v[i] : 0u;
        return r;
    }
    EZ_HD static fe_t dbl(const fe_t& a) { return add(a, a); }

    // Montgomery product a*b*R^-1 mod p, inputs and output fully reduced.
    // Operand-scanning with the reduction interleaved per row ("no-carry" form, valid because the top
    // limb of p is < 2^30): per row i,  t = (t + a*b[i] + m*p) / 2^32  with t < 2p kept in 8 limbs.
    // `mul` is deliberately NOT inlined on the device: one Montgomery product is ~2.5k instructions
    // (~19 KB of code); a point addition inlining 10-14 of them would overflow the 64 KB instruction
    // cache shared by a CU pair.  A call costs ~30 register moves against ~2k cycles of work.
    // Operands travel in VGPRs (8-lane ext vectors by value): struct references would go through scratch.
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    __host__ __device__ __attribute__((noinline)) static u32x8 mul_call(u32x8 a, u32x8 b) {
        fe_t x, y;
#pragma unroll
        for (int i = 0; i < 8; i++) { x.v[i] = a[i]; y.v[i] = b[i]; }
#if defined(__HIP_DEVICE_COMPILE__)
        fe_t t;
        mont_mul_single<P>(x.v, y.v, t.v);      // montmul_gen.hpp: the whole product as one asm statement
        fe_t r = reduce_once(t);
#else
        fe_t r = mul_portable(x, y);
#endif
        u32x8 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = r.v[i];
        return o;
    }
    // ---- lazily reduced forms for the NTT butterflies (Harvey): values live in [0, 2p), which 4p < 2^256 leaves room for ----
    // limb i of 2p
    EZ_HD static constexpr uint32_t mod2(int i) { return (P::MOD[i] << 1) | (i ? (P::MOD[i - 1] >> 31) : 0u); }
    // a in [0, 4p) -> [0, 2p)
    EZ_HD static fe_t reduce_2p(const fe_t& a) {
        fe_t d;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d.v[i] = subb32(a.v[i], mod2(i), br);
        fe_t r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = br ? a.v[i] : d.v[i];
        return r;
    }
    // a, b in [0, 2p): a + b in [0, 2p)
    EZ_HD static fe_t add_lazy(const fe_t& a, const fe_t& b) {
        fe_t s;
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s.v[i] = addc32(a.v[i], b.v[i], c);
        return reduce_2p(s);
    }
    // a, b in [0, 2p): a - b + 2p in (0, 4p) -- no comparison; feed it to mul_lazy or reduce_2p
    EZ_HD static fe_t sub_lazy(const fe_t& a, const fe_t& b) {
        fe_t d;
        uint32_t br = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) d.v[i] = subb32(a.v[i], b.v[i], br);      // mod 2^256; the true value a - b + 2p fits
        uint32_t c = 0;
        fe_t r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = addc32(d.v[i], mod2(i), c);
        return r;
    }
    // a in [0, 4p), b in [0, p) (a twiddle): a b / R in [0, 2p) WITHOUT the final conditional subtraction
    // ((4p p + R p) / R = p (4p/R + 1) < 2p because 4p < 0.76 R)
    __host__ __device__ __attribute__((noinline)) static u32x8 mul_lazy_call(u32x8 a, u32x8 b) {
        fe_t x, y;
#pragma unroll
        for (int i = 0; i < 8; i++) { x.v[i] = a[i]; y.v[i] = b[i]; }
#if defined(__HIP_DEVICE_COMPILE__)
        fe_t r;
        mont_mul_single<P>(x.v, y.v, r.v);
#else
        fe_t r = mul_portable(reduce_once(reduce_2p(x)), y);
#endif
        u32x8 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = r.v[i];
        return o;
    }
    EZ_HD static fe_t mul_lazy(const fe_t& a, const fe_t& b) {
        u32x8 x, y;
#pragma unroll
        for (int i = 0; i < 8; i++) { x[i] = a.v[i]; y[i] = b.v[i]; }
        u32x8 o = mul_lazy_call(x, y);
        fe_t r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = o[i];
        return r;
    }
    EZ_HD static fe_t mul(const fe_t& a, const fe_t& b) {
        u32x8 x, y;
#pragma unroll
        for (int i = 0; i < 8; i++) { x[i] = a.v[i]; y[i] = b.v[i]; }
        u32x8 o = mul_call(x, y);
        fe_t r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = o[i];
        return r;
    }
    EZ_HD static fe_t mul_inl(const fe_t& a, const fe_t& b) {
#if defined(__HIP_DEVICE_COMPILE__)
        return mont_mul_asm<P>(a, b);      // montmul_gen.hpp: v_mad_u64_u32 + v_addc_co_u32 columns
#else
        return mul_portable(a, b);
#endif
    }
    EZ_HD static fe_t mul_portable(const fe_t& a, const fe_t& b) {
        uint32_t t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t bi = b.v[i];
            uint64_t A = mad_wide(a.v[0], bi, t[0]);
            const uint32_t m = (uint32_t)A * P::INV;
            uint64_t C = mad_wide(m, P::MOD[0], (uint32_t)A);
#pragma unroll
            for (int j = 1; j < 8; j++) {
                A = mad_wide(a.v[j], bi, (uint64_t)t[j] + (A >> 32));
                C = mad_wide(m, P::MOD[j], (uint64_t)(uint32_t)A + (C >> 32));
                t[j - 1] = (uint32_t)C;
            }
            t[7] = (uint32_t)(A >> 32) + (uint32_t)(C >> 32);
        }
        fe_t r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = t[i];
        return reduce_once(r);
    }
    EZ_HD static fe_t sqr(const fe_t& a) { return mul(a, a); }

    EZ_HD static fe_t from_mont(const fe_t& a) {
        fe_t o = zero();
        o.v[0] = 1;
        return mul(a, o);
    }
    EZ_HD static fe_t to_mont(const fe_t& a) { return mul(a, r2()); }

    // a^e, e given as 8 x u32 little-endian (host or device; used for inversion / omega powers)
    EZ_HD static fe_t pow(const fe_t& a, const uint32_t e[8]) {
        fe_t acc = one(), base = a;
        for (int i = 0; i < 256; i++) {
            if ((e[i >> 5] >> (i & 31)) & 1) acc = mul(acc, base);
            base = sqr(base);
        }
        return acc;
    }
    EZ_HD static fe_t pow_u64(const fe_t& a, uint64_t e) {
        fe_t acc = one(), base = a;
        while (e) {
            if (e & 1) acc = mul(acc, base);
            base = sqr(base);
            e >>= 1;
        }
        return acc;
    }
    EZ_HD static fe_t inv(const fe_t& a) {   // Fermat, a^(p-2); inv(0) = 0
        uint32_t e[8];
        uint64_t br = 2;
        for (int i = 0; i < 8; i++) {
            uint64_t t = (uint64_t)P::MOD[i] - br;
            e[i] = (uint32_t)t;
            br = (t >> 63) & 1;
        }
        return pow(a, e);
    }
    EZ_HD static fe_t from_u64(uint64_t x) {
        fe_t t = zero();
        t.v[0] = (uint32_t)x;
        t.v[1] = (uint32_t)(x >> 32);
        return to_mont(t);
    }
};

}  // namespace ezkl
namespace ezkl {
#include "montmul_gen.hpp"
template <> __device__ __forceinline__ void mont_mul_single<FqP>(const uint32_t (&a)[8], const uint32_t (&b)[8], uint32_t (&r)[8]) { mont_mul_asm1_fq(a, b, r); }
template <> __device__ __forceinline__ void mont_mul_single<FrP>(const uint32_t (&a)[8], const uint32_t (&b)[8], uint32_t (&r)[8]) { mont_mul_asm1_fr(a, b, r); }
using Fr = Field<FrP>;
using Fq = Field<FqP>;

// 32-byte element global memory access as two 16-byte vector transactions (global_load_dwordx4)
EZ_D fe_t ld_fe(const fe_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    fe_t r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
EZ_D void st_fe(fe_t* p, const fe_t& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

// Fr constants as values
EZ_HD fe_t fr_const(const uint32_t (&c)[8]) {
    fe_t r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = c[i];
    return r;
}
struct FrConst {
    static constexpr uint32_t ROOT[8] = BN32_FR_ROOT_INIT;
    static constexpr uint32_t ZETA[8] = BN32_FR_ZETA_INIT;
    static constexpr uint32_t ZETA2[8] = BN32_FR_ZETA2_INIT;
    static constexpr uint32_t DELTA[8] = BN32_FR_DELTA_INIT;
};
struct FqConst {
    static constexpr uint32_t B3[8] = BN32_FQ_B3_INIT;
};

}  // namespace ezkl
