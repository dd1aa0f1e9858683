// ubench.hip -- on-device microbenchmarks that give the roofline its compute-side context:
// Montgomery products/s (the real ceiling of MSM/NTT), v_mad_u64_u32 issue rate, and HBM copy GB/s.
#include "common.hpp"
#include "curve.hpp"
#include "montmul29_gen.hpp"
#include "curve29.hpp"
#include "host64.hpp"
#include <string.h>
#include <vector>
#include <stdlib.h>

namespace ezkl {

template <int VARIANT>   // 0: shipped path (noinline call of the asm product), 1: portable C product, 2: asm inlined
__global__ __launch_bounds__(256) void ub_modmul_kernel(fe_t* io, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe_t a = ld_fe(io + i), b = a, c = Fr::add(a, a), d = c;
    b.v[0] ^= 5;  b = Fr::reduce_once(b);
    for (int k = 0; k < iters; k++) {      // two independent chains for ILP
        if (VARIANT == 0) { a = Fr::mul(a, b); c = Fr::mul(c, d); b = Fr::mul(b, a); d = Fr::mul(d, c); }
        else if (VARIANT == 1) { a = Fr::mul_portable(a, b); c = Fr::mul_portable(c, d); b = Fr::mul_portable(b, a); d = Fr::mul_portable(d, c); }
        else { a = Fr::mul_inl(a, b); c = Fr::mul_inl(c, d); b = Fr::mul_inl(b, a); d = Fr::mul_inl(d, c); }
    }
    st_fe(io + i, Fr::add(Fr::add(a, b), Fr::add(c, d)));
}
// One step of a batched-affine bucket accumulation as a wave would have to do it: every lane holds the denominator of ITS addition;
// the 64 values are inverted together by Montgomery's trick ACROSS the wave -- inclusive prefix products by 6 shuffle levels, ONE Fermat
// inversion (all lanes run it: divergence-free), then each lane's inverse = prefix[l-1] * inv_total * suffix-corrected by a second
// scan.  Returns nothing useful; the time per (lane, step) against the time of one product is the inversion overhead per addition.
__global__ __launch_bounds__(256) void ub_waveinv_kernel(fe_t* io, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    fe_t x = Fr::reduce_once(ld_fe(io + i));
    if (Fr::is_zero(x)) x = Fr::one();
    for (int k = 0; k < iters; k++) {
        fe_t pre = x;                                   // inclusive prefix product over the wave
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            fe_t up;
#pragma unroll
            for (int q = 0; q < 8; q++) up.v[q] = __shfl_up(pre.v[q], d);
            if (lane >= d) pre = Fr::mul(pre, up);
        }
        fe_t tot;
#pragma unroll
        for (int q = 0; q < 8; q++) tot.v[q] = __shfl(pre.v[q], 63);
        fe_t inv = Fr::inv(tot);                        // 1 / (x_0 ... x_63), computed by every lane
        fe_t suf = x;                                   // inclusive suffix product
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            fe_t dn;
#pragma unroll
            for (int q = 0; q < 8; q++) dn.v[q] = __shfl_down(suf.v[q], d);
            if (lane + d < 64) suf = Fr::mul(suf, dn);
        }
        fe_t before, after;                             // x_l^-1 = inv * prefix[l-1] * suffix[l+1]
#pragma unroll
        for (int q = 0; q < 8; q++) { before.v[q] = __shfl_up(pre.v[q], 1); after.v[q] = __shfl_down(suf.v[q], 1); }
        fe_t r = inv;
        if (lane > 0) r = Fr::mul(r, before);
        if (lane < 63) r = Fr::mul(r, after);
        x = Fr::add(r, x);                              // feed the next step
        if (Fr::is_zero(x)) x = Fr::one();
    }
    st_fe(io + i, x);
}
__global__ __launch_bounds__(256) void ub_addsub_kernel(fe_t* io, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    fe_t a = ld_fe(io + i), b = Fr::reduce_once(a), c = Fr::add(b, b), d = c;
    for (int k = 0; k < iters; k++) { b = Fr::add(b, c); c = Fr::sub(c, d); d = Fr::add(d, b); c = Fr::sub(c, b); }
    st_fe(io + i, Fr::add(Fr::add(a, b), Fr::add(c, d)));
}
__global__ __launch_bounds__(256) void ub_mad64_kernel(uint64_t* io, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = io[i];
    uint64_t a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    uint32_t m = (uint32_t)x | 1u, q = (uint32_t)(x >> 32) | 3u;
    for (int k = 0; k < iters; k++) {
        a0 = mad_wide(m, (uint32_t)a0, a0); a1 = mad_wide(q, (uint32_t)a1, a1);
        a2 = mad_wide(m, (uint32_t)a2, a2); a3 = mad_wide(q, (uint32_t)a3, a3);
        a4 = mad_wide(m, (uint32_t)a4, a4); a5 = mad_wide(q, (uint32_t)a5, a5);
        a6 = mad_wide(m, (uint32_t)a6, a6); a7 = mad_wide(q, (uint32_t)a7, a7);
    }
    io[i] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ __launch_bounds__(256) void ub_mullo_kernel(uint64_t* io, int iters) {       // issue rate of v_mul_lo_u32
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = (uint32_t)io[i];
    uint32_t a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7, m = x | 1u;
    for (int k = 0; k < iters; k++) {
        asm volatile("v_mul_lo_u32 %0, %0, %8\n\tv_mul_lo_u32 %1, %1, %8\n\tv_mul_lo_u32 %2, %2, %8\n\tv_mul_lo_u32 %3, %3, %8\n\t"
                     "v_mul_lo_u32 %4, %4, %8\n\tv_mul_lo_u32 %5, %5, %8\n\tv_mul_lo_u32 %6, %6, %8\n\tv_mul_lo_u32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
    }
    io[i] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ __launch_bounds__(256) void ub_dfma_kernel(double* io, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double x = io[i];
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    const double m = 1.0000001, q = 0.9999999;
    for (int k = 0; k < iters; k++) {
        a0 = __builtin_fma(a0, m, q); a1 = __builtin_fma(a1, q, m);
        a2 = __builtin_fma(a2, m, q); a3 = __builtin_fma(a3, q, m);
        a4 = __builtin_fma(a4, m, q); a5 = __builtin_fma(a5, q, m);
        a6 = __builtin_fma(a6, m, q); a7 = __builtin_fma(a7, q, m);
    }
    io[i] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ __launch_bounds__(256) void ub_copy_kernel(const uint4* in, uint4* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

// random 64-byte record gathers (the access pattern of msm_accumulate_kernel's table reads)
__global__ __launch_bounds__(256) void ub_gather_kernel(const uint4* table, uint32_t mask, uint4* out, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = tid * 2654435761u + 12345u;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int k = 0; k < iters; k++) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const uint4* p = table + (size_t)(x & mask) * 4;
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w; acc.y += a.y + b.x; acc.z ^= c.x + d.y; acc.w += a.w ^ d.z;
    }
    out[tid] = acc;
}

// ---- radix-2^29 Montgomery product (tools/gen_montmul29.py): rate probe and self-check against a portable restatement ----
__device__ __forceinline__ void mont_mul29_ref_fq(const uint32_t (&a)[9], const uint32_t (&b)[9], uint32_t (&r)[9]) {
    const uint32_t MASK = (1u << 29) - 1;
    uint32_t p[9];
    {   // limbs of q in radix 2^29 from the 32-bit limb table
        uint64_t acc = 0; int bits = 0, w = 0;
        for (int i = 0; i < 9; i++) {
            while (bits < 29 && w < 8) { acc |= (uint64_t)FqP::MOD[w++] << bits; bits += 32; }
            p[i] = (uint32_t)acc & MASK; acc >>= 29; bits -= 29;
        }
    }
    uint32_t pinv = 1;                                   // -p^-1 mod 2^29 by Newton iteration
    for (int i = 0; i < 5; i++) pinv *= 2u - p[0] * pinv;
    pinv = (0u - pinv) & MASK;
    uint32_t m[9];
    uint64_t acc = 0;
    for (int k = 0; k < 17; k++) {
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) acc += (uint64_t)a[i] * b[k - i];
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++)
            if (i < k || k >= 9) { if (i != k) acc += (uint64_t)m[i] * p[k - i]; }
        if (k < 9) {
            m[k] = ((uint32_t)acc * pinv) & MASK;
            acc += (uint64_t)m[k] * p[0];
        } else r[k - 9] = (uint32_t)acc & MASK;
        acc >>= 29;
    }
    r[8] = (uint32_t)acc;
}
// the two-chain form of the same product (mont_mul29i_fq, mont_sqr29i_fq): checked against the portable form, timed like the one-chain form
__global__ __launch_bounds__(256) void ub_modmul29i_kernel(uint32_t* io, int iters, uint32_t* mismatches) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a[9], b[9], c[9], d[9], r[9];
    for (int i = 0; i < 9; i++) {
        uint32_t x = io[t * 8 + (i & 7)] * 2654435761u + i;
        a[i] = x & 0x1fffffffu; b[i] = (x >> 3) & 0x1fffffffu; c[i] = (x * 7u) & 0x1fffffffu; d[i] = (x * 13u) & 0x1fffffffu;
    }
    a[8] &= 0x3fffu; b[8] &= 0x3fffu; c[8] &= 0x3fffu; d[8] &= 0x3fffu;
    if (mismatches) {
        uint32_t want[9];
        bool bad = false;
        mont_mul29i_fq(a, b, r);
        mont_mul29_ref_fq(a, b, want);
        for (int i = 0; i < 9; i++) bad |= r[i] != want[i];
        mont_sqr29i_fq(c, r);
        mont_mul29_ref_fq(c, c, want);
        for (int i = 0; i < 9; i++) bad |= r[i] != want[i];
        mont_mul29i_fr(a, d, r);
        mont_mul29_fr(a, d, want);
        for (int i = 0; i < 9; i++) bad |= r[i] != want[i];
        if (bad) atomicAdd(mismatches, 1u);
    }
    for (int k = 0; k < iters; k++) {
        mont_mul29i_fq(a, b, r); for (int i = 0; i < 9; i++) a[i] = r[i];
        mont_mul29i_fq(c, d, r); for (int i = 0; i < 9; i++) c[i] = r[i];
        mont_mul29i_fq(b, a, r); for (int i = 0; i < 9; i++) b[i] = r[i];
        mont_mul29i_fq(d, c, r); for (int i = 0; i < 9; i++) d[i] = r[i];
    }
    uint32_t x = 0;
    for (int i = 0; i < 9; i++) x ^= a[i] ^ b[i] ^ c[i] ^ d[i];
    io[t * 8] = x;
}
__global__ __launch_bounds__(256) void ub_modmul29_kernel(uint32_t* io, int iters, uint32_t* mismatches) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a[9], b[9], c[9], d[9], r[9];
    for (int i = 0; i < 9; i++) {
        uint32_t x = io[t * 8 + (i & 7)] * 2654435761u + i;
        a[i] = x & 0x1fffffffu; b[i] = (x >> 3) & 0x1fffffffu; c[i] = (x * 7u) & 0x1fffffffu; d[i] = (x * 13u) & 0x1fffffffu;
    }
    a[8] &= 0x3fffu; b[8] &= 0x3fffu; c[8] &= 0x3fffu; d[8] &= 0x3fffu;          // < 2^246: far below p
    if (mismatches) {
        uint32_t want[9];
        mont_mul29_fq(a, b, r);
        mont_mul29_ref_fq(a, b, want);
        bool bad = false;
        for (int i = 0; i < 9; i++) bad |= r[i] != want[i];
        if (bad) atomicAdd(mismatches, 1u);
    }
    for (int k = 0; k < iters; k++) {      // two independent chains, as in ub_modmul_kernel
        mont_mul29_fq(a, b, r); for (int i = 0; i < 9; i++) a[i] = r[i];
        mont_mul29_fq(c, d, r); for (int i = 0; i < 9; i++) c[i] = r[i];
        mont_mul29_fq(b, a, r); for (int i = 0; i < 9; i++) b[i] = r[i];
        mont_mul29_fq(d, c, r); for (int i = 0; i < 9; i++) d[i] = r[i];
    }
    uint32_t x = 0;
    for (int i = 0; i < 9; i++) x ^= a[i] ^ b[i] ^ c[i] ^ d[i];
    io[t * 8] = x;
}

// ---- is there a faster field product in the FP64 pipe? ------------------------------------------------------------
// (a) issue probes: 8 independent chains of v_mad_u64_u32 / v_fma_f64 / 4 + 4 of each interleaved (co-issue would show as MORE lane-ops per
//     second than either alone) / v_lshl_add_u64 (gfx940+: one-instruction 64-bit integer add, what a DFMA product accumulates with)
// (b) a complete 5 x 52-bit Montgomery product on the FP64 pipe (Emmart / Zheng / Weems: hi = fma_rz(a, b, 2^104), lo = fma_rz(a, b,
//     (2^104 + 2^52) - hi); the mantissas ARE the 52-bit halves), product-scanning form with 64-bit integer column sums, checked against a
//     portable integer restatement of the same product, timed like ub_modmul29_kernel.
template <int MODE>
__global__ __launch_bounds__(256) void ub_mix_kernel(uint64_t* io, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = io[i];
    uint64_t a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    double d0 = (double)(x & 1023), d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    const double m = 1.0000001, q = 0.9999999;
    uint32_t mi = (uint32_t)x | 1u, qi = (uint32_t)(x >> 32) | 3u;
    for (int k = 0; k < iters; k++) {
        if (MODE == 0) {          // 8 x v_mad_u64_u32
            a0 = mad_wide(mi, (uint32_t)a0, a0); a1 = mad_wide(qi, (uint32_t)a1, a1); a2 = mad_wide(mi, (uint32_t)a2, a2); a3 = mad_wide(qi, (uint32_t)a3, a3);
            a4 = mad_wide(mi, (uint32_t)a4, a4); a5 = mad_wide(qi, (uint32_t)a5, a5); a6 = mad_wide(mi, (uint32_t)a6, a6); a7 = mad_wide(qi, (uint32_t)a7, a7);
        } else if (MODE == 1) {   // 8 x v_fma_f64
            d0 = __builtin_fma(d0, m, q); d1 = __builtin_fma(d1, q, m); d2 = __builtin_fma(d2, m, q); d3 = __builtin_fma(d3, q, m);
            d4 = __builtin_fma(d4, m, q); d5 = __builtin_fma(d5, q, m); d6 = __builtin_fma(d6, m, q); d7 = __builtin_fma(d7, q, m);
        } else if (MODE == 2) {   // 4 + 4 interleaved
            a0 = mad_wide(mi, (uint32_t)a0, a0); d4 = __builtin_fma(d4, m, q); a1 = mad_wide(qi, (uint32_t)a1, a1); d5 = __builtin_fma(d5, q, m);
            a2 = mad_wide(mi, (uint32_t)a2, a2); d6 = __builtin_fma(d6, m, q); a3 = mad_wide(qi, (uint32_t)a3, a3); d7 = __builtin_fma(d7, q, m);
        } else if (MODE == 3) {   // 8 x v_lshl_add_u64 (64-bit add in one instruction)
            asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n\tv_lshl_add_u64 %1, %1, 0, %2\n\tv_lshl_add_u64 %2, %2, 0, %3\n\tv_lshl_add_u64 %3, %3, 0, %4\n\t"
                         "v_lshl_add_u64 %4, %4, 0, %5\n\tv_lshl_add_u64 %5, %5, 0, %6\n\tv_lshl_add_u64 %6, %6, 0, %7\n\tv_lshl_add_u64 %7, %7, 0, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else {                  // 8 x v_add_f64
            asm volatile("v_add_f64 %0, %0, %8\n\tv_add_f64 %1, %1, %9\n\tv_add_f64 %2, %2, %8\n\tv_add_f64 %3, %3, %9\n\t"
                         "v_add_f64 %4, %4, %8\n\tv_add_f64 %5, %5, %9\n\tv_add_f64 %6, %6, %8\n\tv_add_f64 %7, %7, %9"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(m), "v"(q));
        }
    }
    io[i] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint64_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

struct Dfma52 {
    static constexpr uint64_t M52 = (1ull << 52) - 1;
    // q = BN254 base modulus in 5 limbs of 52 bits; PINV = -q^-1 mod 2^52 (both derived at compile time from the 32-bit limb table)
    EZ_HD static constexpr uint64_t limb(int i) {
        uint64_t v = 0;
        for (int b = 0; b < 52; b++) {
            const int bit = 52 * i + b;
            if (bit < 256 && ((FqP::MOD[bit >> 5] >> (bit & 31)) & 1u)) v |= 1ull << b;
        }
        return v;
    }
    EZ_HD static constexpr uint64_t pinv() {
        uint64_t x = 1;
        for (int i = 0; i < 6; i++) x *= 2 - limb(0) * x;      // Newton: q^-1 mod 2^64
        return (0 - x) & M52;
    }
};
// one 52 x 52 -> (hi, lo) limb product on the FP64 pipe, MODE.FP_ROUND (double) = toward zero; returns the RAW bit patterns
// 0x467.. | hi and 0x433.. | lo: callers sum the raw words and remove count x pattern once per column
__device__ __forceinline__ void dfma_hilo(double a, double b, uint64_t& hi_raw, uint64_t& lo_raw) {
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    // (plain fma / subtraction: without fast-math the compiler may not re-associate them, and every operand is a run-time value; inline
    //  asm here made the compiler pad each statement with s_nop for hazards it could not see)
    const double h = __builtin_fma(a, b, C1);
    const double s = C2 - h;
    const double l = __builtin_fma(a, b, s);
    hi_raw = (uint64_t)__double_as_longlong(h);
    lo_raw = (uint64_t)__double_as_longlong(l);
}
__device__ __forceinline__ void mont_mul52_dfma_fq(const double (&a)[5], const double (&b)[5], double (&r)[5]) {
    constexpr uint64_t HI_PAT = 0x4670000000000000ull, LO_PAT = 0x4330000000000000ull, M52 = Dfma52::M52;
    double m[5];
    double pd[5];
#pragma unroll
    for (int i = 0; i < 5; i++) pd[i] = (double)Dfma52::limb(i);
    const double pinv = (double)Dfma52::pinv();
    uint64_t carry = 0, nxt = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        uint64_t acc = carry + nxt, hi, lo;
        nxt = 0;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int j = k - i;
            if (j < 0 || j > 4) continue;
            dfma_hilo(a[i], b[j], hi, lo);
            acc += lo; nxt += hi; cnt++;
            if (i < k && i < 5 && k < 5) {}                 // (m_i known only for i < k in the low half)
        }
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int j = k - i;
            if (j < 0 || j > 4) continue;
            if (k < 5 && i == k) continue;                  // m_k p_0 is added below, once m_k exists
            dfma_hilo(m[i], pd[j], hi, lo);
            acc += lo; nxt += hi; cnt++;
        }
        acc -= (uint64_t)cnt * LO_PAT;
        nxt -= (uint64_t)cnt * HI_PAT;
        if (k < 5) {
            // m_k = (acc mod 2^52) * pinv mod 2^52: the low half of one more FP64 product; acc's low 52 bits enter the pipe as 2^52 + x - 2^52
            const double t = __longlong_as_double((long long)(LO_PAT | (acc & M52))) - 0x1p52;
            dfma_hilo(t, pinv, hi, lo);
            m[k] = __longlong_as_double((long long)(LO_PAT | (lo & M52))) - 0x1p52;
            dfma_hilo(m[k], pd[0], hi, lo);
            acc += lo - LO_PAT; nxt += hi - HI_PAT;
            carry = acc >> 52;                               // the low 52 bits are zero now
        } else {
            r[k - 5] = __longlong_as_double((long long)(LO_PAT | (acc & M52))) - 0x1p52;
            carry = acc >> 52;
        }
    }
    // a, b < q: the result is < 2q and fits 5 limbs (top limb < 2^47): left unreduced, like the radix-2^29 product
}
// the same product with 64-bit integer arithmetic only (portable restatement, the checker of the kernel below)
__device__ inline void mont_mul52_ref_fq(const uint64_t (&a)[5], const uint64_t (&b)[5], uint64_t (&r)[5]) {
    constexpr uint64_t M52 = Dfma52::M52;
    uint64_t m[5], p[5];
    for (int i = 0; i < 5; i++) p[i] = Dfma52::limb(i);
    const uint64_t pinv = Dfma52::pinv();
    unsigned __int128 acc = 0;
    for (int k = 0; k < 10; k++) {
        for (int i = 0; i < 5; i++) {
            const int j = k - i;
            if (j < 0 || j > 4) continue;
            acc += (unsigned __int128)a[i] * b[j];
            if (!(k < 5 && i == k)) acc += (unsigned __int128)m[i] * p[j];
        }
        if (k < 5) {
            m[k] = (((uint64_t)acc & M52) * pinv) & M52;
            acc += (unsigned __int128)m[k] * p[0];
        } else r[k - 5] = (uint64_t)acc & M52;
        acc >>= 52;
    }
}
__global__ __launch_bounds__(256) void ub_dfma52_kernel(uint32_t* io, int iters, uint32_t* mismatches) {
    // double-precision rounding mode of this wave: toward zero (MODE.FP_ROUND bits [3:2] = 3); nothing else in the kernel is floating point
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t ai[5], bi[5], ci[5], di[5];
    for (int i = 0; i < 5; i++) {
        uint64_t x = ((uint64_t)io[t * 8 + (i & 7)] * 2654435761u + i) * 0x9e3779b97f4a7c15ull;
        ai[i] = x & Dfma52::M52; bi[i] = (x >> 7) & Dfma52::M52; ci[i] = (x * 7u) & Dfma52::M52; di[i] = (x * 13u) & Dfma52::M52;
    }
    ai[4] &= (1ull << 44) - 1; bi[4] &= (1ull << 44) - 1; ci[4] &= (1ull << 44) - 1; di[4] &= (1ull << 44) - 1;      // < 2^252 < q
    double a[5], b[5], c[5], d[5], r[5];
    for (int i = 0; i < 5; i++) { a[i] = (double)ai[i]; b[i] = (double)bi[i]; c[i] = (double)ci[i]; d[i] = (double)di[i]; }
    if (mismatches) {
        uint64_t want[5];
        bool bad = false;
        mont_mul52_dfma_fq(a, b, r);
        mont_mul52_ref_fq(ai, bi, want);
        for (int i = 0; i < 5; i++) bad |= (uint64_t)r[i] != want[i];
        mont_mul52_dfma_fq(c, d, r);
        mont_mul52_ref_fq(ci, di, want);
        for (int i = 0; i < 5; i++) bad |= (uint64_t)r[i] != want[i];
        if (bad) atomicAdd(mismatches, 1u);
    }
    for (int k = 0; k < iters; k++) {      // two independent chains, as in ub_modmul29_kernel (operands stay < 2q: limbs < 2^52, top limb < 2^47)
        mont_mul52_dfma_fq(a, b, r); for (int i = 0; i < 5; i++) a[i] = r[i];
        mont_mul52_dfma_fq(c, d, r); for (int i = 0; i < 5; i++) c[i] = r[i];
        mont_mul52_dfma_fq(b, a, r); for (int i = 0; i < 5; i++) b[i] = r[i];
        mont_mul52_dfma_fq(d, c, r); for (int i = 0; i < 5; i++) d[i] = r[i];
    }
    double x = 0;
    for (int i = 0; i < 5; i++) x += a[i] + b[i] + c[i] + d[i];
    io[t * 8] = (uint32_t)(uint64_t)x;
}

// dependent chain of general XYZZ additions: launch latency / cold-instruction-fetch probe for the small tail kernels
__global__ __launch_bounds__(256) void ub_ecadd_kernel(g1x_t* io, int iters) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    g1x_t a = ld_g1x(io + i), b = ld_g1x(io + i + 1);
    for (int k = 0; k < iters; k++) a = g1x_add(a, b);
    st_g1x(io + i, a);
}

// the shape of the reduce tail: one wave per block, a 6-level shuffle tree of general additions
__global__ __launch_bounds__(64) void ub_ectree_kernel(g1x_t* io) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    g1x_t acc = g1x_group_sum(ld_g1x(io + i), 64);
    if (threadIdx.x == 0) st_g1x(io + i, acc);
}

// variants that separate the cost of the lane exchange from the cost of the partial EXEC mask
__global__ __launch_bounds__(64) void ub_ectree_u_kernel(g1x_t* io) {     // all lanes add at every level
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    g1x_t acc = ld_g1x(io + i);
#pragma unroll 1
    for (uint32_t s = 32; s > 0; s >>= 1) acc = g1x_add(acc, g1x_shfl_xor(acc, s ^ 63));
    st_g1x(io + i, acc);
}
__global__ __launch_bounds__(64) void ub_ectree_v_kernel(g1x_t* io, uint32_t lim) {     // no exchange, partial EXEC mask only (lim = 0)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;                           // or the same branch with all lanes in (lim = 64)
    g1x_t acc = ld_g1x(io + i), b = ld_g1x(io + i + 1);
#pragma unroll 1
    for (uint32_t s = 32; s > 0; s >>= 1)
        if (threadIdx.x < (s | lim)) acc = g1x_add(acc, b);
    st_g1x(io + i, acc);
}


// ---- diagnostics of the quad-cooperative addition (curve29.hpp), one wave: out[lane] holds
//   [0..3]   quad_bcast<0..3> of the lane id            [4] special flag of the cooperative addition
//   [8..43]  a + b by g1x29_add      [44..79] a + b by g1x29_add_quad   (a, b = lanes 0 and 1 of the quad's points, so quad-uniform)
//   [80..115] plain butterfly sum of the 64 points   [116..151] cooperative butterfly sum
__global__ __launch_bounds__(64) void ub_coopcheck_kernel(const g1a_t* pts, uint32_t* out) {
    const uint32_t l = threadIdx.x;
    uint32_t* o = out + (size_t)l * 160;
    o[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)l, 0x00, 0xf, 0xf, true);
    o[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)l, 0x55, 0xf, 0xf, true);
    o[2] = (uint32_t)__builtin_amdgcn_mov_dpp((int)l, 0xaa, 0xf, 0xf, true);
    o[3] = (uint32_t)__builtin_amdgcn_mov_dpp((int)l, 0xff, 0xf, 0xf, true);
    g1x29_t x = g1x29_add_mixed(g1x29_identity(), g1a29_unpack(ld_g1a(pts + l)), false);
    x = g1x29_add(x, g1x29_add_mixed(g1x29_identity(), g1a29_unpack(ld_g1a(pts + 64 + l)), false));      // general ZZ
    const g1x29_t a = g1x29_quad_bcast<0>(x), b = g1x29_quad_bcast<1>(x);
    const g1x29_t plain = g1x29_add(a, b);
    bool special;
    const g1x29_t coop = g1x29_add_quad(a, b, special, out + 64 * 160 + (size_t)l * 128);
    o[4] = special ? 1u : 0u;
    {   // experiment: the broadcast-then-subtract of Y3, three ways, on a per-lane value v = limbs of (lane-dependent) x
        const f29_t v = x.y;                                             // differs per lane
        const f29_t b2 = f29_quad_bcast<2>(v), b3 = f29_quad_bcast<3>(v);
        const f29_t y1 = Fq29::sub<1>(b2, b3);                           // DPP combined into the subtraction by the compiler
        f29_t c2 = b2, c3 = b3;
        for (int i = 0; i < 9; i++) { asm volatile("" : "+v"(c2.v[i])); asm volatile("" : "+v"(c3.v[i])); }     // moves kept apart
        const f29_t y2 = Fq29::sub<1>(c2, c3);
        f29_t s2, s3;
        for (int i = 0; i < 9; i++) { s2.v[i] = __shfl(v.v[i], (l & ~3u) | 2u); s3.v[i] = __shfl(v.v[i], (l & ~3u) | 3u); }
        const f29_t y3 = Fq29::sub<1>(s2, s3);
        uint32_t* e = out + 64 * 360 + (size_t)l * 36;
        for (int i = 0; i < 9; i++) { e[i] = y1.v[i]; e[9 + i] = y2.v[i]; e[18 + i] = y3.v[i]; e[27 + i] = v.v[i]; }
    }
    for (int i = 0; i < 36; i++) { out[64 * 160 + 64 * 128 + (size_t)l * 72 + i] = a.x.v[i]; out[64 * 160 + 64 * 128 + (size_t)l * 72 + 36 + i] = b.x.v[i]; }
    for (int i = 0; i < 36; i++) { o[8 + i] = plain.x.v[i]; o[44 + i] = coop.x.v[i]; }
    const g1x29_t s0 = g1x29_group_sum(x, 64), s1 = g1x29_group_sum_coop(x, 64);
    for (int i = 0; i < 36; i++) { o[80 + i] = s0.x.v[i]; o[116 + i] = s1.x.v[i]; }
}

// widths 32 / 8 / 4 and the 256-thread block form: out[t] = 6 points of 36 words (plain, cooperative) x (32, 8, block)
__global__ __launch_bounds__(256) void ub_coopcheck2_kernel(const g1a_t* pts, uint32_t* out, int with_identities) {
    __shared__ uint4 sh[9 * 4];
    const uint32_t t = threadIdx.x;
    g1x29_t x = g1x29_add_mixed(g1x29_identity(), g1a29_unpack(ld_g1a(pts + t)), false);
    x = g1x29_add(x, g1x29_add_mixed(g1x29_identity(), g1a29_unpack(ld_g1a(pts + 256 + t)), false));
    if (with_identities && (t % 5 == 1 || (t >= 64 && t < 72))) x = g1x29_identity();
    uint32_t* o = out + (size_t)t * 216;
    const g1x29_t a0 = g1x29_group_sum(x, 32), a1 = g1x29_group_sum_coop(x, 32);
    const g1x29_t b0 = g1x29_group_sum(x, 8), b1 = g1x29_group_sum_coop(x, 8);
    if (t < 36) sh[t] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const g1x29_t c1 = g1x29_block256_sum_coop(x, sh);      // first, on cleared LDS: nothing left over from the plain form can help it
    const g1x29_t c0 = g1x29_block256_sum(x, sh);
    for (int i = 0; i < 36; i++) { o[i] = a0.x.v[i]; o[36 + i] = a1.x.v[i]; o[72 + i] = b0.x.v[i]; o[108 + i] = b1.x.v[i]; o[144 + i] = c0.x.v[i]; o[180 + i] = c1.x.v[i]; }
}

int ubench(Ctx* c, const char* which, double* out) {
    hipStream_t st = c->stream;
    hipEvent_t e0, e1;
    int rc = ev_pair(c, "ubench", &e0, &e1);
    if (rc) return rc;
    const int blocks = c->num_cus * 16, threads = 256;
    const size_t nthreads = (size_t)blocks * threads;
    float ms = 0.f;
    const bool is_mm = !strncmp(which, "modmul", 6) && strncmp(which, "modmul29", 8);
    if (is_mm || !strcmp(which, "mad64") || !strcmp(which, "mullo") || !strcmp(which, "dfma") || !strcmp(which, "addsub")) {
        void* buf = nullptr;
        EZ_HIP(hipMalloc(&buf, nthreads * 32));
        EZ_HIP(hipMemsetAsync(buf, 0x11, nthreads * 32, st));
        const int iters = is_mm ? 256 : 4096;
        int blocks = c->num_cus * 16;
        if (const char* o = strstr(which, "_o")) blocks = c->num_cus * atoi(o + 2);
        const size_t nthreads = (size_t)blocks * threads;
        for (int rep = 0; rep < 2; rep++) {     // rep 0 = warm-up
            EZ_HIP(hipEventRecord(e0, st));
            if (!strncmp(which, "modmul_o", 8) || !strcmp(which, "modmul")) hipLaunchKernelGGL(ub_modmul_kernel<0>, dim3(blocks), dim3(threads), 0, st, (fe_t*)buf, iters);
            else if (!strcmp(which, "modmul_c")) hipLaunchKernelGGL(ub_modmul_kernel<1>, dim3(blocks), dim3(threads), 0, st, (fe_t*)buf, iters);
            else if (!strcmp(which, "modmul_inl")) hipLaunchKernelGGL(ub_modmul_kernel<2>, dim3(blocks), dim3(threads), 0, st, (fe_t*)buf, iters);
            else if (!strcmp(which, "addsub")) hipLaunchKernelGGL(ub_addsub_kernel, dim3(blocks), dim3(threads), 0, st, (fe_t*)buf, iters);
            else if (!strcmp(which, "mullo")) hipLaunchKernelGGL(ub_mullo_kernel, dim3(blocks), dim3(threads), 0, st, (uint64_t*)buf, iters);
            else if (!strcmp(which, "mad64")) hipLaunchKernelGGL(ub_mad64_kernel, dim3(blocks), dim3(threads), 0, st, (uint64_t*)buf, iters);
            else hipLaunchKernelGGL(ub_dfma_kernel, dim3(blocks), dim3(threads), 0, st, (double*)buf, iters);
            EZ_HIP(hipEventRecord(e1, st));
            EZ_HIP(hipStreamSynchronize(st));
            EZ_HIP(hipEventElapsedTime(&ms, e0, e1));
        }
        EZ_HIP(hipFree(buf));
        const double per_thread = (is_mm || !strcmp(which, "addsub")) ? 4.0 * iters : 8.0 * iters;
        *out = per_thread * (double)nthreads / (ms * 1e-3);
        return EZKL_OK;
    }
    if (!strcmp(which, "waveinv")) {            // wave-batched inversions per second (one per lane and step), full occupancy
        const int threads = 256, blocks = c->num_cus * 16, iters = 16;
        const size_t nthreads = (size_t)threads * blocks;
        void* buf = nullptr;
        EZ_HIP(hipMalloc(&buf, nthreads * sizeof(fe_t)));
        std::vector<fe_t> init(nthreads);
        for (size_t t = 0; t < nthreads; t++) { init[t] = Fr::one(); init[t].v[0] = (uint32_t)(t * 2654435761u) | 1u; init[t].v[7] = 0; }
        EZ_HIP(hipMemcpy(buf, init.data(), nthreads * sizeof(fe_t), hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        EZ_HIP(hipEventCreate(&e0)); EZ_HIP(hipEventCreate(&e1));
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            EZ_HIP(hipEventRecord(e0, st));
            hipLaunchKernelGGL(ub_waveinv_kernel, dim3(blocks), dim3(threads), 0, st, (fe_t*)buf, iters);
            EZ_HIP(hipEventRecord(e1, st));
            EZ_HIP(hipEventSynchronize(e1));
            EZ_HIP(hipEventElapsedTime(&ms, e0, e1));
        }
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        (void)hipFree(buf);
        *out = (double)nthreads * iters / (ms * 1e-3);
        return EZKL_OK;
    }
    if (!strncmp(which, "mix_", 4)) {           // "mix_mad" / "mix_dfma" / "mix_both" / "mix_add64" / "mix_dadd" [_oN]: lane-ops per second
        void* buf = nullptr;
        int blocks = c->num_cus * 16;
        if (const char* o = strstr(which, "_o")) blocks = c->num_cus * atoi(o + 2);
        const size_t nthreads = (size_t)blocks * threads;
        EZ_HIP(hipMalloc(&buf, nthreads * 8));
        EZ_HIP(hipMemsetAsync(buf, 0x11, nthreads * 8, st));
        const int iters = 4096;
        for (int rep = 0; rep < 2; rep++) {
            EZ_HIP(hipEventRecord(e0, st));
            if (!strncmp(which, "mix_mad", 7)) hipLaunchKernelGGL(ub_mix_kernel<0>, dim3(blocks), dim3(threads), 0, st, (uint64_t*)buf, iters);
            else if (!strncmp(which, "mix_dfma", 8)) hipLaunchKernelGGL(ub_mix_kernel<1>, dim3(blocks), dim3(threads), 0, st, (uint64_t*)buf, iters);
            else if (!strncmp(which, "mix_both", 8)) hipLaunchKernelGGL(ub_mix_kernel<2>, dim3(blocks), dim3(threads), 0, st, (uint64_t*)buf, iters);
            else if (!strncmp(which, "mix_add64", 9)) hipLaunchKernelGGL(ub_mix_kernel<3>, dim3(blocks), dim3(threads), 0, st, (uint64_t*)buf, iters);
            else hipLaunchKernelGGL(ub_mix_kernel<4>, dim3(blocks), dim3(threads), 0, st, (uint64_t*)buf, iters);
            EZ_HIP(hipEventRecord(e1, st));
            EZ_HIP(hipStreamSynchronize(st));
            EZ_HIP(hipEventElapsedTime(&ms, e0, e1));
        }
        EZ_HIP(hipFree(buf));
        *out = 8.0 * iters * (double)nthreads / (ms * 1e-3);
        return EZKL_OK;
    }
    if (!strncmp(which, "dfma52", 6)) {         // "dfma52[_oN]": FP64-pipe Montgomery products per second; "dfma52_check": mismatches vs the integer form
        void* buf = nullptr;
        uint32_t* mis = nullptr;
        const bool chk = strstr(which, "check") != nullptr;
        const int iters = chk ? 1 : 256;
        int blocks = c->num_cus * 16;
        if (const char* o = strstr(which, "_o")) blocks = c->num_cus * atoi(o + 2);
        const size_t nthreads = (size_t)blocks * threads;
        EZ_HIP(hipMalloc(&buf, nthreads * 32));
        EZ_HIP(hipMalloc((void**)&mis, 4));
        {   // distinct words per thread: the check covers 2 x nthreads random operand pairs
            std::vector<uint32_t> h(nthreads * 8);
            uint64_t x = 0x9e3779b97f4a7c15ull;
            for (auto& w : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = (uint32_t)(x >> 16); }
            EZ_HIP(hipMemcpyAsync(buf, h.data(), nthreads * 32, hipMemcpyHostToDevice, st));
            EZ_HIP(hipStreamSynchronize(st));
        }
        EZ_HIP(hipMemsetAsync(mis, 0, 4, st));
        for (int rep = 0; rep < 2; rep++) {
            EZ_HIP(hipEventRecord(e0, st));
            hipLaunchKernelGGL(ub_dfma52_kernel, dim3(blocks), dim3(threads), 0, st, (uint32_t*)buf, iters, chk ? mis : nullptr);
            EZ_HIP(hipEventRecord(e1, st));
            EZ_HIP(hipStreamSynchronize(st));
            EZ_HIP(hipEventElapsedTime(&ms, e0, e1));
        }
        uint32_t hm = 0;
        EZ_HIP(hipMemcpy(&hm, mis, 4, hipMemcpyDeviceToHost));
        EZ_HIP(hipFree(buf));
        EZ_HIP(hipFree(mis));
        *out = chk ? (double)hm : 4.0 * iters * (double)nthreads / (ms * 1e-3);
        return EZKL_OK;
    }
    if (!strncmp(which, "modmul29", 8)) {       // "modmul29": products per second; "modmul29_check": mismatches vs the portable form
        void* buf = nullptr;
        uint32_t* mis = nullptr;
        EZ_HIP(hipMalloc(&buf, nthreads * 32));
        EZ_HIP(hipMalloc((void**)&mis, 4));
        EZ_HIP(hipMemsetAsync(buf, 0x5b, nthreads * 32, st));
        EZ_HIP(hipMemsetAsync(mis, 0, 4, st));
        const bool chk = strstr(which, "check") != nullptr;
        const int iters = chk ? 1 : 256;
        int blocks = c->num_cus * 16;                              // "modmul29_oN": N waves per SIMD instead of as many as fit
        if (const char* o = strstr(which, "_o")) blocks = c->num_cus * atoi(o + 2);
        const size_t nthreads = (size_t)blocks * threads;
        for (int rep = 0; rep < 2; rep++) {
            EZ_HIP(hipEventRecord(e0, st));
            if (!strncmp(which, "modmul29i", 9)) hipLaunchKernelGGL(ub_modmul29i_kernel, dim3(blocks), dim3(threads), 0, st, (uint32_t*)buf, iters, chk ? mis : nullptr);
            else hipLaunchKernelGGL(ub_modmul29_kernel, dim3(blocks), dim3(threads), 0, st, (uint32_t*)buf, iters, chk ? mis : nullptr);
            EZ_HIP(hipEventRecord(e1, st));
            EZ_HIP(hipStreamSynchronize(st));
            EZ_HIP(hipEventElapsedTime(&ms, e0, e1));
        }
        uint32_t hm = 0;
        EZ_HIP(hipMemcpy(&hm, mis, 4, hipMemcpyDeviceToHost));
        EZ_HIP(hipFree(buf));
        EZ_HIP(hipFree(mis));
        *out = chk ? (double)hm : 4.0 * iters * (double)nthreads / (ms * 1e-3);
        return EZKL_OK;
    }
    if (!strncmp(which, "ecadd", 5)) {         // "ecadd<iters>[w|f]": microseconds of the 4th launch; w = one wave, f = full GPU
        const int iters = atoi(which + 5);
        const char mode = which[strlen(which) - 1];       // w: one wave, f: 4 waves per SIMD, h: one wave per SIMD, q: one wave per CU
        const bool full = mode == 'f';
        const int eb = full ? c->num_cus * 4 : mode == 'h' ? c->num_cus * 4 : mode == 'q' ? c->num_cus : 1, et = full ? 256 : 64;
        void* buf = nullptr;
        const size_t cnt = (size_t)eb * et + 1;
        EZ_HIP(hipMalloc(&buf, cnt * sizeof(g1x_t)));
        EZ_HIP(hipMemsetAsync(buf, 0, cnt * sizeof(g1x_t), st));
        {   // a valid point (generator (1, 2), zz = zzz = 1 in Montgomery form) everywhere except slot 0 (identity): a += G repeatedly
            std::vector<g1x_t> h(cnt);
            g1x_t g;
            g.x = Fq::one(); g.y = Fq::add(Fq::one(), Fq::one()); g.zz = Fq::one(); g.zzz = Fq::one();
            for (size_t i = 0; i < cnt; i++) h[i] = g;
            for (size_t i = 0; i + 1 < cnt; i += 2) h[i] = g1x_double(g);
            EZ_HIP(hipMemcpyAsync(buf, h.data(), cnt * sizeof(g1x_t), hipMemcpyHostToDevice, st));
            EZ_HIP(hipStreamSynchronize(st));
        }
        for (int rep = 0; rep < 4; rep++) {
            EZ_HIP(hipEventRecord(e0, st));
            if (which[5] == 't') hipLaunchKernelGGL(ub_ectree_kernel, dim3(eb), dim3(64), 0, st, (g1x_t*)buf);
            else if (which[5] == 'u') hipLaunchKernelGGL(ub_ectree_u_kernel, dim3(eb), dim3(64), 0, st, (g1x_t*)buf);
            else if (which[5] == 'v') hipLaunchKernelGGL(ub_ectree_v_kernel, dim3(eb), dim3(64), 0, st, (g1x_t*)buf, 0u);
            else if (which[5] == 'x') hipLaunchKernelGGL(ub_ectree_v_kernel, dim3(eb), dim3(64), 0, st, (g1x_t*)buf, 64u);
            else hipLaunchKernelGGL(ub_ecadd_kernel, dim3(eb), dim3(et), 0, st, (g1x_t*)buf, iters);
            EZ_HIP(hipEventRecord(e1, st));
            EZ_HIP(hipStreamSynchronize(st));
            EZ_HIP(hipEventElapsedTime(&ms, e0, e1));
        }
        EZ_HIP(hipFree(buf));
        *out = ms * 1e3;
        return EZKL_OK;
    }
    if (!strcmp(which, "coopcheck")) {          // result: a bit mask of what failed (0 = all good); details on stderr
        const Bases* dummy = nullptr; (void)dummy;
        g1a_t* pts = nullptr;
        uint32_t* o = nullptr;
        EZ_HIP(hipMalloc(&pts, 512 * sizeof(g1a_t)));
        EZ_HIP(hipMalloc(&o, 64 * (160 + 128 + 72 + 36) * 4));
        EZ_HIP(hipMemsetAsync(o, 0, 64 * (160 + 128 + 72 + 36) * 4, st));
        {   // 128 points k G in the table's form (canonical x 2^261, y 2^261): built on the host
            std::vector<g1a_t> h(512);
            h64::aff g;
            memset(&g, 0, sizeof g);
            h64::fe one; { uint32_t w[8]; for (int q = 0; q < 8; q++) w[q] = Fq::one().v[q]; one = h64::from32(w); }
            g.x = one; g.y = h64::add(one, one);
            h64::xyzz acc = h64::from_affine(g), G = acc;
            h64::fe k261; { uint32_t w[8]; for (int q = 0; q < 8; q++) w[q] = Fq29C::R261_32[q]; k261 = h64::from32(w); }   // canonical 2^261 mod p
            for (int i = 0; i < 512; i++) {
                h64::aff a = h64::to_affine(acc);                    // Montgomery (2^256) form of the affine coordinates
                // table form = canonical value of x * 2^261: mont_mul(x_mont, k261_canonical) = x * 2^261 mod p as a canonical integer
                h64::fe tx = h64::mul(a.x, k261), ty = h64::mul(a.y, k261);
                memcpy(&h[i].x, &tx, 32); memcpy(&h[i].y, &ty, 32);
                acc = h64::add(acc, G);
                acc = h64::add(acc, G);                              // odd multiples apart so that no two inputs coincide
                acc = h64::add(acc, G);
            }
            EZ_HIP(hipMemcpy(pts, h.data(), 512 * sizeof(g1a_t), hipMemcpyHostToDevice));
        }
        hipLaunchKernelGGL(ub_coopcheck_kernel, dim3(1), dim3(64), 0, st, pts, o);
        EZ_HIP(hipStreamSynchronize(st));
        std::vector<uint32_t> h(64 * (160 + 128 + 72 + 36));
        EZ_HIP(hipMemcpy(h.data(), o, h.size() * 4, hipMemcpyDeviceToHost));
        uint32_t mask = 0;
        {
            uint32_t* o2 = nullptr;
            EZ_HIP(hipMalloc(&o2, 256 * 216 * 4));
            std::vector<uint32_t> h2(256 * 216);
            for (int ident = 0; ident < 2; ident++) {
                hipLaunchKernelGGL(ub_coopcheck2_kernel, dim3(1), dim3(256), 0, st, pts, o2, ident);
                EZ_HIP(hipStreamSynchronize(st));
                EZ_HIP(hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost));
                int bad[3] = {0, 0, 0};
                for (uint32_t t = 0; t < 256; t++)
                    for (int f = 0; f < 3; f++) {
                        const h64::aff p0 = h64::to_affine(h64::from_limbs29_point(h2.data() + (size_t)t * 216 + 72 * f));
                        const h64::aff p1 = h64::to_affine(h64::from_limbs29_point(h2.data() + (size_t)t * 216 + 72 * f + 36));
                        if (memcmp(&p0, &p1, 64)) { if (!bad[f]) fprintf(stderr, "[coopcheck] identities=%d form %d (0: width 32, 1: width 8, 2: block): thread %u differs\n", ident, f, t); bad[f]++; }
                    }
                fprintf(stderr, "[coopcheck] identities=%d: width 32: %d bad, width 8: %d bad, block of 256: %d bad\n", ident, bad[0], bad[1], bad[2]);
                if (bad[0] || bad[1] || bad[2]) mask |= 16u << ident;
            }
            EZ_HIP(hipFree(o2));
        }
        EZ_HIP(hipFree(pts)); EZ_HIP(hipFree(o));
        for (uint32_t l = 0; l < 64; l++) {
            const uint32_t* r = h.data() + (size_t)l * 160;
            for (uint32_t sidx = 0; sidx < 4; sidx++)
                if (r[sidx] != (l & ~3u) + sidx) { if (!(mask & 1)) fprintf(stderr, "[coopcheck] bcast<%u> lane %u got %u\n", sidx, l, r[sidx]); mask |= 1; }
            if (r[4]) { if (!(mask & 2)) fprintf(stderr, "[coopcheck] lane %u: special flag set\n", l); mask |= 2; }
            const h64::aff p0 = h64::to_affine(h64::from_limbs29_point(r + 8)), p1 = h64::to_affine(h64::from_limbs29_point(r + 44));
            if (memcmp(&p0, &p1, 64)) {
                if (!(mask & 4)) {
                    fprintf(stderr, "[coopcheck] lane %u: add_quad != add (x %s, y %s)\n", l, memcmp(&p0.x, &p1.x, 32) ? "differs" : "same", memcmp(&p0.y, &p1.y, 32) ? "differs" : "same");
                    for (int q = 0; q < 4; q++) {
                        const h64::fe c0 = h64::from_limbs29(r + 8 + 9 * q), c1 = h64::from_limbs29(r + 44 + 9 * q);
                        fprintf(stderr, "[coopcheck]   coordinate %d: %s\n", q, memcmp(&c0, &c1, 32) ? "differs" : "same residue");
                    }
                }
                mask |= 4;
            }
            const h64::aff t0 = h64::to_affine(h64::from_limbs29_point(r + 80)), t1 = h64::to_affine(h64::from_limbs29_point(r + 116));
            if (memcmp(&t0, &t1, 64)) { if (!(mask & 8)) fprintf(stderr, "[coopcheck] lane %u: cooperative butterfly sum != plain\n", l); mask |= 8; }
        }
        {   // the broadcast-subtract experiment: which forms agree with the host
            int bad[3] = {0, 0, 0};
            static const uint32_t SUBC1[9] = {0x21f3f51cu, 0x241182dau, 0x31ca8d3bu, 0x2b548b42u, 0x361765dfu, 0x2b6d0301u, 0x229b8503u, 0x397098cfu, 0x00c19138u};
            for (uint32_t l = 0; l < 64; l++) {
                const uint32_t* e = h.data() + 64 * 360 + (size_t)l * 36;
                const uint32_t* v2 = h.data() + 64 * 360 + (size_t)((l & ~3u) | 2u) * 36 + 27;
                const uint32_t* v3 = h.data() + 64 * 360 + (size_t)((l & ~3u) | 3u) * 36 + 27;
                for (int f = 0; f < 3; f++)
                    for (int i = 0; i < 9; i++)
                        if (e[9 * f + i] != v2[i] + (SUBC1[i] - v3[i])) { if (!bad[f]) fprintf(stderr, "[coopcheck] form %d lane %u limb %d: got %08x want %08x\n", f, l, i, e[9 * f + i], v2[i] + (SUBC1[i] - v3[i])); bad[f]++; }
            }
            fprintf(stderr, "[coopcheck] broadcast-subtract: combined %d bad, kept apart %d bad, shfl %d bad\n", bad[0], bad[1], bad[2]);
        }
        if (mask & 4) {
            const uint32_t* r0 = h.data();
            fprintf(stderr, "[coopcheck] plain.y:"); for (int i = 0; i < 9; i++) fprintf(stderr, " %08x", r0[8 + 9 + i]); fprintf(stderr, "\n");
            fprintf(stderr, "[coopcheck] coop.y: "); for (int i = 0; i < 9; i++) fprintf(stderr, " %08x", r0[44 + 9 + i]); fprintf(stderr, "\n");
            for (uint32_t q = 0; q < 4; q++) { fprintf(stderr, "[coopcheck] lane %u coop.y:", q); for (int i = 0; i < 9; i++) fprintf(stderr, " %08x", h[(size_t)q * 160 + 44 + 9 + i]); fprintf(stderr, "\n"); }
            for (int k = 0; k < 14; k++) for (uint32_t q = 0; q < 4; q++) { fprintf(stderr, "[coopcheck] t%d[q%u]: ", k, q); for (int i = 0; i < 9; i++) fprintf(stderr, " %08x", h[64 * 160 + q * 128 + 9 * k + i]); fprintf(stderr, "\n"); }
            for (uint32_t q = 0; q < 4; q++) { fprintf(stderr, "[coopcheck] m4[q%u]: ", q); for (int i = 0; i < 9; i++) fprintf(stderr, " %08x", h[64 * 160 + q * 128 + 9 * 13 + i]); fprintf(stderr, "\n"); }
        }
        if (mask & 4) {      // replay the levels of the first quad on the host and name the first intermediate that is wrong
            static const char* names[14] = {"m1", "u1", "s1", "p", "r", "m2", "pp", "rr", "m3", "ppp", "qq", "x3", "d", "m4"};
            const uint32_t* ab = h.data() + 64 * 160 + 64 * 128;
            const h64::xyzz A = h64::from_limbs29_point(ab), Bp = h64::from_limbs29_point(ab + 36);
            const h64::fe U1 = h64::mul(A.x, Bp.zz), U2 = h64::mul(Bp.x, A.zz), S1 = h64::mul(A.y, Bp.zzz), S2 = h64::mul(Bp.y, A.zzz);
            const h64::fe P = h64::sub(U2, U1), Rr = h64::sub(S2, S1), PP = h64::sqr(P), RR = h64::sqr(Rr), ZZ = h64::mul(A.zz, Bp.zz), ZZZ = h64::mul(A.zzz, Bp.zzz);
            const h64::fe PPP = h64::mul(P, PP), Q = h64::mul(U1, PP), X3 = h64::sub(h64::sub(RR, PPP), h64::dbl(Q)), D = h64::sub(Q, X3);
            const h64::fe ZZ3 = h64::mul(ZZ, PP), ZZZ3 = h64::mul(ZZZ, PPP), RD = h64::mul(Rr, D), E = h64::mul(S1, PPP);
            // expected value of intermediate k in lane q (nullptr: an idle lane, anything goes)
            for (uint32_t q = 0; q < 4; q++) {
                const uint32_t* dq = h.data() + 64 * 160 + (size_t)q * 128;
                const h64::fe m1e[4] = {U1, U2, S1, S2}, m2e[4] = {ZZ, ZZZ, PP, RR}, m3e[4] = {ZZ3, PPP, Q, Q}, m4e[4] = {ZZZ3, ZZZ3, RD, E};
                const h64::fe* want[14] = {&m1e[q], &U1, &S1, &P, &Rr, &m2e[q], &PP, &RR, q == 3 ? nullptr : &m3e[q], &PPP, &Q, &X3, &D, q == 0 ? nullptr : &m4e[q]};
                for (int k = 0; k < 14; k++) {
                    if (!want[k]) continue;
                    const h64::fe got = h64::from_limbs29(dq + 9 * k);
                    if (memcmp(&got, want[k], 32)) fprintf(stderr, "[coopcheck] lane %u: %s is wrong (limbs %08x %08x .. %08x)\n", q, names[k], dq[9 * k], dq[9 * k + 1], dq[9 * k + 8]);
                }
            }
        }
        fprintf(stderr, "[coopcheck] mask %u\n", mask);
        *out = (double)mask;
        return EZKL_OK;
    }
    if (!strcmp(which, "gather64")) {
        const size_t recs = (size_t)1 << 24;                 // 1 GiB table, far beyond L2 + MALL
        void *tab = nullptr, *o = nullptr;
        EZ_HIP(hipMalloc(&tab, recs * 64));
        EZ_HIP(hipMalloc(&o, nthreads * 16));
        EZ_HIP(hipMemsetAsync(tab, 3, recs * 64, st));
        const int iters = 32;
        for (int rep = 0; rep < 2; rep++) {
            EZ_HIP(hipEventRecord(e0, st));
            hipLaunchKernelGGL(ub_gather_kernel, dim3(blocks), dim3(threads), 0, st, (const uint4*)tab, (uint32_t)(recs - 1), (uint4*)o, iters);
            EZ_HIP(hipEventRecord(e1, st));
            EZ_HIP(hipStreamSynchronize(st));
            EZ_HIP(hipEventElapsedTime(&ms, e0, e1));
        }
        EZ_HIP(hipFree(tab));
        EZ_HIP(hipFree(o));
        *out = 64.0 * iters * (double)nthreads / (ms * 1e-3);     // gathered bytes per second
        return EZKL_OK;
    }
    if (!strcmp(which, "copy")) {
        const size_t bytes = (size_t)1 << 30;
        void *a = nullptr, *b = nullptr;
        EZ_HIP(hipMalloc(&a, bytes));
        EZ_HIP(hipMalloc(&b, bytes));
        EZ_HIP(hipMemsetAsync(a, 1, bytes, st));
        for (int rep = 0; rep < 3; rep++) {
            EZ_HIP(hipEventRecord(e0, st));
            hipLaunchKernelGGL(ub_copy_kernel, dim3(c->num_cus * 8), dim3(256), 0, st, (const uint4*)a, (uint4*)b, bytes / 16);
            EZ_HIP(hipEventRecord(e1, st));
            EZ_HIP(hipStreamSynchronize(st));
            EZ_HIP(hipEventElapsedTime(&ms, e0, e1));
        }
        EZ_HIP(hipFree(a));
        EZ_HIP(hipFree(b));
        *out = 2.0 * (double)bytes / (ms * 1e-3);    // read + write bytes per second
        return EZKL_OK;
    }
    return EZKL_ERR_INVALID;
}

}  // namespace ezkl
