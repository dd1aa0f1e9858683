#!/usr/bin/env python3
"""Generate ezkl_amd/csrc/montmul29_gen.hpp: Montgomery product in radix 2^29 (9 limbs, R' = 2^261) for BN254 Fq / Fr.

Why another radix: with 32-bit limbs every v_mad_u64_u32 partial product needs a v_addc_co_u32 to catch the carry out of
the 64-bit accumulator (128 + 128 instructions of the 261-slot product are carry handling).  With 29-bit limbs a column
of 9 + 9 products of < 2^58 fits a 64-bit accumulator, so the carries disappear: 162 mads + 9 (m = acc * p' mod 2^29)
+ 17 shifts + 17 masks.  Values stay lazily reduced in [0, 2p) (R' = 2^261 >> p), so there is no final subtraction.
Operand limbs may exceed 2^29 as long as 18 * max_a_limb * max_b_limb < 2^64."""
import os

FQ = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
FR = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
MASK = (1 << 29) - 1


def gen(name, mod, square=False, ilp2=False):
    """ilp2=True: two accumulator chains.  The product above is ONE dependent chain of ~190 instructions through v[16:17]; a dependent
    v_mad_u64_u32 issues 16 cycles after its producer, so a wave alone runs at a quarter of the issue rate and three waves per SIMD (the
    MSM's accumulate kernel) at three quarters.  The partial products a_i b_(k-i) of a column do not depend on the reduction: they are
    summed in a second accumulator v[18:19] (a fresh chain per column, emitted interleaved with the previous column's reduction chain)
    and enter the main chain with one 64-bit addition: 17 more instructions, critical path 141 instead of 190.
    square=True: r = a^2 / 2^261.  Column k of a square is sum_{i<j} (2 a_i) a_j + a_{k/2}^2: with the doubled limbs d_i = 2 a_i
    prepared once (8 shifts; limbs < 2^31) the 81 partial products become 45 -- 159 issue slots instead of 186."""
    p = [(mod >> (29 * i)) & MASK for i in range(9)]
    pinv = (-pow(mod, -1, 1 << 29)) % (1 << 29)
    A0, A1 = 16, 17                      # accumulator pair (even aligned): the only fixed VGPRs
    T0, T1 = 18, 19                      # ilp2: the a*b column sums
    # m_k lives in the register of output r_k: m_k is last read in column k + 8, r_k is written in column k + 9
    M = ["%%[r%d]" % i for i in range(9)]
    SP = [16 + i for i in range(9)]
    SINV, SMASK = 25, 26
    L = []
    for i in range(9):
        L.append("s_mov_b32 s%d, 0x%08x" % (SP[i], p[i]))
    L.append("s_mov_b32 s%d, 0x%08x" % (SINV, pinv))
    L.append("s_mov_b32 s%d, 0x%08x" % (SMASK, MASK))
    first = [True]

    def mac(s0, s1):
        src2 = "0" if first[0] else "v[%d:%d]" % (A0, A1)
        first[0] = False
        L.append("v_mad_u64_u32 v[%d:%d], vcc, %s, %s, %s" % (A0, A1, s0, s1, src2))

    if square:
        for i in range(8):
            L.append("v_lshlrev_b32 %%[d%d], 1, %%[a%d]" % (i, i))
    if ilp2:
        def tchain(k):
            out, fresh = [], True
            for i in range(max(0, k - 8), min(k, 8) + 1):
                if not square: ops = ("%%[a%d]" % i, "%%[b%d]" % (k - i))
                elif i < k - i: ops = ("%%[d%d]" % i, "%%[a%d]" % (k - i))
                elif i == k - i: ops = ("%%[a%d]" % i, "%%[a%d]" % i)
                else: continue
                out.append("v_mad_u64_u32 v[%d:%d], vcc, %s, %s, %s" % (T0, T1, ops[0], ops[1], "0" if fresh else "v[%d:%d]" % (T0, T1)))
                fresh = False
            return out
        def achain(k):
            out = []
            if k == 0: out.append("v_mov_b32 v%d, v%d" % (A0, T0)); out.append("v_mov_b32 v%d, v%d" % (A1, T1))
            else: out.append("v_lshl_add_u64 v[%d:%d], v[%d:%d], 0, v[%d:%d]" % (A0, A1, T0, T1, A0, A1))
            for i in (range(0, k) if k < 9 else range(k - 8, 9)):
                out.append("v_mad_u64_u32 v[%d:%d], vcc, %s, s%d, v[%d:%d]" % (A0, A1, M[i], SP[k - i], A0, A1))
            if k < 9:
                out.append("v_mul_lo_u32 %s, v%d, s%d" % (M[k], A0, SINV))
                out.append("v_and_b32 %s, s%d, %s" % (M[k], SMASK, M[k]))
                out.append("v_mad_u64_u32 v[%d:%d], vcc, %s, s%d, v[%d:%d]" % (A0, A1, M[k], SP[0], A0, A1))
            else:
                out.append("v_and_b32 %%[r%d], s%d, v%d" % (k - 9, SMASK, A0))
            out.append("v_lshrrev_b64 v[%d:%d], 29, v[%d:%d]" % (A0, A1, A0, A1))
            return out
        L += tchain(0)
        for k in range(17):
            a, t = achain(k), (tchain(k + 1) if k < 16 else [])
            # the first instruction of the reduction chain reads the finished column sum: it goes first, then the two chains alternate
            L.append(a[0])
            if k == 0: L.append(a[1]); a = a[1:]
            ai, ti = 1, 0
            while ai < len(a) or ti < len(t):
                if ti < len(t): L.append(t[ti]); ti += 1
                if ai < len(a): L.append(a[ai]); ai += 1
    for k in range(17 if not ilp2 else 0):
        for i in range(max(0, k - 8), min(k, 8) + 1):
            if not square:
                mac("%%[a%d]" % i, "%%[b%d]" % (k - i))
            elif i < k - i:
                mac("%%[d%d]" % i, "%%[a%d]" % (k - i))
            elif i == k - i:
                mac("%%[a%d]" % i, "%%[a%d]" % i)
        for i in (range(0, k) if k < 9 else range(k - 8, 9)):
            mac(M[i], "s%d" % SP[k - i])
        if k < 9:
            L.append("v_mul_lo_u32 %s, v%d, s%d" % (M[k], A0, SINV))
            L.append("v_and_b32 %s, s%d, %s" % (M[k], SMASK, M[k]))
            mac(M[k], "s%d" % SP[0])
        else:
            L.append("v_and_b32 %%[r%d], s%d, v%d" % (k - 9, SMASK, A0))
        L.append("v_lshrrev_b64 v[%d:%d], 29, v[%d:%d]" % (A0, A1, A0, A1))
    L.append("v_mov_b32 %%[r8], v%d" % A0)
    if square:
        o = ["// %s: r = a^2 / 2^261 mod p, limbs of 29 bits (a's limbs < 2^31); v16, v17 and s16..s26 clobbered" % name,
             "__device__ __forceinline__ void %s(const uint32_t (&a)[9], uint32_t (&r)[9]) {" % name,
             "    uint32_t d[8];"]
    else:
        o = ["// %s: r = a * b / 2^261 mod p, limbs of 29 bits; v16, v17 and s16..s26 clobbered" % name,
             "__device__ __forceinline__ void %s(const uint32_t (&a)[9], const uint32_t (&b)[9], uint32_t (&r)[9]) {" % name]
    body = "\n        ".join('"%s\\n\\t"' % l for l in L)
    o.append("    asm(" + body)
    o.append("        : " + ", ".join('[r%d] "=&v"(r[%d])' % (i, i) for i in range(9)) + (", " + ", ".join('[d%d] "=&v"(d[%d])' % (i, i) for i in range(8)) if square else ""))
    o.append("        : " + ", ".join('[a%d] "v"(a[%d])' % (i, i) for i in range(9)) + ("" if square else ", " + ", ".join('[b%d] "v"(b[%d])' % (i, i) for i in range(9))))
    o.append('        : "v16", "v17", ' + ('"v18", "v19", ' if ilp2 else "") + ", ".join('"s%d"' % s for s in range(16, 27)) + ', "vcc");')
    o.append("}")
    return "\n".join(o) + "\n"


def gen_mul2(name, mod):
    """r = (a b + c d) / 2^261 mod p with ONE reduction: column k sums a_i b_(k-i) + c_i d_(k-i) before the m p terms, 243 mads instead of
    2 x 162 (and no subtraction / carry pass between two products).  Limb bound: 9 (A B + C D) 2^58 + 9 2^58 + carry < 2^64, i.e.
    A B + C D < 6.1 in units of 2^29 per limb; value bound (alpha beta + gamma delta) / 169 + 1."""
    p = [(mod >> (29 * i)) & MASK for i in range(9)]
    pinv = (-pow(mod, -1, 1 << 29)) % (1 << 29)
    A0, A1 = 16, 17
    M = ["%%[r%d]" % i for i in range(9)]
    SP = [16 + i for i in range(9)]
    SINV, SMASK = 25, 26
    L = []
    for i in range(9):
        L.append("s_mov_b32 s%d, 0x%08x" % (SP[i], p[i]))
    L.append("s_mov_b32 s%d, 0x%08x" % (SINV, pinv))
    L.append("s_mov_b32 s%d, 0x%08x" % (SMASK, MASK))
    first = [True]

    def mac(s0, s1):
        src2 = "0" if first[0] else "v[%d:%d]" % (A0, A1)
        first[0] = False
        L.append("v_mad_u64_u32 v[%d:%d], vcc, %s, %s, %s" % (A0, A1, s0, s1, src2))

    for k in range(17):
        for i in range(max(0, k - 8), min(k, 8) + 1):
            mac("%%[a%d]" % i, "%%[b%d]" % (k - i))
            mac("%%[c%d]" % i, "%%[d%d]" % (k - i))
        for i in (range(0, k) if k < 9 else range(k - 8, 9)):
            mac(M[i], "s%d" % SP[k - i])
        if k < 9:
            L.append("v_mul_lo_u32 %s, v%d, s%d" % (M[k], A0, SINV))
            L.append("v_and_b32 %s, s%d, %s" % (M[k], SMASK, M[k]))
            mac(M[k], "s%d" % SP[0])
        else:
            L.append("v_and_b32 %%[r%d], s%d, v%d" % (k - 9, SMASK, A0))
        L.append("v_lshrrev_b64 v[%d:%d], 29, v[%d:%d]" % (A0, A1, A0, A1))
    L.append("v_mov_b32 %%[r8], v%d" % A0)
    o = ["// %s: r = (a * b + c * d) / 2^261 mod p, limbs of 29 bits, one reduction; v16, v17 and s16..s26 clobbered" % name,
         "__device__ __forceinline__ void %s(const uint32_t (&a)[9], const uint32_t (&b)[9], const uint32_t (&c)[9], const uint32_t (&d)[9], uint32_t (&r)[9]) {" % name]
    body = "\n        ".join('"%s\\n\\t"' % l for l in L)
    o.append("    asm(" + body)
    o.append("        : " + ", ".join('[r%d] "=&v"(r[%d])' % (i, i) for i in range(9)))
    o.append("        : " + ", ".join('[%s%d] "v"(%s[%d])' % (n, i, n, i) for n in "abcd" for i in range(9)))
    o.append('        : "v16", "v17", ' + ", ".join('"s%d"' % s for s in range(16, 27)) + ', "vcc");')
    o.append("}")
    return "\n".join(o) + "\n"


def consts(tag, mod):
    """limb tables for the lazy radix-2^29 arithmetic of field29.hpp"""
    def limbs29(x, n=9):
        return [(x >> (29 * i)) & MASK if i < n - 1 else (x >> (29 * i)) for i in range(n)]
    def arr(v):
        return "{" + ", ".join("0x%08xu" % x for x in v) + "}"
    o = ["struct %s29C {" % tag]
    o.append("    static constexpr uint32_t P[9] = %s;" % arr(limbs29(mod)))
    o.append("    static constexpr uint32_t ONE[9] = %s;            // 2^261 mod p" % arr(limbs29((1 << 261) % mod)))
    o.append("    static constexpr uint32_t P0INV = 0x%08xu;        // p^-1 mod 2^29" % pow(mod, -1, 1 << 29))
    # subtraction constants: K*p with 2^29 lent from every limb to the one below, so that C[i] - b[i] >= 0 limb-wise for any
    # normalized b < (K-1)*p
    rows = []
    for K in (2, 4, 8, 16, 32, 64, 128):
        k = limbs29(K * mod)
        c = [k[0] + (1 << 29)] + [k[i] + (1 << 29) - 1 for i in range(1, 8)] + [k[8] - 1]
        assert sum(ci << (29 * i) for i, ci in enumerate(c)) == K * mod and all(0 < ci < (1 << 31) for ci in c)
        assert c[8] >= ((K - 1) * mod) >> 232
        rows.append("        %s,   // %d p" % (arr(c), K))
    o.append("    static constexpr uint32_t SUBC[7][9] = {\n" + "\n".join(rows) + "\n    };")
    # conditional subtraction of K*p (K = 1, 2, 4, 8): x + (2^261 - K p) carries into bit 261 exactly when x >= K p
    rows = []
    for K in (1, 2, 4, 8):
        d = limbs29((1 << 261) - K * mod)
        assert all(x < (1 << 29) for x in d)
        rows.append("        %s,   // 2^261 - %d p" % (arr(d), K))
    o.append("    static constexpr uint32_t CSUB[4][9] = {\n" + "\n".join(rows) + "\n    };")
    k261 = (1 << 261) % mod
    o.append("    static constexpr uint32_t R261_32[8] = %s;    // 2^261 mod p as 8 x 32-bit limbs (table conversion R -> R')" %
             arr([(k261 >> (32 * i)) & 0xffffffff for i in range(8)]))
    o.append("};")
    return "\n".join(o) + "\n"


hdr = "// GENERATED by tools/gen_montmul29.py -- do not edit\n#pragma once\n#ifndef __HIPCC_RTC__            // hiprtc (the eval_h JIT) supplies the fixed-width types itself\n#include <stdint.h>\n#endif\nnamespace ezkl {\n"
text = (hdr + consts("Fq", FQ) + consts("Fr", FR) + gen("mont_mul29_fq", FQ) + gen("mont_mul29_fr", FR) + gen("mont_sqr29_fq", FQ, True) + gen("mont_sqr29_fr", FR, True) +
        gen_mul2("mont_mul2add29_fq", FQ) + gen("mont_mul29i_fq", FQ, ilp2=True) + gen("mont_sqr29i_fq", FQ, True, ilp2=True) + gen("mont_mul29i_fr", FR, ilp2=True) + "}  // namespace ezkl\n")
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ezkl_amd", "csrc", "montmul29_gen.hpp")
open(path, "w").write(text)
print("wrote", path, len(text.splitlines()), "lines")
