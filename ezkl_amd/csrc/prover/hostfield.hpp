// Host-side BN254 field arithmetic for the prover's scalar glue (challenges, evaluation points, interpolation, the
// transcript's canonical encodings).  O(1) work per proof step: every O(n) operation runs on the GPU through the C ABI.
// Values are 4 x u64 little-endian limbs in Montgomery form (R = 2^256), the representation of halo2curves and of
// every buffer that crosses include/ezkl_hip.h.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>

namespace ezkl_prover {

using U256 = std::array<uint64_t, 4>;
typedef unsigned __int128 u128;

struct Modulus {
    U256 p, one, r2;
    uint64_t inv;      // -p^-1 mod 2^64
};
constexpr Modulus FR = {{0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
                        {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full},
                        {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull},
                        0xc2e1f593efffffffull};
constexpr Modulus FQ = {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull},
                        {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full},
                        {0xf32cfc5b538afa89ull, 0xb5e71911d44501fbull, 0x47ab1eff0a417ff6ull, 0x06d89f71cab8351full},
                        0x87d20782e4866389ull};

inline int cmp(const U256& a, const U256& b) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return -1;
    }
    return 0;
}
inline uint64_t add_raw(U256& r, const U256& a, const U256& b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
inline uint64_t sub_raw(U256& r, const U256& a, const U256& b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - br;
        r[i] = (uint64_t)d;
        br = (uint64_t)(d >> 64) & 1;
    }
    return br;
}
inline U256 mont_mul(const U256& a, const U256& b, const Modulus& M) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * M.inv;
        c = (u128)m * M.p[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * M.p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    U256 r = {t[0], t[1], t[2], t[3]};
    if (t[4] || cmp(r, M.p) >= 0) sub_raw(r, r, M.p);
    return r;
}

// An Fr element in Montgomery form with value semantics
struct Fe {
    U256 v{};
    static Fe zero() { return Fe{}; }
    static Fe one() { return Fe{FR.one}; }
    static Fe from_canonical(const U256& c) { return Fe{mont_mul(c, FR.r2, FR)}; }
    static Fe from_u64(uint64_t x) { return from_canonical(U256{x, 0, 0, 0}); }
    U256 canonical() const { return mont_mul(v, U256{1, 0, 0, 0}, FR); }
    bool operator==(const Fe& o) const { return v == o.v; }
    bool operator!=(const Fe& o) const { return v != o.v; }
    bool is_zero() const { return v == U256{0, 0, 0, 0}; }
    Fe operator*(const Fe& o) const { return Fe{mont_mul(v, o.v, FR)}; }
    Fe operator+(const Fe& o) const {
        Fe r;
        uint64_t c = add_raw(r.v, v, o.v);
        if (c || cmp(r.v, FR.p) >= 0) sub_raw(r.v, r.v, FR.p);
        return r;
    }
    Fe operator-(const Fe& o) const {
        Fe r;
        if (sub_raw(r.v, v, o.v)) add_raw(r.v, r.v, FR.p);
        return r;
    }
    Fe operator-() const { return zero() - *this; }
    Fe pow(const U256& e) const {
        int top = 255;                                   // squarings stop at the exponent's highest set bit
        while (top >= 0 && !((e[top >> 6] >> (top & 63)) & 1)) top--;
        Fe acc = one(), b = *this;
        for (int i = 0; i <= top; i++) {
            if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * b;
            if (i < top) b = b * b;
        }
        return acc;
    }
    Fe pow(uint64_t e) const { return pow(U256{e, 0, 0, 0}); }
    Fe inv() const {
        U256 e = FR.p;
        e[0] -= 2;
        return pow(e);
    }
};
// canonical 2^28-th root of unity ROOT = 7^((r-1)/2^28) and DELTA = 7^(2^28), Montgomery form
constexpr U256 FR_ROOT = {0x9632c7c5b639feb8ull, 0x985ce3400d0ff299ull, 0xb2dd880001b0ecd8ull, 0x1d69070d6d98ce29ull};
constexpr U256 FR_DELTA = {0x9a0c322befd78855ull, 0x46e82d14249b563cull, 0x5983a663e0b0b7a7ull, 0x22ab452baaa111adull};
// ZETA: the primitive cube root of unity halo2 uses as the generator of the extended coset (EvaluationDomain::g_coset), Montgomery form
constexpr U256 FR_ZETA = {0x0363f29955fcd653ull, 0x73e7950b5fc1e200ull, 0xc5fce83e576d9d24ull, 0x059c805da1c3a4d4ull};
inline Fe omega(uint32_t k) {
    Fe w{FR_ROOT};
    for (uint32_t i = k; i < 28; i++) w = w * w;
    return w;
}

// 32-byte big-endian encodings of the EVM transcript
inline void to_be32(const U256& c, uint8_t out[32]) {
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) out[31 - (8 * i + b)] = (uint8_t)(c[i] >> (8 * b));
}
inline U256 from_be32(const uint8_t in[32]) {
    U256 c{};
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) c[i] |= (uint64_t)in[31 - (8 * i + b)] << (8 * b);
    return c;
}
// a 256-bit integer reduced mod r (2^256 < 6r)
inline U256 reduce_fr(U256 c) {
    while (cmp(c, FR.p) >= 0) sub_raw(c, c, FR.p);
    return c;
}
// affine point (Montgomery Fq, 64 B) -> canonical (x, y)
struct G1 {
    U256 x{}, y{};          // Montgomery limbs, (0, 0) = identity
    void canonical(U256& cx, U256& cy) const {
        cx = mont_mul(x, U256{1, 0, 0, 0}, FQ);
        cy = mont_mul(y, U256{1, 0, 0, 0}, FQ);
    }
};

}  // namespace ezkl_prover
