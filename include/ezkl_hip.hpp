// ezkl_hip.hpp -- C++ host-side mirror of the halo2 interfaces on the `ezkl prove` hot path, over the C ABI of
// include/ezkl_hip.h.  The reference's host code is Rust (absent from this image); this header is what the Rust glue
// of INTEGRATION.md does, written in C++ with the reference's own names so that call sites and tests read the same:
//
//   ParamsKZG::{read, commit, commit_lagrange}              halo2_proofs::poly::kzg::commitment::ParamsKZG
//                                                           (in-tree call site /root/reference/src/circuit/modules/polycommit.rs:71,
//                                                            loader /root/reference/src/pfsys/srs.rs:40-47)
//   EvaluationDomain::{new, lagrange_to_coeff, coeff_to_lagrange, coeff_to_extended, extended_to_coeff,
//                      divide_by_vanishing_poly}            halo2_proofs::poly::EvaluationDomain (polycommit.rs:52)
//   GraphEvaluator::{add_constant, add_rotation, add_calculation, evaluate_h}
//                                                           halo2_proofs::plonk::evaluation::GraphEvaluator, with
//                                                           ValueSource / Calculation carrying halo2's variant names
//   polycommit_commit                                       PolyCommitChip::commit (polycommit.rs:46-81)
//
// Header-only, C++17, no dependency beyond libezkl_hip.so.  The C ABI never throws; this convenience layer turns a
// non-zero status into ezkl_hip::Error (the analogue of the `unwrap()` in the halo2 fork).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "ezkl_hip.h"

namespace ezkl_hip {

using Fr = std::array<uint64_t, 4>;        // Montgomery, little-endian limbs: the bytes halo2curves holds
using G1Affine = std::array<uint64_t, 8>;  // x || y, identity = all zero

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what + ": " + ezkl_hip_strerror(c)), code(c) {}
};
inline void check(int rc, const char* what) {
    if (rc != EZKL_OK) throw Error(rc, what);
}

// ---- minimal host Fr arithmetic (domain constants only: omega, its inverse, n^-1) ----
// the runtime gate: ENABLE_HIP_GPU set and k > HIP_SMALL_K (default 8) -- what ENABLE_ICICLE_GPU / ICICLE_SMALL_K decide in the reference
inline bool enabled(uint32_t k) { return ezkl_hip_enabled(k) != 0; }

namespace fr {
typedef unsigned __int128 u128;
constexpr Fr MOD = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
constexpr Fr ONE = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};      // 2^256 mod r
constexpr Fr R2 = {0x1bb8e645ae216da7ull, 0x53fe3ab1e35c59e3ull, 0x8c49833d53bb8085ull, 0x0216d0b17f4e44a5ull};       // 2^512 mod r
constexpr Fr ROOT = {0xd34f1ed960c37c9cull, 0x3215cf6dd39329c8ull, 0x98865ea93dd31f74ull, 0x03ddb9f5166d18b7ull};    // canonical 2^28-th root
constexpr uint64_t INV = 0xc2e1f593efffffffull;                                                                        // -r^-1 mod 2^64
inline bool geq(const Fr& a, const Fr& b) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > b[i]) return true;
        if (a[i] < b[i]) return false;
    }
    return true;
}
inline Fr mul(const Fr& a, const Fr& b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * INV;
        c = (u128)m * MOD[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * MOD[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fr r = {t[0], t[1], t[2], t[3]};
    if (t[4] || geq(r, MOD)) {
        uint64_t br = 0;
        for (int i = 0; i < 4; i++) { u128 d = (u128)r[i] - MOD[i] - br; r[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    }
    return r;
}
inline Fr to_mont(const Fr& canonical) { return mul(canonical, R2); }
inline Fr pow(Fr base, const Fr& e) {
    Fr acc = ONE;
    for (int i = 0; i < 256; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base);
        base = mul(base, base);
    }
    return acc;
}
inline Fr inv(const Fr& a) {
    Fr e = MOD;
    e[0] -= 2;                       // r - 2 (low limb of r is ...0001, no borrow)
    return pow(a, e);
}
inline Fr from_u64(uint64_t x) { return to_mont(Fr{x, 0, 0, 0}); }
}  // namespace fr

// ---- resident column ----
class DeviceColumn {
  public:
    DeviceColumn() = default;
    explicit DeviceColumn(size_t n) : n_(n) { check(ezkl_hip_malloc(&p_, n * 32), "ezkl_hip_malloc"); }
    explicit DeviceColumn(const std::vector<Fr>& v) : DeviceColumn(v.size()) {
        check(ezkl_hip_memcpy_h2d(p_, v.data(), v.size() * 32), "ezkl_hip_memcpy_h2d");
    }
    DeviceColumn(const DeviceColumn&) = delete;
    DeviceColumn& operator=(const DeviceColumn&) = delete;
    DeviceColumn(DeviceColumn&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DeviceColumn& operator=(DeviceColumn&& o) noexcept {
        if (this != &o) { reset(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    ~DeviceColumn() { reset(); }
    std::vector<Fr> to_host() const {
        std::vector<Fr> v(n_);
        check(ezkl_hip_memcpy_d2h(v.data(), p_, n_ * 32), "ezkl_hip_memcpy_d2h");
        return v;
    }
    void* ptr() const { return p_; }
    size_t len() const { return n_; }

  private:
    void reset() {
        if (p_) ezkl_hip_free(p_);
        p_ = nullptr;
    }
    void* p_ = nullptr;
    size_t n_ = 0;
};

// ---- ParamsKZG ----
class ParamsKZG {
  public:
    // raw-bytes SRS: u32 LE k | 2^k G1 g | 2^k G1 g_lagrange | G2 g2 | G2 s_g2
    static ParamsKZG read(const uint8_t* buf, size_t len) {
        if (len < 4) throw Error(EZKL_ERR_INVALID, "ParamsKZG::read");
        uint32_t k;
        std::memcpy(&k, buf, 4);
        const size_t n = (size_t)1 << k;
        if (len != 4 + 128 * n + 256) throw Error(EZKL_ERR_INVALID, "ParamsKZG::read: length");
        ParamsKZG p;
        p.k_ = k;
        check(ezkl_hip_bases_upload(buf + 4, n, &p.g_), "ezkl_hip_bases_upload(g)");
        check(ezkl_hip_bases_upload(buf + 4 + 64 * n, n, &p.gl_), "ezkl_hip_bases_upload(g_lagrange)");
        std::memcpy(p.g2_.data(), buf + 4 + 128 * n, 128);
        std::memcpy(p.s_g2_.data(), buf + 4 + 128 * n + 128, 128);
        return p;
    }
    ParamsKZG(ParamsKZG&& o) noexcept { *this = std::move(o); }
    ParamsKZG& operator=(ParamsKZG&& o) noexcept {
        release();
        k_ = o.k_; g_ = o.g_; gl_ = o.gl_; g2_ = o.g2_; s_g2_ = o.s_g2_;
        o.g_ = o.gl_ = nullptr;
        return *this;
    }
    ~ParamsKZG() { release(); }
    uint32_t k() const { return k_; }
    uint64_t n() const { return (uint64_t)1 << k_; }
    // commit_lagrange(&Polynomial<Fr, LagrangeCoeff>, Blind): the blind is ignored by KZG; the result is the canonical
    // affine point, i.e. what batch_normalize of the projective commitment gives (polycommit.rs:76)
    G1Affine commit_lagrange(const std::vector<Fr>& poly) const { return msm(gl_, poly); }
    G1Affine commit(const std::vector<Fr>& poly) const { return msm(g_, poly); }
    G1Affine commit_lagrange(const DeviceColumn& col) const { return msm_dev(gl_, col); }
    G1Affine commit(const DeviceColumn& col) const { return msm_dev(g_, col); }
    // one prover phase: all columns against the Lagrange basis, pipelined over the library's stream slots
    std::vector<G1Affine> commit_lagrange_batch(const std::vector<const DeviceColumn*>& cols) const {
        std::vector<const void*> ptrs;
        for (auto* c : cols) ptrs.push_back(c->ptr());
        std::vector<G1Affine> out(cols.size());
        if (!cols.empty())
            check(ezkl_hip_msm_g1_batch_dev(gl_, 0, ptrs.data(), ptrs.size(), cols[0]->len(), out.data(), nullptr), "ezkl_hip_msm_g1_batch_dev");
        return out;
    }

  private:
    ParamsKZG() = default;
    void release() {
        if (g_) ezkl_hip_bases_free(g_);
        if (gl_) ezkl_hip_bases_free(gl_);
        g_ = gl_ = nullptr;
    }
    static G1Affine msm(ezkl_bases_t b, const std::vector<Fr>& s) {
        G1Affine out{};
        check(ezkl_hip_msm_g1(b, s.data(), s.size(), out.data()), "ezkl_hip_msm_g1");
        return out;
    }
    static G1Affine msm_dev(ezkl_bases_t b, const DeviceColumn& c) {
        G1Affine out{};
        check(ezkl_hip_msm_g1_dev(b, 0, c.ptr(), c.len(), out.data(), nullptr), "ezkl_hip_msm_g1_dev");
        return out;
    }
    uint32_t k_ = 0;
    ezkl_bases_t g_ = nullptr, gl_ = nullptr;
    std::array<uint8_t, 128> g2_{}, s_g2_{};
};

// ---- EvaluationDomain ----
class EvaluationDomain {
  public:
    // EvaluationDomain::new(j, k): quotient_poly_degree = j - 1, extended_k = k + ceil(log2(j - 1))
    EvaluationDomain(uint32_t j, uint32_t k) : k_(k), extended_k_(k) {
        while (((uint64_t)1 << extended_k_) < ((uint64_t)1 << k) * (j - 1)) extended_k_++;
        omega_ = root_of_unity(k);
        omega_inv_ = fr::inv(omega_);
        extended_omega_ = root_of_unity(extended_k_);
        extended_omega_inv_ = fr::inv(extended_omega_);
    }
    uint32_t k() const { return k_; }
    uint32_t extended_k() const { return extended_k_; }
    const Fr& get_omega() const { return omega_; }
    const Fr& get_omega_inv() const { return omega_inv_; }
    const Fr& get_extended_omega() const { return extended_omega_; }
    void lagrange_to_coeff(std::vector<Fr>& a) const { check(ezkl_hip_ntt(a.data(), k_, omega_inv_.data(), 1), "ezkl_hip_ntt"); }
    void coeff_to_lagrange(std::vector<Fr>& a) const { check(ezkl_hip_ntt(a.data(), k_, omega_.data(), 0), "ezkl_hip_ntt"); }
    std::vector<Fr> coeff_to_extended(const std::vector<Fr>& a) const {
        std::vector<Fr> out((size_t)1 << extended_k_);
        const void* in = a.data();
        void* o = out.data();
        check(ezkl_hip_coset_ntt_batch(&in, &o, 1, k_, extended_k_, 0), "ezkl_hip_coset_ntt_batch");
        return out;
    }
    // returns all 2^extended_k coefficients; halo2 truncates to n * quotient_poly_degree
    std::vector<Fr> extended_to_coeff(const std::vector<Fr>& a) const {
        std::vector<Fr> out((size_t)1 << extended_k_);
        const void* in = a.data();
        void* o = out.data();
        check(ezkl_hip_coset_ntt_batch(&in, &o, 1, k_, extended_k_, 1), "ezkl_hip_coset_ntt_batch");
        return out;
    }
    void lagrange_to_coeff(DeviceColumn& c) const { check(ezkl_hip_ntt_dev(c.ptr(), k_, omega_inv_.data(), 1, 1, c.len(), nullptr), "ezkl_hip_ntt_dev"); }
    DeviceColumn coeff_to_extended(const DeviceColumn& c) const {
        DeviceColumn out((size_t)1 << extended_k_);
        check(ezkl_hip_coset_ntt_dev(c.ptr(), out.ptr(), 1, c.len(), out.len(), k_, extended_k_, 0, nullptr), "ezkl_hip_coset_ntt_dev");
        return out;
    }
    void divide_by_vanishing_poly(DeviceColumn& ext) const {
        check(ezkl_hip_divide_by_vanishing_dev(ext.ptr(), k_, extended_k_, nullptr), "ezkl_hip_divide_by_vanishing_dev");
    }
    static Fr root_of_unity(uint32_t k) {          // ROOT^(2^(28-k))
        Fr w = fr::to_mont(fr::ROOT);
        for (uint32_t i = k; i < 28; i++) w = fr::mul(w, w);
        return w;
    }

  private:
    uint32_t k_, extended_k_;
    Fr omega_, omega_inv_, extended_omega_, extended_omega_inv_;
};

// ---- GraphEvaluator ----
struct ValueSource {
    enum Kind { Constant, Intermediate, Fixed, Advice, Instance, Challenge, Beta, Gamma, Theta, Y, PreviousValue } kind;
    uint32_t a = 0, b = 0;     // (index) or (column index, rotation index)
    static ValueSource constant(uint32_t i) { return {Constant, i, 0}; }
    static ValueSource intermediate(uint32_t i) { return {Intermediate, i, 0}; }
    static ValueSource fixed(uint32_t col, uint32_t rot) { return {Fixed, col, rot}; }
    static ValueSource advice(uint32_t col, uint32_t rot) { return {Advice, col, rot}; }
    static ValueSource instance(uint32_t col, uint32_t rot) { return {Instance, col, rot}; }
    static ValueSource challenge(uint32_t i) { return {Challenge, i, 0}; }
    static ValueSource beta() { return {Beta, 0, 0}; }
    static ValueSource gamma() { return {Gamma, 0, 0}; }
    static ValueSource theta() { return {Theta, 0, 0}; }
    static ValueSource y() { return {Y, 0, 0}; }
    static ValueSource previous_value() { return {PreviousValue, 0, 0}; }
};
struct Calculation {
    enum Op { Add, Sub, Mul, Square, Double, Negate, Horner, Store } op;
    ValueSource s0{ValueSource::Constant, 0, 0}, s1{ValueSource::Constant, 0, 0};
    std::vector<ValueSource> parts;        // Horner(start = s0, parts, factor = s1)
};

class GraphEvaluator {
  public:
    GraphEvaluator(uint32_t n_fixed, uint32_t n_advice, uint32_t n_instance, uint32_t n_challenges)
        : nf_(n_fixed), na_(n_advice), ni_(n_instance), nc_(n_challenges) {}
    uint32_t add_constant(const Fr& c) {
        for (size_t i = 0; i < constants_.size(); i++)
            if (constants_[i] == c) return (uint32_t)i;
        constants_.push_back(c);
        return (uint32_t)constants_.size() - 1;
    }
    uint32_t add_rotation(int32_t rot) {
        for (size_t i = 0; i < rotations_.size(); i++)
            if (rotations_[i] == rot) return (uint32_t)i;
        rotations_.push_back(rot);
        return (uint32_t)rotations_.size() - 1;
    }
    // returns ValueSource::Intermediate(target), like GraphEvaluator::add_calculation
    ValueSource add_calculation(const Calculation& c) {
        const uint32_t target = num_intermediates_++;
        if (c.op == Calculation::Horner) {
            emit(EZKL_OP_STORE, target, c.s0, c.s0);
            for (const auto& p : c.parts) emit(EZKL_OP_HORNER_STEP, target, p, c.s1);
        } else {
            static const uint32_t map[] = {EZKL_OP_ADD, EZKL_OP_SUB, EZKL_OP_MUL, EZKL_OP_SQUARE, EZKL_OP_DOUBLE, EZKL_OP_NEGATE, 0, EZKL_OP_STORE};
            emit(map[c.op], target, c.s0, c.s1);
        }
        return ValueSource::intermediate(target);
    }
    // values[r] = program(row r) with PreviousValue = old values[r], over the extended domain.
    // Columns are resident cosets in the order fixed | advice | instance; challenges = user challenges | beta, gamma, theta, y
    void evaluate_h(uint32_t k, uint32_t extended_k, const std::vector<const DeviceColumn*>& fixed, const std::vector<const DeviceColumn*>& advice,
                    const std::vector<const DeviceColumn*>& instance, const std::vector<Fr>& challenges, const Fr& beta, const Fr& gamma,
                    const Fr& theta, const Fr& y, DeviceColumn& values) const {
        std::vector<const void*> cols;
        for (auto* c : fixed) cols.push_back(c->ptr());
        for (auto* c : advice) cols.push_back(c->ptr());
        for (auto* c : instance) cols.push_back(c->ptr());
        std::vector<Fr> ch(challenges);
        ch.resize(nc_);
        ch.push_back(beta); ch.push_back(gamma); ch.push_back(theta); ch.push_back(y);
        ezkl_program_t p{};
        p.code = code_.data();
        p.n_instr = (uint32_t)(code_.size() / 8);
        p.n_intermediates = num_intermediates_;
        p.constants = constants_.data();
        p.n_constants = (uint32_t)constants_.size();
        p.rotations = rotations_.data();
        p.n_rotations = (uint32_t)rotations_.size();
        p.columns = cols.data();
        p.n_columns = (uint32_t)cols.size();
        p.challenges = ch.data();
        p.n_challenges = (uint32_t)ch.size();
        p.k = k;
        p.ext_k = extended_k;
        check(ezkl_hip_eval_h_dev(&p, values.ptr(), nullptr), "ezkl_hip_eval_h_dev");
    }
    uint32_t num_intermediates() const { return num_intermediates_; }
    const std::vector<uint32_t>& code() const { return code_; }
    const std::vector<Fr>& constants() const { return constants_; }
    const std::vector<int32_t>& rotations() const { return rotations_; }

  private:
    void lower(const ValueSource& s, uint32_t out[3]) const {
        switch (s.kind) {
        case ValueSource::Constant: out[0] = EZKL_SRC_CONST; out[1] = s.a; out[2] = 0; break;
        case ValueSource::Intermediate: out[0] = EZKL_SRC_INTERMEDIATE; out[1] = s.a; out[2] = 0; break;
        case ValueSource::Fixed: out[0] = EZKL_SRC_COLUMN; out[1] = s.a; out[2] = s.b; break;
        case ValueSource::Advice: out[0] = EZKL_SRC_COLUMN; out[1] = nf_ + s.a; out[2] = s.b; break;
        case ValueSource::Instance: out[0] = EZKL_SRC_COLUMN; out[1] = nf_ + na_ + s.a; out[2] = s.b; break;
        case ValueSource::Challenge: out[0] = EZKL_SRC_CHALLENGE; out[1] = s.a; out[2] = 0; break;
        case ValueSource::Beta: out[0] = EZKL_SRC_CHALLENGE; out[1] = nc_; out[2] = 0; break;
        case ValueSource::Gamma: out[0] = EZKL_SRC_CHALLENGE; out[1] = nc_ + 1; out[2] = 0; break;
        case ValueSource::Theta: out[0] = EZKL_SRC_CHALLENGE; out[1] = nc_ + 2; out[2] = 0; break;
        case ValueSource::Y: out[0] = EZKL_SRC_CHALLENGE; out[1] = nc_ + 3; out[2] = 0; break;
        default: out[0] = EZKL_SRC_PREVIOUS; out[1] = 0; out[2] = 0; break;
        }
    }
    void emit(uint32_t op, uint32_t target, const ValueSource& s0, const ValueSource& s1) {
        uint32_t a[3], b[3];
        lower(s0, a);
        lower(s1, b);
        const uint32_t w[8] = {op, target, a[0], a[1], a[2], b[0], b[1], b[2]};
        code_.insert(code_.end(), w, w + 8);
    }
    uint32_t nf_, na_, ni_, nc_;
    std::vector<Fr> constants_;
    std::vector<int32_t> rotations_;
    std::vector<uint32_t> code_;
    uint32_t num_intermediates_ = 0;
};

// ---- halo2_proofs::arithmetic helpers of the opening step, on resident columns ----
// eval_polynomial(poly, point)
inline Fr eval_polynomial(const DeviceColumn& poly, const Fr& point) {
    Fr out;
    check(ezkl_hip_eval_poly_dev(poly.ptr(), poly.len(), point.data(), out.data(), nullptr), "ezkl_hip_eval_poly_dev");
    return out;
}
// kate_division(a, z): a(X) / (X - z), remainder dropped; in place (the top coefficient becomes 0)
inline void kate_division(DeviceColumn& a, const Fr& z) {
    check(ezkl_hip_kate_division_dev(a.ptr(), z.data(), a.ptr(), a.len(), nullptr), "ezkl_hip_kate_division_dev");
}

// ---- PolyCommitChip::commit (/root/reference/src/circuit/modules/polycommit.rs:46-81) ----
inline std::vector<G1Affine> polycommit_commit(const std::vector<Fr>& message, uint32_t num_unusable_rows, const ParamsKZG& params) {
    const uint64_t rows = params.n(), n = rows - num_unusable_rows;
    const size_t num_poly = message.size() / n + 1;
    std::vector<std::vector<Fr>> poly(num_poly, std::vector<Fr>(rows, Fr{0, 0, 0, 0}));
    for (auto& p : poly)
        for (uint32_t i = 0; i < num_unusable_rows; i++) p[n + i] = fr::ONE;      // Blind::default().0 == Fr::ONE
    for (size_t i = 0; i < message.size(); i++) poly[i / n][i % n] = message[i];
    std::vector<G1Affine> out;
    for (auto& p : poly) out.push_back(params.commit_lagrange(p));
    return out;
}

}  // namespace ezkl_hip
