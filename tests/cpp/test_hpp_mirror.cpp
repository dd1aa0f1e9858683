// Parity test of the C++ host mirror (include/ezkl_hip.hpp) against the C oracle and the reference's golden fixtures.
// Reads like the reference's own module test (/root/reference/src/circuit/modules/polycommit.rs:46-81 commit(),
// tests around src/pfsys/srs.rs): read params, build a domain, commit, compare.
//
//   usage: test_hpp_mirror <kzg_k6.srs> <fixture_dir>      fixture_dir holds fixed_values.bin (C x 64 x 32 B),
//          fixed_polys.bin (C x 64 x 32 B), fixed_cosets.bin (C x 512 x 32 B), exported from pk_k6_subset.npz by the
//          pytest wrapper (tests/test_cpp_mirror.py).  Exit code 0 = all checks passed.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include "ezkl_hip.hpp"

using namespace ezkl_hip;

// ---- the oracle (liboracle.so; test infrastructure) ----
extern "C" {
void oracle_msm(const void* scalars, const void* bases, size_t n, void* out);
void oracle_omega(unsigned k, void* o);
void oracle_fr_mul(const void* a, const void* b, void* o);
void oracle_fr_inv(const void* a, void* o);
void oracle_lagrange_to_coeff(void* a, unsigned k);
void oracle_coeff_to_extended(const void* in, unsigned k, unsigned ext_k, void* out);
void oracle_extended_to_coeff(void* a, unsigned ext_k);
void oracle_divide_by_vanishing(void* a, unsigned k, unsigned ext_k);
void oracle_kate_div(void* a, size_t n, const void* z);
void oracle_eval_poly(const void* c, size_t n, const void* x, void* out);
struct oracle_program {
    const uint32_t* code; uint32_t n_instr; uint32_t n_intermediates;
    const void* constants; uint32_t n_constants;
    const int32_t* rotations; uint32_t n_rotations;
    const void* const* columns; uint32_t n_columns;
    const void* challenges; uint32_t n_challenges;
    uint32_t k, ext_k;
};
void oracle_eval_program(const oracle_program* p, void* out);
}

static int failures = 0;
#define EXPECT(cond)                                                          \
    do {                                                                      \
        if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

static std::vector<uint8_t> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { std::printf("cannot open %s\n", path.c_str()); std::exit(2); }
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static std::vector<std::vector<Fr>> columns(const std::vector<uint8_t>& raw, size_t rows) {
    std::vector<std::vector<Fr>> out(raw.size() / (rows * 32), std::vector<Fr>(rows));
    for (size_t c = 0; c < out.size(); c++) std::memcpy(out[c].data(), raw.data() + c * rows * 32, rows * 32);
    return out;
}
static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static Fr random_fr() {       // canonical value < 2^253 (< r), then into Montgomery form
    Fr c;
    for (auto& l : c) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; l = rng_state; }
    c[3] &= 0x1fffffffffffffffull;
    return fr::to_mont(c);
}

int main(int argc, char** argv) {
    if (argc < 3) { std::printf("usage: %s <srs> <fixture_dir>\n", argv[0]); return 2; }
    check(ezkl_hip_init(-1), "ezkl_hip_init");
    const auto srs = slurp(argv[1]);
    const std::string dir = argv[2];
    const uint32_t K = 6, EXT_K = 9;
    const size_t N = 64, NE = 512;

    // ---- ParamsKZG::read, error behaviour ----
    ParamsKZG params = ParamsKZG::read(srs.data(), srs.size());
    EXPECT(params.k() == K);
    try { ParamsKZG::read(srs.data(), srs.size() - 1); EXPECT(false); } catch (const Error& e) { EXPECT(e.code == EZKL_ERR_INVALID); }
    const uint8_t* g = srs.data() + 4;
    const uint8_t* gl = g + 64 * N;

    // sum of the Lagrange basis == g[0]  (the SRS fixture's own invariant)
    {
        std::vector<Fr> ones(N, fr::ONE);
        G1Affine c = params.commit_lagrange(ones);
        EXPECT(std::memcmp(c.data(), g, 64) == 0);
    }

    // ---- host Fr vs oracle ----
    {
        Fr a = random_fr(), b = random_fr(), o, inv, oi;
        oracle_fr_mul(a.data(), b.data(), o.data());
        EXPECT(fr::mul(a, b) == o);
        oracle_fr_inv(a.data(), oi.data());
        inv = fr::inv(a);
        EXPECT(inv == oi);
        EXPECT(fr::mul(a, inv) == fr::ONE);
        EXPECT(fr::from_u64(1) == fr::ONE);
    }

    // ---- EvaluationDomain::new(j, k) ----
    EvaluationDomain domain(9, K);                 // quotient degree 8 -> extended_k = 9 (the fixture's pk)
    EXPECT(domain.extended_k() == EXT_K);
    EXPECT(EvaluationDomain(3, K).extended_k() == K + 1);
    EXPECT(EvaluationDomain(5, K).extended_k() == K + 2);
    {
        Fr w;
        oracle_omega(K, w.data());
        EXPECT(domain.get_omega() == w);
        oracle_omega(EXT_K, w.data());
        EXPECT(domain.get_extended_omega() == w);
        EXPECT(fr::mul(domain.get_omega(), domain.get_omega_inv()) == fr::ONE);
    }

    // ---- golden pk columns: values -> polys -> cosets, commitments ----
    const auto values = columns(slurp(dir + "/fixed_values.bin"), N);
    const auto polys = columns(slurp(dir + "/fixed_polys.bin"), N);
    const auto cosets = columns(slurp(dir + "/fixed_cosets.bin"), NE);
    EXPECT(values.size() >= 3 && values.size() == polys.size() && polys.size() == cosets.size());
    std::vector<DeviceColumn> coset_dev;
    for (size_t c = 0; c < values.size(); c++) {
        std::vector<Fr> a = values[c];
        domain.lagrange_to_coeff(a);
        EXPECT(a == polys[c]);
        std::vector<Fr> back = a;
        domain.coeff_to_lagrange(back);
        EXPECT(back == values[c]);
        std::vector<Fr> ext = domain.coeff_to_extended(a);
        EXPECT(ext == cosets[c]);
        std::vector<Fr> coeffs = domain.extended_to_coeff(ext);
        bool ok = true;
        for (size_t i = 0; i < NE; i++) ok &= (coeffs[i] == (i < N ? a[i] : Fr{0, 0, 0, 0}));
        EXPECT(ok);
        // resident path
        DeviceColumn dv(values[c]);
        domain.lagrange_to_coeff(dv);
        EXPECT(dv.to_host() == polys[c]);
        DeviceColumn de = domain.coeff_to_extended(dv);
        EXPECT(de.to_host() == cosets[c]);
        coset_dev.push_back(std::move(de));
        // commit_lagrange(values) == commit(iNTT(values)) == oracle MSM
        G1Affine c1 = params.commit_lagrange(values[c]), c2 = params.commit(polys[c]), c3 = params.commit(dv), want;
        oracle_msm(values[c].data(), gl, N, want.data());
        EXPECT(c1 == want);
        EXPECT(c2 == want);
        EXPECT(c3 == want);
    }
    {   // one prover phase
        std::vector<DeviceColumn> dv;
        for (auto& v : values) dv.emplace_back(v);
        std::vector<const DeviceColumn*> ptrs;
        for (auto& d : dv) ptrs.push_back(&d);
        auto commits = params.commit_lagrange_batch(ptrs);
        for (size_t c = 0; c < values.size(); c++) EXPECT(commits[c] == params.commit_lagrange(values[c]));
    }

    // ---- polycommit_commit (polycommit.rs:46-81): message over 2 polynomials, unusable rows blinded with Fr::ONE ----
    {
        const uint32_t unusable = 6;
        std::vector<Fr> message(N - unusable + 10);
        for (auto& m : message) m = random_fr();
        auto commits = polycommit_commit(message, unusable, params);
        EXPECT(commits.size() == 2);
        for (size_t p = 0; p < 2; p++) {
            std::vector<Fr> poly(N, Fr{0, 0, 0, 0});
            for (size_t i = 0; i < N - unusable; i++)
                if (p * (N - unusable) + i < message.size()) poly[i] = message[p * (N - unusable) + i];
            for (size_t i = N - unusable; i < N; i++) poly[i] = fr::ONE;
            G1Affine want;
            oracle_msm(poly.data(), gl, N, want.data());
            EXPECT(commits[p] == want);
        }
    }

    // ---- GraphEvaluator: the shape halo2's Evaluator::new builds for custom gates ----
    //   gate0 = fixed0 * (advice0 * advice1(next) - advice2(prev));  gate1 = fixed1 * (advice0 + 5 - challenge0 * advice1)
    //   value = ((previous * y + gate0) * y + gate1), then a permutation-like term with beta / gamma / theta
    {
        GraphEvaluator ev(2, 3, 0, 1);
        const uint32_t cur = ev.add_rotation(0), next = ev.add_rotation(1), prev = ev.add_rotation(-1);
        EXPECT(ev.add_rotation(1) == next);
        const uint32_t five = ev.add_constant(fr::from_u64(5));
        EXPECT(ev.add_constant(fr::from_u64(5)) == five);
        auto ab = ev.add_calculation({Calculation::Mul, ValueSource::advice(0, cur), ValueSource::advice(1, next), {}});
        auto d0 = ev.add_calculation({Calculation::Sub, ab, ValueSource::advice(2, prev), {}});
        auto g0 = ev.add_calculation({Calculation::Mul, ValueSource::fixed(0, cur), d0, {}});
        auto a5 = ev.add_calculation({Calculation::Add, ValueSource::advice(0, cur), ValueSource::constant(five), {}});
        auto cb = ev.add_calculation({Calculation::Mul, ValueSource::challenge(0), ValueSource::advice(1, cur), {}});
        auto d1 = ev.add_calculation({Calculation::Sub, a5, cb, {}});
        auto g1 = ev.add_calculation({Calculation::Mul, ValueSource::fixed(1, cur), d1, {}});
        auto h = ev.add_calculation({Calculation::Horner, ValueSource::previous_value(), ValueSource::y(), {g0, g1}});
        auto sq = ev.add_calculation({Calculation::Square, h, h, {}});
        auto db = ev.add_calculation({Calculation::Double, sq, sq, {}});
        auto ng = ev.add_calculation({Calculation::Negate, db, db, {}});
        auto bt = ev.add_calculation({Calculation::Mul, ValueSource::beta(), ValueSource::advice(2, cur), {}});
        auto gm = ev.add_calculation({Calculation::Add, bt, ValueSource::gamma(), {}});
        auto th = ev.add_calculation({Calculation::Mul, gm, ValueSource::theta(), {}});
        auto st = ev.add_calculation({Calculation::Store, th, th, {}});
        ev.add_calculation({Calculation::Add, ng, st, {}});

        const Fr ch0 = random_fr(), beta = random_fr(), gamma = random_fr(), theta = random_fr(), y = random_fr();
        std::vector<Fr> prev_values(NE);
        for (auto& v : prev_values) v = random_fr();
        DeviceColumn out(prev_values);
        std::vector<const DeviceColumn*> fixed = {&coset_dev[0], &coset_dev[1]}, advice = {&coset_dev[2], &coset_dev[3 % coset_dev.size()], &coset_dev[4 % coset_dev.size()]};
        ev.evaluate_h(K, EXT_K, fixed, advice, {}, {ch0}, beta, gamma, theta, y, out);

        std::vector<const void*> cols = {cosets[0].data(), cosets[1].data(), cosets[2].data(), cosets[3 % cosets.size()].data(), cosets[4 % cosets.size()].data()};
        std::vector<Fr> chal = {ch0, beta, gamma, theta, y};
        oracle_program p{ev.code().data(), (uint32_t)(ev.code().size() / 8), ev.num_intermediates(), ev.constants().data(), (uint32_t)ev.constants().size(),
                         ev.rotations().data(), (uint32_t)ev.rotations().size(), cols.data(), (uint32_t)cols.size(), chal.data(), (uint32_t)chal.size(), K, EXT_K};
        std::vector<Fr> want = prev_values;
        oracle_eval_program(&p, want.data());
        EXPECT(out.to_host() == want);

        // h(X) tail: divide by the vanishing polynomial, back to coefficients
        domain.divide_by_vanishing_poly(out);
        oracle_divide_by_vanishing(want.data(), K, EXT_K);
        std::vector<Fr> quotient = out.to_host();
        EXPECT(quotient == want);
        std::vector<Fr> hc = domain.extended_to_coeff(quotient);
        oracle_extended_to_coeff(want.data(), EXT_K);
        EXPECT(hc == want);
    }

    // ---- the opening step's arithmetic: eval_polynomial, kate_division; and the runtime gate ----
    {
        std::vector<Fr> a(5000);
        for (auto& v : a) v = random_fr();
        const Fr z = random_fr();
        DeviceColumn col(a);
        Fr want_eval;
        oracle_eval_poly(a.data(), a.size(), z.data(), want_eval.data());
        EXPECT(eval_polynomial(col, z) == want_eval);
        kate_division(col, z);
        oracle_kate_div(a.data(), a.size(), z.data());
        EXPECT(col.to_host() == a);
        unsetenv("ENABLE_HIP_GPU");
        EXPECT(!enabled(20));
        setenv("ENABLE_HIP_GPU", "1", 1);
        EXPECT(enabled(20) && !enabled(8));
    }

    std::printf(failures ? "%d check(s) FAILED\n" : "all checks passed\n", failures);
    return failures ? 1 : 0;
}
