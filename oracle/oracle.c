/*
 * oracle.c -- CPU restatement of the BN254 / halo2-KZG primitives on the `ezkl prove` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.  The product (ezkl_amd/,
 * libezkl_hip.so) never links, loads or calls it.
 *
 * Where the algorithm comes from.  ezkl itself holds none of this arithmetic: the call
 * sites are /root/reference/src/pfsys/mod.rs:456-463 (create_proof), src/circuit/modules/polycommit.rs:52,71
 * (EvaluationDomain / commit_lagrange) and the arithmetic lives in un-vendored crates pinned by
 * /root/reference/Cargo.lock: halo2curves 0.7.0 @ b753a832 (bn256::{Fr,Fq,G1}, msm, fft) and
 * zkonduit/halo2 @ 01c88842 (EvaluationDomain, ParamsKZG, plonk::evaluation).  Neither is on disk, so
 * this file restates their published algorithms:
 *   - Montgomery arithmetic, R = 2^256, 4x64-bit limbs (halo2curves field macros)
 *   - G1 Jacobian add / mixed add / double for y^2 = x^3 + 3 (halo2curves new_curve_impl!)
 *   - msm: windowed Pippenger with signed (Booth) digits (halo2curves msm.rs)
 *   - best_fft: bit-reversal + iterative radix-2 DIT butterflies, natural order in/out (halo2curves fft.rs)
 *   - EvaluationDomain::{ifft, coeff_to_extended, extended_to_coeff, divide_by_vanishing_poly}
 *   - plonk::evaluation::GraphEvaluator::evaluate (straight-line program over rotated columns)
 * PINNING: value-level, against the reference's own fixtures (files under tests/golden, extracted from
 * /root/reference/tests/assets by tests/golden/make_golden.py): pk.key fixed_values -> fixed_polys ->
 * fixed_cosets (iNTT + coset NTT), SRS self-consistency (MSM/iNTT), vk.key sigma commitments against the
 * public k=1 SRS (MSM).  The GraphEvaluator restatement has no fixture in the reference: "parity unpinned"
 * for that function; it is cross-checked against oracle/pyref.py only.
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared; no -march=native: the .so travels to the GPU box)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "bn254_constants.h"
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } fe;
typedef struct { fe mod, one, r2; uint64_t inv; } fparams;

static const fparams FQ = { {BN64_FQ_MOD_INIT}, {BN64_FQ_R_INIT}, {BN64_FQ_R2_INIT}, BN64_FQ_INV };
static const fparams FR = { {BN64_FR_MOD_INIT}, {BN64_FR_R_INIT}, {BN64_FR_R2_INIT}, BN64_FR_INV };
static const fe FR_ZETA = {BN64_FR_ZETA_INIT}, FR_ZETA2 = {BN64_FR_ZETA2_INIT};
static const fe FQ_B3 = {BN64_FQ_B3_INIT};

/* ------------------------------------------------------------------ field ------------------- */
static inline int f_is_zero(const fe *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static inline int f_eq(const fe *a, const fe *b) { return memcmp(a, b, 32) == 0; }
static inline int f_geq(const fe *a, const fe *m) {
    for (int i = 3; i >= 0; i--) { if (a->v[i] > m->v[i]) return 1; if (a->v[i] < m->v[i]) return 0; }
    return 1;
}
static inline uint64_t sub4(fe *o, const fe *a, const fe *b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a->v[i] - b->v[i] - br; o->v[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    return br;
}
static inline uint64_t add4(fe *o, const fe *a, const fe *b) {
    uint64_t c = 0;
    for (int i = 0; i < 4; i++) { u128 s = (u128)a->v[i] + b->v[i] + c; o->v[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    return c;
}
static inline void f_add(fe *o, const fe *a, const fe *b, const fparams *F) {
    fe t; add4(&t, a, b);              /* < 2p < 2^255: no carry out */
    if (f_geq(&t, &F->mod)) sub4(&t, &t, &F->mod);
    *o = t;
}
static inline void f_sub(fe *o, const fe *a, const fe *b, const fparams *F) {
    fe t; if (sub4(&t, a, b)) add4(&t, &t, &F->mod);
    *o = t;
}
static inline void f_neg(fe *o, const fe *a, const fparams *F) {
    if (f_is_zero(a)) { *o = *a; return; }
    sub4(o, &F->mod, a);
}
static inline void f_dbl(fe *o, const fe *a, const fparams *F) { f_add(o, a, a, F); }
/* CIOS Montgomery product a*b*R^-1 mod p */
static inline void f_mul(fe *o, const fe *a, const fe *b, const fparams *F) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->v[j] * b->v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->inv;
        c = (u128)m * F->mod.v[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * F->mod.v[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || f_geq(&r, &F->mod)) sub4(&r, &r, &F->mod);
    *o = r;
}
static inline void f_sqr(fe *o, const fe *a, const fparams *F) { f_mul(o, a, a, F); }
static void f_pow(fe *o, const fe *a, const uint64_t e[4], const fparams *F) {
    fe acc = F->one, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e[i >> 6] >> (i & 63)) & 1) f_mul(&acc, &acc, &base, F);
        f_sqr(&base, &base, F);
    }
    *o = acc;
}
static void f_inv(fe *o, const fe *a, const fparams *F) {    /* a^(p-2); 0 -> 0 */
    fe e = F->mod; fe two = {{2, 0, 0, 0}}; sub4(&e, &e, &two);
    f_pow(o, a, e.v, F);
}
static inline void f_from_mont(fe *o, const fe *a, const fparams *F) { fe one = {{1, 0, 0, 0}}; f_mul(o, a, &one, F); }
static inline void f_to_mont(fe *o, const fe *a, const fparams *F) { f_mul(o, a, &F->r2, F); }
static void f_from_u64(fe *o, uint64_t x, const fparams *F) { fe t = {{x, 0, 0, 0}}; f_to_mont(o, &t, F); }

/* exported scalar helpers (used by tests to cross-check against pyref) */
void oracle_fr_mul(const fe *a, const fe *b, fe *o) { f_mul(o, a, b, &FR); }
void oracle_fq_mul(const fe *a, const fe *b, fe *o) { f_mul(o, a, b, &FQ); }
void oracle_fr_add(const fe *a, const fe *b, fe *o) { f_add(o, a, b, &FR); }
void oracle_fr_sub(const fe *a, const fe *b, fe *o) { f_sub(o, a, b, &FR); }
void oracle_fr_inv(const fe *a, fe *o) { f_inv(o, a, &FR); }
void oracle_fq_inv(const fe *a, fe *o) { f_inv(o, a, &FQ); }
void oracle_fr_pow(const fe *a, const uint64_t e[4], fe *o) { f_pow(o, a, e, &FR); }
void oracle_fr_from_mont(const fe *a, fe *o) { f_from_mont(o, a, &FR); }
void oracle_fr_to_mont(const fe *a, fe *o) { f_to_mont(o, a, &FR); }

/* ------------------------------------------------------------------ G1 ---------------------- */
typedef struct { fe x, y; } g1a;        /* affine, (0,0) = identity (raw-bytes convention) */
typedef struct { fe x, y, z; } g1j;     /* Jacobian, z = 0 identity */

static inline int g1a_is_id(const g1a *p) { return f_is_zero(&p->x) && f_is_zero(&p->y); }
static inline void g1j_set_id(g1j *p) { memset(p, 0, sizeof *p); p->y = FQ.one; }
static inline int g1j_is_id(const g1j *p) { return f_is_zero(&p->z); }

static void g1j_double(g1j *o, const g1j *p) {       /* dbl-2009-l, a = 0 */
    if (g1j_is_id(p)) { *o = *p; return; }
    fe a, b, c, d, e, f, t, x3, y3, z3;
    f_sqr(&a, &p->x, &FQ); f_sqr(&b, &p->y, &FQ); f_sqr(&c, &b, &FQ);
    f_add(&d, &p->x, &b, &FQ); f_sqr(&d, &d, &FQ); f_sub(&d, &d, &a, &FQ); f_sub(&d, &d, &c, &FQ); f_dbl(&d, &d, &FQ);
    f_dbl(&e, &a, &FQ); f_add(&e, &e, &a, &FQ);
    f_sqr(&f, &e, &FQ);
    f_mul(&z3, &p->z, &p->y, &FQ); f_dbl(&z3, &z3, &FQ);
    f_dbl(&t, &d, &FQ); f_sub(&x3, &f, &t, &FQ);
    f_dbl(&c, &c, &FQ); f_dbl(&c, &c, &FQ); f_dbl(&c, &c, &FQ);
    f_sub(&t, &d, &x3, &FQ); f_mul(&y3, &e, &t, &FQ); f_sub(&y3, &y3, &c, &FQ);
    o->x = x3; o->y = y3; o->z = z3;
}
static void g1j_add_mixed(g1j *o, const g1j *p, const g1a *q) {   /* madd-2007-bl with edge cases */
    if (g1a_is_id(q)) { *o = *p; return; }
    if (g1j_is_id(p)) { o->x = q->x; o->y = q->y; o->z = FQ.one; return; }
    fe z1z1, u2, s2, h, hh, i, j, r, v, t, x3, y3, z3;
    f_sqr(&z1z1, &p->z, &FQ); f_mul(&u2, &q->x, &z1z1, &FQ);
    f_mul(&s2, &q->y, &p->z, &FQ); f_mul(&s2, &s2, &z1z1, &FQ);
    if (f_eq(&u2, &p->x)) {
        if (f_eq(&s2, &p->y)) { g1j_double(o, p); return; }
        g1j_set_id(o); return;
    }
    f_sub(&h, &u2, &p->x, &FQ); f_sqr(&hh, &h, &FQ);
    f_dbl(&i, &hh, &FQ); f_dbl(&i, &i, &FQ);
    f_mul(&j, &h, &i, &FQ);
    f_sub(&r, &s2, &p->y, &FQ); f_dbl(&r, &r, &FQ);
    f_mul(&v, &p->x, &i, &FQ);
    f_sqr(&x3, &r, &FQ); f_sub(&x3, &x3, &j, &FQ); f_sub(&x3, &x3, &v, &FQ); f_sub(&x3, &x3, &v, &FQ);
    f_sub(&t, &v, &x3, &FQ); f_mul(&y3, &r, &t, &FQ);
    f_mul(&t, &p->y, &j, &FQ); f_dbl(&t, &t, &FQ); f_sub(&y3, &y3, &t, &FQ);
    f_add(&z3, &p->z, &h, &FQ); f_sqr(&z3, &z3, &FQ); f_sub(&z3, &z3, &z1z1, &FQ); f_sub(&z3, &z3, &hh, &FQ);
    o->x = x3; o->y = y3; o->z = z3;
}
static void g1j_add(g1j *o, const g1j *p, const g1j *q) {         /* add-2007-bl with edge cases */
    if (g1j_is_id(p)) { *o = *q; return; }
    if (g1j_is_id(q)) { *o = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t, x3, y3, z3;
    f_sqr(&z1z1, &p->z, &FQ); f_sqr(&z2z2, &q->z, &FQ);
    f_mul(&u1, &p->x, &z2z2, &FQ); f_mul(&u2, &q->x, &z1z1, &FQ);
    f_mul(&s1, &p->y, &q->z, &FQ); f_mul(&s1, &s1, &z2z2, &FQ);
    f_mul(&s2, &q->y, &p->z, &FQ); f_mul(&s2, &s2, &z1z1, &FQ);
    if (f_eq(&u1, &u2)) {
        if (f_eq(&s1, &s2)) { g1j_double(o, p); return; }
        g1j_set_id(o); return;
    }
    f_sub(&h, &u2, &u1, &FQ); f_dbl(&i, &h, &FQ); f_sqr(&i, &i, &FQ);
    f_mul(&j, &h, &i, &FQ);
    f_sub(&r, &s2, &s1, &FQ); f_dbl(&r, &r, &FQ);
    f_mul(&v, &u1, &i, &FQ);
    f_sqr(&x3, &r, &FQ); f_sub(&x3, &x3, &j, &FQ); f_sub(&x3, &x3, &v, &FQ); f_sub(&x3, &x3, &v, &FQ);
    f_sub(&t, &v, &x3, &FQ); f_mul(&y3, &r, &t, &FQ);
    f_mul(&t, &s1, &j, &FQ); f_dbl(&t, &t, &FQ); f_sub(&y3, &y3, &t, &FQ);
    f_add(&z3, &p->z, &q->z, &FQ); f_sqr(&z3, &z3, &FQ); f_sub(&z3, &z3, &z1z1, &FQ); f_sub(&z3, &z3, &z2z2, &FQ);
    f_mul(&z3, &z3, &h, &FQ);
    o->x = x3; o->y = y3; o->z = z3;
}
static void g1j_to_affine(g1a *o, const g1j *p) {
    if (g1j_is_id(p)) { memset(o, 0, sizeof *o); return; }
    fe zi, zi2, zi3;
    f_inv(&zi, &p->z, &FQ); f_sqr(&zi2, &zi, &FQ); f_mul(&zi3, &zi2, &zi, &FQ);
    f_mul(&o->x, &p->x, &zi2, &FQ); f_mul(&o->y, &p->y, &zi3, &FQ);
}
int oracle_g1_on_curve(const g1a *p) {
    if (g1a_is_id(p)) return 1;
    fe l, r; f_sqr(&l, &p->y, &FQ); f_sqr(&r, &p->x, &FQ); f_mul(&r, &r, &p->x, &FQ); f_add(&r, &r, &FQ_B3, &FQ);
    return f_eq(&l, &r);
}
void oracle_g1_add_affine(const g1a *p, const g1a *q, g1a *o) {
    g1j a; if (g1a_is_id(p)) g1j_set_id(&a); else { a.x = p->x; a.y = p->y; a.z = FQ.one; }
    g1j_add_mixed(&a, &a, q); g1j_to_affine(o, &a);
}
/* double-and-add scalar multiplication, scalar in Montgomery form (the slow obviously-right path) */
void oracle_g1_mul(const g1a *p, const fe *scalar_mont, g1a *o) {
    fe s; f_from_mont(&s, scalar_mont, &FR);
    g1j acc; g1j_set_id(&acc);
    for (int i = 255; i >= 0; i--) {
        g1j_double(&acc, &acc);
        if ((s.v[i >> 6] >> (i & 63)) & 1) g1j_add_mixed(&acc, &acc, p);
    }
    g1j_to_affine(o, &acc);
}
/* naive MSM: sum of independent double-and-add products (reference for Pippenger itself) */
void oracle_msm_naive(const fe *scalars, const g1a *bases, size_t n, g1a *out) {
    g1j acc; g1j_set_id(&acc);
    for (size_t k = 0; k < n; k++) {
        fe s; f_from_mont(&s, &scalars[k], &FR);
        g1j t; g1j_set_id(&t);
        for (int i = 255; i >= 0; i--) {
            g1j_double(&t, &t);
            if ((s.v[i >> 6] >> (i & 63)) & 1) g1j_add_mixed(&t, &t, &bases[k]);
        }
        g1j_add(&acc, &acc, &t);
    }
    g1j_to_affine(out, &acc);
}

/* ------------------------------------------------------------------ MSM (Pippenger) --------- */
static inline uint32_t get_bits(const fe *s, unsigned lo, unsigned c) {   /* c <= 31 bits starting at lo */
    if (lo >= 256) return 0;
    unsigned w = lo >> 6, sh = lo & 63;
    uint64_t x = s->v[w] >> sh;
    if (sh + c > 64 && w + 1 < 4) x |= s->v[w + 1] << (64 - sh);
    return (uint32_t)(x & ((1ull << c) - 1));
}
static unsigned msm_window(size_t n) {
    if (n < 4) return 1;
    if (n < 32) return 3;
    unsigned lg = 0; while ((1ull << (lg + 1)) <= n) lg++;
    unsigned c = (lg * 69 + 50) / 100 + 2;      /* ~ ln(n) + 2, the usual Pippenger heuristic */
    return c > 16 ? 16 : c;
}
/* Signed-digit windowed Pippenger.  Scalars: Montgomery Fr (as halo2 holds them); bases: affine
 * Montgomery Fq; out: affine (the caller of commit_lagrange batch-normalises, polycommit.rs:76). */
void oracle_msm(const fe *scalars, const g1a *bases, size_t n, g1a *out) {
    if (n == 0) { memset(out, 0, sizeof *out); return; }
    unsigned c = msm_window(n);
    unsigned nwin = (254 + c) / c + 1;            /* one spare window for the signed carry */
    size_t nb = (size_t)1 << (c - 1);
    fe *canon = (fe *)malloc(n * sizeof(fe));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) f_from_mont(&canon[i], &scalars[i], &FR);
    /* signed digits d in [-2^(c-1), 2^(c-1)], stored as int32 per (window, point) */
    int32_t *dig = (int32_t *)malloc((size_t)nwin * n * sizeof(int32_t));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        uint32_t carry = 0;
        for (unsigned w = 0; w < nwin; w++) {
            uint32_t raw = get_bits(&canon[i], w * c, c) + carry;
            if (raw > nb) { dig[(size_t)w * n + i] = (int32_t)raw - (int32_t)(1u << c); carry = 1; }
            else { dig[(size_t)w * n + i] = (int32_t)raw; carry = 0; }
        }
    }
    g1j *wsum = (g1j *)malloc(nwin * sizeof(g1j));
#pragma omp parallel for schedule(dynamic, 1)
    for (unsigned w = 0; w < nwin; w++) {
        g1j *bk = (g1j *)malloc(nb * sizeof(g1j));
        for (size_t b = 0; b < nb; b++) g1j_set_id(&bk[b]);
        const int32_t *d = dig + (size_t)w * n;
        for (size_t i = 0; i < n; i++) {
            if (d[i] > 0) g1j_add_mixed(&bk[d[i] - 1], &bk[d[i] - 1], &bases[i]);
            else if (d[i] < 0) { g1a nq = bases[i]; f_neg(&nq.y, &nq.y, &FQ); g1j_add_mixed(&bk[-d[i] - 1], &bk[-d[i] - 1], &nq); }
        }
        g1j run, acc; g1j_set_id(&run); g1j_set_id(&acc);
        for (size_t b = nb; b-- > 0;) { g1j_add(&run, &run, &bk[b]); g1j_add(&acc, &acc, &run); }
        wsum[w] = acc;
        free(bk);
    }
    g1j tot; g1j_set_id(&tot);
    for (unsigned w = nwin; w-- > 0;) {
        for (unsigned k = 0; k < c; k++) g1j_double(&tot, &tot);
        g1j_add(&tot, &tot, &wsum[w]);
    }
    g1j_to_affine(out, &tot);
    free(wsum); free(dig); free(canon);
}

/* ------------------------------------------------------------------ FFT --------------------- */
static inline uint32_t bitrev(uint32_t x, unsigned bits) {
    uint32_t r = 0; for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r;
}
/* best_fft(a, omega, log_n): in place, natural order in and out, no scaling.
 * Same dataflow as halo2curves' serial path (bit reversal, then log_n radix-2 DIT stages over a table of
 * omega powers); the loops are spread over the host cores so that it is a fair CPU baseline. */
void oracle_fft(fe *a, unsigned log_n, const fe *omega) {
    size_t n = (size_t)1 << log_n;
#pragma omp parallel for schedule(static) if (n >= 65536)
    for (size_t i = 0; i < n; i++) { size_t j = bitrev((uint32_t)i, log_n); if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; } }
    if (log_n == 0) return;
    fe *tw = (fe *)malloc((n / 2) * sizeof(fe));
    {   /* twiddles in independent chunks: each chunk starts from omega^(start) by square-and-multiply */
        size_t half = n / 2, chunk = half < 4096 ? half : 4096, nchunks = (half + chunk - 1) / chunk;
#pragma omp parallel for schedule(static) if (nchunks > 1)
        for (size_t c = 0; c < nchunks; c++) {
            size_t lo = c * chunk, hi = lo + chunk < half ? lo + chunk : half;
            uint64_t e[4] = {lo, 0, 0, 0};
            fe cur; f_pow(&cur, omega, e, &FR);
            for (size_t i = lo; i < hi; i++) { tw[i] = cur; f_mul(&cur, &cur, omega, &FR); }
        }
    }
    for (size_t m = 1; m < n; m <<= 1) {
        size_t step = n / (2 * m), nblk = n / (2 * m);
        if (nblk >= 64) {                       /* many small blocks: one thread per run of blocks */
#pragma omp parallel for schedule(static) if (n >= 4096)
            for (size_t blk = 0; blk < nblk; blk++) {
                fe *lo = a + blk * 2 * m, *hi = lo + m;
                for (size_t j = 0; j < m; j++) {
                    fe t; if (j == 0) t = hi[0]; else f_mul(&t, &hi[j], &tw[j * step], &FR);
                    fe u = lo[j];
                    f_add(&lo[j], &u, &t, &FR); f_sub(&hi[j], &u, &t, &FR);
                }
            }
        } else {                                /* few large blocks: split the butterflies of each block */
            for (size_t blk = 0; blk < nblk; blk++) {
                fe *lo = a + blk * 2 * m, *hi = lo + m;
#pragma omp parallel for schedule(static) if (m >= 4096)
                for (size_t j = 0; j < m; j++) {
                    fe t; if (j == 0) t = hi[0]; else f_mul(&t, &hi[j], &tw[j * step], &FR);
                    fe u = lo[j];
                    f_add(&lo[j], &u, &t, &FR); f_sub(&hi[j], &u, &t, &FR);
                }
            }
        }
    }
    free(tw);
}
static void fr_omega(fe *o, unsigned k) {          /* ROOT^(2^(28-k)) */
    fe w = {BN64_FR_ROOT_INIT};
    for (unsigned i = k; i < 28; i++) f_sqr(&w, &w, &FR);
    *o = w;
}
void oracle_omega(unsigned k, fe *o) { fr_omega(o, k); }
/* EvaluationDomain::ifft / lagrange_to_coeff : best_fft(omega^-1) then * n^-1 */
void oracle_lagrange_to_coeff(fe *a, unsigned k) {
    fe w, wi, ninv; fr_omega(&w, k); f_inv(&wi, &w, &FR);
    oracle_fft(a, k, &wi);
    f_from_u64(&ninv, (uint64_t)1 << k, &FR); f_inv(&ninv, &ninv, &FR);
    size_t n = (size_t)1 << k;
#pragma omp parallel for schedule(static) if (n >= 65536)
    for (size_t i = 0; i < n; i++) f_mul(&a[i], &a[i], &ninv, &FR);
}
void oracle_coeff_to_lagrange(fe *a, unsigned k) { fe w; fr_omega(&w, k); oracle_fft(a, k, &w); }
/* EvaluationDomain::coeff_to_extended: in[n] coeffs -> out[2^ext_k] evaluations on the zeta-coset */
void oracle_coeff_to_extended(const fe *in, unsigned k, unsigned ext_k, fe *out) {
    size_t n = (size_t)1 << k, ne = (size_t)1 << ext_k;
#pragma omp parallel for schedule(static) if (n >= 65536)
    for (size_t i = 0; i < n; i++) {
        switch (i % 3) { case 0: out[i] = in[i]; break; case 1: f_mul(&out[i], &in[i], &FR_ZETA, &FR); break;
                         default: f_mul(&out[i], &in[i], &FR_ZETA2, &FR); }
    }
    memset(out + n, 0, (ne - n) * sizeof(fe));
    fe w; fr_omega(&w, ext_k); oracle_fft(out, ext_k, &w);
}
/* EvaluationDomain::extended_to_coeff: in place on 2^ext_k values; caller truncates */
void oracle_extended_to_coeff(fe *a, unsigned ext_k) {
    size_t ne = (size_t)1 << ext_k;
    fe w, wi, ninv; fr_omega(&w, ext_k); f_inv(&wi, &w, &FR);
    oracle_fft(a, ext_k, &wi);
    f_from_u64(&ninv, (uint64_t)ne, &FR); f_inv(&ninv, &ninv, &FR);
    for (size_t i = 0; i < ne; i++) {
        f_mul(&a[i], &a[i], &ninv, &FR);
        if (i % 3 == 1) f_mul(&a[i], &a[i], &FR_ZETA2, &FR);      /* zeta^-1 = zeta^2 */
        else if (i % 3 == 2) f_mul(&a[i], &a[i], &FR_ZETA, &FR);  /* zeta^-2 = zeta   */
    }
}
/* EvaluationDomain::divide_by_vanishing_poly: a[i] *= t_evaluations[i mod 2^(ext_k-k)],
 * t_evaluations[j] = 1 / ((zeta * omega_ext^j)^n - 1) */
void oracle_divide_by_vanishing(fe *a, unsigned k, unsigned ext_k) {
    size_t ne = (size_t)1 << ext_k, period = (size_t)1 << (ext_k - k);
    fe *t = (fe *)malloc(period * sizeof(fe));
    fe w; fr_omega(&w, ext_k);
    fe cur = FR_ZETA;
    uint64_t e[4] = {(uint64_t)1 << k, 0, 0, 0};
    for (size_t j = 0; j < period; j++) {
        fe p; f_pow(&p, &cur, e, &FR); f_sub(&p, &p, &FR.one, &FR); f_inv(&t[j], &p, &FR);
        f_mul(&cur, &cur, &w, &FR);
    }
    for (size_t i = 0; i < ne; i++) f_mul(&a[i], &a[i], &t[i & (period - 1)], &FR);
    free(t);
}

/* ------------------------------------------------------------------ vec ops ----------------- */
void oracle_vec_mul(const fe *a, const fe *b, fe *o, size_t n) { for (size_t i = 0; i < n; i++) f_mul(&o[i], &a[i], &b[i], &FR); }
void oracle_vec_add(const fe *a, const fe *b, fe *o, size_t n) { for (size_t i = 0; i < n; i++) f_add(&o[i], &a[i], &b[i], &FR); }
void oracle_vec_sub(const fe *a, const fe *b, fe *o, size_t n) { for (size_t i = 0; i < n; i++) f_sub(&o[i], &a[i], &b[i], &FR); }
void oracle_vec_scale(const fe *a, const fe *s, fe *o, size_t n) { for (size_t i = 0; i < n; i++) f_mul(&o[i], &a[i], s, &FR); }
/* Montgomery batch inversion (halo2 BatchInvert semantics: zeros stay zero) */
void oracle_batch_invert(fe *a, size_t n) {
    fe *pre = (fe *)malloc(n * sizeof(fe));
    fe acc = FR.one;
    for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!f_is_zero(&a[i])) f_mul(&acc, &acc, &a[i], &FR); }
    f_inv(&acc, &acc, &FR);
    for (size_t i = n; i-- > 0;) {
        if (f_is_zero(&a[i])) continue;
        fe t; f_mul(&t, &acc, &pre[i], &FR); f_mul(&acc, &acc, &a[i], &FR); a[i] = t;
    }
    free(pre);
}

/* q(X) = p(X) / (X - z), remainder dropped (synthetic division: halo2_proofs::arithmetic::kate_division, un-vendored; the
 * quotients of the KZG / SHPLONK openings behind /root/reference/src/pfsys/mod.rs:456-463), in place; the top coefficient becomes 0 */
void oracle_kate_div(fe *a, size_t n, const fe *z) {
    fe carry = {{0, 0, 0, 0}};
    for (size_t i = n; i-- > 0;) {
        fe cur = a[i];            /* q_{i-1} = p_i + z * q_i */
        a[i] = carry;
        f_mul(&carry, &carry, z, &FR);
        f_add(&carry, &carry, &cur, &FR);
    }
}

/* eval_polynomial(poly, x): Horner from the top coefficient */
void oracle_eval_poly(const fe *c, size_t n, const fe *x, fe *out) {
    fe acc = {{0, 0, 0, 0}};
    for (size_t i = n; i-- > 0;) { f_mul(&acc, &acc, x, &FR); f_add(&acc, &acc, &c[i], &FR); }
    *out = acc;
}

/* running sum / product (permutation grand product z, mv-lookup grand sum phi); op 0 = add, 2 = mul */
void oracle_prefix_scan(const fe *in, fe *out, size_t n, int op, int exclusive) {
    fe acc = op == 0 ? (fe){{0, 0, 0, 0}} : FR.one;
    for (size_t i = 0; i < n; i++) {
        fe x = in[i];
        if (exclusive) out[i] = acc;
        if (op == 0) f_add(&acc, &acc, &x, &FR); else f_mul(&acc, &acc, &x, &FR);
        if (!exclusive) out[i] = acc;
    }
}

/* ------------------------------------------------------------------ GraphEvaluator ---------- */
/* Program format (shared with include/ezkl_hip.h): 8 x u32 per instruction
 *   [op, target, s0.kind, s0.idx, s0.rot, s1.kind, s1.idx, s1.rot]
 * ops / source kinds mirror halo2 plonk::evaluation::{Calculation, ValueSource}.  Horner(start, parts,
 * factor) is lowered by the host into STORE(start) followed by one HORNER_STEP per part
 * (target = target*s1 + s0), which is the loop body of Calculation::Horner. */
enum { OP_ADD = 0, OP_SUB, OP_MUL, OP_SQUARE, OP_DOUBLE, OP_NEGATE, OP_STORE, OP_HORNER_STEP };
enum { SRC_CONST = 0, SRC_INTERMEDIATE, SRC_COLUMN, SRC_CHALLENGE, SRC_PREVIOUS };
typedef struct {
    const uint32_t *code; uint32_t n_instr; uint32_t n_intermediates;
    const fe *constants; uint32_t n_constants;
    const int32_t *rotations; uint32_t n_rotations;   /* in units of original-domain rows */
    const fe *const *columns; uint32_t n_columns;      /* each 2^ext_k long */
    const fe *challenges; uint32_t n_challenges;
    uint32_t k, ext_k;
} oracle_program;

/* out[r] = last intermediate of the program run on extended row r; `previous` is out's old value
 * (ValueSource::PreviousValue), exactly as evaluate_h threads `values[idx]` through the gate program */
void oracle_eval_program(const oracle_program *p, fe *out) {
    size_t ne = (size_t)1 << p->ext_k;
    int64_t rot_scale = (int64_t)1 << (p->ext_k - p->k);
#pragma omp parallel
    {
        fe *im = (fe *)malloc((p->n_intermediates ? p->n_intermediates : 1) * sizeof(fe));
#pragma omp for schedule(static)
        for (size_t r = 0; r < ne; r++) {
            fe prev = out[r];
            uint32_t last = 0;
            for (uint32_t ii = 0; ii < p->n_instr; ii++) {
                const uint32_t *I = p->code + 8 * (size_t)ii;
                fe s[2];
                for (int q = 0; q < 2; q++) {
                    uint32_t kind = I[2 + 3 * q], idx = I[3 + 3 * q], rot = I[4 + 3 * q];
                    switch (kind) {
                    case SRC_CONST: s[q] = p->constants[idx]; break;
                    case SRC_INTERMEDIATE: s[q] = im[idx]; break;
                    case SRC_COLUMN: {
                        int64_t rr = ((int64_t)r + (int64_t)p->rotations[rot] * rot_scale) % (int64_t)ne;
                        if (rr < 0) rr += ne;
                        s[q] = p->columns[idx][rr]; break; }
                    case SRC_CHALLENGE: s[q] = p->challenges[idx]; break;
                    default: s[q] = prev; break;
                    }
                }
                fe *t = &im[I[1]];
                switch (I[0]) {
                case OP_ADD: f_add(t, &s[0], &s[1], &FR); break;
                case OP_SUB: f_sub(t, &s[0], &s[1], &FR); break;
                case OP_MUL: f_mul(t, &s[0], &s[1], &FR); break;
                case OP_SQUARE: f_sqr(t, &s[0], &FR); break;
                case OP_DOUBLE: f_dbl(t, &s[0], &FR); break;
                case OP_NEGATE: f_neg(t, &s[0], &FR); break;
                case OP_STORE: *t = s[0]; break;
                default: { fe m; f_mul(&m, t, &s[1], &FR); f_add(t, &m, &s[0], &FR); } break;
                }
                last = I[1];
            }
            if (p->n_instr) out[r] = im[last];
        }
        free(im);
    }
}

/* ------------------------------------------------------------------ ChaCha20 field sampler ---------- */
/* ChaCha20 block function (D. J. Bernstein; RFC 8439 section 2.3 with the counter/nonce words regrouped as a 64-bit
 * block counter in words 12,13 and a 64-bit stream id in words 14,15), pinned on the RFC 8439 2.3.2 test vector in
 * tests/test_oracle.py.  The sampler restates ezkl_hip_chacha20_fr_dev (include/ezkl_hip.h): rejection sampling of
 * 254-bit candidates against r. */
void oracle_chacha20_block(const uint32_t key[8], uint64_t counter, uint64_t stream, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t x[16];
    memcpy(x, s, sizeof s);
#define ROTL(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define QR(a, b, c, d) a += b; d ^= a; d = ROTL(d, 16); c += d; b ^= c; b = ROTL(b, 12); a += b; d ^= a; d = ROTL(d, 8); c += d; b ^= c; b = ROTL(b, 7);
    for (int r = 0; r < 10; r++) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
#undef QR
#undef ROTL
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
void oracle_chacha20_fr(const uint32_t key[8], uint64_t stream, size_t first, size_t n, fe *out) {
    for (size_t i = 0; i < n; i++) {
        fe v;
        memset(&v, 0, sizeof v);
        int done = 0;
        for (uint32_t b = 0; b < 16 && !done; b++) {
            uint32_t blk[16];
            oracle_chacha20_block(key, (uint64_t)(first + i) * 16 + b, stream, blk);
            for (int h = 0; h < 2 && !done; h++) {
                fe cand;
                for (int q = 0; q < 4; q++) cand.v[q] = (uint64_t)blk[8 * h + 2 * q] | ((uint64_t)blk[8 * h + 2 * q + 1] << 32);
                cand.v[3] &= 0x3fffffffffffffffull;
                if (!f_geq(&cand, &FR.mod)) { v = cand; done = 1; }
            }
        }
        out[i] = v;
    }
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ synthetic inputs -------- */
/* Deterministic G1 points for tests/bench (try-and-increment on y^2 = x^3 + 3).
 * x0 = 4 splitmix64 words keyed by (seed, i), masked to 254 bits, reduced once, read as a Montgomery
 * residue; x += 1 until x^3 + 3 is a square; y = rhs^((q+1)/4), negated if its low Montgomery limb is odd.
 * The HIP library has the same generator (ezkl_hip_gen_bases) so the two can be byte-compared. */
static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31);
}
void oracle_gen_bases(uint64_t seed, size_t first, size_t n, g1a *out) {
    static const uint64_t sqrt_e[4] = BN64_FQ_SQRT_EXP_INIT;
#pragma omp parallel for schedule(static)
    for (size_t k = 0; k < n; k++) {
        uint64_t st = seed ^ ((uint64_t)(first + k) * 0xd1342543de82ef95ull);
        fe x; for (int j = 0; j < 4; j++) x.v[j] = splitmix64(&st);
        x.v[3] &= 0x3fffffffffffffffull;
        if (f_geq(&x, &FQ.mod)) sub4(&x, &x, &FQ.mod);
        for (;;) {
            fe rhs, y, yy; f_sqr(&rhs, &x, &FQ); f_mul(&rhs, &rhs, &x, &FQ); f_add(&rhs, &rhs, &FQ_B3, &FQ);
            f_pow(&y, &rhs, sqrt_e, &FQ); f_sqr(&yy, &y, &FQ);
            if (f_eq(&yy, &rhs) && !f_is_zero(&y)) { if (y.v[0] & 1) f_neg(&y, &y, &FQ); out[k].x = x; out[k].y = y; break; }
            f_add(&x, &x, &FQ.one, &FQ);
        }
    }
}
