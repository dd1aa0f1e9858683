// g2.hip -- BN254 G2 (the sextic twist y^2 = x^3 + 3 / (9 + u) over Fq2 = Fq[u] / (u^2 + 1)) on the device.
//
// The north star names "multi-scalar multiplication over BN254 G1/G2".  On the reference's prove path G2 is DATA: the two points g2 and
// s_g2 of the SRS file (/root/reference/src/pfsys/srs.rs:14-16: gen_srs -> ParamsKZG::setup computes s_g2 = [s] g2 once), consumed by the
// verifier's pairing; no G2 MSM runs in `prove`.  This file is therefore a plain, correct G2 group law and a small MSM -- one
// double-and-add per (point, scalar) pair, a strided partial sum per thread, one LDS tree -- for SRS generation and for callers that hold G2
// commitments; it shares nothing with the G1 Pippenger pipeline and is not a hot path (n is 1 for every use the reference has).
// Layout: a G2 affine point is the 128 bytes of the SRS file -- x.c0, x.c1, y.c0, y.c1, each a 32-byte little-endian Montgomery Fq --
// (0, 0) = identity.
#include "common.hpp"
#include <string.h>

namespace ezkl {

struct fq2_t {
    fe_t a, b;      // a + b u
};
struct g2a_t {
    fq2_t x, y;
};
struct g2j_t {      // Jacobian: x = X / Z^2, y = Y / Z^3; identity: Z = 0
    fq2_t x, y, z;
};

EZ_D fq2_t f2_zero() { return fq2_t{Fq::zero(), Fq::zero()}; }
EZ_D fq2_t f2_one() { return fq2_t{Fq::one(), Fq::zero()}; }
EZ_D bool f2_is_zero(const fq2_t& x) { return Fq::is_zero(x.a) && Fq::is_zero(x.b); }
EZ_D fq2_t f2_add(const fq2_t& x, const fq2_t& y) { return fq2_t{Fq::add(x.a, y.a), Fq::add(x.b, y.b)}; }
EZ_D fq2_t f2_sub(const fq2_t& x, const fq2_t& y) { return fq2_t{Fq::sub(x.a, y.a), Fq::sub(x.b, y.b)}; }
EZ_D fq2_t f2_dbl(const fq2_t& x) { return f2_add(x, x); }
EZ_D fq2_t f2_neg(const fq2_t& x) { return fq2_t{Fq::neg(x.a), Fq::neg(x.b)}; }
EZ_D fq2_t f2_mul(const fq2_t& x, const fq2_t& y) {                   // Karatsuba: 3 products
    const fe_t aa = Fq::mul(x.a, y.a), bb = Fq::mul(x.b, y.b);
    const fe_t cross = Fq::mul(Fq::add(x.a, x.b), Fq::add(y.a, y.b));
    return fq2_t{Fq::sub(aa, bb), Fq::sub(Fq::sub(cross, aa), bb)};
}
EZ_D fq2_t f2_sqr(const fq2_t& x) {                                    // (a + b)(a - b) + 2ab u: 2 products
    const fe_t t = Fq::mul(Fq::add(x.a, x.b), Fq::sub(x.a, x.b));
    const fe_t ab = Fq::mul(x.a, x.b);
    return fq2_t{t, Fq::dbl(ab)};
}
EZ_D fq2_t f2_inv(const fq2_t& x) {                                    // conj(x) / (a^2 + b^2)
    const fe_t n = Fq::inv(Fq::add(Fq::sqr(x.a), Fq::sqr(x.b)));
    return fq2_t{Fq::mul(x.a, n), Fq::neg(Fq::mul(x.b, n))};
}

EZ_D g2j_t g2_identity() { return g2j_t{f2_zero(), f2_zero(), f2_zero()}; }
EZ_D bool g2_is_id(const g2j_t& p) { return f2_is_zero(p.z); }
EZ_D g2j_t g2_from_affine(const g2a_t& p) {
    if (f2_is_zero(p.x) && f2_is_zero(p.y)) return g2_identity();
    return g2j_t{p.x, p.y, f2_one()};
}
// dbl-2009-l (a = 0)
EZ_D g2j_t g2_double(const g2j_t& p) {
    if (g2_is_id(p)) return p;
    const fq2_t A = f2_sqr(p.x), B = f2_sqr(p.y), C = f2_sqr(B);
    const fq2_t D = f2_dbl(f2_sub(f2_sub(f2_sqr(f2_add(p.x, B)), A), C));
    const fq2_t E = f2_add(f2_dbl(A), A), F = f2_sqr(E);
    g2j_t r;
    r.x = f2_sub(F, f2_dbl(D));
    const fq2_t C8 = f2_dbl(f2_dbl(f2_dbl(C)));
    r.y = f2_sub(f2_mul(E, f2_sub(D, r.x)), C8);
    r.z = f2_dbl(f2_mul(p.y, p.z));
    return r;
}
// add-2007-bl with the identity / doubling / inverse cases
EZ_D g2j_t g2_add(const g2j_t& p, const g2j_t& q) {
    if (g2_is_id(p)) return q;
    if (g2_is_id(q)) return p;
    const fq2_t z1z1 = f2_sqr(p.z), z2z2 = f2_sqr(q.z);
    const fq2_t u1 = f2_mul(p.x, z2z2), u2 = f2_mul(q.x, z1z1);
    const fq2_t s1 = f2_mul(f2_mul(p.y, q.z), z2z2), s2 = f2_mul(f2_mul(q.y, p.z), z1z1);
    const fq2_t h = f2_sub(u2, u1), rr = f2_dbl(f2_sub(s2, s1));
    if (f2_is_zero(h)) {
        if (f2_is_zero(rr)) return g2_double(p);
        return g2_identity();
    }
    const fq2_t i = f2_sqr(f2_dbl(h)), j = f2_mul(h, i), v = f2_mul(u1, i);
    g2j_t r;
    r.x = f2_sub(f2_sub(f2_sqr(rr), j), f2_dbl(v));
    r.y = f2_sub(f2_mul(rr, f2_sub(v, r.x)), f2_dbl(f2_mul(s1, j)));
    r.z = f2_mul(f2_sub(f2_sub(f2_sqr(f2_add(p.z, q.z)), z1z1), z2z2), h);
    return r;
}
EZ_D g2a_t g2_to_affine(const g2j_t& p) {
    if (g2_is_id(p)) return g2a_t{f2_zero(), f2_zero()};
    const fq2_t zi = f2_inv(p.z), zi2 = f2_sqr(zi);
    return g2a_t{f2_mul(p.x, zi2), f2_mul(p.y, f2_mul(zi2, zi))};
}
EZ_D g2j_t g2_scalar_mul(const g2j_t& p, const fe_t& s_canon) {
    g2j_t acc = g2_identity();
    bool started = false;
#pragma unroll 1
    for (int b = 253; b >= 0; b--) {
        if (started) acc = g2_double(acc);
        if ((s_canon.v[b >> 5] >> (b & 31)) & 1) {
            acc = started ? g2_add(acc, p) : p;
            started = true;
        }
    }
    return acc;
}
EZ_D g2a_t ld_g2a(const g2a_t* p) {
    g2a_t r;
    r.x.a = ld_fe(&p->x.a); r.x.b = ld_fe(&p->x.b); r.y.a = ld_fe(&p->y.a); r.y.b = ld_fe(&p->y.b);
    return r;
}

// thread t: sum over i = t, t + T, ... of scalars[i] * points[i]  ->  partial[t] (Jacobian)
__global__ __launch_bounds__(64) void g2_msm_partial_kernel(const g2a_t* pts, const fe_t* scalars, size_t n, g2j_t* partial, uint32_t T) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    g2j_t acc = g2_identity();
    for (size_t i = t; i < n; i += T) {
        const fe_t s = Fr::from_mont(ld_fe(scalars + i));
        acc = g2_add(acc, g2_scalar_mul(g2_from_affine(ld_g2a(pts + i)), s));
    }
    partial[t] = acc;
}
// one workgroup folds the T <= 1024 partial sums (strided per thread, then an LDS tree) and normalises
__global__ __launch_bounds__(64) void g2_msm_reduce_kernel(const g2j_t* partial, uint32_t T, g2a_t* out) {
    __shared__ g2j_t sh[64];
    g2j_t acc = g2_identity();
    for (uint32_t i = threadIdx.x; i < T; i += 64) acc = g2_add(acc, partial[i]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 32; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = g2_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = g2_to_affine(sh[0]);
}

int g2_msm(Ctx* c, hipStream_t st, const void* pts_host, const void* scalars_host, size_t n, void* out_host) {
    (void)c;
    if (n == 0) { memset(out_host, 0, 128); return EZKL_OK; }
    g2a_t *d_pts = nullptr, *d_out = nullptr;
    fe_t* d_sc = nullptr;
    g2j_t* d_part = nullptr;
    const uint32_t T = (uint32_t)(n < 1024 ? n : 1024);
    int rc = EZKL_OK;
    hipError_t e = hipMalloc(&d_pts, n * sizeof(g2a_t));
    if (e == hipSuccess) e = hipMalloc(&d_sc, n * sizeof(fe_t));
    if (e == hipSuccess) e = hipMalloc(&d_part, (size_t)T * sizeof(g2j_t));
    if (e == hipSuccess) e = hipMalloc(&d_out, sizeof(g2a_t));
    if (e == hipSuccess) e = hipMemcpyAsync(d_pts, pts_host, n * sizeof(g2a_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_sc, scalars_host, n * sizeof(fe_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(g2_msm_partial_kernel, dim3(cdiv(T, 64)), dim3(64), 0, st, (const g2a_t*)d_pts, (const fe_t*)d_sc, n, d_part, T);
        hipLaunchKernelGGL(g2_msm_reduce_kernel, dim3(1), dim3(64), 0, st, (const g2j_t*)d_part, T, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_host, d_out, sizeof(g2a_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = set_hip_error(e, "g2_msm", __FILE__, __LINE__);
    if (d_pts) (void)hipFree(d_pts);
    if (d_sc) (void)hipFree(d_sc);
    if (d_part) (void)hipFree(d_part);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

}  // namespace ezkl
