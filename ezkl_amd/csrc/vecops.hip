// vecops.hip -- element-wise Fr kernels on device-resident columns (the icicle "vec-ops" surface the
// halo2 fork uses between MSM/NTT calls, SURVEY.md §2 kernel inventory) plus divide_by_vanishing_poly
// and Montgomery batch inversion (mv-lookup / permutation helpers).
// All are streaming kernels: one 32-byte element per lane per access (2 x global_load_dwordx4),
// grid-stride, HBM-bound except batch inversion.
#include "common.hpp"

namespace ezkl {

template <int OP>
__global__ __launch_bounds__(256) void vec_op_kernel(const fe_t* a, const fe_t* b, fe_t* o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        fe_t x = ld_fe(a + i), y = ld_fe(b + i);
        fe_t r = OP == EZKL_VEC_ADD ? Fr::add(x, y) : OP == EZKL_VEC_SUB ? Fr::sub(x, y) : Fr::mul(x, y);
        st_fe(o + i, r);
    }
}
__global__ __launch_bounds__(256) void vec_scale_kernel(const fe_t* a, fe_t s, fe_t* o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        st_fe(o + i, Fr::mul(ld_fe(a + i), s));
}
__global__ __launch_bounds__(256) void vec_periodic_mul_kernel(fe_t* a, const fe_t* t, uint32_t mask, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        st_fe(a + i, Fr::mul(ld_fe(a + i), ld_fe(t + (i & mask))));
}

static unsigned stream_grid(Ctx* c, size_t n) {
    size_t want = (n + 255) / 256, cap = (size_t)c->num_cus * 8;
    return (unsigned)(want < cap ? (want ? want : 1) : cap);
}

int vec_op(Ctx* c, hipStream_t st, int op, const fe_t* a, const fe_t* b, fe_t* o, size_t n) {
    if (n == 0) return EZKL_OK;
    dim3 g(stream_grid(c, n)), blk(256);
    if (op == EZKL_VEC_ADD) hipLaunchKernelGGL(vec_op_kernel<EZKL_VEC_ADD>, g, blk, 0, st, a, b, o, n);
    else if (op == EZKL_VEC_SUB) hipLaunchKernelGGL(vec_op_kernel<EZKL_VEC_SUB>, g, blk, 0, st, a, b, o, n);
    else hipLaunchKernelGGL(vec_op_kernel<EZKL_VEC_MUL>, g, blk, 0, st, a, b, o, n);
    EZ_HIP(hipGetLastError());
    return EZKL_OK;
}
int vec_scale(Ctx* c, hipStream_t st, const fe_t* a, const fe_t& s, fe_t* o, size_t n) {
    if (n == 0) return EZKL_OK;
    hipLaunchKernelGGL(vec_scale_kernel, dim3(stream_grid(c, n)), dim3(256), 0, st, a, s, o, n);
    EZ_HIP(hipGetLastError());
    return EZKL_OK;
}

// EvaluationDomain::divide_by_vanishing_poly: a[i] *= t[i mod 2^(ext_k-k)],
// t[j] = ((zeta * w_ext^j)^(2^k) - 1)^-1
int divide_by_vanishing(Ctx* c, hipStream_t st, fe_t* a, uint32_t k, uint32_t ext_k) {
    const size_t period = (size_t)1 << (ext_k - k), ne = (size_t)1 << ext_k;
    std::vector<fe_t> t(period);
    fe_t w = fr_const(FrConst::ROOT);
    for (uint32_t i = ext_k; i < 28; i++) w = Fr::sqr(w);
    fe_t cur = fr_const(FrConst::ZETA);
    for (size_t j = 0; j < period; j++) {
        fe_t p = cur;
        for (uint32_t q = 0; q < k; q++) p = Fr::sqr(p);
        t[j] = Fr::inv(Fr::sub(p, Fr::one()));
        cur = Fr::mul(cur, w);
    }
    fe_t* dt = nullptr;
    EZ_HIP(hipMalloc(&dt, period * sizeof(fe_t)));
    EZ_HIP(hipMemcpyAsync(dt, t.data(), period * sizeof(fe_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(vec_periodic_mul_kernel, dim3(stream_grid(c, ne)), dim3(256), 0, st, a, dt, (uint32_t)(period - 1), ne);
    EZ_HIP(hipGetLastError());
    EZ_HIP(hipStreamSynchronize(st));
    EZ_HIP(hipFree(dt));
    return EZKL_OK;
}

// Montgomery batch inversion.  T threads; thread t owns elements {t + j*T}: a coalesced strided chain.
// forward: pre[t + j*T] = prod_{i<j} a_i (zeros skipped); invert the chain product once (Fermat);
// backward: a_j^-1 = acc * pre_j ; acc *= a_j.
__global__ __launch_bounds__(256) void batch_invert_kernel(fe_t* a, fe_t* pre, size_t n, size_t T) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    fe_t acc = Fr::one();
    size_t last = t;
    for (size_t i = t; i < n; i += T) {
        st_fe(pre + i, acc);
        fe_t x = ld_fe(a + i);
        if (!Fr::is_zero(x)) acc = Fr::mul(acc, x);
        last = i;
    }
    acc = Fr::inv(acc);
    for (size_t i = last;; i -= T) {
        fe_t x = ld_fe(a + i);
        if (!Fr::is_zero(x)) {
            st_fe(a + i, Fr::mul(acc, ld_fe(pre + i)));
            acc = Fr::mul(acc, x);
        }
        if (i < T) break;
    }
}
int batch_invert(Ctx* c, hipStream_t st, fe_t* a, size_t n) {
    if (n == 0) return EZKL_OK;
    size_t T = n / 64;                       // >= 64 elements per chain amortises the Fermat inversion
    size_t cap = (size_t)c->num_cus * 256 * 4;
    if (T > cap) T = cap;
    if (T < 1) T = 1;
    fe_t* pre = nullptr;
    EZ_HIP(hipMalloc(&pre, n * sizeof(fe_t)));
    hipLaunchKernelGGL(batch_invert_kernel, dim3(cdiv(T, 256)), dim3(256), 0, st, a, pre, n, T);
    EZ_HIP(hipGetLastError());
    EZ_HIP(hipStreamSynchronize(st));
    EZ_HIP(hipFree(pre));
    return EZKL_OK;
}

}  // namespace ezkl
