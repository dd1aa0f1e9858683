// msm.hip -- fixed-base Pippenger MSM over BN254 G1 for gfx950.
//
// Replaces ParamsKZG::{commit, commit_lagrange} -> halo2curves::msm (in-tree call
// site /root/reference/src/circuit/modules/polycommit.rs:71).  Result is the canonical affine point, so it
// equals batch_normalize(commit_lagrange(..)) byte for byte (polycommit.rs:76).
//
// Design (MI355X-first; the KZG bases are FIXED for the life of the SRS, and HBM is 288 GB):
//   * At first use of a base set we precompute T[w][i] = 2^(c*w) * P_i in HBM (W = ceil(255/c) tables).
//     Every (point, window) digit then lands in ONE shared bucket set of 2^(c-1) buckets: no per-window
//     bucket reduction, no final window Horner, and c can be large (few adds per point).
//   * digits kernel : scalar -> canonical -> (s > r/2 ? r - s, negated) -> signed c-bit digits; one
//                     (bucket, table index | sign) pair per non-zero digit; bucket histogram by atomics.
//                     The r - s trick turns witness-like small negative values (src/fieldutils.rs:9-17)
//                     into single-digit scalars.
//   * counting sort  : exclusive scan of the histogram (hipCUB), scatter of the pair payloads.
//   * accumulate     : one lane per bucket walks its run, gathering 64-byte affine points from T and
//                     mixed-adding into an XYZZ accumulator held in VGPRs (the dominant kernel).
//                     Buckets longer than HEAVY are handed to a workgroup-per-bucket kernel with an LDS tree.
//   * reduce         : sum_b (b+1)*B_b via 8-bucket running sums + small scalar multiples, then a
//                     workgroup tree sum staged through LDS; final to-affine on the device.
// Order of additions differs from the CPU Pippenger, the group element (and its canonical affine bytes) does not.
#include "common.hpp"
#include "curve.hpp"
#include <hipcub/hipcub.hpp>
#include <string.h>

namespace ezkl {

static constexpr uint32_t MSM_SKIP = 0xffffffffu;
static constexpr uint32_t MSM_HEAVY = 192;       // runs longer than this go to the workgroup kernel
static constexpr uint32_t MSM_CHUNK = 8;         // buckets per running-sum chunk in the reduce phase

struct MsmTable {
    g1a_t* tab = nullptr;    // W x n affine
    uint32_t c = 0, W = 0;
    size_t n = 0;
};
static std::map<const Bases*, MsmTable> g_tables;   // guarded by the ctx mutex

static uint32_t pick_window(size_t n) {
    // aim at ~24 points per bucket: n*W / 2^(c-1) ~ 24, W = ceil(255/c)
    uint32_t best = 2;
    for (uint32_t c = 2; c <= 22; c++) {
        double W = (255 + c - 1) / c;
        double per = (double)n * W / (double)((size_t)1 << (c - 1));
        if (per >= 20.0) best = c;
    }
    return best;
}

// ---- table precompute: T[w] = 2^c * T[w-1] ------------------------------------------------------
__global__ __launch_bounds__(256) void msm_precompute_kernel(const g1a_t* prev, g1a_t* next, size_t n, uint32_t c) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1a_t p = ld_g1a(prev + i);
    g1x_t a = g1x_from_affine(p);
    for (uint32_t k = 0; k < c; k++) a = g1x_double(a);
    st_g1a(next + i, g1x_to_affine(a));
}

// ---- digits ------------------------------------------------------------------------------------
// half = (r-1)/2 as plain integer limbs
__device__ __forceinline__ bool gt_half_r(const fe_t& s) {
    // r = FrP::MOD ; compare s > (r-1)/2  <=>  2s > r - 1  <=> 2s >= r
    uint64_t c = 0;
    uint32_t d[9];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)s.v[i] * 2;
        d[i] = (uint32_t)c;
        c >>= 32;
    }
    d[8] = (uint32_t)c;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)d[i] - FrP::MOD[i] - br;
        br = (t >> 63) & 1;
    }
    return d[8] != 0 || br == 0;
}
__device__ __forceinline__ uint32_t get_bits(const fe_t& s, uint32_t lo, uint32_t c) {
    if (lo >= 256) return 0;
    uint32_t w = lo >> 5, sh = lo & 31;
    uint64_t x = s.v[w];
    if (w + 1 < 8) x |= (uint64_t)s.v[w + 1] << 32;
    return (uint32_t)((x >> sh) & (((uint64_t)1 << c) - 1));
}

__global__ __launch_bounds__(256) void msm_digits_kernel(const fe_t* scalars, size_t n, uint32_t c, uint32_t W,
                                                         uint32_t* keys, uint32_t* hist) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_t s = Fr::from_mont(ld_fe(scalars + i));
    uint32_t neg = 0;
    if (gt_half_r(s)) {          // s*P = (r-s)*(-P)
        uint64_t br = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint64_t t = (uint64_t)FrP::MOD[k] - s.v[k] - br;
            s.v[k] = (uint32_t)t;
            br = (t >> 63) & 1;
        }
        neg = 1;
    }
    const uint32_t half = 1u << (c - 1);
    uint32_t carry = 0;
    for (uint32_t w = 0; w < W; w++) {
        uint32_t raw = get_bits(s, w * c, c) + carry;
        uint32_t key = MSM_SKIP;
        if (raw > half) {
            carry = 1;
            uint32_t mag = (1u << c) - raw;           // digit = -(mag)
            key = (mag - 1) | ((neg ^ 1u) << 31);
        } else {
            carry = 0;
            if (raw) key = (raw - 1) | (neg << 31);
        }
        keys[(size_t)w * n + i] = key;
        if (key != MSM_SKIP) atomicAdd(&hist[key & 0x7fffffffu], 1u);
    }
}

__global__ __launch_bounds__(256) void msm_scatter_kernel(const uint32_t* keys, size_t n, uint32_t W, size_t base_offset,
                                                          size_t tab_stride, uint32_t* cursor, uint32_t* vals) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * W) return;
    uint32_t key = keys[idx];
    if (key == MSM_SKIP) return;
    size_t w = idx / n, i = idx - w * n;
    uint32_t pos = atomicAdd(&cursor[key & 0x7fffffffu], 1u);
    vals[pos] = (uint32_t)(w * tab_stride + base_offset + i) | (key & 0x80000000u);
}

// ---- bucket accumulation (dominant kernel) ------------------------------------------------------
__global__ __launch_bounds__(256) void msm_accumulate_kernel(const g1a_t* tab, const uint32_t* offsets, const uint32_t* vals,
                                                             uint32_t nb, g1x_t* buckets, uint32_t* heavy_list,
                                                             uint32_t* heavy_count) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    uint32_t beg = offsets[b], end = offsets[b + 1];
    if (end - beg > MSM_HEAVY) {
        heavy_list[atomicAdd(heavy_count, 1u)] = b;
        return;
    }
    g1x_t acc = g1x_identity();
    for (uint32_t k = beg; k < end; k++) {
        uint32_t v = vals[k];
        g1a_t p = ld_g1a(tab + (v & 0x7fffffffu));
        if (v >> 31) p.y = Fq::neg(p.y);
        acc = g1x_add_mixed(acc, p);
    }
    st_g1x(buckets + b, acc);
}

// one workgroup per heavy bucket: strided accumulation, then an LDS tree of XYZZ adds
__global__ __launch_bounds__(256) void msm_heavy_kernel(const g1a_t* tab, const uint32_t* offsets, const uint32_t* vals,
                                                        const uint32_t* heavy_list, const uint32_t* heavy_count,
                                                        g1x_t* buckets) {
    __shared__ g1x_t sh[256];
  for (uint32_t h = blockIdx.x; h < *heavy_count; h += gridDim.x) {
    uint32_t b = heavy_list[h];
    uint32_t beg = offsets[b], end = offsets[b + 1];
    g1x_t acc = g1x_identity();
    for (uint32_t k = beg + threadIdx.x; k < end; k += 256) {
        uint32_t v = vals[k];
        g1a_t p = ld_g1a(tab + (v & 0x7fffffffu));
        if (v >> 31) p.y = Fq::neg(p.y);
        acc = g1x_add_mixed(acc, p);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = g1x_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) st_g1x(buckets + b, sh[0]);
    __syncthreads();
  }
}

// ---- reduce: sum_b (b+1) * B_b -------------------------------------------------------------------
// thread t owns buckets [t*m, t*m + m): L = sum_j (j+1) * B_{tm+j} (running sums), A = sum_j B_{tm+j};
// contribution = L + (t*m) * A, the small multiple by double-and-add.
__global__ __launch_bounds__(256) void msm_chunk_reduce_kernel(const g1x_t* buckets, uint32_t nb, g1x_t* out, uint32_t nchunks) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    uint32_t lo = t * MSM_CHUNK, hi = lo + MSM_CHUNK;
    if (hi > nb) hi = nb;
    g1x_t run = g1x_identity(), acc = g1x_identity();
    for (uint32_t b = hi; b-- > lo;) {
        run = g1x_add(run, ld_g1x(buckets + b));
        acc = g1x_add(acc, run);
    }
    // (t*m) * run
    uint32_t k = lo;
    if (k && !g1x_is_id(run)) {
        g1x_t m = g1x_identity();
        for (int bit = 31 - __clz(k); bit >= 0; bit--) {
            m = g1x_double(m);
            if ((k >> bit) & 1) m = g1x_add(m, run);
        }
        acc = g1x_add(acc, m);
    }
    st_g1x(out + t, acc);
}
// out[blockIdx] = sum of in[blockIdx*256*per .. ) : per-thread serial partial, then LDS tree
__global__ __launch_bounds__(256) void msm_sum_kernel(const g1x_t* in, uint32_t n, g1x_t* out, uint32_t per) {
    __shared__ g1x_t sh[256];
    uint32_t base = (blockIdx.x * 256 + threadIdx.x) * per;
    g1x_t acc = g1x_identity();
    for (uint32_t k = 0; k < per; k++)
        if (base + k < n) acc = g1x_add(acc, ld_g1x(in + base + k));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = g1x_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) st_g1x(out + blockIdx.x, sh[0]);
}
__global__ void msm_finalize_kernel(const g1x_t* in, g1a_t* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) st_g1a(out, g1x_to_affine(ld_g1x(in)));
}

static int table_get(Ctx* c, hipStream_t st, const Bases* b, MsmTable** out) {
    auto it = g_tables.find(b);
    if (it != g_tables.end()) { *out = &it->second; return EZKL_OK; }
    MsmTable t;
    t.n = b->n;
    t.c = pick_window(b->n);
    t.W = (255 + t.c - 1) / t.c;
    if ((size_t)t.W * t.n >= ((size_t)1 << 31)) return EZKL_ERR_UNSUPPORTED;
    EZ_HIP(hipMalloc(&t.tab, (size_t)t.W * t.n * sizeof(g1a_t)));
    EZ_HIP(hipMemcpyAsync(t.tab, b->pts, t.n * sizeof(g1a_t), hipMemcpyDeviceToDevice, st));
    for (uint32_t w = 1; w < t.W; w++)
        hipLaunchKernelGGL(msm_precompute_kernel, dim3(cdiv(t.n, 256)), dim3(256), 0, st, t.tab + (size_t)(w - 1) * t.n,
                           t.tab + (size_t)w * t.n, t.n, t.c);
    EZ_HIP(hipGetLastError());
    EZ_HIP(hipStreamSynchronize(st));
    g_tables[b] = t;
    *out = &g_tables[b];
    return EZKL_OK;
}
void msm_table_drop(const Bases* b) {
    auto it = g_tables.find(b);
    if (it != g_tables.end()) {
        (void)hipFree(it->second.tab);
        g_tables.erase(it);
    }
}

int msm_run(Ctx* c, hipStream_t st, const Bases* b, size_t base_offset, const fe_t* scalars, size_t n, void* out_host) {
    if (n == 0) { memset(out_host, 0, 64); return EZKL_OK; }
    MsmTable* T = nullptr;
    int rc = table_get(c, st, b, &T);
    if (rc) return rc;
    const uint32_t cw = T->c, W = T->W;
    const uint32_t nb = 1u << (cw - 1);
    const size_t npairs = n * W;
    const uint32_t nchunks = cdiv(nb, MSM_CHUNK);
    // ---- carve scratch ----
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o_keys = 0;
    size_t o_vals = o_keys + al(npairs * 4);
    size_t o_hist = o_vals + al(npairs * 4);
    size_t o_offs = o_hist + al(((size_t)nb + 1) * 4);
    size_t o_curs = o_offs + al(((size_t)nb + 1) * 4);
    size_t o_heavy = o_curs + al(((size_t)nb + 1) * 4);
    size_t o_hcnt = o_heavy + al((size_t)nb * 4);
    size_t o_bkt = o_hcnt + al(256);
    size_t o_red = o_bkt + al((size_t)nb * sizeof(g1x_t));
    size_t o_red2 = o_red + al((size_t)nchunks * sizeof(g1x_t));
    size_t o_out = o_red2 + al((size_t)nchunks * sizeof(g1x_t));
    size_t cub_bytes = 0;
    EZ_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, cub_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (int)(nb + 1), st));
    size_t o_cub = o_out + al(256);
    size_t total = o_cub + al(cub_bytes);
    uint8_t* S = nullptr;
    rc = scratch_reserve(c, total, (void**)&S);
    if (rc) return rc;
    uint32_t* keys = (uint32_t*)(S + o_keys);
    uint32_t* vals = (uint32_t*)(S + o_vals);
    uint32_t* hist = (uint32_t*)(S + o_hist);
    uint32_t* offs = (uint32_t*)(S + o_offs);
    uint32_t* curs = (uint32_t*)(S + o_curs);
    uint32_t* heavy = (uint32_t*)(S + o_heavy);
    uint32_t* hcnt = (uint32_t*)(S + o_hcnt);
    g1x_t* bkt = (g1x_t*)(S + o_bkt);
    g1x_t* red = (g1x_t*)(S + o_red);
    g1x_t* red2 = (g1x_t*)(S + o_red2);
    g1a_t* dout = (g1a_t*)(S + o_out);

    hipEvent_t m0, m1, a0, a1;
    if ((rc = ev_pair(c, "msm", &m0, &m1))) return rc;
    if ((rc = ev_pair(c, "msm_accumulate", &a0, &a1))) return rc;
    EZ_HIP(hipEventRecord(m0, st));
    EZ_HIP(hipMemsetAsync(hist, 0, ((size_t)nb + 1) * 4, st));
    EZ_HIP(hipMemsetAsync(hcnt, 0, 4, st));
    hipLaunchKernelGGL(msm_digits_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, scalars, n, cw, W, keys, hist);
    EZ_HIP(hipcub::DeviceScan::ExclusiveSum(S + o_cub, cub_bytes, hist, offs, (int)(nb + 1), st));
    EZ_HIP(hipMemcpyAsync(curs, offs, ((size_t)nb + 1) * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(msm_scatter_kernel, dim3(cdiv(npairs, 256)), dim3(256), 0, st, keys, n, W, base_offset, T->n, curs, vals);
    EZ_HIP(hipEventRecord(a0, st));
    hipLaunchKernelGGL(msm_accumulate_kernel, dim3(cdiv(nb, 256)), dim3(256), 0, st, T->tab, offs, vals, nb, bkt, heavy, hcnt);
    EZ_HIP(hipEventRecord(a1, st));
    {   // heavy buckets (skewed witnesses): block-stride over the device-side list, no host round trip
        size_t max_heavy = npairs / MSM_HEAVY + 1;
        unsigned hb = (unsigned)(max_heavy < (size_t)c->num_cus * 4 ? max_heavy : (size_t)c->num_cus * 4);
        hipLaunchKernelGGL(msm_heavy_kernel, dim3(hb), dim3(256), 0, st, T->tab, offs, vals, heavy, hcnt, bkt);
    }
    hipLaunchKernelGGL(msm_chunk_reduce_kernel, dim3(cdiv(nchunks, 256)), dim3(256), 0, st, bkt, nb, red, nchunks);
    // tree-sum nchunks partials down to one
    g1x_t *src = red, *dst = red2;
    uint32_t cur = nchunks;
    while (cur > 1) {
        uint32_t per = cur >= 256 * 64 ? 8 : (cur > 256 ? 2 : 1);
        uint32_t blocks = cdiv(cur, 256 * per);
        hipLaunchKernelGGL(msm_sum_kernel, dim3(blocks), dim3(256), 0, st, src, cur, dst, per);
        g1x_t* tmp = src; src = dst; dst = tmp;
        cur = blocks;
    }
    hipLaunchKernelGGL(msm_finalize_kernel, dim3(1), dim3(64), 0, st, src, dout);
    EZ_HIP(hipGetLastError());
    EZ_HIP(hipEventRecord(m1, st));
    EZ_HIP(hipMemcpyAsync(out_host, dout, 64, hipMemcpyDeviceToHost, st));
    EZ_HIP(hipStreamSynchronize(st));
    return EZKL_OK;
}

// ---- synthetic bases (bench / tests): same deterministic function as oracle_gen_bases -----------
__device__ __forceinline__ uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void msm_gen_bases_kernel(uint64_t seed, size_t first, size_t n, g1a_t* out) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t sqrt_e[8] = BN32_FQ_SQRT_EXP_INIT;
    uint64_t st = seed ^ ((uint64_t)(first + k) * 0xd1342543de82ef95ull);
    fe_t x;
    for (int j = 0; j < 4; j++) {
        uint64_t w = splitmix64(st);
        x.v[2 * j] = (uint32_t)w;
        x.v[2 * j + 1] = (uint32_t)(w >> 32);
    }
    x.v[7] &= 0x3fffffffu;
    x = Fq::reduce_once(x);
    const fe_t b3 = fr_const(FqConst::B3);
    for (;;) {
        fe_t rhs = Fq::add(Fq::mul(Fq::sqr(x), x), b3);
        fe_t y = Fq::one(), base = rhs;
        for (int i = 0; i < 254; i++) {
            if ((sqrt_e[i >> 5] >> (i & 31)) & 1) y = Fq::mul(y, base);
            base = Fq::sqr(base);
        }
        if (Fq::eq(Fq::sqr(y), rhs) && !Fq::is_zero(y)) {
            if (y.v[0] & 1) y = Fq::neg(y);
            g1a_t p;
            p.x = x;
            p.y = y;
            st_g1a(out + k, p);
            return;
        }
        x = Fq::add(x, Fq::one());
    }
}
int gen_bases(Ctx* c, hipStream_t st, uint64_t seed, size_t first, size_t n, void* out_dev) {
    if (n == 0) return EZKL_OK;
    hipLaunchKernelGGL(msm_gen_bases_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, seed, first, n, (g1a_t*)out_dev);
    EZ_HIP(hipGetLastError());
    EZ_HIP(hipStreamSynchronize(st));
    return EZKL_OK;
}

void g1_add_affine_host(const void* a, const void* b, void* out) {
    g1a_t p, q;
    memcpy(&p, a, 64);
    memcpy(&q, b, 64);
    g1a_t r = g1x_to_affine(g1x_add_mixed(g1x_from_affine(p), q));
    memcpy(out, &r, 64);
}

}  // namespace ezkl
