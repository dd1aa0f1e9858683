// vecops.hip -- element-wise Fr kernels on device-resident columns (the icicle "vec-ops" surface the
// halo2 fork uses between MSM/NTT calls, SURVEY.md §2 kernel inventory) plus divide_by_vanishing_poly
// and Montgomery batch inversion (mv-lookup / permutation helpers).
// All are streaming kernels: one 32-byte element per lane per access (2 x global_load_dwordx4),
// grid-stride, HBM-bound except batch inversion.
#include "common.hpp"
#include "field29.hpp"
#include <string.h>

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

// sigma column of the permutation argument from the cycle successor map: out[r] = delta^(t >> log_n) * omega^(t & (n - 1)), t = next[r]
__global__ __launch_bounds__(256) void perm_sigma_kernel(const uint32_t* next, const fe_t* omega_col, const fe_t* delta_pows, uint32_t log_n, uint32_t m,
                                                         fe_t* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t t = next[i], c = t >> log_n, r = t & (uint32_t)(n - 1);
        st_fe(out + i, c < m ? Fr::mul(ld_fe(omega_col + r), ld_fe(delta_pows + c)) : Fr::zero());     // a successor outside the m columns: no read
    }
}
__global__ __launch_bounds__(256) void vec_fill_kernel(fe_t* o, fe_t v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) st_fe(o + i, v);
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
int vec_fill(Ctx* c, hipStream_t st, fe_t* o, const fe_t& v, size_t n) {
    if (n == 0) return EZKL_OK;
    hipLaunchKernelGGL(vec_fill_kernel, dim3(stream_grid(c, n)), dim3(256), 0, st, o, v, n);
    EZ_HIP(hipGetLastError());
    return EZKL_OK;
}
int perm_sigma(Ctx* c, hipStream_t st, const uint32_t* next, const fe_t* omega_col, const fe_t* delta_pows, uint32_t log_n, uint32_t m, fe_t* out) {
    const size_t n = (size_t)1 << log_n;
    hipLaunchKernelGGL(perm_sigma_kernel, dim3(stream_grid(c, n)), dim3(256), 0, st, next, omega_col, delta_pows, log_n, m, out, n);
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
    int rc = arena_reserve(c->aux, period * sizeof(fe_t), st, (void**)&dt);
    if (rc) return rc;
    EZ_HIP(hipMemcpyAsync(dt, t.data(), period * sizeof(fe_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(vec_periodic_mul_kernel, dim3(stream_grid(c, ne)), dim3(256), 0, st, a, dt, (uint32_t)(period - 1), ne);
    EZ_HIP(hipGetLastError());
    if ((rc = arena_done(c->aux, st))) return rc;
    EZ_HIP(hipStreamSynchronize(st));          // `t` (pageable host memory) is copied asynchronously
    return EZKL_OK;
}

// Montgomery batch inversion.  T threads; thread t owns elements {t + j*T}: a coalesced strided chain.
// forward: pre[t + j*T] = prod_{i<j} a_i (zeros skipped); invert the chain product once (Fermat);
// backward: a_j^-1 = acc * pre_j ; acc *= a_j.
// Thread t owns the strided chain a[t], a[t + T], ...: prefix products forward, then the 256 chain products of a workgroup are combined
// (an inclusive prefix scan and an inclusive suffix scan of them in LDS: 1 / acc_t = 1 / total * prefix_(t-1) * suffix_(t+1)), ONE lane
// of the workgroup runs the Fermat inversion of the total, and every chain walks back.  Until round 4 every chain ended in its own
// inversion (~380 dependent products), which capped the launch at one wave per SIMD -- more chains would have bought shorter chains with
// more inversions -- and left a 16-column batch of a k = 20 proof at 1.25 ms of mostly dependent-issue latency
// (profiles/r04au_events.csv); one inversion per 256 chains lets four waves per SIMD share the issue slots.
__global__ __launch_bounds__(256) void batch_invert_kernel(fe_t* a, fe_t* pre, size_t n, size_t T) {
    __shared__ fe_t pfx[256], sfx[256];
    __shared__ fe_t inv_total;
    const uint32_t tid = threadIdx.x;
    const size_t t = (size_t)blockIdx.x * blockDim.x + tid;
    fe_t acc = Fr::one();
    size_t last = t;
    const bool have = t < T && t < n;
    if (have)
        for (size_t i = t; i < n; i += T) {
            st_fe(pre + i, acc);
            fe_t x = ld_fe(a + i);
            if (!Fr::is_zero(x)) acc = Fr::mul(acc, x);
            last = i;
        }
    pfx[tid] = acc;
    sfx[tid] = acc;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {           // inclusive Hillis-Steele, prefix and suffix at once
        fe_t p = pfx[tid], q = sfx[tid];
        if (tid >= d) p = Fr::mul(pfx[tid - d], p);
        if (tid + d < 256) q = Fr::mul(q, sfx[tid + d]);
        __syncthreads();
        pfx[tid] = p;
        sfx[tid] = q;
        __syncthreads();
    }
    if (tid == 0) inv_total = Fr::inv(pfx[255]);       // no chain product is zero: zero elements are skipped
    __syncthreads();
    if (!have) return;
    acc = inv_total;
    if (tid > 0) acc = Fr::mul(acc, pfx[tid - 1]);
    if (tid < 255) acc = Fr::mul(acc, sfx[tid + 1]);
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
    // chains of >= 16 elements, at most four workgroups (16 waves) per CU: one Fermat inversion per workgroup, so the chains can be many
    size_t T = n / 16;
    size_t cap = (size_t)c->num_cus * 1024;
    if (T > cap) T = cap;
    if (T < 1) T = 1;
    fe_t* pre = nullptr;
    int rc = arena_reserve(c->aux, n * sizeof(fe_t), st, (void**)&pre);
    if (rc) return rc;
    hipLaunchKernelGGL(batch_invert_kernel, dim3(cdiv(T, 256)), dim3(256), 0, st, a, pre, n, T);
    EZ_HIP(hipGetLastError());
    return arena_done(c->aux, st);
}

// ---- prefix scan over Fr (grand product of the permutation argument, grand sum of mv-lookup: SURVEY §8(a) A13) ----
// Three-phase scan: each workgroup scans a 2048-element chunk (8 consecutive elements per lane serially, lane
// totals by a Hillis-Steele scan in LDS), chunk totals are scanned recursively, then chunk prefixes are applied.
// 3 field operations per element; for the product scan that is 3 Montgomery products per element.
static constexpr uint32_t SCAN_E = 8, SCAN_CHUNK = 256 * SCAN_E;

template <int OP> EZ_D fe_t scan_op(const fe_t& a, const fe_t& b) { return OP == EZKL_VEC_ADD ? Fr::add(a, b) : Fr::mul(a, b); }
template <int OP> EZ_D fe_t scan_id() { return OP == EZKL_VEC_ADD ? Fr::zero() : Fr::one(); }

template <int OP>
__global__ __launch_bounds__(256) void scan_chunk_kernel(const fe_t* in, fe_t* out, size_t n, fe_t* totals, int exclusive) {
    __shared__ fe_t sh[256];
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)threadIdx.x * SCAN_E;
    fe_t x[SCAN_E];
    fe_t run = scan_id<OP>();
#pragma unroll
    for (uint32_t e = 0; e < SCAN_E; e++) {
        x[e] = base + e < n ? ld_fe(in + base + e) : scan_id<OP>();
        run = scan_op<OP>(run, x[e]);
        x[e] = run;                                    // inclusive within the lane
    }
    sh[threadIdx.x] = run;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {           // inclusive Hillis-Steele over lane totals
        fe_t v = sh[threadIdx.x];
        if (threadIdx.x >= d) v = scan_op<OP>(sh[threadIdx.x - d], v);
        __syncthreads();
        sh[threadIdx.x] = v;
        __syncthreads();
    }
    const fe_t lane_prefix = threadIdx.x ? sh[threadIdx.x - 1] : scan_id<OP>();
    if (threadIdx.x == 255) st_fe(totals + blockIdx.x, sh[255]);
    fe_t prev = lane_prefix;
#pragma unroll
    for (uint32_t e = 0; e < SCAN_E; e++) {
        fe_t inc = scan_op<OP>(lane_prefix, x[e]);
        if (base + e < n) st_fe(out + base + e, exclusive ? prev : inc);
        prev = inc;
    }
}
template <int OP>
__global__ __launch_bounds__(256) void scan_apply_kernel(fe_t* out, size_t n, const fe_t* chunk_prefix) {
    const size_t chunk = blockIdx.x + 1;               // chunk 0 needs no prefix
    const fe_t p = ld_fe(chunk_prefix + chunk);
    for (uint32_t e = threadIdx.x; e < SCAN_CHUNK; e += 256) {
        size_t i = chunk * SCAN_CHUNK + e;
        if (i < n) st_fe(out + i, scan_op<OP>(p, ld_fe(out + i)));
    }
}
template <int OP>
static int scan_rec(Ctx* c, hipStream_t st, const fe_t* in, fe_t* out, size_t n, int exclusive, fe_t* scratch) {
    const unsigned chunks = cdiv(n, SCAN_CHUNK);
    fe_t* totals = scratch;                            // chunks entries, then exclusive-scanned in place
    hipLaunchKernelGGL(scan_chunk_kernel<OP>, dim3(chunks), dim3(256), 0, st, in, out, n, totals, exclusive);
    if (chunks > 1) {
        int rc = scan_rec<OP>(c, st, totals, totals, chunks, 1, scratch + (((size_t)chunks + 63) & ~(size_t)63));
        if (rc) return rc;
        hipLaunchKernelGGL(scan_apply_kernel<OP>, dim3(chunks - 1), dim3(256), 0, st, out, n, totals);
    }
    EZ_HIP(hipGetLastError());
    return EZKL_OK;
}
int prefix_scan(Ctx* c, hipStream_t st, int op, int exclusive, const fe_t* in, fe_t* out, size_t n) {
    if (n == 0) return EZKL_OK;
    size_t need = 0;
    for (size_t m = n; m > 1;) { m = (m + SCAN_CHUNK - 1) / SCAN_CHUNK; need += (m + 63) & ~(size_t)63; }
    fe_t* scratch = nullptr;
    int rc = arena_reserve(c->aux, (need + 64) * sizeof(fe_t), st, (void**)&scratch);
    if (rc) return rc;
    rc = op == EZKL_VEC_ADD ? scan_rec<EZKL_VEC_ADD>(c, st, in, out, n, exclusive, scratch)
                            : scan_rec<EZKL_VEC_MUL>(c, st, in, out, n, exclusive, scratch);
    if (rc) return rc;
    return arena_done(c->aux, st);
}

// ---- kate_division: q(X) = a(X) / (X - z) without the remainder (halo2_proofs::arithmetic::kate_division, the quotients of
// the KZG / SHPLONK openings): q[n-1] = 0, q[i-1] = a[i] + z q[i], i.e. q[i] = sum_{j>i} a[j] z^(j-i-1) -----------------------
// A descending first-order recurrence with a constant coefficient.  A workgroup owns KD_CHUNK consecutive coefficients, a
// thread KD_E of them: Horner inside the thread, a doubling scan of the thread totals with ratio r = z^KD_E across the
// workgroup, and the carry from the chunks above -- which is the SAME division applied to the chunk totals with Z = z^KD_CHUNK,
// so the host recurses.  The coefficients are read twice and written once (96 B per element; the formulation with resident
// columns of z^j and z^-(j+1) around a prefix sum moved ~400 B), ~4 products per element.
static constexpr uint32_t KD_E = 8, KD_CHUNK = 256 * KD_E;
struct KdPowers {
    fe_t z;             // the point
    fe_t lev[8];        // r^(2^l), r = z^KD_E
};
// I[t] = sum_{t' >= t} T_t' r^(t' - t) over the 256 thread totals (T_t = the thread's coefficients evaluated at z), with the
// carry c from the chunks above folded into the last thread as T_255 + r c
__device__ __forceinline__ void kd_load_and_scan(const fe_t* a, size_t n, size_t base, const KdPowers& pw, const fe_t& carry, fe_t (&x)[KD_E], fe_t* I) {
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (uint32_t e = 0; e < KD_E; e++) {
        const size_t i = base + (size_t)t * KD_E + e;
        x[e] = i < n ? ld_fe(a + i) : Fr::zero();
    }
    fe_t h = x[KD_E - 1];
#pragma unroll
    for (int e = (int)KD_E - 2; e >= 0; e--) h = Fr::add(x[e], Fr::mul(pw.z, h));
    if (t == 255) h = Fr::add(h, Fr::mul(pw.lev[0], carry));
    I[t] = h;
    __syncthreads();
#pragma unroll
    for (uint32_t l = 0; l < 8; l++) {                       // unrolled: pw.lev[l] with a run-time l goes through scratch
        const uint32_t d = 1u << l;
        fe_t up = Fr::zero();
        const bool have = t + d < 256;
        if (have) up = I[t + d];
        __syncthreads();
        if (have) I[t] = Fr::add(I[t], Fr::mul(pw.lev[l], up));
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void kate_totals_kernel(const fe_t* a, size_t n, KdPowers pw, fe_t* totals) {
    __shared__ fe_t I[256];
    fe_t x[KD_E];
    kd_load_and_scan(a, n, (size_t)blockIdx.x * KD_CHUNK, pw, Fr::zero(), x, I);
    if (threadIdx.x == 0) st_fe(totals + blockIdx.x, I[0]);
}
// carries[b] = q at the last coefficient of chunk b (null: a single chunk, nothing above it)
__global__ __launch_bounds__(256) void kate_apply_kernel(const fe_t* a, size_t n, KdPowers pw, const fe_t* carries, fe_t* out) {
    __shared__ fe_t I[256];
    fe_t x[KD_E];
    const size_t base = (size_t)blockIdx.x * KD_CHUNK;
    const fe_t carry = carries ? ld_fe(carries + blockIdx.x) : Fr::zero();
    kd_load_and_scan(a, n, base, pw, carry, x, I);
    const uint32_t t = threadIdx.x;
    fe_t q = t < 255 ? I[t + 1] : carry;                     // q at the thread's last coefficient: everything above it
#pragma unroll
    for (int e = (int)KD_E - 1; e >= 0; e--) {
        const size_t i = base + (size_t)t * KD_E + e;
        if (i < n) st_fe(out + i, q);
        q = Fr::add(x[e], Fr::mul(pw.z, q));                 // q[i-1] = a[i] + z q[i]
    }
}
static int kate_rec(Ctx* c, hipStream_t st, const fe_t* a, fe_t* out, size_t n, const fe_t& z, fe_t* scratch) {
    KdPowers pw;
    pw.z = z;
    fe_t r = z;
    for (uint32_t e = 1; e < KD_E; e <<= 1) r = Fr::sqr(r);                  // z^KD_E (KD_E is a power of two)
    for (int l = 0; l < 8; l++) { pw.lev[l] = r; r = Fr::sqr(r); }           // r ends as z^KD_CHUNK
    const unsigned chunks = cdiv(n, KD_CHUNK);
    fe_t* carries = nullptr;
    if (chunks > 1) {
        carries = scratch;
        hipLaunchKernelGGL(kate_totals_kernel, dim3(chunks), dim3(256), 0, st, a, n, pw, carries);
        int rc = kate_rec(c, st, carries, carries, chunks, r, scratch + (((size_t)chunks + 63) & ~(size_t)63));
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kate_apply_kernel, dim3(chunks), dim3(256), 0, st, a, n, pw, carries, out);
    EZ_HIP(hipGetLastError());
    return EZKL_OK;
}
int kate_division(Ctx* c, hipStream_t st, const fe_t* a, const fe_t& z, fe_t* out, size_t n) {
    if (n == 0) return EZKL_OK;
    size_t need = 0;
    for (size_t m = n; m > KD_CHUNK;) { m = (m + KD_CHUNK - 1) / KD_CHUNK; need += (m + 63) & ~(size_t)63; }
    fe_t* scratch = nullptr;
    int rc = arena_reserve(c->aux, (need + 64) * sizeof(fe_t), st, (void**)&scratch);
    if (rc) return rc;
    if ((rc = kate_rec(c, st, a, out, n, z, scratch))) return rc;
    return arena_done(c->aux, st);
}

// ---- mv-lookup multiplicities (A13: [UPSTREAM] mv_lookup::prover::prepare builds m(X) with a BTreeMap from table
//      value to its FIRST row, then counts every input occurrence) ----
// Open-addressing hash table over 256-bit keys in HBM: slot = first table row holding the key (atomicCAS to claim,
// atomicMin among equal keys); inputs probe the same sequence and atomically count into the owning row.
static constexpr uint32_t HT_EMPTY = 0xffffffffu;
EZ_D uint32_t ht_hash(const fe_t& v, uint32_t mask) {
    uint32_t h = v.v[0] * 0x9e3779b1u ^ v.v[1] * 0x85ebca77u ^ v.v[2] * 0xc2b2ae3du ^ v.v[3] * 0x27d4eb2fu ^ v.v[5] * 0x165667b1u ^ v.v[7];
    h ^= h >> 15;
    return h & mask;
}
// Every kernel serves a BATCH of lookup arguments (blockIdx.y): a proof builds one multiplicity column per argument, each call of the
// one-argument form was three sub-0.2 ms launches that leave most of the machine waiting on random 4- and 32-byte reads, and a k = 20 MLP
// proof spent 3.3 ms on its eight arguments one after the other (profiles/r04au_events.csv).  tables[l] / slots + l * cap / counts +
// l * cstride belong to argument l; input item y (blockIdx.y of the count pass) probes the table of argument which[y].
__global__ __launch_bounds__(256) void ht_build_kernel(const fe_t* const* tables, uint32_t usable, uint32_t* slots, uint32_t cap) {
    const fe_t* table = tables[blockIdx.y];
    slots += (size_t)blockIdx.y * cap;
    const uint32_t mask = cap - 1;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < usable;
    const fe_t key = live ? ld_fe(table + i) : Fr::zero();
    // Only the first row of a run of equal values goes to the table: a lookup table padded with one repeated tuple (ezkl pads
    // every table to the column height) would otherwise send a million lanes to ONE slot -- measured 3.4 ms of atomics for a
    // 2^15-row table in a 2^20-row column.  (The lane before me holds the previous row; lane 0 always proceeds.)
    bool same_as_prev = true;
#pragma unroll
    for (int q = 0; q < 8; q++) same_as_prev &= (__shfl_up(key.v[q], 1) == key.v[q]);
    if (!live || ((threadIdx.x & 63) != 0 && same_as_prev)) return;
    uint32_t h = ht_hash(key, mask);
    for (;;) {
        uint32_t cur = slots[h];
        if (cur == HT_EMPTY) {
            cur = atomicCAS(&slots[h], HT_EMPTY, i);
            if (cur == HT_EMPTY) return;                 // claimed
        }
        if (Fr::eq(ld_fe(table + cur), key)) {           // same value already present: keep the first row
            if (i < cur) atomicMin(&slots[h], i);        // (a table padded with one repeated value must not serialise here)
            return;
        }
        h = (h + 1) & mask;
    }
}
__global__ __launch_bounds__(256) void ht_count_kernel(const fe_t* const* inputs, const uint32_t* which, uint32_t rows, const fe_t* const* tables,
                                                       const uint32_t* slots, uint32_t cap, uint32_t* counts, uint32_t cstride, uint32_t* missing) {
    const uint32_t l = which[blockIdx.y], mask = cap - 1;
    const fe_t* input = inputs[blockIdx.y];
    const fe_t* table = tables[l];
    slots += (size_t)l * cap;
    counts += (size_t)l * cstride;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // target: the table row this lane's input equals (HT_EMPTY = not in the table), HT_EMPTY - 1 = lane past the end
    uint32_t target = HT_EMPTY - 1;
    if (i < rows) {
        const fe_t key = ld_fe(input + i);
        uint32_t h = ht_hash(key, mask);
        for (;;) {
            const uint32_t cur = slots[h];
            if (cur == HT_EMPTY) { target = HT_EMPTY; break; }
            if (Fr::eq(ld_fe(table + cur), key)) { target = cur; break; }
            h = (h + 1) & mask;
        }
    }
    // A witness column is mostly ONE value on the rows where its lookup is switched off (the table's first element, chip.rs:560-575)
    // and small digits elsewhere: a million lanes adding 1 to the same counter serialise (1.5 ms per 2^17-row column measured).  The
    // lanes of a wave that hit the same row add once: three leader rounds take out the common values, the rest go one by one.
    const uint32_t lane = threadIdx.x & 63;
    bool pending = target != HT_EMPTY - 1;
    for (int round = 0; round < 3; round++) {
        const uint64_t act = __ballot(pending);
        if (!act) break;
        const int leader = __ffsll((unsigned long long)act) - 1;
        const uint32_t lt = __shfl(target, leader);
        const uint64_t same = __ballot(pending && target == lt);
        if ((int)lane == leader) atomicAdd(lt == HT_EMPTY ? missing : &counts[lt], (uint32_t)__popcll(same));
        if (pending && target == lt) pending = false;
    }
    if (pending) atomicAdd(target == HT_EMPTY ? missing : &counts[target], 1u);
}
__global__ __launch_bounds__(256) void counts_to_fr_kernel(const uint32_t* counts, uint32_t n, uint32_t cstride, fe_t* const* outs) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = counts[(size_t)blockIdx.y * cstride + i];
    fe_t t = Fr::zero();
    t.v[0] = v;
    st_fe(outs[blockIdx.y] + i, v ? Fr::to_mont(t) : t);
}
// n_lookups arguments at once: item y of inputs (n_items of them, any number per argument) belongs to argument which[y]
int lookup_multiplicity_batch(Ctx* c, hipStream_t st, const fe_t* const* inputs, const uint32_t* which, uint32_t n_items, const fe_t* const* tables,
                              uint32_t n_lookups, uint32_t n_rows, uint32_t usable, fe_t* const* m_outs, uint32_t* missing_host, uint32_t* missing_dev) {
    // missing_dev (device u32, ACCUMULATED into): the stream-ordered form -- nothing comes back to the host, so the call does not
    // synchronise; the caller reads the counter once after queuing all its lookups
    if (usable > n_rows || n_lookups == 0 || n_lookups > 65535 || n_items > 65535) return EZKL_ERR_INVALID;
    for (uint32_t y = 0; y < n_items; y++)
        if (which[y] >= n_lookups) return EZKL_ERR_INVALID;
    uint32_t cap = 16;
    while (cap < 2 * (usable ? usable : 1)) cap <<= 1;
    const uint32_t cstride = n_rows;
    // pointer arrays first (8-byte aligned), then the u32 arrays: which[], slots, counts, the missing counter
    const size_t n_ptr = (size_t)n_items + 2 * (size_t)n_lookups;
    const size_t words = (size_t)n_items + (size_t)n_lookups * cap + (size_t)n_lookups * cstride + 1;
    uint8_t* d = nullptr;
    int rc = arena_reserve(c->aux, n_ptr * 8 + words * 4, st, (void**)&d);
    if (rc) return rc;
    const fe_t** d_inputs = reinterpret_cast<const fe_t**>(d);
    const fe_t** d_tables = d_inputs + n_items;
    fe_t** d_outs = const_cast<fe_t**>(d_tables + n_lookups);
    uint32_t* d_which = reinterpret_cast<uint32_t*>(d + n_ptr * 8);
    uint32_t *slots = d_which + n_items, *counts = slots + (size_t)n_lookups * cap, *missing = missing_dev ? missing_dev : counts + (size_t)n_lookups * cstride;
    // the caller's pointer / index arrays are temporaries (std::vector, ctypes arrays) and the stream-ordered form returns without
    // synchronising: they travel through ONE pinned block the library owns (the layout of the device block's head: inputs, tables, outs,
    // which), as the gate programs' arguments do (evalh.hip); without a block the copies are synchronised before returning
    const size_t head = n_ptr * 8 + (size_t)n_items * 4;
    void* stg = nullptr;
    uint8_t* H = staging_acquire(c, head, &stg);
    bool copies_pending_on_caller_memory = false;
    if (H) {
        if (n_items) memcpy(H, inputs, (size_t)n_items * 8);
        memcpy(H + (size_t)n_items * 8, tables, (size_t)n_lookups * 8);
        memcpy(H + ((size_t)n_items + n_lookups) * 8, m_outs, (size_t)n_lookups * 8);
        if (n_items) memcpy(H + n_ptr * 8, which, (size_t)n_items * 4);
        EZ_HIP(hipMemcpyAsync(d, H, head, hipMemcpyHostToDevice, st));
        if ((rc = staging_release(stg, st))) return rc;
    } else {
        if (n_items) {
            EZ_HIP(hipMemcpyAsync(d_inputs, inputs, (size_t)n_items * 8, hipMemcpyHostToDevice, st));
            EZ_HIP(hipMemcpyAsync(d_which, which, (size_t)n_items * 4, hipMemcpyHostToDevice, st));
        }
        EZ_HIP(hipMemcpyAsync(d_tables, tables, (size_t)n_lookups * 8, hipMemcpyHostToDevice, st));
        EZ_HIP(hipMemcpyAsync(d_outs, m_outs, (size_t)n_lookups * 8, hipMemcpyHostToDevice, st));
        copies_pending_on_caller_memory = true;
    }
    EZ_HIP(hipMemsetAsync(slots, 0xff, (size_t)n_lookups * cap * 4, st));
    EZ_HIP(hipMemsetAsync(counts, 0, ((size_t)n_lookups * cstride + (missing_dev ? 0 : 1)) * 4, st));
    if (usable) {
        hipLaunchKernelGGL(ht_build_kernel, dim3(cdiv(usable, 256), n_lookups), dim3(256), 0, st, d_tables, usable, slots, cap);
        if (n_items)
            hipLaunchKernelGGL(ht_count_kernel, dim3(cdiv(usable, 256), n_items), dim3(256), 0, st, d_inputs, (const uint32_t*)d_which, usable, d_tables,
                               (const uint32_t*)slots, cap, counts, cstride, missing);
    }
    hipLaunchKernelGGL(counts_to_fr_kernel, dim3(cdiv(n_rows, 256), n_lookups), dim3(256), 0, st, (const uint32_t*)counts, n_rows, cstride, (fe_t* const*)d_outs);
    hipError_t e = hipGetLastError();
    if (missing_dev) {
        if (e == hipSuccess && copies_pending_on_caller_memory) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return set_hip_error(e, "lookup_multiplicity", __FILE__, __LINE__);
        return arena_done(c->aux, st);
    }
    uint32_t miss = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&miss, missing, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return set_hip_error(e, "lookup_multiplicity", __FILE__, __LINE__);
    if ((rc = arena_done(c->aux, st))) return rc;
    if (missing_host) *missing_host = miss;
    return EZKL_OK;
}
int lookup_multiplicity(Ctx* c, hipStream_t st, const fe_t* const* inputs, uint32_t n_inputs, const fe_t* table, uint32_t n_rows,
                        uint32_t usable, fe_t* m_out, uint32_t* missing_host, uint32_t* missing_dev) {
    const std::vector<uint32_t> which(n_inputs ? n_inputs : 1, 0u);
    return lookup_multiplicity_batch(c, st, inputs, which.data(), n_inputs, &table, 1, n_rows, usable, &m_out, missing_host, missing_dev);
}

// ---- polynomial evaluation at a point (halo2 eval_polynomial: hundreds of O(n) Horner reductions per proof,
//      create_proof step 10 in SURVEY.md §3.1; A14) ----
// lane t evaluates its 32-coefficient segment by Horner, scales it by x^(32 t) (square-and-multiply on the lane
// index), workgroups tree-sum in LDS; a second launch of the same tree folds the per-workgroup partials.
static constexpr uint32_t EVP_SEG = 32;
// Horner over m <= 32 coefficients at x: the segment's 31 dependent products are the kernel, so they run on the radix-2^29 lazy limbs of
// field29.hpp (205 instructions a product against 281; x carries the 2^261 of that product: x_r261 = 32 x in the Montgomery domain of the
// data, which is left alone); one canonicalisation at the end of the segment brings the value back to 8 x 32-bit words
EZ_D fe_t evp_segment(const fe_t* c, uint32_t m, const fe_t& x_r261) {
    const f29_t x29 = Fr29::unpack(x_r261);
    f29_t acc = Fr29::unpack(ld_fe(c + m - 1));
    for (uint32_t j = m - 1; j-- > 0;) acc = Fr29::add(Fr29::mul(acc, x29), Fr29::unpack(ld_fe(c + j)));     // < 2p + p, limbs below 2 units
    return Fr29::pack(Fr29::canonical(Fr29::normalize(acc)));
}
__global__ __launch_bounds__(256) void eval_poly_kernel(const fe_t* coeffs, size_t n, fe_t x_r261, const fe_t* xpow2, uint32_t npow, fe_t* partial) {
    __shared__ fe_t sh[256];
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, base = t * EVP_SEG;
    fe_t acc = Fr::zero();
    if (base < n) {
        const uint32_t m = n - base < EVP_SEG ? (uint32_t)(n - base) : EVP_SEG;
        acc = evp_segment(coeffs + base, m, x_r261);
        // x^(32 t): bits of t select precomputed x^(32 * 2^b)
        for (uint32_t b = 0; b < npow; b++)
            if ((t >> b) & 1) acc = Fr::mul(acc, ld_fe(xpow2 + b));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = Fr::add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) st_fe(partial + blockIdx.x, sh[0]);
}
__global__ __launch_bounds__(256) void sum_kernel(const fe_t* in, size_t n, fe_t* out) {
    __shared__ fe_t sh[256];
    fe_t acc = Fr::zero();
    for (size_t i = threadIdx.x; i < n; i += 256) acc = Fr::add(acc, ld_fe(in + i));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = Fr::add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) st_fe(out, sh[0]);
}
int eval_poly(Ctx* c, hipStream_t st, const fe_t* coeffs, size_t n, const fe_t& x, void* out_host) {
    if (n == 0) { memset(out_host, 0, 32); return EZKL_OK; }
    const size_t nseg = (n + EVP_SEG - 1) / EVP_SEG;
    const unsigned blocks = cdiv(nseg, 256);
    uint32_t npow = 0;
    while (((size_t)1 << npow) < nseg) npow++;
    std::vector<fe_t> pw(npow ? npow : 1);
    fe_t p = x;
    for (int i = 0; i < 5; i++) p = Fr::sqr(p);          // x^32
    for (uint32_t b = 0; b < npow; b++) { pw[b] = p; p = Fr::sqr(p); }
    fe_t* d = nullptr;
    int rc = arena_reserve(c->aux, (pw.size() + blocks + 1) * sizeof(fe_t), st, (void**)&d);
    if (rc) return rc;
    fe_t *d_pw = d, *d_part = d + pw.size(), *d_out = d_part + blocks;
    EZ_HIP(hipMemcpyAsync(d_pw, pw.data(), pw.size() * sizeof(fe_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(eval_poly_kernel, dim3(blocks), dim3(256), 0, st, coeffs, n, Fr::mul(x, Fr::from_u64(32)), d_pw, npow, d_part);
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, d_part, (size_t)blocks, d_out);
    hipError_t e = hipMemcpyAsync(out_host, d_out, 32, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return set_hip_error(e, "eval_poly", __FILE__, __LINE__);
    return arena_done(c->aux, st);
}

// m evaluations (polynomial j at point xs[j], all of length n): ONE launch for all of them (blockIdx.y = j -- a single
// 2^20-coefficient polynomial only fills 128 workgroups), one upload of the x-power tables, one download, one stream
// synchronisation.  create_proof's step 10 issues ~50 of these.
struct EvalItem {
    const fe_t* coeffs;
    fe_t x;
};
__global__ __launch_bounds__(256) void eval_poly_multi_kernel(const EvalItem* items, size_t n, const fe_t* pws, uint32_t pw_stride, uint32_t npow,
                                                              fe_t* partial) {
    __shared__ fe_t sh[256];
    const EvalItem it = items[blockIdx.y];
    const fe_t* xpow2 = pws + (size_t)blockIdx.y * pw_stride;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, base = t * EVP_SEG;
    fe_t acc = Fr::zero();
    if (base < n) {
        const uint32_t m = n - base < EVP_SEG ? (uint32_t)(n - base) : EVP_SEG;
        acc = evp_segment(it.coeffs + base, m, it.x);
        for (uint32_t b = 0; b < npow; b++)
            if ((t >> b) & 1) acc = Fr::mul(acc, ld_fe(xpow2 + b));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = Fr::add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) st_fe(partial + (size_t)blockIdx.y * gridDim.x + blockIdx.x, sh[0]);
}
__global__ __launch_bounds__(256) void sum_multi_kernel(const fe_t* in, size_t per, fe_t* out) {
    __shared__ fe_t sh[256];
    const fe_t* src = in + (size_t)blockIdx.x * per;
    fe_t acc = Fr::zero();
    for (size_t i = threadIdx.x; i < per; i += 256) acc = Fr::add(acc, ld_fe(src + i));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = Fr::add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) st_fe(out + blockIdx.x, sh[0]);
}
int eval_poly_batch(Ctx* c, hipStream_t st, const fe_t* const* coeffs, const fe_t* xs, uint32_t m, size_t n, void* out_host) {
    if (m == 0) return EZKL_OK;
    if (n == 0) { memset(out_host, 0, 32 * (size_t)m); return EZKL_OK; }
    const size_t nseg = (n + EVP_SEG - 1) / EVP_SEG;
    const unsigned blocks = cdiv(nseg, 256);
    uint32_t npow = 0;
    while (((size_t)1 << npow) < nseg) npow++;
    const size_t pws = npow ? npow : 1;
    // one staging buffer: [items | power tables], so a single upload
    const size_t items_fe = ((size_t)m * sizeof(EvalItem) + sizeof(fe_t) - 1) / sizeof(fe_t);
    std::vector<fe_t> host(items_fe + pws * m);
    EvalItem* items = reinterpret_cast<EvalItem*>(host.data());
    for (uint32_t j = 0; j < m; j++) {
        items[j].coeffs = coeffs[j];
        items[j].x = Fr::mul(xs[j], Fr::from_u64(32));       // x in the 2^261 domain of the segment's radix-2^29 Horner (evp_segment)
        fe_t p = xs[j];
        for (int i = 0; i < 5; i++) p = Fr::sqr(p);          // x^32
        for (uint32_t b = 0; b < npow; b++) { host[items_fe + j * pws + b] = p; p = Fr::sqr(p); }
    }
    fe_t* d = nullptr;
    int rc = arena_reserve(c->aux, (host.size() + (size_t)m * blocks + m) * sizeof(fe_t), st, (void**)&d);
    if (rc) return rc;
    fe_t *d_pw = d + items_fe, *d_part = d + host.size(), *d_out = d_part + (size_t)m * blocks;
    EZ_HIP(hipMemcpyAsync(d, host.data(), host.size() * sizeof(fe_t), hipMemcpyHostToDevice, st));
    for (uint32_t j0 = 0; j0 < m; j0 += 65535) {             // gridDim.y limit
        const uint32_t mj = m - j0 < 65535 ? m - j0 : 65535;
        hipLaunchKernelGGL(eval_poly_multi_kernel, dim3(blocks, mj), dim3(256), 0, st, reinterpret_cast<const EvalItem*>(d) + j0, n, d_pw + (size_t)j0 * pws,
                           (uint32_t)pws, npow, d_part + (size_t)j0 * blocks);
    }
    hipLaunchKernelGGL(sum_multi_kernel, dim3(m), dim3(256), 0, st, d_part, (size_t)blocks, d_out);
    hipError_t e = hipMemcpyAsync(out_host, d_out, 32 * (size_t)m, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return set_hip_error(e, "eval_poly_batch", __FILE__, __LINE__);
    return arena_done(c->aux, st);
}

// ---- fused linear combination: out[i] (+)= sum_j coeff[j] * in_j[i]  (SHPLONK's per-rotation-set combinations and
//      final L(X), the x^n-Horner over the quotient pieces): every input is read once, the output written once,
//      instead of a scale + add pass (160 B of traffic per element) per term ----
static constexpr uint32_t LINCOMB_MAX = 16;
struct LincombArgs {
    const fe_t* in[LINCOMB_MAX];
    fe_t coeff[LINCOMB_MAX];
    uint32_t m;
    int accumulate;
};
__global__ __launch_bounds__(256) void lincomb_kernel(LincombArgs a, fe_t* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        fe_t acc = a.accumulate ? ld_fe(out + i) : Fr::zero();
        for (uint32_t j = 0; j < a.m; j++) acc = Fr::add(acc, Fr::mul(ld_fe(a.in[j] + i), a.coeff[j]));
        st_fe(out + i, acc);
    }
}
int lincomb(Ctx* c, hipStream_t st, const fe_t* const* in, const fe_t* coeffs, uint32_t m, fe_t* out, size_t n, int accumulate) {
    if (n == 0) return EZKL_OK;
    if (m == 0 && !accumulate) return vec_fill(c, st, out, Fr::zero(), n);
    for (uint32_t j0 = 0; j0 < m; j0 += LINCOMB_MAX) {
        LincombArgs a;
        a.m = m - j0 < LINCOMB_MAX ? m - j0 : LINCOMB_MAX;
        a.accumulate = (accumulate || j0 > 0) ? 1 : 0;
        for (uint32_t j = 0; j < a.m; j++) { a.in[j] = in[j0 + j]; a.coeff[j] = coeffs[j0 + j]; }
        hipLaunchKernelGGL(lincomb_kernel, dim3(stream_grid(c, n)), dim3(256), 0, st, a, out, n);
    }
    EZ_HIP(hipGetLastError());
    return EZKL_OK;
}

// ---- uniform field elements from ChaCha20 (blinding rows, the vanishing argument's random polynomial: halo2 draws them
//      from OsRng on the host, 32 MiB per random polynomial at k = 20; here the host supplies a 256-bit key and the keystream
//      is expanded where the column lives) ----
// DJB layout: words 12,13 = 64-bit block counter, words 14,15 = 64-bit stream id.  Element i owns blocks 16 i .. 16 i + 15;
// candidate j (0..31) = 8 words of block j >> 1 (upper half for odd j), top two bits cleared; the first candidate < r wins
// (rejection sampling: uniform on [0, r), acceptance 0.76 per try; all 32 failing -- probability 1e-20 -- yields 0).
__host__ __device__ inline void chacha20_block(const uint32_t key[8], uint64_t counter, uint64_t stream, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t x[16];
    for (int i = 0; i < 16; i++) x[i] = s[i];
#define EZ_ROTL(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define EZ_QR(a, b, c, d) \
    a += b; d ^= a; d = EZ_ROTL(d, 16); c += d; b ^= c; b = EZ_ROTL(b, 12); a += b; d ^= a; d = EZ_ROTL(d, 8); c += d; b ^= c; b = EZ_ROTL(b, 7);
    for (int r = 0; r < 10; r++) {
        EZ_QR(x[0], x[4], x[8], x[12]) EZ_QR(x[1], x[5], x[9], x[13]) EZ_QR(x[2], x[6], x[10], x[14]) EZ_QR(x[3], x[7], x[11], x[15])
        EZ_QR(x[0], x[5], x[10], x[15]) EZ_QR(x[1], x[6], x[11], x[12]) EZ_QR(x[2], x[7], x[8], x[13]) EZ_QR(x[3], x[4], x[9], x[14])
    }
#undef EZ_QR
#undef EZ_ROTL
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
struct ChachaKey {
    uint32_t k[8];
};
__global__ __launch_bounds__(256) void chacha20_fr_kernel(ChachaKey key, uint64_t stream, size_t first, fe_t* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        fe_t v = Fr::zero();
        bool done = false;
        for (uint32_t b = 0; b < 16 && !done; b++) {
            uint32_t blk[16];
            chacha20_block(key.k, (uint64_t)(first + i) * 16 + b, stream, blk);
            for (uint32_t h = 0; h < 2 && !done; h++) {
                fe_t cand;
                for (int q = 0; q < 8; q++) cand.v[q] = blk[8 * h + q];
                cand.v[7] &= 0x3fffffffu;
                uint32_t br = 0;
                for (int q = 0; q < 8; q++) (void)subb32(cand.v[q], FrP::MOD[q], br);
                if (br) { v = cand; done = true; }          // cand < r
            }
        }
        st_fe(out + i, v);
    }
}
int chacha20_fr(Ctx* c, hipStream_t st, const uint32_t key[8], uint64_t stream, size_t first, fe_t* out, size_t n) {
    if (n == 0) return EZKL_OK;
    ChachaKey k;
    memcpy(k.k, key, 32);
    hipLaunchKernelGGL(chacha20_fr_kernel, dim3(stream_grid(c, n)), dim3(256), 0, st, k, stream, first, out, n);
    EZ_HIP(hipGetLastError());
    return EZKL_OK;
}

}  // namespace ezkl
