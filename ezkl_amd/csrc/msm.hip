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
//   * window plan    : W signed-digit windows of BALANCED widths covering 254 bits (2^20 points: 13 windows of 20 / 19 bits),
//                      chosen by the cost model n*W + 4*#buckets.
//   * hist / scans / partition : scalar -> canonical -> (s > r/2 ? r - s, negated) -> signed digits (the r - s trick turns
//                      witness-like small negative values, src/fieldutils.rs:9-17, into one-digit scalars).  No global atomics:
//                      every workgroup histograms its slice in LDS, a column scan gives (workgroup, partition) start slots, the
//                      partition pass ranks its (bucket, table index | sign) pairs in 104 KiB of LDS by the LOW bucket bits and
//                      writes them out in staged order; bucket 0 (+-1 digits, carries) is a partition of its own.
//   * binsort        : one workgroup per partition, counting sort on the remaining bucket bits in LDS, contiguous output, bucket
//                      offsets + empty-bucket marks; oversized partitions (skewed witnesses) go through bigsort_{count,scatter}.
//   * accumulate     : balanced SEGMENTED lanes -- lane t owns sorted pairs [t*L, (t+1)*L) whatever the bucket boundaries, L
//                      chosen on the device from the number of pairs that exist; 64-byte table records gathered (next record
//                      prefetched under the current add) and mixed-added (8M + 2S, radix-2^29 lazy limbs, curve29.hpp) into an
//                      XYZZ accumulator in VGPRs: the dominant kernel.  fixup_boundary folds buckets cut by lane boundaries
//                      (one thread per boundary), fixup_heavy1/2 the buckets cut more than 16 times.
//   * reduce         : sum_b weight(b) * B_b with the position split into three bit-fields: reduce1 (column sums), reduce2 (row
//                      sums, <= 1 wave per SIMD), planes (per-bit plane sums, one workgroup per plane); the last ~40 dependent
//                      point operations (plane Horner + to-affine) run on the host (host64.hpp): 0.5 us per link there, ~9 us on
//                      one GPU wave.
//   * batches        : msm_run_batch pipelines independent MSMs over stream slots so the latency-bound sort / reduce tails of
//                      one overlap the accumulation of the next.
// Order of additions differs from the CPU Pippenger, the group element (and its canonical affine bytes) does not.
#include "common.hpp"
#include <deque>
#include <thread>
#include <chrono>
#include "curve.hpp"
#include "curve29.hpp"
#include "host64.hpp"
#include <string.h>

namespace ezkl {

// (The compile-time variants of rounds 4-5 -- the loop without the one-iteration-ahead loads, the fused "lean" chain, unpack-first, the full
// accumulator reset, conditional gathers -- were measured and removed; their A/B logs are profiles/r05q_msm_ab.log, r05y_msm_ab.log and
// DESIGN.md §4.1.  What is here is the one shipped form.)
static constexpr uint32_t MSM_MAX_PART_BITS = 12;   // <= 4096 partitions in the first sorting pass (1024 up to 2^20 points: msm_part_bits)
static constexpr uint32_t MSM_SPAN_HEAVY = 16;      // buckets cut by more lane boundaries than this are folded by a whole workgroup
static constexpr uint32_t MSM_PART_STAGE = 13312;     // pairs a partition workgroup stages in LDS (104 KiB): 1024 scalars x 13 windows
static constexpr uint32_t MSM_BINSORT_STAGE = 15360;  // payloads a sort workgroup stages in LDS (60 KiB): 2 workgroups per CU
static constexpr uint32_t MSM_HEAVY_CHUNK = 256;     // lane partials folded by one WAVE in the first heavy pass (four serial additions per lane + the wave tree)
static constexpr uint32_t MSM_DIGIT_E = 8;          // serial elements per lane in the first reduce stage
static constexpr uint32_t MSM_LMIN = 8;             // shortest lane of the accumulate kernel
static constexpr uint32_t MSM_MAX_BIG = 64;         // oversized partitions sorted by several workgroups each (the rest: one workgroup)
static constexpr uint32_t MSM_BIG_BLOCKS = 64;      // workgroups per oversized partition
static constexpr uint32_t MSM_BIG_ROWS = 8;         // rows of such workgroups in a launch: row r takes the oversized partitions r, r + 8, ...

// Partition of a bucket in the first sorting pass: its low bits -- except bucket 0, which gets a partition of its own
// (index 0; ordinary partition p is index p + 1).  Bucket 0 holds the digits +-1: every carry of the signed recoding into an
// otherwise empty window, every boolean, every one.  For a witness of 20-bit values that is 2^19 pairs in ONE bucket, and the
// workgroup sorting its partition used to walk them alone (0.89 of that MSM's 1.87 ms).  A one-bucket partition needs no
// sorting: the partition pass writes its payloads straight to their final place.
__device__ __forceinline__ uint32_t msm_part_of(uint32_t bucket, uint32_t NP) { return bucket ? 1u + (bucket & (NP - 1u)) : 0u; }
// Lane length of the accumulate kernel, decided ON THE DEVICE from the number of pairs that actually exist: the host sizes
// the launch for n * W pairs, but small witness values have one or two non-zero digits, and with the host's lane length
// such a column filled a third of the CUs with one wave each (0.45 ms for 1.5 M pairs; 1.1 ms for 13.6 M).
__device__ __forceinline__ uint32_t msm_lane_len(const uint32_t* offsets, uint32_t nb, uint32_t nlanes, uint32_t lmin) {
    const uint32_t l = (uint32_t)(((uint64_t)offsets[nb] + nlanes - 1) / nlanes);
    return l < lmin ? lmin : l;
}

// Window plan: W signed-digit windows covering 254 bits (253-bit magnitudes after the r - s fold + the last carry),
// the first `rem` windows base+1 bits wide, the others base bits.  Balanced widths instead of "c, c, ..., short top
// window": a 14-bit top window under c = 20 pours 128 extra pairs into each of 2^13 buckets, which then get cut by
// several lane boundaries of the accumulate kernel; with 7 x 20 + 6 x 19 bits no bucket outgrows a lane.
// Batched launches: one launch serves `gridDim.z` MSMs of the same size whose scratch slabs have the same layout `bstride` bytes
// apart; a kernel shifts every scratch pointer it is given by blockIdx.z * bstride (bstride = 0 / gridDim.z = 1: a single MSM).
template <class T> __device__ __forceinline__ T* bshift(T* p, size_t bytes) {
    return reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + bytes);
}
#define BSH(p) p = bshift(p, _bo)
#define BOFF() const size_t _bo = (size_t)blockIdx.z * bstride

struct WinPlan {
    uint32_t W, base, rem;
    __host__ __device__ uint32_t width(uint32_t w) const { return base + (w < rem ? 1u : 0u); }
    __host__ __device__ uint32_t offset(uint32_t w) const { return w * base + (w < rem ? w : rem); }
    __host__ __device__ uint32_t cmax() const { return base + (rem ? 1u : 0u); }
};
struct MsmTable {
    g1a_t* tab = nullptr;    // W x n affine: T[w][i] = 2^offset(w) * P_i
    WinPlan wp{0, 0, 0};
    size_t n = 0;
    hipEvent_t ready = nullptr;   // recorded after the build
    bool synced = false;          // a host wait on `ready` has happened
};
// ---- per-context state: window tables, stream slots, the open upload phase / batch (Ctx::msm_state; one per context, so that the
// contexts of a single-process multi-GPU prover never share device objects).  The names below are what the code uses.
struct MsmSlot;
struct MsmUpload;
struct MsmBatch;
struct MsmState;
static MsmState& msm_state();
#define g_tables (msm_state().tables)
#define g_table_stream (msm_state().table_stream)
#define g_slots (msm_state().slots)
#define g_call_slots (msm_state().call_slots)
#define g_call_claimed (msm_state().call_claimed)
#define g_call_finishing (msm_state().call_finishing)
#define g_call_owner (msm_state().call_owner)
#define g_call_order_ev (msm_state().call_order_ev)
#define g_copy_st (msm_state().copy_st)
#define g_tail_pinned (msm_state().tail_pinned)
#define g_tail_pinned_elems (msm_state().tail_pinned_elems)
#define g_open_upload (msm_state().open_upload)
#define g_open_batch (msm_state().open_batch)

// One in-flight MSM: its own scratch arena, a pinned landing buffer for the plane sums and a completion event.
// A batch (one commit phase of the prover) is pipelined over MSM_SLOTS of these on separate streams so that the
// latency-bound sort / reduce tails of one MSM and the host Horner overlap the accumulate kernel of the next.
static constexpr int MSM_MAX_SLOTS = 16;
// how many groups of a batch are in flight: EZKL_MSM_SLOTS (1..16) overrides the default.  Four since round 4 (six before): with commit
// batches fused in groups of 4 / 6 a phase has two or three groups, and every slot is one more stream next to the library, side, copy and
// call-slot streams for the runtime to fold onto its 8 hardware queues (k = 20 MLP, early random commitment, groups of 1 / 2 / 4: 74.1-75.0
// ms with four slots, 75.4-77.7 with three, 74.1-76.2 with two, 73.1-77.2 with six: profiles/r04u_*, r04v_slots_ab.log)
static int msm_slots_init() {
    int v = 4;
    if (const char* e = getenv("EZKL_MSM_SLOTS")) {
        const int x = atoi(e);
        if (x >= 1 && x <= MSM_MAX_SLOTS) v = x;
    }
    return v;
}
static const int MSM_SLOTS = msm_slots_init();
struct MsmSlot {
    hipStream_t st = nullptr;
    uint8_t* scratch = nullptr;
    size_t scratch_bytes = 0;
    uint32_t* pinned = nullptr;       // per MSM of the group: 1 + 22 planes of 36 limbs (g1x29_t), 32 records apart
    void* pinned_dev = nullptr;       // the same buffer as the device addresses it (hipHostGetDevicePointer): the planes kernel writes into it
    size_t pinned_msms = 0;
    const fe_t** list_pinned = nullptr;   // the group's scalar-column pointers, staged for the device
    hipEvent_t done = nullptr;
    uint32_t bits = 0;
    uint32_t count = 0;               // MSMs in flight in this slot (one fused group)
    void* out = nullptr;              // where msm_finish puts the group's `count` affine results (64 B each)
    bool busy = false;
};
static constexpr int MSM_CALL_SLOTS = 4;
struct MsmState {
    std::map<const Bases*, MsmTable> tables;
    hipStream_t table_stream = nullptr;
    MsmSlot slots[MSM_MAX_SLOTS];
    MsmSlot call_slots[MSM_CALL_SLOTS];
    bool call_claimed[MSM_CALL_SLOTS] = {false, false, false, false};
    bool call_finishing[MSM_CALL_SLOTS] = {false, false, false, false};   // a finish is running outside the lock: a second finish of the token is refused
    std::thread::id call_owner[MSM_CALL_SLOTS];
    hipEvent_t call_order_ev = nullptr;
    hipStream_t copy_st = nullptr;
    fe_t* tail_pinned = nullptr;
    size_t tail_pinned_elems = 0;
    MsmUpload* open_upload = nullptr;
    MsmBatch* open_batch = nullptr;
    bool attrs_set = false;
    int acc_blocks_per_cu = 0;
};
static MsmState& msm_state() {
    Ctx* c = ctx();                   // the calling thread's context: every user of this state runs under that context's lock (EZ_CTX)
    if (!c->msm_state) c->msm_state = new MsmState();
    return *static_cast<MsmState*>(c->msm_state);
}


static WinPlan pick_plan(size_t n) {
    // cost model in point additions: n*W bucket additions + ~4 per bucket for the reduce phase (2^(cmax-1) buckets);
    // ties go to the smaller W (fewer gathers, smaller table).  n = 2^20: W = 13 (7 x 20 + 6 x 19 bits, 26 pairs/bucket).
    WinPlan best{0, 0, 0};
    double best_cost = 0;
    for (uint32_t W = 10; W <= 127; W++) {
        WinPlan p{W, 254 / W, 254 % W};
        if (p.cmax() > 23) continue;
        const double cost = (double)n * W + 4.0 * (double)((size_t)1 << (p.cmax() - 1));
        if (!best.W || cost < best_cost) { best = p; best_cost = cost; }
    }
    return best;
}

// ---- table precompute: T[w] = 2^width(w-1) * T[w-1] --------------------------------------------
__global__ __launch_bounds__(256) void msm_precompute_kernel(const g1a_t* prev, g1a_t* next, size_t n, uint32_t c) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    g1a_t p = ld_g1a(prev + i);
    g1x_t a = g1x_from_affine(p);
    for (uint32_t k = 0; k < c; k++) a = g1x_double(a);
    st_g1a(next + i, g1x_to_affine(a));
}

// the accumulate kernel works in Montgomery form R' = 2^261 (field29.hpp); the table is private to the MSM, so it is
// converted once: coordinate * (2^261 mod p) / 2^256 in the old arithmetic = the canonical value of x * 2^261
__global__ __launch_bounds__(256) void msm_table_to_r261_kernel(g1a_t* tab, size_t count) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    fe_t k;
#pragma unroll
    for (int q = 0; q < 8; q++) k.v[q] = Fq29C::R261_32[q];
    g1a_t p = ld_g1a(tab + i);
    if (g1a_is_id(p)) return;
    p.x = Fq::mul(p.x, k);
    p.y = Fq::mul(p.y, k);
    st_g1a(tab + i, p);
}

// ---- signed-digit decomposition ------------------------------------------------------------------
__device__ __forceinline__ bool gt_half_r(const fe_t& s) {      // s > (r-1)/2  <=>  2s >= r
    uint32_t d[8], c = 0, br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = addc32(s.v[i], s.v[i], c);
#pragma unroll
    for (int i = 0; i < 8; i++) (void)subb32(d[i], FrP::MOD[i], br);
    return c != 0 || br == 0;
}
// canonical scalar, negated into [0, (r-1)/2] when that is shorter: s*P = (r-s)*(-P).  Small negative
// witness values (src/fieldutils.rs:9-17) thereby become single-digit scalars.
__device__ __forceinline__ fe_t msm_canon(const fe_t* scalars, size_t i, uint32_t& neg) {
    fe_t s = Fr::from_mont(ld_fe(scalars + i));
    neg = 0;
    if (gt_half_r(s)) {
        uint32_t br = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) s.v[k] = subb32(FrP::MOD[k], s.v[k], br);
        neg = 1;
    }
    return s;
}
// the low c bits of s (c < 32), then s >>= c.  The windows are consumed by SHIFTING the scalar down: indexing its limbs with a
// run-time window offset (s.v[off >> 5]) put every scalar of the sort passes into scratch memory (144 B per lane in the
// histogram pass) and each digit cost two scratch loads.
__device__ __forceinline__ uint32_t msm_take_bits(fe_t& s, uint32_t c) {
    const uint32_t d = s.v[0] & ((1u << c) - 1u);
#pragma unroll
    for (int k = 0; k < 7; k++) s.v[k] = __builtin_amdgcn_alignbit(s.v[k + 1], s.v[k], c);
    s.v[7] >>= c;
    return d;
}
// calls f(w, bucket, sign) for every non-zero signed c-bit digit (bucket = |digit| - 1)
template <class F>
__device__ __forceinline__ void msm_foreach_digit(fe_t s, uint32_t neg, const WinPlan& wp, F&& f) {
    uint32_t carry = 0;
    for (uint32_t w = 0; w < wp.W; w++) {
        const uint32_t c = wp.width(w), half = 1u << (c - 1);
        uint32_t raw = msm_take_bits(s, c) + carry;
        if (raw > half) {                       // negative digit raw - 2^c, carry into the next window
            carry = 1;
            if (raw != (1u << c)) f(w, (1u << c) - raw - 1u, neg ^ 1u);   // raw == 2^c: digit 0 with a carry
        } else {
            carry = 0;
            if (raw) f(w, raw - 1u, neg);
        }
    }
}

// one window of the recoding above (consumed from s), for loops that interleave several scalars: returns whether the digit is non-zero
__device__ __forceinline__ bool msm_digit_step(fe_t& s, uint32_t neg, uint32_t c, uint32_t& carry, uint32_t& bucket, uint32_t& sign) {
    const uint32_t half = 1u << (c - 1);
    const uint32_t raw = msm_take_bits(s, c) + carry;
    if (raw > half) {
        carry = 1;
        bucket = (1u << c) - raw - 1u;
        sign = neg ^ 1u;
        return raw != (1u << c);
    }
    carry = 0;
    bucket = raw - 1u;
    sign = neg;
    return raw != 0;
}

// inclusive prefix sum of one value per thread over the workgroup (blockDim.x a multiple of 64, at most 1024): shuffles inside the waves, the
// <= 16 wave totals through `wsum` (>= 16 words of LDS), three barriers in all -- the Hillis-Steele loops this replaces took two barriers per
// doubling step (20 for the 1024 threads of the partition pass, whose 16 waves are alone on their CU).  total = the sum over the workgroup.
__device__ __forceinline__ uint32_t msm_block_scan(uint32_t v, uint32_t* wsum, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    if (wave == 0) {
        uint32_t s = lane < nw ? wsum[lane] : 0u;
#pragma unroll
        for (uint32_t d = 1; d < 16; d <<= 1) {
            const uint32_t y = __shfl_up(s, d, 64);
            if (lane >= d) s += y;
        }
        if (lane < nw) wsum[lane] = s;
    }
    __syncthreads();
    total = wsum[nw - 1];
    const uint32_t r = x + (wave ? wsum[wave - 1] : 0u);
    __syncthreads();                                   // wsum may be reused by the caller
    return r;
}

// ---- sort pass 1: partition (bucket, payload) pairs by the LOW bits of the bucket id -------------
// (low bits are uniformly populated even when the top window or a skewed witness concentrates the bucket
// values in a small numeric range, so the partitions stay balanced).  Buckets are stored at position
// pos = (bucket & (NP-1)) * 2^LB + (bucket >> PB); the reduce phase weights positions accordingly.
// No global atomics: every workgroup histograms its slice of scalars in LDS and stores the row; a column scan turns the
// rows into per-(workgroup, partition) start slots; the partition pass ranks its pairs with LDS atomics into an LDS
// staging area grouped by partition and writes them out in staged order (runs of one partition leave as whole cache lines).
__global__ __launch_bounds__(256) void msm_hist_kernel(const fe_t* scalars, size_t n, size_t per_block, WinPlan wp,
                                                       uint32_t LB, uint32_t NP, uint32_t* wg_hist, uint32_t* wg_cnt,
                                                       const fe_t* const* scal_list, uint32_t* zero_base, uint32_t zero_words, size_t bstride) {
    BOFF(); BSH(wg_hist); BSH(wg_cnt);
    if (scal_list) scalars = scal_list[blockIdx.z];
    if (zero_base) {                                                      // the chain's counters, bin totals and planes start at zero
        BSH(zero_base);
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < zero_words; i += gridDim.x * 256) zero_base[i] = 0;
    }
    __shared__ uint32_t lh[(1u << MSM_MAX_PART_BITS) + 1];
    const uint32_t NQ = NP + 1;                                           // + the bucket-0 partition
    for (uint32_t p = threadIdx.x; p < NQ; p += 256) lh[p] = 0;
    __syncthreads();
    size_t lo = (size_t)blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
    for (size_t i0 = lo + threadIdx.x; i0 < hi; i0 += 4 * 256) {          // four scalars in flight per thread
        fe_t s[4];
        uint32_t neg[4], carry[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const size_t i = i0 + (size_t)q * 256;
            if (i < hi) s[q] = msm_canon(scalars, i, neg[q]);
            else { s[q] = Fr::zero(); neg[q] = 0; }
        }
        for (uint32_t w = 0; w < wp.W; w++) {
            const uint32_t c = wp.width(w);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t bucket, sign;
                if (msm_digit_step(s[q], neg[q], c, carry[q], bucket, sign)) atomicAdd(&lh[msm_part_of(bucket, NP)], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < NQ; p += 256) {                 // row of this workgroup, coalesced; wg_hist is scanned in place later,
        wg_hist[(size_t)blockIdx.x * NQ + p] = lh[p];                   // wg_cnt keeps the raw counts for the partition pass's local ranking
        wg_cnt[(size_t)blockIdx.x * NQ + p] = lh[p];
    }
}
// wg_hist[g][p] (G workgroups x NP partitions) -> in place, the exclusive prefix over g of column p; part_count[p] = column
// total.  One workgroup per 32 columns: thread (c, j) sums the j-th chunk of G/32 rows of column c (a row segment of 32
// columns is one 128-byte line), the 32 chunk sums of a column are scanned in LDS, then the rows are rewritten.
__global__ __launch_bounds__(1024) void msm_hist_scan_kernel(uint32_t* wg_hist, uint32_t G, uint32_t NP, uint32_t* part_count, size_t bstride) {
    BOFF(); BSH(wg_hist); BSH(part_count);
    __shared__ uint32_t sums[32][33];
    const uint32_t c = threadIdx.x & 31, j = threadIdx.x >> 5;
    const uint32_t p = blockIdx.x * 32 + c;
    const uint32_t chunk = (G + 31) / 32, g0 = j * chunk, g1 = g0 + chunk < G ? g0 + chunk : G;
    uint32_t s = 0;
    if (p < NP)
        for (uint32_t g = g0; g < g1; g++) s += wg_hist[(size_t)g * NP + p];
    sums[j][c] = s;
    __syncthreads();
    if (j == 0) {                                  // 32 threads, one per column: serial exclusive scan over the 32 chunks
        uint32_t run = 0;
        for (uint32_t q = 0; q < 32; q++) {
            uint32_t v = sums[q][c];
            sums[q][c] = run;
            run += v;
        }
        if (p < NP) part_count[p] = run;
    }
    __syncthreads();
    if (p < NP) {
        uint32_t run = sums[j][c];
        for (uint32_t g = g0; g < g1; g++) {
            uint32_t v = wg_hist[(size_t)g * NP + p];
            wg_hist[(size_t)g * NP + p] = run;
            run += v;
        }
    }
}
// exclusive scan of <= 5120 partition counts (five per thread); part_base[NQ] = the number of pairs.  Also lists the oversized
// ordinary partitions (big_flag[p] = 1 + slot, big_list[slot] = p, big_count[0] = how many asked for a slot).
__device__ __forceinline__ void msm_part_scan_body(const uint32_t* part_count, uint32_t NQ, uint32_t* part_base, uint32_t* big_flag,
                                                   uint32_t* big_list, uint32_t* big_count, uint32_t* sh) {
    const uint32_t t = threadIdx.x;
    constexpr uint32_t PER = 5;                                          // 5 x 1024 threads >= 4097 partitions
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (uint32_t q = 0; q < PER; q++) {
        const uint32_t i = PER * t + q;
        v[q] = i < NQ ? part_count[i] : 0;
        sum += v[q];
    }
    uint32_t total;
    uint32_t run = msm_block_scan(sum, sh, total) - sum;
#pragma unroll
    for (uint32_t q = 0; q < PER; q++) {
        const uint32_t i = PER * t + q;
        if (i < NQ) part_base[i] = run;
        run += v[q];
    }
    if (t == 1023) part_base[NQ] = total;
#pragma unroll
    for (uint32_t q = 0; q < PER; q++) {
        const uint32_t i = PER * t + q;
        if (i == 0 || i >= NQ) continue;                                 // index 0 is bucket 0's partition: never sorted
        uint32_t slot = 0;
        if (v[q] > MSM_BINSORT_STAGE) {
            slot = atomicAdd(big_count, 1u);
            if (slot < MSM_MAX_BIG) big_list[slot] = i - 1;
            slot = slot < MSM_MAX_BIG ? slot + 1 : 0;
        }
        big_flag[i - 1] = slot;
    }
}
__global__ __launch_bounds__(1024) void msm_part_scan_kernel(const uint32_t* part_count, uint32_t NQ, uint32_t* part_base, uint32_t* big_flag,
                                                             uint32_t* big_list, uint32_t* big_count, size_t bstride) {
    BOFF(); BSH(part_count); BSH(part_base); BSH(big_flag); BSH(big_list); BSH(big_count);
    __shared__ uint32_t sh[1024];
    msm_part_scan_body(part_count, NQ, part_base, big_flag, big_list, big_count, sh);
}
// One pass, one scalar per thread.  A workgroup first ranks its (up to MSM_PART_STAGE) pairs into LDS grouped by partition
// (start[p] = exclusive scan of its own histogram row, cursors advanced with LDS atomics), then writes them out in staged order:
// consecutive lanes hold consecutive pairs of the same partition, i.e. consecutive global slots
//     slot = part_base[p] + (pairs of partition p in earlier workgroups) + (index - start[p]).
// The 13.6 M pairs of a 2^20-point MSM used to leave as 13.6 M scattered 8-byte stores, one L2 request each (~100 of the
// kernel's 146 us); staged, a run of ~13 pairs of one partition is two cache lines.
__global__ __launch_bounds__(1024) void msm_partition_kernel(const fe_t* scalars, size_t n, size_t per_block, WinPlan wp,
                                                             uint32_t LB, uint32_t NP, size_t base_offset, size_t tab_stride,
                                                             const uint32_t* part_base, const uint32_t* wg_hist, const uint32_t* wg_cnt, uint2* entries,
                                                             uint32_t* vals, const fe_t* const* scal_list, size_t bstride) {
    BOFF(); BSH(part_base); BSH(wg_hist); BSH(wg_cnt); BSH(entries); BSH(vals);
    if (scal_list) scalars = scal_list[blockIdx.z];
    extern __shared__ uint32_t plds[];
    __shared__ uint32_t tsum[1024];
    const uint32_t NQ = NP + 1;                // partitions incl. bucket 0's (msm_part_of); the arrays below are padded to NQ + 1 words
    uint32_t* start = plds;                    // NQ: first staged index of partition p
    uint32_t* cursor = plds + (NQ + 1);        // NQ: next free staged index
    uint32_t* gbase = plds + 2 * (NQ + 1);     // NQ: global slot of the workgroup's first pair of partition p
    uint2* stage = reinterpret_cast<uint2*>(plds + 3 * (NQ + 1));
    const uint32_t t = threadIdx.x, T = blockDim.x;
    // exclusive scan of this workgroup's histogram row: K consecutive partitions per thread, then a scan over threads
    const uint32_t K = (NQ + T - 1) / T;
    uint32_t loc = 0;
    for (uint32_t q = 0; q < K; q++) {
        const uint32_t p = t * K + q;
        if (p < NQ) loc += wg_cnt[(size_t)blockIdx.x * NQ + p];
    }
    uint32_t total;
    uint32_t run = msm_block_scan(loc, tsum, total) - loc;
    for (uint32_t q = 0; q < K; q++) {
        const uint32_t p = t * K + q;
        if (p < NQ) {
            start[p] = run;
            cursor[p] = run;
            gbase[p] = part_base[p] + wg_hist[(size_t)blockIdx.x * NQ + p];
            run += wg_cnt[(size_t)blockIdx.x * NQ + p];
        }
    }
    __syncthreads();
    const size_t lo = (size_t)blockIdx.x * per_block, hi = lo + per_block < n ? lo + per_block : n;
    const size_t i = lo + t;
    if (i < hi) {
        uint32_t neg;
        fe_t s = msm_canon(scalars, i, neg);
        msm_foreach_digit(s, neg, wp, [&](uint32_t w, uint32_t bucket, uint32_t sign) {
            const uint32_t r = atomicAdd(&cursor[msm_part_of(bucket, NP)], 1u);
            stage[r] = make_uint2((uint32_t)(w * tab_stride + base_offset + i) | (sign << 31), bucket);
        });
    }
    __syncthreads();
    const uint32_t PB = 31 - __clz(NP);
    for (uint32_t idx = t; idx < total; idx += T) {
        const uint2 e = stage[idx];
        const uint32_t p = msm_part_of(e.y, NP);
        const uint32_t slot = gbase[p] + (idx - start[p]);
        if (p) entries[slot] = make_uint2(e.x, e.y >> PB);
        else vals[slot] = e.x;                              // bucket 0: already in its final place (its partition comes first)
    }
}
// atomicAdd(&cnt[key], 1) for every active lane, with ONE atomic when the whole wave holds the same key (a heavy bucket:
// 2^19 increments of one LDS word serialise otherwise)
__device__ __forceinline__ uint32_t msm_wave_rank(uint32_t* cnt, uint32_t key) {
    const uint64_t active = __ballot(1);
    const uint32_t first = __builtin_amdgcn_readfirstlane(key);
    if (__ballot(key == first) == active) {                          // wave-uniform branch
        const uint32_t rank = (uint32_t)__popcll(active & ((1ull << (threadIdx.x & 63u)) - 1ull));
        uint32_t base = 0;
        if (rank == 0) base = atomicAdd(&cnt[first], (uint32_t)__popcll(active));
        return (uint32_t)__builtin_amdgcn_readfirstlane(base) + rank;
    }
    return atomicAdd(&cnt[key], 1u);
}
// ---- sort pass 2: one workgroup per partition, counting sort on the low bucket bits in LDS ---------
// cnt[0..nbins) counts -> exclusive bases in place (512 threads, <= 2048 bins); with `emit` also the partition's bucket offsets
// and the empty-bucket marks (an empty bucket is the identity, ZZ = 0: no 75 MB memset of the whole bucket array)
__device__ __forceinline__ void msm_bins_scan(uint32_t* cnt, uint32_t* tsum, uint32_t nbins, uint32_t p, uint32_t beg, const uint32_t* part_base,
                                              bool emit, uint32_t* offsets, g1x29_t* buckets) {
    const uint32_t t = threadIdx.x;
    uint32_t loc[4], s = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {                        // 4 consecutive bins per thread
        uint32_t j = t * 4 + q;
        loc[q] = j < nbins ? cnt[j] : 0;
        s += loc[q];
    }
    uint32_t total;
    uint32_t run = msm_block_scan(s, tsum, total) - s;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t j = t * 4 + q;
        if (j < nbins) {
            cnt[j] = run;
            if (emit) {
                const bool b0 = (p == 0 && j == 0);                                        // bucket 0 lives in partition index 0
                offsets[(size_t)p * nbins + j] = b0 ? part_base[0] : beg + run;
                if (b0 ? part_base[1] == part_base[0] : loc[q] == 0) st_f29(&buckets[(size_t)p * nbins + j].zz, Fq29::zero());
            }
        }
        run += loc[q];
    }
    __syncthreads();
}
__device__ __forceinline__ void msm_bigsort_count_body(const uint2* entries, const uint32_t* part_base, uint32_t LB, const uint32_t* big_list,
                                                       const uint32_t* big_count, uint32_t* bin_total, uint32_t* block_off, uint32_t* cnt, uint32_t bx,
                                                       uint32_t by, uint32_t rows);
__global__ __launch_bounds__(512) void msm_binsort_kernel(const uint2* entries, const uint32_t* part_base, uint32_t LB, uint32_t NP,
                                                          const uint32_t* big_flag, uint32_t* offsets, uint32_t* vals, g1x29_t* buckets,
                                                          const uint32_t* big_list, const uint32_t* big_count, uint32_t* bin_total, uint32_t* block_off,
                                                          size_t bstride) {
    BOFF(); BSH(entries); BSH(part_base); BSH(big_flag); BSH(offsets); BSH(vals); BSH(buckets);
    __shared__ uint32_t cnt[2048];
    __shared__ uint32_t tsum[512];
    extern __shared__ uint32_t stage[];                      // MSM_BINSORT_STAGE sorted payloads: written out as one contiguous run
    if (blockIdx.x >= NP) {                                  // the counting half of the multi-workgroup sort of oversized partitions
        BSH(big_list); BSH(big_count); BSH(bin_total); BSH(block_off);
        const uint32_t w = blockIdx.x - NP;
        msm_bigsort_count_body(entries, part_base, LB, big_list, big_count, bin_total, block_off, cnt, w % MSM_BIG_BLOCKS, w / MSM_BIG_BLOCKS, MSM_BIG_ROWS);
        return;
    }
    const uint32_t p = blockIdx.x, t = threadIdx.x, nbins = 1u << LB;
    const uint32_t beg = part_base[p + 1], end = part_base[p + 2];       // index 0 is bucket 0's own partition (msm_part_of)
    if (p == NP - 1 && t == 0) offsets[(size_t)NP * nbins] = end;
    if (big_flag[p]) return;                                             // oversized: msm_bigsort_{count,scatter}_kernel
    const bool staged = (end - beg) <= MSM_BINSORT_STAGE;                // uniform over the workgroup
    for (uint32_t j = t; j < nbins; j += 512) cnt[j] = 0;
    __syncthreads();
    if (staged) {
        // A partition of ordinary size (<= MSM_BINSORT_STAGE = 30 x 512 pairs) is read ONCE: every thread keeps its <= 30 pairs in registers
        // between the counting pass and the ranking pass (round 5; the two passes used to read the 109 MB of pairs twice), all loads of a
        // thread in flight at once.  It is ranked into LDS and leaves as one contiguous run (scattered 4-byte stores cost one L2 request each).
        constexpr int PER = MSM_BINSORT_STAGE / 512;
        uint2 v[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const uint32_t e = beg + t + (uint32_t)q * 512u;
            v[q] = e < end ? entries[e] : make_uint2(0u, 0xffffffffu);
        }
#pragma unroll
        for (int q = 0; q < PER; q++)
            if (v[q].y != 0xffffffffu) atomicAdd(&cnt[v[q].y], 1u);
        __syncthreads();
        msm_bins_scan(cnt, tsum, nbins, p, beg, part_base, true, offsets, buckets);
#pragma unroll
        for (int q = 0; q < PER; q++)
            if (v[q].y != 0xffffffffu) stage[atomicAdd(&cnt[v[q].y], 1u)] = v[q].x;
        __syncthreads();
        for (uint32_t i = t; i < end - beg; i += 512) vals[beg + i] = stage[i];
    } else {
        // An oversized partition that did not get a slot in the multi-workgroup sort below (more than MSM_MAX_BIG of them) is ranked straight
        // into HBM, a whole wave of equal keys with one LDS atomic (msm_wave_rank).
        for (uint32_t e = beg + t; e < end; e += 512) (void)msm_wave_rank(cnt, entries[e].y);
        __syncthreads();
        msm_bins_scan(cnt, tsum, nbins, p, beg, part_base, true, offsets, buckets);
        for (uint32_t e = beg + t; e < end; e += 512) {
            const uint2 v = entries[e];
            vals[beg + msm_wave_rank(cnt, v.y)] = v.x;
        }
    }
}
// Oversized partitions (skew: a column with long constant runs -- a permutation product over rows without copy constraints,
// a padded column -- puts most of its pairs into one bucket per window, i.e. into one partition each; a single workgroup
// walking 2^20 pairs took 1.3-1.5 ms).  They are sorted by MSM_BIG_BLOCKS workgroups each, in two kernels: every workgroup
// histograms its slice and reserves, per bin, a range inside the bin with ONE global atomic (the order of pairs inside a bucket
// is irrelevant); after the kernel boundary the bin totals are complete, every workgroup scans them and scatters its slice.
__device__ __forceinline__ void msm_big_slice(uint32_t beg, uint32_t end, uint32_t& s0, uint32_t& s1, uint32_t bx) {
    const uint32_t chunk = (end - beg + MSM_BIG_BLOCKS - 1) / MSM_BIG_BLOCKS;
    s0 = beg + bx * chunk;
    s1 = s0 + chunk < end ? s0 + chunk : end;
    if (s0 > end) s0 = end;
}
// (round 5) both kernels run MSM_BIG_BLOCKS x MSM_BIG_ROWS workgroups that LOOP over the oversized partitions (row r takes partitions r,
// r + MSM_BIG_ROWS, ...): the launch of an MSM without any (the common case) is 512 workgroups that read one counter and leave, not 4096,
// and a skewed column still has two workgroups per CU at work.
// the counting half, for workgroup (bx, by) of MSM_BIG_BLOCKS x `rows`: since round 5 these workgroups ride at the END of the binsort launch
// (they need nothing but the partition scan's output, like the binsort workgroups), one launch fewer per MSM
__device__ __forceinline__ void msm_bigsort_count_body(const uint2* entries, const uint32_t* part_base, uint32_t LB, const uint32_t* big_list,
                                                       const uint32_t* big_count, uint32_t* bin_total, uint32_t* block_off, uint32_t* cnt, uint32_t bx,
                                                       uint32_t by, uint32_t rows) {
    const uint32_t t = threadIdx.x, nbins = 1u << LB;
    const uint32_t nbig = big_count[0] < MSM_MAX_BIG ? big_count[0] : MSM_MAX_BIG;
    for (uint32_t y = by; y < nbig; y += rows) {
        const uint32_t p = big_list[y];
        uint32_t s0, s1;
        msm_big_slice(part_base[p + 1], part_base[p + 2], s0, s1, bx);
        for (uint32_t j = t; j < nbins; j += 512) cnt[j] = 0;
        __syncthreads();
        for (uint32_t e = s0 + t; e < s1; e += 512) (void)msm_wave_rank(cnt, entries[e].y);
        __syncthreads();
        for (uint32_t j = t; j < nbins; j += 512)                            // this workgroup's range inside bin j starts at the returned value
            block_off[((size_t)y * MSM_BIG_BLOCKS + bx) * nbins + j] = cnt[j] ? atomicAdd(&bin_total[(size_t)y * nbins + j], cnt[j]) : 0u;
        __syncthreads();
    }
}
__global__ __launch_bounds__(512) void msm_bigsort_scatter_kernel(const uint2* entries, const uint32_t* part_base, uint32_t LB, const uint32_t* big_list,
                                                                  const uint32_t* big_count, const uint32_t* bin_total, const uint32_t* block_off,
                                                                  uint32_t* offsets, uint32_t* vals, g1x29_t* buckets, size_t bstride) {
    BOFF(); BSH(entries); BSH(part_base); BSH(big_list); BSH(big_count); BSH(bin_total); BSH(block_off); BSH(offsets); BSH(vals); BSH(buckets);
    __shared__ uint32_t cnt[2048];
    __shared__ uint32_t tsum[512];
    const uint32_t t = threadIdx.x, nbins = 1u << LB;
    const uint32_t nbig = big_count[0] < MSM_MAX_BIG ? big_count[0] : MSM_MAX_BIG;
    for (uint32_t y = blockIdx.y; y < nbig; y += gridDim.y) {
        const uint32_t p = big_list[y], beg = part_base[p + 1];
        uint32_t s0, s1;
        msm_big_slice(beg, part_base[p + 2], s0, s1, blockIdx.x);
        for (uint32_t j = t; j < nbins; j += 512) cnt[j] = bin_total[(size_t)y * nbins + j];
        __syncthreads();
        msm_bins_scan(cnt, tsum, nbins, p, beg, part_base, blockIdx.x == 0, offsets, buckets);     // cnt[j] = start of bin j in the partition
        const uint32_t* mine = block_off + ((size_t)y * MSM_BIG_BLOCKS + blockIdx.x) * nbins;      // + this workgroup's range inside each bin
        for (uint32_t j = t; j < nbins; j += 512) cnt[j] += mine[j];
        __syncthreads();
        for (uint32_t e = s0 + t; e < s1; e += 512) {
            const uint2 v = entries[e];
            vals[beg + msm_wave_rank(cnt, v.y)] = v.x;
        }
        __syncthreads();
    }
}

// ---- bucket accumulation (dominant kernel) ------------------------------------------------------
// Perfectly balanced segmented accumulation: lane t owns sorted pairs [t*L, (t+1)*L) regardless of bucket
// boundaries.  A bucket that lies inside one lane's range is written directly; a bucket cut by a lane
// boundary leaves partial sums in tail[t] (continues into lane t+1) / head[t] (started before lane t), which
// msm_fixup_boundary_kernel folds.  Each pair costs one 64-byte gather from T and one mixed add (8M + 2S).
// a gathered table record: 64 bytes, canonical coordinates in R' = 2^261 Montgomery form (msm_table_to_r261_kernel); the
// sign bit of the pair selects -P, applied inside the mixed addition
struct MsmRec {
    g1a_t p;
    bool neg;
};
__device__ __forceinline__ MsmRec msm_fetch(const g1a_t* tab, uint32_t v) {
    MsmRec r;
    r.p = ld_g1a(tab + (v & 0x7fffffffu));
    r.neg = (v >> 31) != 0;
    return r;
}
__global__ __launch_bounds__(256, 3) void msm_accumulate_kernel(const g1a_t* tab, const uint32_t* offsets, const uint32_t* vals,
                                                             uint32_t nb, uint32_t nlanes, g1x29_t* buckets, g1x29_t* head, g1x29_t* tail,
                                                             uint32_t* lane_first, uint32_t lmin, size_t bstride) {
    BOFF(); BSH(offsets); BSH(vals); BSH(buckets); BSH(head); BSH(tail); BSH(lane_first);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = offsets[nb], L = msm_lane_len(offsets, nb, nlanes, lmin);
    const uint64_t k0w = (uint64_t)t * L;
    if (k0w >= total) return;
    const uint32_t k0 = (uint32_t)k0w;
    const uint32_t k1 = (uint64_t)k0 + L < total ? k0 + L : total;
    uint32_t lo = 0, hi = nb;                   // offsets[lo] <= k0 < offsets[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= k0) lo = mid; else hi = mid;
    }
    uint32_t b = lo, bin_end = offsets[b + 1];
    // everything the next iteration needs is loaded ONE ITERATION AHEAD: the end of the bucket after this one, the payload of the pair
    // after next (the address of the next gather) and the next table record -- no load sits on the path of the iteration that uses it
    uint32_t end_next = offsets[b + 2 <= nb ? b + 2 : nb];
    lane_first[t] = b;                          // the bucket holding this lane's first pair (msm_fixup_boundary_kernel)
    bool started_before = offsets[b] < k0;
    g1x29_t acc = g1x29_identity();
    MsmRec nxt = msm_fetch(tab, vals[k0]);
    uint32_t vn = k0 + 1 < k1 ? vals[k0 + 1] : 0u;
    // (round 5, measured and dropped: two pairs per loop iteration with the roles of the two record registers swapped -- 41 register
    //  copies fewer per pair, but the 5.7 k-instruction body ran 2-10 % SLOWER (instruction cache); a "predicated" mixed addition that
    //  computes the common path for every lane and overwrites the rare cases: slower as well.  profiles/r05q_msm_ab.log)
    for (uint32_t k = k0; k < k1; k++) {
        if (k == bin_end) {                     // bucket b is finished inside this lane
            if (started_before) st_g1x29(head + t, acc); else st_g1x29(buckets + b, acc);
            acc.zz = Fq29::zero();              // the identity is ZZ = 0 whatever X, Y, ZZZ hold (g1x29_is_id): 9 moves instead of 36 at every bucket end
            started_before = false;
            b++;
            bin_end = end_next;
            // an empty bucket follows: a few linear probes (the common case), then a binary search -- a sparse column
            // (e.g. m(X): a handful of blinding rows scattered over 2^19 buckets) must not walk every empty bucket
            if (bin_end == k) {
                uint32_t probes = 1;
                while (bin_end == k && probes < 4) { b++; bin_end = offsets[b + 1]; probes++; }
                if (bin_end == k) {
                    uint32_t lo2 = b + 1, hi2 = nb;          // offsets[lo2] == k, offsets[hi2] = total > k
                    while (hi2 - lo2 > 1) {
                        uint32_t mid = (lo2 + hi2) >> 1;
                        if (offsets[mid] <= k) lo2 = mid; else hi2 = mid;
                    }
                    b = lo2;
                    bin_end = offsets[b + 1];
                }
            }
            end_next = offsets[b + 2 <= nb ? b + 2 : nb];
        }
        const MsmRec cur = nxt;
        // no divergent region around the loads (round 5: the `if (k + 1 < k1)` around them cost 3.7 % of the kernel, 0.916 -> 0.886 ms,
        // profiles/r05y_msm_ab.log): past the lane's last pair a record is fetched and never used
        nxt = msm_fetch(tab, vn);
        vn = k + 2 < k1 ? vals[k + 2] : 0u;
        acc = g1x29_add_mixed(acc, g1a29_unpack(cur.p), cur.neg);
    }
    if (k1 == bin_end) {
        if (started_before) st_g1x29(head + t, acc); else st_g1x29(buckets + b, acc);
    } else {
        if (started_before) st_g1x29(head + t, acc); else st_g1x29(tail + t, acc);
    }
}
// A bucket cut by lane boundaries is tail[t1] + head[t1+1 .. t2].  One thread per LANE BOUNDARY (not per bucket: with
// ~26 pairs per bucket and ~69 per lane nearly every boundary cuts a bucket, so the waves are full).  The common case,
// a bucket cut once, is one addition; a bucket cut a few times is folded serially by its first boundary; anything
// longer (a skewed witness) is queued for msm_fixup_heavy{1,2}_kernel.
// The partials of a bucket are summed by a SEGMENTED TREE inside the wave (round 5; the serial kernel of rounds 1-4 is gone: profiles/r05ar_tree{0,1}_msm_columns_serial.txt).  A witness-shaped column (small values: one
// non-zero digit per scalar, lanes of 8 pairs) cuts almost every bucket 5-16 times, and the thread of a bucket's first boundary then folded up
// to 16 partials one after the other while its neighbours -- the bucket's other boundaries -- idled: 180-290 us per column
// (profiles/r04o_msm_columns_serial.txt).  Here the 64 boundaries of a wave run a segmented suffix sum: log2(longest run in the wave)
// shuffle-and-add steps, every lane adding the lane d to its right when that lane belongs to the same bucket; the first boundary of a bucket
// then holds the sum of all its partials inside the wave, adds the bucket's tail and -- when the bucket runs past the end of the wave -- the
// few partials beyond it one by one.  A wave whose buckets are all cut once takes no step at all: the uniform case costs what it did.
EZ_D g1x29_t g1x29_shfl_down(const g1x29_t& p, uint32_t d) {
    g1x29_t r;
    const uint32_t* s = p.x.v;
    uint32_t* o = r.x.v;
#pragma unroll
    for (int k = 0; k < 36; k++) o[k] = __shfl_down(s[k], d);
    return r;
}
__global__ __launch_bounds__(256) void msm_fixup_boundary_tree_kernel(const uint32_t* offsets, uint32_t nb, uint32_t nlanes,
                                                                      const uint32_t* lane_first, const g1x29_t* head, const g1x29_t* tail, g1x29_t* buckets,
                                                                      uint32_t* heavy_list, uint32_t* heavy_count, uint32_t* chunk_list, uint32_t lmin,
                                                                      uint32_t span_heavy, size_t bstride) {
    BOFF(); BSH(offsets); BSH(lane_first); BSH(head); BSH(tail); BSH(buckets); BSH(heavy_list); BSH(heavy_count); BSH(chunk_list);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x + 1;          // boundary between lanes t-1 and t
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t L = msm_lane_len(offsets, nb, nlanes, lmin);
    const uint64_t k0 = (uint64_t)t * L;
    bool cut = t < nlanes && k0 < offsets[nb];
    uint32_t b = 0, t1 = 0, t2 = 0;
    if (cut) {
        b = lane_first[t];
        const uint32_t beg = offsets[b], end = offsets[b + 1];
        cut = beg < k0;                                                    // otherwise the bucket starts exactly on the boundary
        t1 = beg / L; t2 = (end - 1) / L;
    }
    if (cut && t2 - t1 > span_heavy) {                                     // skewed witness: the whole-workgroup passes below
        if (t == t1 + 1 && t2 - t1 > MSM_HEAVY_CHUNK) heavy_list[atomicAdd(heavy_count, 1u)] = b;     // more than one chunk: the second pass
        if ((t - t1 - 1) % MSM_HEAVY_CHUNK == 0) chunk_list[atomicAdd(heavy_count + 1, 1u)] = t;
        cut = false;
    }
    // what is left of my bucket from this boundary on, inside this wave (0: not a boundary of a bucket that is folded here)
    const uint32_t wave_last = t + (63u - lane);
    const uint32_t run = cut ? (t2 < wave_last ? t2 : wave_last) - t + 1u : 0u;
    uint32_t longest = run;
#pragma unroll
    for (uint32_t o = 32; o > 0; o >>= 1) {
        const uint32_t m = __shfl_xor(longest, o);
        longest = m > longest ? m : longest;
    }
    if (longest == 0) return;                                              // wave-uniform
    if (longest <= 2) {                                                    // wave-uniform: nothing in this wave is cut more than twice -- the serial fold
        if (!cut || t != t1 + 1) return;                                   // (uniform scalars: 26 pairs per bucket, 69 per lane)
        g1x29_t acc = g1x29_add(ld_g1x29(tail + t1), ld_g1x29(head + t));
        for (uint32_t u = t + 1; u <= t2; u++) acc = g1x29_add(acc, ld_g1x29(head + u));
        st_g1x29(buckets + b, acc);
        return;
    }
    g1x29_t v = cut ? ld_g1x29(head + t) : g1x29_identity();
    uint32_t have = cut ? 1u : 0u;                                         // partials of my bucket summed into v so far (from t rightwards)
#pragma unroll 1
    for (uint32_t d = 1; d < longest; d <<= 1) {                          // wave-uniform trip count
        const g1x29_t w = g1x29_shfl_down(v, d);
        const uint32_t hw = __shfl_down(have, d);
        // the lane d to my right continues my run exactly when I already hold d partials and my run is longer than that
        if (have == d && run > d) {
            v = g1x29_add(v, w);
            have += hw;
        }
    }
    if (!cut || t != t1 + 1) return;                                       // only the first boundary of a bucket writes it
    g1x29_t acc = g1x29_add(ld_g1x29(tail + t1), v);
    for (uint32_t u = t + run; u <= t2; u++) acc = g1x29_add(acc, ld_g1x29(head + u));   // the part of the bucket beyond this wave
    st_g1x29(buckets + b, acc);
}
// Heavily skewed buckets (thousands of equal witness values -- a constant column is ONE bucket cut by every lane boundary).
// Pass 1: one WAVE per chunk of MSM_HEAVY_CHUNK lane partials folds head[start .. start + chunk) into head[start] -- or, for a bucket of one
// chunk (almost all of them: the medium buckets of a witness column, 17-256 partials), finishes the bucket, tail included.
// Pass 2: one workgroup per bucket of SEVERAL chunks folds tail[t1] and the chunk sums.
// (Until round 6 a chunk was 1024 partials folded by a whole 256-thread workgroup, one per CU at this kernel's register count: a witness
// column's hundreds of medium buckets queued behind each other, 62-229 us per advice column and 8.5 ms of a k = 20 MLP proof's kernel time
// (profiles/r05au_msm_columns_serial.txt, r05az_timeline.txt).  A wave needs no LDS and no barrier, four chunks run per workgroup, and the
// tree is the cooperative wave sum alone.)
__global__ __launch_bounds__(256) void msm_fixup_heavy1_kernel(const uint32_t* offsets, uint32_t nb, uint32_t nlanes, const uint32_t* lane_first, g1x29_t* head,
                                                               const g1x29_t* tail, g1x29_t* buckets, const uint32_t* chunk_list, const uint32_t* counts, uint32_t lmin,
                                                               uint32_t coop, size_t bstride) {
    BOFF(); BSH(offsets); BSH(lane_first); BSH(head); BSH(tail); BSH(buckets); BSH(chunk_list); BSH(counts);
    const uint32_t nchunks = counts[1], L = msm_lane_len(offsets, nb, nlanes, lmin);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t ci = blockIdx.x * 4u + wave; ci < nchunks; ci += gridDim.x * 4u) {           // wave-uniform
        const uint32_t start = chunk_list[ci], b = lane_first[start];
        const uint32_t t2 = (offsets[b + 1] - 1) / L;
        const uint32_t stop = start + MSM_HEAVY_CHUNK - 1 < t2 ? start + MSM_HEAVY_CHUNK - 1 : t2;
        const uint32_t t1 = offsets[b] / L;
        const bool whole = t2 - t1 <= MSM_HEAVY_CHUNK;                     // the bucket's only chunk (start == t1 + 1): finished here, tail included
        // the tail enters as the LAST lane's first operand (that lane has the fewest partials of the chunk): no addition after the tree
        g1x29_t acc = (whole && lane == 63u) ? ld_g1x29(tail + t1) : g1x29_identity();
        for (uint32_t t = start + lane; t <= stop; t += 64) acc = g1x29_add(acc, ld_g1x29(head + t));
        acc = coop ? g1x29_group_sum_coop(acc, 64) : g1x29_group_sum(acc, 64);   // every lane's reads of head[start..stop] precede the sum it feeds
        if (lane == 0) st_g1x29(whole ? buckets + b : head + start, acc);
    }
}
__global__ __launch_bounds__(256) void msm_fixup_heavy2_kernel(const uint32_t* offsets, uint32_t nb, uint32_t nlanes, const g1x29_t* head, const g1x29_t* tail,
                                                               const uint32_t* heavy_list, const uint32_t* counts, g1x29_t* buckets, uint32_t lmin, uint32_t coop, size_t bstride) {
    BOFF(); BSH(offsets); BSH(head); BSH(tail); BSH(heavy_list); BSH(counts); BSH(buckets);
    const uint32_t L = msm_lane_len(offsets, nb, nlanes, lmin);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (uint32_t h = blockIdx.x * 4u + wave; h < counts[0]; h += gridDim.x * 4u) {            // one WAVE per bucket of several chunks (wave-uniform)
        const uint32_t b = heavy_list[h];
        const uint32_t t1 = offsets[b] / L, t2 = (offsets[b + 1] - 1) / L;
        g1x29_t acc = lane == 63u ? ld_g1x29(tail + t1) : g1x29_identity();
        for (uint32_t t = t1 + 1 + lane * MSM_HEAVY_CHUNK; t <= t2; t += 64 * MSM_HEAVY_CHUNK) acc = g1x29_add(acc, ld_g1x29(head + t));
        acc = coop ? g1x29_group_sum_coop(acc, 64) : g1x29_group_sum(acc, 64);
        if (lane == 0) st_g1x29(buckets + b, acc);
    }
}

// ---- reduce: sum_pos weight(pos) * B_pos ---------------------------------------------------------
// A bucket position splits into three bit-fields A (lowest), B, C; its weight is 1 + dA*2^wsA + dB*2^wsB +
// dC*2^wsC.  So the weighted sum is TOTAL + sum over fields of 2^ws * sum_d d * S_field[d], with S_field[d] the
// plain sum of the buckets whose field equals d.  S_A comes from column sums; S_B and S_C come from the row
// sums T[dC,dB] = sum_dA B: two passes over the buckets, each a shallow reduction (no running sums, whose
// dependent chains are latency-bound on a GPU), then per-bit plane sums; the final Horner over <= 22 planes
// runs on the host.
struct ReduceGeom {
    uint32_t wA, wB, wC;          // field widths (pos bits: A = [0,wA), B = [wA,wA+wB), C = rest)
    uint32_t wsA, wsB, wsC;       // weight shifts of the fields
    uint32_t EA, GA, ET, GT;      // serial elements per lane / groups for column sums (A) and row sums (T)
};
__global__ __launch_bounds__(256, 2) void msm_reduce1_kernel(const g1x29_t* buckets, ReduceGeom g, g1x29_t* partA, g1x29_t* partT, size_t bstride) {
    BOFF(); BSH(buckets); BSH(partA); BSH(partT);
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    g1x29_t acc = g1x29_identity();
    if (blockIdx.y == 0) {            // column sums: S_A[dA] partials
        if (idx >= (g.GA << g.wA)) return;
        const uint32_t dA = idx & ((1u << g.wA) - 1u), grp = idx >> g.wA;
        for (uint32_t e = 0; e < g.EA; e++) acc = g1x29_add(acc, ld_g1x29(buckets + ((((size_t)grp * g.EA + e) << g.wA) | dA)));
        st_g1x29(partA + (size_t)dA * g.GA + grp, acc);
    } else {                          // row sums: T[t] partials
        const uint32_t nT = 1u << (g.wB + g.wC);
        if (idx >= nT * g.GT) return;
        const uint32_t grp = idx % g.GT, t = idx / g.GT;
        for (uint32_t e = 0; e < g.ET; e++) acc = g1x29_add(acc, ld_g1x29(buckets + (((size_t)t << g.wA) | (e * g.GT + grp))));   // lanes = consecutive buckets
        st_g1x29(partT + (size_t)t * g.GT + grp, acc);
    }
}
// S_A[d] = sum_g partA[d][g] (blocks [0, blocksA)),  T[t] = sum_g partT[t][g] (the rest): `lanes` lanes of a wave per output
// (a few serial additions, then a shuffle tree).  The lane counts are chosen by the host so that the launch has at most one
// wave per SIMD: these chains are latency-bound, and a second wave on a SIMD doubles the latency of both.
__global__ __launch_bounds__(64) void msm_reduce2_kernel(const g1x29_t* partA, const g1x29_t* partT, ReduceGeom g, g1x29_t* SA, g1x29_t* T, uint32_t lanesA,
                                                         uint32_t lanesT, uint32_t blocksA, uint32_t coop, size_t bstride) {
    BOFF(); BSH(partA); BSH(partT); BSH(SA); BSH(T);
    const bool isA = blockIdx.x < blocksA;
    const uint32_t lanes = isA ? lanesA : lanesT, per = 64 / lanes;
    const uint32_t o = (isA ? blockIdx.x : blockIdx.x - blocksA) * per + threadIdx.x / lanes, j = threadIdx.x % lanes;
    const uint32_t nout = isA ? (1u << g.wA) : (1u << (g.wB + g.wC)), G = isA ? g.GA : g.GT;
    g1x29_t acc = g1x29_identity();
    if (o < nout) {
        const g1x29_t* src = (isA ? partA : partT) + (size_t)o * G;
        for (uint32_t i = j; i < G; i += lanes) acc = g1x29_add(acc, ld_g1x29(src + i));
    }
    acc = (coop && lanes >= 4) ? g1x29_group_sum_coop(acc, lanes) : g1x29_group_sum(acc, lanes);   // uniform over the workgroup
    if (j == 0 && o < nout) st_g1x29((isA ? SA : T) + o, acc);
}
// one workgroup per plane: planes[0] = TOTAL; planes[1 + ws + j] = sum of the field sums whose digit has bit j
// planes_host (optional): the group's landing buffer in page-locked host memory, 32 records per MSM -- the plane sums are written THERE as
// well, straight from the kernel (144 bytes per plane over PCIe), so the host tail needs no copy command after the last kernel
__global__ __launch_bounds__(256) void msm_planes_kernel(const g1x29_t* SA, const g1x29_t* T, ReduceGeom g, g1x29_t* planes, g1x29_t* planes_host, uint32_t coop,
                                                         size_t bstride) {
    BOFF(); BSH(SA); BSH(T); BSH(planes);
    __shared__ uint4 sh[9 * 4];
    uint32_t id = blockIdx.x, field = 0, j = 0;      // field 0: TOTAL, 1: A, 2: B, 3: C
    if (id > 0) {
        id -= 1;
        if (id < g.wA) { field = 1; j = id; }
        else if (id < g.wA + g.wB) { field = 2; j = id - g.wA; }
        else { field = 3; j = id - g.wA - g.wB; }
    }
    const uint32_t nA = 1u << g.wA, nT = 1u << (g.wB + g.wC);
    g1x29_t acc = g1x29_identity();
    // every thread walks indices that HAVE the bit (k-th such index: k with a 1 inserted at the bit position), so no
    // lane idles through the serial part
    auto with_bit = [](uint32_t k, uint32_t bit) { return ((k >> bit) << (bit + 1)) | (1u << bit) | (k & ((1u << bit) - 1u)); };
    if (field == 0) {
        for (uint32_t d = threadIdx.x; d < nA; d += 256) acc = g1x29_add(acc, ld_g1x29(SA + d));
    } else if (field == 1) {
        for (uint32_t k = threadIdx.x; k < nA / 2; k += 256) acc = g1x29_add(acc, ld_g1x29(SA + with_bit(k, j)));
    } else {
        const uint32_t sh_bits = field == 2 ? j : g.wB + j;          // t = (dC << wB) | dB
        for (uint32_t k = threadIdx.x; k < nT / 2; k += 256) acc = g1x29_add(acc, ld_g1x29(T + with_bit(k, sh_bits)));
    }
    acc = (coop & 8u) ? g1x29_block256_sum_coop_full(acc, sh) : coop ? g1x29_block256_sum_coop(acc, sh) : g1x29_block256_sum(acc, sh);
    if (threadIdx.x == 0) {
        const uint32_t ws = field == 1 ? g.wsA : field == 2 ? g.wsB : g.wsC;
        const uint32_t slot = field == 0 ? 0u : 1u + ws + j;
        st_g1x29(planes + slot, acc);
        if (planes_host) st_g1x29(planes_host + (size_t)blockIdx.z * 32 + slot, acc);
    }
}

// The table is built once per base set: on first use (blocking), or ahead of time by msm_table_prepare (ezkl_hip_bases_prepare) on the
// library's table stream WITHOUT waiting -- a one-shot prover uploads the SRS, starts the build and goes on reading the proving key;
// the first MSM only waits for the `ready` event if the build is still running.
static int table_build(Ctx* c, hipStream_t st, const Bases* b, bool wait) {
    MsmTable t;
    t.n = b->n;
    t.wp = pick_plan(b->n);
    if ((size_t)t.wp.W * t.n >= ((size_t)1 << 31)) return EZKL_ERR_UNSUPPORTED;
    EZ_HIP(hipMalloc(&t.tab, (size_t)t.wp.W * t.n * sizeof(g1a_t)));
    EZ_HIP(hipMemcpyAsync(t.tab, b->pts, t.n * sizeof(g1a_t), hipMemcpyDeviceToDevice, st));
    for (uint32_t w = 1; w < t.wp.W; w++)
        hipLaunchKernelGGL(msm_precompute_kernel, dim3(cdiv(t.n, 256)), dim3(256), 0, st, t.tab + (size_t)(w - 1) * t.n,
                           t.tab + (size_t)w * t.n, t.n, t.wp.width(w - 1));
    hipLaunchKernelGGL(msm_table_to_r261_kernel, dim3(cdiv((size_t)t.wp.W * t.n, 256)), dim3(256), 0, st, t.tab, (size_t)t.wp.W * t.n);
    EZ_HIP(hipGetLastError());
    EZ_HIP(hipEventCreateWithFlags(&t.ready, hipEventDisableTiming));
    EZ_HIP(hipEventRecord(t.ready, st));
    if (wait) {
        EZ_HIP(hipStreamSynchronize(st));
        t.synced = true;
    }
    g_tables[b] = t;
    return EZKL_OK;
}
static int table_get(Ctx* c, hipStream_t st, const Bases* b, MsmTable** out) {
    auto it = g_tables.find(b);
    if (it == g_tables.end()) {
        int rc = table_build(c, st, b, true);
        if (rc) return rc;
        it = g_tables.find(b);
    }
    if (!it->second.synced) {                     // prepared ahead of time: make sure the build has finished
        EZ_HIP(hipEventSynchronize(it->second.ready));
        it->second.synced = true;
    }
    *out = &it->second;
    return EZKL_OK;
}
int msm_table_prepare(Ctx* c, const Bases* b) {
    if (g_tables.count(b)) return EZKL_OK;
    if (!g_table_stream) EZ_HIP(hipStreamCreateWithFlags(&g_table_stream, hipStreamNonBlocking));
    return table_build(c, g_table_stream, b, false);
}
void msm_table_drop(const Bases* b) {
    auto it = g_tables.find(b);
    if (it != g_tables.end()) {
        if (it->second.ready) { (void)hipEventSynchronize(it->second.ready); (void)hipEventDestroy(it->second.ready); }
        (void)hipFree(it->second.tab);
        g_tables.erase(it);
    }
}

static constexpr size_t MSM_MAX_GROUP = 16;       // MSMs fused into one sequence of launches (gridDim.z)
static int slot_prepare(MsmSlot& sl, size_t bytes, size_t msms = 1) {
    if (!sl.st) {
        EZ_HIP(stream_create_prio(&sl.st, "EZKL_HIP_PRIO_MSM", 0));
        EZ_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        EZ_HIP(hipHostMalloc((void**)&sl.list_pinned, MSM_MAX_GROUP * sizeof(void*), hipHostMallocDefault));
    }
    if (msms > sl.pinned_msms) {
        if (sl.pinned) {
            EZ_HIP(hipStreamSynchronize(sl.st));
            EZ_HIP(hipHostFree(sl.pinned));
            sl.pinned = nullptr;
        }
        EZ_HIP(hipHostMalloc((void**)&sl.pinned, msms * 32 * sizeof(g1x29_t), hipHostMallocDefault));
        EZ_HIP(hipHostGetDevicePointer(&sl.pinned_dev, sl.pinned, 0));
        sl.pinned_msms = msms;
    }
    if (bytes > sl.scratch_bytes) {
        if (sl.scratch) {
            EZ_HIP(hipStreamSynchronize(sl.st));
            EZ_HIP(hipFree(sl.scratch));
            sl.scratch = nullptr;
            sl.scratch_bytes = 0;
        }
        size_t want = bytes + (bytes >> 4);
        EZ_HIP(hipMalloc((void**)&sl.scratch, want));
        sl.scratch_bytes = want;
    }
    return EZKL_OK;
}
// host tail: result = TOTAL + sum_k 2^k * plane[1+k] (Horner from the top bit), then canonical affine
static int msm_finish(MsmSlot& sl, void* out_host = nullptr, bool release = true) {
    if (!out_host) out_host = sl.out;
    EZ_HIP(hipEventSynchronize(sl.done));
    for (uint32_t j = 0; j < sl.count; j++) {
        const uint32_t* hp = sl.pinned + (size_t)j * 32 * 36;   // planes in the kernels' radix-2^29 form -> canonical 64-bit limbs
        h64::xyzz acc = h64::identity();
        for (int k = (int)sl.bits - 1; k >= 0; k--) {
            acc = h64::dbl(acc);
            acc = h64::add(acc, h64::from_limbs29_point(hp + 36 * (1 + k)));
        }
        acc = h64::add(acc, h64::from_limbs29_point(hp));
        h64::aff r = h64::to_affine(acc);
        memcpy((uint8_t*)out_host + 64 * (size_t)j, &r, 64);
    }
    if (release) sl.busy = false;
    return EZKL_OK;
}

// how many MSMs of this size are fused into one group (gridDim.z): small MSMs are launch- and latency-bound (at 2^17 points a lone MSM
// is 0.12 ms of accumulation inside a 0.6 ms chain of ~17 launches)
static size_t msm_group_size(const MsmTable* T, size_t n, bool small_scalars, size_t batch = 0) {
    if (const char* e = getenv("EZKL_MSM_GROUP")) {
        const int v = atoi(e);
        if (v >= 1 && v <= (int)MSM_MAX_GROUP) return (size_t)v;
    }
    // Round 4, measured on whole proofs (tools/ab.sh group | circuits | matrix | slots, profiles/r04[e-v]_*; 8 proofs per setting, same bytes):
    //   * general scalars (z, phi, h pieces): groups of FOUR at every size (round 3 left 2^20-point batches unfused: a lone fused group is
    //     1.50 against 1.35 ms per MSM).  With the false dependency between the NTT stream and the helper programs gone (one scratch arena
    //     per stream, capi.hip / common.hpp) groups of 1 / 2 / 4 on 2 / 3 / 4 / 6 slots all land within +-1.5 ms on the k = 20 MLP proof
    //     (73-77 ms, box-to-box variation included); four keeps a phase to two or three streams and is best or tied at k = 20 (einsum, MLP)
    //     and k = 22.
    //   * witness-shaped columns (advice, multiplicities: one or two non-zero digits per scalar, every kernel of the chain latency-bound
    //     whatever n is): groups of SIX (12 advice columns = 2 chains, 8 multiplicity columns = 6 + 2): 81.0 against 82.0 ms with four.
    // EZKL_MSM_GROUP_SMALL / EZKL_MSM_GROUP_BIG override one class, EZKL_MSM_GROUP both.
    (void)T; (void)n; (void)batch;
    if (small_scalars) {
        static const int small = [] { const char* e = getenv("EZKL_MSM_GROUP_SMALL"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= (int)MSM_MAX_GROUP ? v : 6; }();
        return (size_t)small;
    }
    static const int big = [] { const char* e = getenv("EZKL_MSM_GROUP_BIG"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= (int)MSM_MAX_GROUP ? v : 4; }();
    return (size_t)big;
}
// `count` MSMs of n points each (scalar columns cols[0..count)) as ONE sequence of launches with gridDim.z = count
static int msm_enqueue(Ctx* c, MsmSlot& sl, hipStream_t st, MsmTable* T, size_t base_offset, const fe_t* const* cols, size_t count, size_t n, bool timed) {
    int rc = EZKL_OK;
    if (count < 1 || count > MSM_MAX_GROUP) return EZKL_ERR_INVALID;
    const unsigned Z = (unsigned)count;
    const fe_t* scalars = cols[0];
    const WinPlan wp = T->wp;
    const uint32_t W = wp.W, bits = wp.cmax() - 1;
    const uint32_t nb = 1u << bits;
    const size_t npairs = n * W;
    // partitions of the first sorting pass: 1024, or as many more (<= 4096) as it takes to keep a partition inside the second pass's LDS
    // stage -- beyond 2^20 points a partition of the 1024 outgrew it and the second pass fell back to scattered stores (0.9 of the 6 ms of
    // a 2^22-point MSM: profiles/r06u_size_sweep.log, tools/msm22_profile.py)
    uint32_t PB = bits < 10 ? bits : 10;
    while (PB < bits && PB < MSM_MAX_PART_BITS && (npairs >> PB) > (size_t)MSM_BINSORT_STAGE * 9 / 10) PB++;
    const uint32_t LB = bits - PB, NP = 1u << PB;
    // ---- lane length for the accumulate kernel: fill the resident lanes an integer number of times ----
    int& acc_blocks_per_cu = msm_state().acc_blocks_per_cu;
    if (!acc_blocks_per_cu) {
        EZ_HIP(hipFuncSetAttribute((const void*)msm_binsort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        EZ_HIP(hipFuncSetAttribute((const void*)msm_partition_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
        EZ_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&acc_blocks_per_cu, msm_accumulate_kernel, 256, 0));
        if (acc_blocks_per_cu < 1) acc_blocks_per_cu = 1;
    }
    const size_t resident = (size_t)acc_blocks_per_cu * 256 * c->num_cus;
    // 40..80 pairs per lane at 2^20 points: few cut buckets, whole waves of work.  Larger MSMs have more pairs per BUCKET (104 at 2^22), and a lane
    // shorter than a bucket cuts every bucket several times (the boundary fold was 0.52 of a 2^22-point MSM's 6 ms): the lanes grow with the load
    const size_t load = npairs >> bits, per_lane = load * 9 / 10 > 40 ? load * 9 / 10 : 40;
    size_t rounds = npairs / (resident * per_lane);
    if (rounds < 1) rounds = 1;
    uint32_t L = (uint32_t)((npairs + resident * rounds - 1) / (resident * rounds));
    if (L < 8) L = 8;
    if (const char* e = getenv("EZKL_MSM_L")) {                        // tuning knob (tools/gpu_probe.py); nonsense keeps the default
        const int v = atoi(e);
        if (v > 0 && v < (1 << 20)) L = (uint32_t)v;
    }
    const uint32_t nlanes = cdiv(npairs, L);
    // device-side floor of the lane length (columns with few non-zero digits) and the cut count above which a bucket takes the heavy path
    static const uint32_t lmin = [] { const char* e = getenv("EZKL_MSM_LMIN"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 1 && v <= 4096 ? v : (int)MSM_LMIN); }();
    // cooperative (quad) additions in the latency-bound trees (curve29.hpp: g1x29_add_quad) -- bit 0: reduce2, 1: planes, 2: heavy; bit 3 (the
    // planes kernel's four wave totals cooperatively as well) was measured level and stays off (profiles/r05aj_coop15.log)
    constexpr uint32_t coop = 7;
    // (the planes kernel writes its sums into the slot's page-locked landing buffer itself: no copy command after it)
    static const uint32_t span_heavy = [] { const char* e = getenv("EZKL_MSM_SPAN"); const int v = e ? atoi(e) : 0; return (uint32_t)(v >= 1 && v <= 4096 ? v : (int)MSM_SPAN_HEAVY); }();
    // ---- field geometry of the reduce phase (positions: pos = (bucket & (NP-1)) << LB | bucket >> PB) ----
    ReduceGeom rg;
    memset(&rg, 0, sizeof rg);
    if (LB > 0) {
        rg.wA = LB; rg.wsA = PB;                          // A = high bucket bits
        rg.wB = (PB + 1) / 2; rg.wsB = 0;                 // B, C = low bucket bits
        rg.wC = PB - rg.wB; rg.wsC = rg.wB;
    } else {                                              // small MSM: pos == bucket
        rg.wA = (bits + 2) / 3; rg.wsA = 0;
        rg.wB = (bits - rg.wA + 1) / 2; rg.wsB = rg.wA;
        rg.wC = bits - rg.wA - rg.wB; rg.wsC = rg.wA + rg.wB;
    }
    {
        const uint32_t rows = 1u << (rg.wB + rg.wC), cols = 1u << rg.wA;
        uint32_t E = MSM_DIGIT_E;
        if (const char* e = getenv("EZKL_MSM_E")) {                          // tuning knob: a power of two; nonsense keeps the default
            const int v = atoi(e);
            if (v > 0 && v <= 1024 && (v & (v - 1)) == 0) E = (uint32_t)v;
        }
        rg.EA = rows < E ? rows : E; rg.GA = rows / rg.EA;
        rg.ET = cols < E ? cols : E; rg.GT = cols / rg.ET;
    }
    const uint32_t nA = 1u << rg.wA, nT = 1u << (rg.wB + rg.wC);
    const uint32_t n_partA = nA * rg.GA, n_partT = nT * rg.GT;
    const uint32_t nplanes = 1 + bits;
    // ---- sort geometry: sgrid workgroups, each owning per_block consecutive scalars ----
    // (one scalar per thread of the partition pass; all of a workgroup's pairs must fit its LDS staging area)
    size_t per_block = MSM_PART_STAGE / W / 64 * 64;
    {
        const size_t lds_words = (144u << 10) / 4, fixed = 3 * ((size_t)NP + 2);       // the partition kernel's 144 KiB: three arrays of NQ + 1 words, then 2 W words per scalar
        const size_t fit = lds_words > fixed ? (lds_words - fixed) / (2 * (size_t)W) / 64 * 64 : 64;
        if (per_block > fit) per_block = fit;
    }
    if (per_block > 1024) per_block = 1024;
    if (per_block < 64) per_block = 64;
    const unsigned sgrid = cdiv(n, per_block);
    // ---- carve scratch ----
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    size_t o_list = carve(MSM_MAX_GROUP * sizeof(void*));   // (first slab only) the group's scalar-column pointers
    size_t o_ent = carve(npairs * 8), o_vals = carve(npairs * 4), o_offs = carve(((size_t)nb + 1) * 4);
    const uint32_t NQ = NP + 1;                          // + bucket 0's own partition (msm_part_of)
    size_t o_pcnt = carve((NQ + 1) * 4), o_pbase = carve((NQ + 1) * 4), o_wgh = carve((size_t)sgrid * NQ * 4), o_wgc = carve((size_t)sgrid * NQ * 4);
    size_t o_heavy = carve((size_t)nb * 4), o_chunks = carve(((size_t)nlanes + 1) * 4);
    const size_t nbins = (size_t)1 << LB;
    // ONE region that starts every chain at zero: the counters (hcnt[0..2]), the bin totals of the multi-workgroup
    // sort, the planes
    size_t o_hcnt = carve(256), o_btot = carve(MSM_MAX_BIG * nbins * 4), o_planes = carve((size_t)nplanes * sizeof(g1x29_t));
    const size_t zero_bytes = off - o_hcnt;
    size_t o_bflag = carve((size_t)NP * 4), o_blist = carve(MSM_MAX_BIG * 4),
           o_boff = carve((size_t)MSM_MAX_BIG * MSM_BIG_BLOCKS * nbins * 4);
    size_t o_lfirst = carve((size_t)nlanes * 4);
    size_t o_bkt = carve((size_t)nb * sizeof(g1x29_t));
    size_t o_head = carve((size_t)nlanes * sizeof(g1x29_t)), o_tail = carve((size_t)nlanes * sizeof(g1x29_t));
    size_t o_partA = carve((size_t)n_partA * sizeof(g1x29_t)), o_partT = carve((size_t)n_partT * sizeof(g1x29_t));
    size_t o_SA = carve((size_t)nA * sizeof(g1x29_t)), o_T = carve((size_t)nT * sizeof(g1x29_t));
    const size_t bstride = count > 1 ? off : 0;           // every MSM of the group owns one slab of this layout
    rc = slot_prepare(sl, off * count, count);
    if (rc) return rc;
    uint8_t* S = sl.scratch;
    const fe_t* const* scal_list = nullptr;
    if (count > 1) {
        for (size_t j = 0; j < count; j++) sl.list_pinned[j] = cols[j];
        EZ_HIP(hipMemcpyAsync(S + o_list, sl.list_pinned, count * sizeof(void*), hipMemcpyHostToDevice, st));
        scal_list = (const fe_t* const*)(S + o_list);
    }
    uint2* entries = (uint2*)(S + o_ent);
    uint32_t* vals = (uint32_t*)(S + o_vals);
    uint32_t* offs = (uint32_t*)(S + o_offs);
    uint32_t *pcnt = (uint32_t*)(S + o_pcnt), *pbase = (uint32_t*)(S + o_pbase), *wghist = (uint32_t*)(S + o_wgh), *wgcnt = (uint32_t*)(S + o_wgc);
    uint32_t *heavy = (uint32_t*)(S + o_heavy), *hcnt = (uint32_t*)(S + o_hcnt), *chunks = (uint32_t*)(S + o_chunks);   // hcnt[0] buckets, [1] chunks
    uint32_t* lfirst = (uint32_t*)(S + o_lfirst);
    uint32_t *bflag = (uint32_t*)(S + o_bflag), *blist = (uint32_t*)(S + o_blist), *btot = (uint32_t*)(S + o_btot), *boff = (uint32_t*)(S + o_boff);
    uint32_t* bcnt = hcnt + 2;                                // hcnt[0] heavy buckets, [1] chunks, [2] oversized partitions
    g1x29_t *bkt = (g1x29_t*)(S + o_bkt), *head = (g1x29_t*)(S + o_head), *tail = (g1x29_t*)(S + o_tail);
    g1x29_t *partA = (g1x29_t*)(S + o_partA), *partT = (g1x29_t*)(S + o_partT), *SA = (g1x29_t*)(S + o_SA), *TT = (g1x29_t*)(S + o_T);
    g1x29_t* planes = (g1x29_t*)(S + o_planes);

    hipEvent_t m0 = nullptr, m1 = nullptr, a0 = nullptr, a1 = nullptr;
    // EZKL_HIP_TIMING (read at every call): which event pairs a synchronous call records -- "all" (default): the whole chain ("msm") and the
    // dominant kernel ("msm_accumulate"); "kernel": the dominant kernel only; "none".  An event record is a barrier packet with a timestamp on
    // the stream: the four of a call cost ~15 us of a 1.31 ms step (profiles/r05aa_wall_probe.log), so a caller that only needs the
    // kernel's time -- bench.py's timed region -- asks for two.
    bool timed_chain = timed;
    if (timed) {
        const char* tm = getenv("EZKL_HIP_TIMING");
        if (tm && !strcmp(tm, "none")) timed = timed_chain = false;
        else if (tm && !strcmp(tm, "kernel")) timed_chain = false;
    }
    if (timed) {
        if (timed_chain && (rc = ev_pair(c, "msm", &m0, &m1))) return rc;
        if ((rc = ev_pair(c, "msm_accumulate", &a0, &a1))) return rc;
        if (timed_chain) EZ_HIP(hipEventRecord(m0, st));
    }
    // the chain's counters, bin totals and planes are zeroed by the histogram kernel (no memset command in front of the chain: round 5)
    hipLaunchKernelGGL(msm_hist_kernel, dim3(sgrid, 1, Z), dim3(256), 0, st, scalars, n, per_block, wp, LB, NP, wghist, wgcnt, scal_list, hcnt,
                       (uint32_t)(zero_bytes / 4), bstride);
    hipLaunchKernelGGL(msm_hist_scan_kernel, dim3(cdiv(NQ, 32), 1, Z), dim3(1024), 0, st, wghist, sgrid, NQ, pcnt, bstride);
    hipLaunchKernelGGL(msm_part_scan_kernel, dim3(1, 1, Z), dim3(1024), 0, st, pcnt, NQ, pbase, bflag, blist, bcnt, bstride);
    hipLaunchKernelGGL(msm_partition_kernel, dim3(sgrid, 1, Z), dim3((unsigned)per_block), (3 * ((size_t)NQ + 1) + 2 * per_block * W) * 4, st, scalars, n, per_block, wp,
                       LB, NP, base_offset, T->n, pbase, wghist, wgcnt, entries, vals, scal_list, bstride);
    hipLaunchKernelGGL(msm_binsort_kernel, dim3(NP + MSM_BIG_BLOCKS * MSM_BIG_ROWS, 1, Z), dim3(512), MSM_BINSORT_STAGE * 4, st, entries, pbase, LB, NP, bflag, offs,
                       vals, bkt, blist, bcnt, btot, boff, bstride);
    hipLaunchKernelGGL(msm_bigsort_scatter_kernel, dim3(MSM_BIG_BLOCKS, MSM_BIG_ROWS, Z), dim3(512), 0, st, entries, pbase, LB, blist, bcnt, btot, boff, offs,
                       vals, bkt, bstride);
    // accumulate
    if (timed) EZ_HIP(hipEventRecord(a0, st));
    hipLaunchKernelGGL(msm_accumulate_kernel, dim3(cdiv(nlanes, 256), 1, Z), dim3(256), 0, st, T->tab, offs, vals, nb, nlanes, bkt, head, tail,
                       lfirst, lmin, bstride);
    if (timed) EZ_HIP(hipEventRecord(a1, st));
    if (nlanes > 1)
    {
        hipLaunchKernelGGL(msm_fixup_boundary_tree_kernel, dim3(cdiv(nlanes - 1, 256), 1, Z), dim3(256), 0, st, offs, nb, nlanes, lfirst, head, tail, bkt,
                           heavy, hcnt, chunks, lmin, span_heavy, bstride);
    }
    {
        size_t max_heavy = nlanes / span_heavy + 1;
        const size_t heavy_wgs = (max_heavy + 3) / 4;                           // four buckets (waves) per workgroup
        unsigned hb = (unsigned)(heavy_wgs < (size_t)c->num_cus * 4 ? heavy_wgs : (size_t)c->num_cus * 4);
        size_t max_chunks = nlanes / MSM_HEAVY_CHUNK + max_heavy;
        const size_t chunk_wgs = (max_chunks + 3) / 4;                         // four chunks (waves) per workgroup
        unsigned cb = (unsigned)(chunk_wgs < (size_t)c->num_cus * 4 ? chunk_wgs : (size_t)c->num_cus * 4);
        hipLaunchKernelGGL(msm_fixup_heavy1_kernel, dim3(cb, 1, Z), dim3(256), 0, st, offs, nb, nlanes, lfirst, head, tail, bkt, chunks, hcnt, lmin, coop & 4u, bstride);
        hipLaunchKernelGGL(msm_fixup_heavy2_kernel, dim3(hb, 1, Z), dim3(256), 0, st, offs, nb, nlanes, head, tail, heavy, hcnt, bkt, lmin, coop & 4u, bstride);
    }
    // reduce
    {
        const uint32_t th = n_partA > n_partT ? n_partA : n_partT;
        hipLaunchKernelGGL(msm_reduce1_kernel, dim3(cdiv(th, 256), 2, Z), dim3(256), 0, st, bkt, rg, partA, partT, bstride);
        auto pow2_le = [](uint32_t x) { uint32_t p = 1; while (p * 2 <= x) p *= 2; return p; };
        uint32_t lanesA = pow2_le(rg.GA < 64 ? rg.GA : 64), lanesT = pow2_le(rg.GT < 64 ? rg.GT : 64);
        auto waves = [&](uint32_t nout, uint32_t lanes) { return cdiv(nout, 64 / lanes); };
        while (waves(nA, lanesA) + waves(nT, lanesT) > (unsigned)c->num_cus * 4 && (lanesA > 1 || lanesT > 1)) {
            if (lanesT > 1 && waves(nT, lanesT) >= waves(nA, lanesA)) lanesT >>= 1; else if (lanesA > 1) lanesA >>= 1; else lanesT >>= 1;
        }
        const uint32_t blocksA = waves(nA, lanesA);
        hipLaunchKernelGGL(msm_reduce2_kernel, dim3(blocksA + waves(nT, lanesT), 1, Z), dim3(64), 0, st, partA, partT, rg, SA, TT, lanesA, lanesT, blocksA, coop & 1u, bstride);
        hipLaunchKernelGGL(msm_planes_kernel, dim3(nplanes, 1, Z), dim3(256), 0, st, SA, TT, rg, planes, (g1x29_t*)sl.pinned_dev,
                           coop & 10u, bstride);
    }
    EZ_HIP(hipGetLastError());
    if (getenv("EZKL_MSM_DEBUG")) {
        uint32_t hc = 0;
        EZ_HIP(hipMemcpyAsync(&hc, hcnt, 4, hipMemcpyDeviceToHost, st));
        EZ_HIP(hipStreamSynchronize(st));
        fprintf(stderr, "[msm] n=%zu W=%u bits=%u L=%u nlanes=%u heavy=%u\n", n, W, bits, L, nlanes, hc);
    }
    if (timed_chain) EZ_HIP(hipEventRecord(m1, st));
    EZ_HIP(hipEventRecord(sl.done, st));
    sl.bits = bits;
    sl.count = (uint32_t)count;
    sl.busy = true;
    return EZKL_OK;
}
// A batch of `batch` MSMs as fused groups pipelined over the slot streams: group g runs on slot g % MSM_SLOTS; a slot is retired
// (host Horner, results written to their place in out_host) before it is reused.  wait_ev (optional): column j may only be read
// after wait_ev[j] (the upload phase's copy events).
static int msm_run_groups(Ctx* c, MsmTable* T, size_t base_offset, const fe_t* const* cols, size_t batch, size_t n, void* out_host,
                          const hipEvent_t* wait_ev, bool small_scalars) {
    const size_t G = msm_group_size(T, n, small_scalars, batch);
    static const bool dbg = getenv("EZKL_MSM_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[ezkl_hip] msm batch: %zu columns of %zu points, %s scalars, groups of %zu\n", batch, n, small_scalars ? "witness-shaped" : "general", G);
    int rc = EZKL_OK;
    size_t gi = 0;
    // An upload phase (wait_ev: column j becomes available when ITS copy lands, 0.6 ms apart at 2^20 rows over PCIe) is cut into TAPERED
    // groups -- G, then half of what is left, ... down to single columns -- so that what remains to be done after the last copy has landed
    // is the short chain of one column, not of a full group: twelve advice columns go as 6 + 3 + 2 + 1 (the groups start at 3.6, 5.4, 6.6
    // and 7.2 ms and overlap on their slots) instead of 6 + 6 (the second group could not start before 7.2 ms and then ran for ~3 ms).
    for (size_t j0 = 0; j0 < batch && !rc; gi++) {
        size_t cnt = batch - j0 < G ? batch - j0 : G;
        if (wait_ev && j0 > 0) {
            const size_t left = batch - j0;
            cnt = left <= 2 ? 1 : std::min(G, (left + 1) / 2);
        }
        MsmSlot& sl = g_slots[gi % MSM_SLOTS];
        if (sl.busy) rc = msm_finish(sl);
        if (!rc) rc = slot_prepare(sl, 0);
        for (size_t j = 0; wait_ev && j < cnt && !rc; j++)
            if (hipStreamWaitEvent(sl.st, wait_ev[j0 + j], 0) != hipSuccess) rc = EZKL_ERR_HIP;
        if (!rc) {
            sl.out = (uint8_t*)out_host + 64 * j0;
            rc = msm_enqueue(c, sl, sl.st, T, base_offset, cols + j0, cnt, n, false);
        }
        j0 += cnt;
    }
    for (size_t k = 0; k < (size_t)MSM_SLOTS; k++) {     // drain in launch order
        MsmSlot& sl = g_slots[(gi + k) % MSM_SLOTS];
        if (!sl.busy) continue;
        if (!rc) rc = msm_finish(sl);
        else { (void)hipStreamSynchronize(sl.st); sl.busy = false; }   // a failed batch leaves no slot busy
    }
    return rc;
}

struct MsmBatch;
static MsmBatch* g_open_batch_fwd();
int msm_run(Ctx* c, hipStream_t st, const Bases* b, size_t base_offset, const fe_t* scalars, size_t n, void* out_host) {
    if (g_open_batch_fwd()) return EZKL_ERR_INVALID;
    if (n == 0) { memset(out_host, 0, 64); return EZKL_OK; }
    MsmTable* T = nullptr;
    int rc = table_get(c, st, b, &T);
    if (rc) return rc;
    MsmSlot& sl = g_slots[0];
    if (sl.busy) return EZKL_ERR_INVALID;
    static const bool host_timing = getenv("EZKL_MSM_HOST_TIMING") != nullptr;      // where a synchronous call spends its host time
    if (!host_timing) {
        if ((rc = msm_enqueue(c, sl, st, T, base_offset, &scalars, 1, n, true))) return rc;
        return msm_finish(sl, out_host);
    }
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    static double acc[3] = {0, 0, 0};
    static int calls = 0;
    const double t0 = now();
    if ((rc = msm_enqueue(c, sl, st, T, base_offset, &scalars, 1, n, true))) return rc;
    const double t1 = now();
    EZ_HIP(hipEventSynchronize(sl.done));
    const double t2 = now();
    rc = msm_finish(sl, out_host);
    const double t3 = now();
    acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2;
    if (++calls % 50 == 0) {
        fprintf(stderr, "[msm host] n=%zu per call: enqueue %.1f us, wait for the GPU %.1f us, host tail %.1f us\n", n, acc[0] / 50, acc[1] / 50, acc[2] / 50);
        acc[0] = acc[1] = acc[2] = 0;
    }
    return rc;
}

// Concurrent callers (halo2 commits from rayon workers; VERDICT r01 "one global mutex serialises every call, including the host-side
// Horner that finishes each MSM"): a single MSM on the library stream holds the context lock only while its launches are queued.  The
// wait for the GPU and the host tail run outside the lock, on a call slot of the caller's own (streams, scratch, pinned planes), so
// the latency-bound tails of one thread's MSM overlap the accumulation of another's -- the overlap the batch entry points give a
// single-threaded caller.  The slots are separate from the batch slots: a batch never finishes (or reuses) a caller's slot.
// the two halves of it, for a caller that has other work to queue in between (ezkl_hip_msm_g1_start_dev / _finish: the prover commits
// the vanishing argument's random polynomial -- which depends on nothing but the randomness -- under the upload of the witness):
// start claims a call slot, orders it behind the library stream and queues the launches; finish waits and runs the host tail.
int msm_call_start(Ctx* c, std::unique_lock<std::recursive_mutex>& lk, const Bases* b, size_t base_offset, const fe_t* scalars, size_t n, int* slot_out) {
    if (g_open_batch_fwd() || n == 0) return EZKL_ERR_INVALID;
    MsmTable* T = nullptr;
    int rc = table_get(c, c->stream, b, &T);
    if (rc) return rc;
    int k = -1;
    for (;;) {
        for (int i = 0; i < MSM_CALL_SLOTS && k < 0; i++)
            if (!g_call_claimed[i]) k = i;
        if (k >= 0) break;
        bool all_mine = true;                         // ... unless they are all in THIS thread's hands: nobody else will ever finish one
        for (int i = 0; i < MSM_CALL_SLOTS; i++) all_mine = all_mine && g_call_owner[i] == std::this_thread::get_id();
        if (all_mine) return EZKL_ERR_BUSY;
        lk.unlock();                                  // every call slot is in another thread's hands: let one finish
        std::this_thread::yield();
        lk.lock();
    }
    MsmSlot& sl = g_call_slots[k];
    g_call_claimed[k] = true;
    g_call_finishing[k] = false;
    g_call_owner[k] = std::this_thread::get_id();
    rc = slot_prepare(sl, 0);
    if (!rc) {                                        // ordered after what the library stream has been asked to do so far (the scalar column)
        hipError_t e = hipSuccess;
        // ... unless that stream is idle (round 5): then there is nothing to order behind, and the cross-queue event (a barrier packet on the
        // slot's queue that waits for a signal of the library's) only delays the first kernel of a synchronous call
        const hipError_t q = hipStreamQuery(c->stream);
        if (q != hipSuccess) {
            (void)hipGetLastError();                       // hipErrorNotReady is not an error
            if (!g_call_order_ev) e = hipEventCreateWithFlags(&g_call_order_ev, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(g_call_order_ev, c->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(sl.st, g_call_order_ev, 0);
        }
        if (e != hipSuccess) rc = set_hip_error(e, "msm_call_start", __FILE__, __LINE__);
    }
    if (!rc) rc = msm_enqueue(c, sl, sl.st, T, base_offset, &scalars, 1, n, true);
    if (rc) { g_call_claimed[k] = false; g_call_owner[k] = std::thread::id(); return rc; }
    *slot_out = k;
    return EZKL_OK;
}
int msm_call_finish(Ctx* c, std::unique_lock<std::recursive_mutex>& lk, int k, void* out_host) {
    (void)c;
    if (k < 0 || k >= MSM_CALL_SLOTS || !g_call_claimed[k] || g_call_finishing[k] || !g_call_slots[k].busy) return EZKL_ERR_INVALID;
    MsmSlot& sl = g_call_slots[k];
    g_call_finishing[k] = true;                       // the token is spent: a second finish (this thread or another) is refused
    lk.unlock();
    const int rc = msm_finish(sl, out_host, false);   // GPU wait + host Horner: no library state touched
    lk.lock();
    sl.busy = false;
    g_call_finishing[k] = false;
    g_call_claimed[k] = false;
    g_call_owner[k] = std::thread::id();
    return rc;
}
int msm_run_concurrent(Ctx* c, std::unique_lock<std::recursive_mutex>& lk, const Bases* b, size_t base_offset, const fe_t* scalars, size_t n, void* out_host) {
    if (g_open_batch_fwd()) return EZKL_ERR_INVALID;
    if (n == 0) { memset(out_host, 0, 64); return EZKL_OK; }
    int k = -1;
    static const bool host_timing = getenv("EZKL_MSM_HOST_TIMING") != nullptr;      // where a synchronous call spends its host time
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = host_timing ? now() : 0;
    int rc = msm_call_start(c, lk, b, base_offset, scalars, n, &k);
    if (rc) return rc;
    if (!host_timing) return msm_call_finish(c, lk, k, out_host);
    static double acc[3] = {0, 0, 0};
    static int calls = 0;
    const double t1 = now();
    (void)hipEventSynchronize(g_call_slots[k].done);
    const double t2 = now();
    rc = msm_call_finish(c, lk, k, out_host);
    const double t3 = now();
    acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2;
    if (++calls % 50 == 0) {
        fprintf(stderr, "[msm host] n=%zu per call: enqueue %.1f us, wait for the GPU %.1f us, host tail %.1f us\n", n, acc[0] / 50, acc[1] / 50, acc[2] / 50);
        acc[0] = acc[1] = acc[2] = 0;
    }
    return rc;
}

// `batch` independent MSMs against the same bases (the advice-column commits of one prover phase), pipelined
// over MSM_SLOTS streams.  `st` orders the batch after prior work on the caller's stream.
int msm_run_batch(Ctx* c, hipStream_t st, const Bases* b, size_t base_offset, const fe_t* const* scalars, size_t batch, size_t n,
                  void* out_host, bool small_scalars) {
    if (g_open_batch_fwd()) return EZKL_ERR_INVALID;
    if (batch == 0) return EZKL_OK;
    if (n == 0) { memset(out_host, 0, 64 * batch); return EZKL_OK; }
    MsmTable* T = nullptr;
    int rc = table_get(c, st, b, &T);
    if (rc) return rc;
    EZ_HIP(hipStreamSynchronize(st));          // inputs produced on the caller's stream are complete
    return msm_run_groups(c, T, base_offset, scalars, batch, n, out_host, nullptr, small_scalars);
}

// One prover phase in one call: upload `batch` host columns into the caller's device columns, overwrite their tail rows
// (blinding) and commit each.  All copies are queued up front on a dedicated copy stream (column j, then its tail rows, then
// an event); the MSM of column j waits for that event on its slot stream, so PCIe traffic for column j+1.. runs under the
// kernels of column j.  The host columns should be page-locked (ezkl_hip_host_malloc) for the copies to be truly asynchronous;
// pageable memory still works (the runtime stages it).
// One upload phase in flight: the copies are queued by begin(), a caller stream can be made to wait for a column
// (wait), the columns are committed (commit: each MSM waits for its own copy) and end() drains the copy stream.
struct MsmUpload {
    std::vector<hipEvent_t> ev;
    std::vector<fe_t*> dev_cols;
    size_t n = 0;
    bool broken = false;
    uint64_t* stage = nullptr;       // integer columns as they crossed PCIe (msm_expand_integer_rep_kernel reads them); freed by msm_upload_end
};
int msm_upload_end(MsmUpload* u) {
    if (!u || u != g_open_upload) return EZKL_ERR_INVALID;
    // Wait on the copies' own events, newest first, not on the stream: with more streams than hardware queues a stream synchronise
    // queues a marker behind whatever shares the copy stream's queue (measured: 5.3 ms behind the auxiliary stream's NTTs, long after
    // the last copy had landed -- profiles/r03w_hosttrace.txt); an event that has already fired returns at once.
    hipError_t e = u->broken && g_copy_st ? hipStreamSynchronize(g_copy_st) : hipSuccess;   // a copy without its event: drain the stream
    for (size_t j = u->ev.size(); j-- > 0;)
        if (u->ev[j]) {
            const hipError_t ej = hipEventSynchronize(u->ev[j]);
            if (e == hipSuccess) e = ej;
            (void)hipEventDestroy(u->ev[j]);
        }
    g_open_upload = nullptr;
    if (u->stage) (void)ezkl_hip_free(u->stage);     // back to the column pool; every expansion kernel has run (its column's event was waited for above)
    delete u;
    if (e != hipSuccess) return set_hip_error(e, "msm_upload_end", __FILE__, __LINE__);
    return EZKL_OK;
}
// ---- witness columns handed over as INTEGERS (ezkl's IntegerRep, /root/reference/src/fieldutils.rs:6-17: every cell of an ezkl advice column
// is integer_rep_to_felt of an i128 -- the quantised tensor values -- before it becomes 32 bytes of Montgomery form).  PCIe is the
// scarce link of the advice phase (12 columns x 32 MB at ~53 GB/s = 7 ms of a 67 ms proof), so a column may cross it as 8 or 16 bytes
// per cell and is expanded HERE: x >= 0 -> x, x < 0 -> r - |x| (integer_rep_to_felt), then x * 2^256 mod r (one product per cell).
// format: 0 = 32-byte Montgomery words (halo2's Vec<Fp>), 1 = int64, 2 = int128 (little-endian two's complement).
__global__ __launch_bounds__(256) void msm_expand_integer_rep_kernel(const uint64_t* src, uint32_t words, fe_t* dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t lo = src[i * words], hi = words == 2 ? src[i * 2 + 1] : (uint64_t)((int64_t)lo >> 63);
    const bool neg = (int64_t)hi < 0;
    if (neg) {                                            // |x| (two's complement; IntegerRep::MIN = 2^127 is its own magnitude)
        lo = ~lo + 1;
        hi = ~hi + (lo == 0 ? 1 : 0);
    }
    fe_t v = Fr::zero();
    v.v[0] = (uint32_t)lo; v.v[1] = (uint32_t)(lo >> 32); v.v[2] = (uint32_t)hi; v.v[3] = (uint32_t)(hi >> 32);
    if (neg && (lo | hi)) {                               // r - |x|
        uint32_t br = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) v.v[k] = subb32(FrP::MOD[k], v.v[k], br);
    }
    st_fe(dst + i, Fr::to_mont(v));
}
int msm_upload_begin(Ctx* c, const fe_t* const* host_cols, fe_t* const* dev_cols, size_t batch, size_t n, const fe_t* const* tails, size_t tail_start,
                     size_t tail_count, MsmUpload** out, const uint8_t* formats) {
    if (g_open_upload || g_open_batch_fwd()) return EZKL_ERR_INVALID;
    if (n == 0 || tail_start + tail_count > n) return EZKL_ERR_INVALID;
    if (!g_copy_st) EZ_HIP(hipStreamCreateWithFlags(&g_copy_st, hipStreamNonBlocking));
    if (tails && tail_count && batch * tail_count > g_tail_pinned_elems) {
        if (g_tail_pinned) EZ_HIP(hipHostFree(g_tail_pinned));
        g_tail_pinned = nullptr;
        g_tail_pinned_elems = batch * tail_count * 2;
        EZ_HIP(hipHostMalloc((void**)&g_tail_pinned, g_tail_pinned_elems * sizeof(fe_t), hipHostMallocDefault));
    }
    EZ_HIP(hipStreamSynchronize(c->stream));          // the destination columns may have been touched on the library stream
    MsmUpload* u = new MsmUpload();
    u->n = n;
    u->ev.assign(batch, nullptr);
    u->dev_cols.assign(dev_cols, dev_cols + batch);
    g_open_upload = u;
    size_t stage_words = 0, stage_off = 0;
    for (size_t j = 0; formats && j < batch; j++) {
        if (formats[j] > 2) { g_open_upload = nullptr; delete u; return EZKL_ERR_INVALID; }
        stage_words += n * formats[j];
    }
    if (stage_words) {
        // from the column pool (ezkl_hip_malloc: a block of this size is parked after the first proof; hipMalloc / hipFree would synchronise the device)
        const int rc_ = ezkl_hip_malloc((void**)&u->stage, stage_words * 8);
        if (rc_) { g_open_upload = nullptr; delete u; return rc_; }
    }
    for (size_t j = 0; j < batch; j++) {
        hipError_t e = hipEventCreateWithFlags(&u->ev[j], hipEventDisableTiming);
        const uint32_t words = formats ? formats[j] : 0;                 // 64-bit words per cell of an integer column (0: 32-byte field elements)
        if (e == hipSuccess && words) {
            // a staging block per integer column (expanding inside the destination column would race its own reads)
            uint64_t* stage = u->stage + stage_off;
            stage_off += n * words;
            e = hipMemcpyAsync(stage, host_cols[j], n * words * 8, hipMemcpyHostToDevice, g_copy_st);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(msm_expand_integer_rep_kernel, dim3(cdiv(n, 256)), dim3(256), 0, g_copy_st, (const uint64_t*)stage, words, dev_cols[j], n);
                e = hipGetLastError();
            }
        } else if (e == hipSuccess) e = hipMemcpyAsync(dev_cols[j], host_cols[j], n * sizeof(fe_t), hipMemcpyHostToDevice, g_copy_st);
        if (e == hipSuccess && tails && tail_count) {
            memcpy(g_tail_pinned + j * tail_count, tails[j], tail_count * sizeof(fe_t));
            e = hipMemcpyAsync(dev_cols[j] + tail_start, g_tail_pinned + j * tail_count, tail_count * sizeof(fe_t), hipMemcpyHostToDevice, g_copy_st);
        }
        if (e == hipSuccess) e = hipEventRecord(u->ev[j], g_copy_st);
        if (e != hipSuccess) {
            u->broken = true;
            (void)msm_upload_end(u);
            return set_hip_error(e, "msm_upload_begin", __FILE__, __LINE__);
        }
    }
    *out = u;
    return EZKL_OK;
}
int msm_upload_wait(MsmUpload* u, size_t j, hipStream_t st) {
    if (!u || u != g_open_upload || j >= u->ev.size() || !st) return EZKL_ERR_INVALID;
    EZ_HIP(hipStreamWaitEvent(st, u->ev[j], 0));
    return EZKL_OK;
}
int msm_upload_commit(Ctx* c, MsmUpload* u, const Bases* b, size_t commit_first, size_t commit_count, void* out_host) {
    if (!u || u != g_open_upload) return EZKL_ERR_INVALID;
    const size_t batch = u->ev.size();
    if (commit_first + commit_count > u->n) return EZKL_ERR_INVALID;
    if (batch == 0) return EZKL_OK;
    if (commit_count == 0) {                               // empty slice: the identity
        memset(out_host, 0, 64 * batch);
        return EZKL_OK;
    }
    MsmTable* T = nullptr;
    int rc = table_get(c, c->stream, b, &T);
    if (rc) return rc;
    std::vector<const fe_t*> cols(batch);
    for (size_t j = 0; j < batch; j++) cols[j] = u->dev_cols[j] + commit_first;
    return msm_run_groups(c, T, 0, cols.data(), batch, commit_count, out_host, u->ev.data(), true);     // an upload phase commits witness columns
}
bool msm_upload_is_open() { return g_open_upload != nullptr; }

// The same pipeline fed one column at a time: the caller uploads column j+1 (a blocking PCIe copy) while the slots
// run the MSMs of the columns pushed so far.  One batch may be open per device; other MSM entry points refuse
// (EZKL_ERR_INVALID) until it is finished.
struct MsmBatch {
    const Bases* b = nullptr;
    MsmTable* T = nullptr;
    size_t base_offset = 0, n = 0, pushed = 0;      // pushed: columns
    // one entry per fused group in flight or retired; a slot writes its group's affine results here (std::deque: stable addresses)
    struct Group {
        uint8_t out[64 * MSM_MAX_GROUP];
        uint32_t count = 0;
    };
    std::deque<Group> groups;
    size_t retired = 0;                             // groups
    hipEvent_t order_ev = nullptr;                  // "the pushed columns are final": recorded on the producing stream at every push
};
static MsmBatch* g_open_batch_fwd() { return g_open_batch; }
bool msm_batch_is_open() { return g_open_batch != nullptr; }
int msm_batch_begin(Ctx* c, hipStream_t st, const Bases* b, size_t base_offset, size_t n, MsmBatch** out) {
    if (g_open_batch) return EZKL_ERR_INVALID;
    for (auto& sl : g_slots)
        if (sl.busy) return EZKL_ERR_INVALID;
    MsmTable* T = nullptr;
    int rc = n ? table_get(c, st, b, &T) : EZKL_OK;
    if (rc) return rc;
    MsmBatch* mb = new MsmBatch();
    mb->b = b; mb->T = T; mb->base_offset = base_offset; mb->n = n;
    g_open_batch = mb;
    *out = mb;
    return EZKL_OK;
}
static int msm_batch_retire_one(MsmBatch* mb) {
    int rc = msm_finish(g_slots[mb->retired % MSM_SLOTS]);          // -> the group's `out`
    if (rc) return rc;
    mb->retired++;
    return EZKL_OK;
}
// `count` resident columns, produced by work queued on `after` so far: their MSMs start behind it WITHOUT a host synchronisation (an
// event), as fused groups on the slot streams -- the caller goes on queuing unrelated work on `after` (the prover: the next argument's
// helper chain) while they run
int msm_batch_push_many(Ctx* c, MsmBatch* mb, const fe_t* const* cols, size_t count, hipStream_t after) {
    if (mb != g_open_batch) return EZKL_ERR_INVALID;
    if (count == 0) return EZKL_OK;
    if (mb->n == 0) { mb->pushed += count; return EZKL_OK; }
    if (!mb->order_ev) EZ_HIP(hipEventCreateWithFlags(&mb->order_ev, hipEventDisableTiming));
    EZ_HIP(hipEventRecord(mb->order_ev, after));
    const size_t G = msm_group_size(mb->T, mb->n, false);
    int rc = EZKL_OK;
    for (size_t j0 = 0; j0 < count; j0 += G) {
        const size_t cnt = count - j0 < G ? count - j0 : G;
        if (mb->groups.size() - mb->retired >= (size_t)MSM_SLOTS && (rc = msm_batch_retire_one(mb))) return rc;
        MsmSlot& sl = g_slots[mb->groups.size() % MSM_SLOTS];
        if ((rc = slot_prepare(sl, 0))) return rc;
        EZ_HIP(hipStreamWaitEvent(sl.st, mb->order_ev, 0));
        mb->groups.emplace_back();
        mb->groups.back().count = (uint32_t)cnt;
        sl.out = mb->groups.back().out;
        if ((rc = msm_enqueue(c, sl, sl.st, mb->T, mb->base_offset, cols + j0, cnt, mb->n, false))) {
            mb->groups.pop_back();
            return rc;
        }
        mb->pushed += cnt;
    }
    return EZKL_OK;
}
int msm_batch_push(Ctx* c, MsmBatch* mb, const fe_t* scalars_dev) { return msm_batch_push_many(c, mb, &scalars_dev, 1, c->stream); }
// drains the pipeline, writes `pushed` affine results (capacity checked) and closes the batch -- also on error
int msm_batch_finish(Ctx* c, MsmBatch* mb, void* out_host, size_t capacity) {
    (void)c;
    if (mb != g_open_batch) return EZKL_ERR_INVALID;
    int rc = EZKL_OK;
    if (mb->n == 0) {
        if (capacity < mb->pushed) rc = EZKL_ERR_INVALID; else memset(out_host, 0, 64 * mb->pushed);
    } else {
        while (!rc && mb->retired < mb->groups.size()) rc = msm_batch_retire_one(mb);
        if (!rc && capacity < mb->pushed) rc = EZKL_ERR_INVALID;
        if (!rc) {
            uint8_t* o = (uint8_t*)out_host;
            for (auto& g : mb->groups) {
                memcpy(o, g.out, 64 * (size_t)g.count);
                o += 64 * (size_t)g.count;
            }
        }
        if (rc)                                              // leave no slot marked busy behind a failed batch
            for (auto& sl : g_slots)
                if (sl.busy) { (void)hipStreamSynchronize(sl.st); sl.busy = false; }
    }
    if (mb->order_ev) (void)hipEventDestroy(mb->order_ev);
    g_open_batch = nullptr;
    delete mb;
    return rc;
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

// ---- out[i] = scalars[i] * P for one fixed point P (SRS generation: g[i] = s^i G, g_lagrange[i] = L_i(s) G;
//      the reference's gen_srs -> ParamsKZG::setup, /root/reference/src/pfsys/srs.rs:14-16) ----
__global__ __launch_bounds__(256) void g1_mul_fixed_kernel(g1a_t base, const fe_t* scalars, size_t n, g1a_t* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_t s = Fr::from_mont(ld_fe(scalars + i));
    g1x_t acc = g1x_identity();
    bool started = false;
    for (int b = 253; b >= 0; b--) {
        if (started) acc = g1x_double(acc);
        if ((s.v[b >> 5] >> (b & 31)) & 1) {
            acc = g1x_add_mixed(acc, base);
            started = true;
        }
    }
    st_g1a(out + i, g1x_to_affine(acc));
}
int g1_mul_fixed(Ctx* c, hipStream_t st, const void* base_host, const fe_t* scalars, size_t n, void* out_dev) {
    if (n == 0) return EZKL_OK;
    g1a_t b;
    memcpy(&b, base_host, 64);
    hipLaunchKernelGGL(g1_mul_fixed_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, b, scalars, n, (g1a_t*)out_dev);
    EZ_HIP(hipGetLastError());
    return EZKL_OK;
}

// ---- ParamsKZG::downsize: g_lagrange of a smaller domain = the INVERSE NTT OVER G1 of the first 2^k' points of g ----------------------
// halo2's ParamsKZG::downsize(new_k) (called by load_params_prover, /root/reference/src/execute.rs:1739-1750, whenever the SRS file is
// larger than the circuit -- the normal case with a shared kzg22.srs) truncates g and rebuilds g_lagrange with g_to_lagrange: an FFT on
// projective points with omega^-1 followed by a scaling by 1 / n'.  Here: decimation in time on XYZZ points in HBM (bit-reversed load,
// natural-order result), one thread per butterfly (P, Q) -> (P + w Q, P - w Q) with w Q by double-and-add over the canonical bits of
// the twiddle, then one more pass that scales by 1 / n' and normalises to affine.  n' / 2 * k' + n' scalar multiplications of ~380 group
// operations each: 0.5 s at k' = 20 -- a load-time operation (the CPU reference spends minutes here), bound by the field-product rate.
__device__ __forceinline__ g1x_t g1x_scalar_mul(const g1x_t& p, const fe_t& s_canon) {
    g1x_t acc = g1x_identity();
    bool started = false;
#pragma unroll 1
    for (int b = 253; b >= 0; b--) {
        if (started) acc = g1x_double(acc);
        if ((s_canon.v[b >> 5] >> (b & 31)) & 1) {
            acc = started ? g1x_add(acc, p) : p;
            started = true;
        }
    }
    return acc;
}
__global__ __launch_bounds__(256) void ecntt_load_kernel(const g1a_t* g, g1x_t* work, uint32_t log_n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (1u << log_n)) return;
    const uint32_t r = log_n ? (__brev(i) >> (32 - log_n)) : 0u;
    st_g1x(work + r, g1x_from_affine(ld_g1a(g + i)));
}
// stage s (1-based): blocks of len = 2^s; butterfly j of a block takes twiddle tw[j << (log_n - s)] (tw[e] = omega^-e, canonical)
__global__ __launch_bounds__(256) void ecntt_stage_kernel(g1x_t* work, const fe_t* tw, uint32_t log_n, uint32_t s) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (1u << (log_n - 1))) return;
    const uint32_t half = 1u << (s - 1), j = t & (half - 1), blk = t >> (s - 1);
    const uint32_t lo = (blk << s) + j, hi = lo + half;
    const g1x_t P = ld_g1x(work + lo);
    g1x_t Q = ld_g1x(work + hi);
    if (j) Q = g1x_scalar_mul(Q, ld_fe(tw + ((size_t)j << (log_n - s))));       // j = 0: twiddle 1 (wave-uniform only in the first stages; correct everywhere)
    g1x_t nQ = Q;
    nQ.y = Fq::neg(Q.y);
    st_g1x(work + lo, g1x_add(P, Q));
    st_g1x(work + hi, g1x_add(P, nQ));
}
__global__ __launch_bounds__(256) void ecntt_finish_kernel(const g1x_t* work, fe_t ninv_canon, g1a_t* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st_g1a(out + i, g1x_to_affine(g1x_scalar_mul(ld_g1x(work + i), ninv_canon)));
}
__global__ __launch_bounds__(256) void ecntt_twiddle_kernel(fe_t* tw, uint32_t count) {      // Montgomery -> canonical, in place
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) st_fe(tw + i, Fr::from_mont(ld_fe(tw + i)));
}
// out (n' affine points) = g_to_lagrange(g[0 .. n')), n' = 2^log_n.  tw_mont: n' / 2 resident Montgomery powers omega^-e, e < n' / 2
// (consumed: converted in place); ninv_mont = 1 / n'
int g1_to_lagrange(Ctx* c, hipStream_t st, const void* g_dev, uint32_t log_n, fe_t* tw_mont, const fe_t& ninv_mont, void* out_dev) {
    (void)c;
    const uint32_t n = 1u << log_n;
    g1x_t* work = nullptr;
    EZ_HIP(hipMalloc(&work, (size_t)n * sizeof(g1x_t)));
    hipLaunchKernelGGL(ecntt_load_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, (const g1a_t*)g_dev, work, log_n);
    if (n > 1) hipLaunchKernelGGL(ecntt_twiddle_kernel, dim3(cdiv(n / 2, 256)), dim3(256), 0, st, tw_mont, n / 2);
    for (uint32_t s = 1; s <= log_n; s++)
        hipLaunchKernelGGL(ecntt_stage_kernel, dim3(cdiv(n / 2, 256)), dim3(256), 0, st, work, (const fe_t*)tw_mont, log_n, s);
    hipLaunchKernelGGL(ecntt_finish_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, (const g1x_t*)work, Fr::from_mont(ninv_mont), (g1a_t*)out_dev, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(work);
    if (e != hipSuccess) return set_hip_error(e, "g1_to_lagrange", __FILE__, __LINE__);
    return EZKL_OK;
}

void g1_add_affine_host(const void* a, const void* b, void* out) {
    h64::aff p, q;
    memcpy(&p, a, 64);
    memcpy(&q, b, 64);
    h64::aff r = h64::to_affine(h64::add(h64::from_affine(p), h64::from_affine(q)));
    memcpy(out, &r, 64);
}

}  // namespace ezkl
