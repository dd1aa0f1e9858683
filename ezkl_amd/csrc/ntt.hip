// ntt.hip -- radix-2^r multi-pass NTT over BN254 Fr for gfx950.
//
// Replaces halo2curves::fft::best_fft and the EvaluationDomain wrappers around it (SURVEY.md §8(a) A11;
// type used in-tree at /root/reference/src/circuit/modules/polycommit.rs:52).  Natural order in and out.
//
// Decomposition (N = R_1 * ... * R_P, R_p = 2^r_p <= 256, Cooley-Tukey across the passes):
//   pass p works on contiguous blocks of size M_p = N / (R_1..R_{p-1}); inside a block, index
//   i = i1 * S + i2 (S = M_p / R_p).  A workgroup owns a TILE of C = TILE/R_p adjacent columns i2 (so each
//   global row segment is C*32 B contiguous -> coalesced), loads it into LDS, runs the R_p-point column
//   transforms there (decimation in time, radix-2^29 lazy limbs: see ntt_pass_kernel), multiplies by the
//   inter-pass twiddle w_M^(i2*k1) (table streamed alongside the data) and writes back.  The last pass
//   (S = 1) owns C whole blocks chosen with consecutive leading digit so that its digit-reversed
//   (natural-order) output is written C*32 B at a time as well.
// HBM traffic per element: P x (32 B read + 32 B write) + 32 B twiddle in non-last passes; algorithmic
// minimum is 64 B.  Arithmetic: (log2 N)/2 + (P-1) Montgomery products per element --
// the kernel is integer-VALU bound, not HBM bound.
#include "common.hpp"
#include "field29.hpp"
#include <string.h>

// (Round 5's compile-time variants are resolved: the stage-2 butterflies whose twiddle is w_4^0 = 1 take no product (kept); loads without a
// divergent region per element were measured level and removed -- profiles/r05ad_ntt_ab.log.)

namespace ezkl {

// workgroup sizes: 256 threads for tiles of 1024 elements (and single-pass transforms), 512 / 1024 for the 2048- / 4096-element tiles of radix 2^9 / 2^10 passes
static constexpr uint32_t NTT_LOG_TILE = 10;   // multi-pass tile: 1024 elements = 36 KiB of LDS (9 limbs each) -> 4 workgroups (16 waves) per CU
static constexpr uint32_t NTT_LOG_SINGLE = 11; // a transform up to 2^11 runs as ONE pass in a 72 KiB tile

struct PassArgs {
    const fe_t* in;
    fe_t* out;
    size_t in_stride, out_stride;
    const fe_t* tw_inter;   // M entries: w_M^(i2*k1) at k1*S + i2 (null in last pass)
    uint32_t log_n, log_r, log_m, log_tile;
    uint32_t last, first;
    uint32_t in_log_len;    // first pass: elements at index >= 2^in_log_len read as zero
    uint32_t coset_pre;     // first pass: multiply element i by zeta^(i mod 3)
    uint32_t post;          // last pass: multiply output i by post_c[i mod 3]
    uint32_t npass, log_radix[4];
    uint32_t k1_major;      // last pass: tile owns C blocks with consecutive leading digit
    fe_t zeta[2];           // zeta, zeta^2
    fe_t post_c[3];
    // coset-major extended evaluation (coset_cm_run): blockIdx.y = column * E + coset; the first pass is a TWISTED transform (stage
    // twiddles of coset b at tw_pre29 + b * R) and its inter-pass table is per coset (tw_inter + coset * 2^log_n holds w^(i2 k1) * c_b^i2):
    // the coset scaling c_b^j costs no product
    // a RANGE of cosets (ezkl_hip_coeff_to_cosets_range_dev: a rank of a sharded prover holds only the cosets of the key columns that
    // it sweeps): blockIdx.y = column * 2^cm_log_cnt + local coset, coset = cm_first + local; the output holds 2^cm_log_cnt cosets
    uint32_t cm, cm_log_e, cm_first, cm_log_cnt;
    // stage-major twiddles w_(2^s)^o at 2^(s-1) - 1 + o as unpacked limbs in the 2^261 domain --
    // one table serves every pass of a plan (tw_stage29); the twisted first pass of the coset-major transform takes coset b's at
    // tw_pre29 + b * R.  Stages 1 .. lds_stages are staged in LDS, the later ones are read from the table.
    const f29_t* tw_stage29;
    const f29_t* tw_pre29;
    uint32_t lds_stages;
};

// ---- the pass: column transforms in radix 2^29 (field29.hpp), decimation in time -------------------------------------------------------
// Until round 4 the pass ran a decimation-in-frequency butterfly on 8 x 32-bit limbs: 87 % of what it issued was its 281-slot Montgomery
// product; the lazily reduced 9 x 29-bit product takes 205 instructions.  Round 2's attempt to use it lost the gain to
// conversions and range control because in a DIF butterfly the sums double in size every stage.  Here every column transform runs
// decimation-in-TIME: t = w v, (u, v) <- (u + t, u - t + 4p).  The product comes first, so t < 2p whatever v was, a value grows by at most 4p per stage (< 54p after the 11 stages of the largest tile,
// far below the 1000p the product accepts) and limbs grow by at most 2 units of 2^29 per stage: the only range control is ONE carry
// propagation per element at the end of a two-stage register group -- no comparison, no conditional subtraction inside the transform.
// (Round 5, measured and dropped: the work buffer and the inter-pass tables as UNPACKED 36-byte elements -- no pack / unpack between passes,
// 2.6 % fewer VALU instructions -- ran 2 % SLOWER, 0.443-0.448 against 0.434-0.437 ms per 2^22 transform on one box: 36-byte elements are
// nine dword accesses per lane at 4-byte alignment instead of two aligned 16-byte ones.  profiles/r05ac_ntt_ab.log)
// (Round 5: half of the stage-2 butterflies of a plain transform have the twiddle 1 as well and take a carry pass instead of the product;
// values then stay below 82p.)
// Elements are 8 x 32-bit words in HBM (the files' 2^256 Montgomery domain, untouched: the twiddles carry the 2^261 of the product) and
// 9 limbs in LDS (four 8-byte planes + one 4-byte plane); the inter-pass product brings a value back below 2p for the work buffer, the last
// pass subtracts floor(top limb / (p >> 232)) p and once more p conditionally.  The stage-1 twiddles of a plain transform are 1: no product.
// tools/ntt29_model.py replays this arithmetic with every register checked for overflow (tests/test_ntt29_model.py).
EZ_D void lds_put29(uint2* d, uint32_t* d8, uint32_t tile, uint32_t e, const f29_t& x) {
#pragma unroll
    for (int q = 0; q < 4; q++) d[q * tile + e] = make_uint2(x.v[2 * q], x.v[2 * q + 1]);
    d8[e] = x.v[8];
}
EZ_D f29_t lds_get29(const uint2* d, const uint32_t* d8, uint32_t tile, uint32_t e) {
    f29_t x;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint2 t = d[q * tile + e];
        x.v[2 * q] = t.x;
        x.v[2 * q + 1] = t.y;
    }
    x.v[8] = d8[e];
    return x;
}
// G consecutive DIT stages (s .. s+G-1, 1-based: stage s joins blocks of 2^(s-1) rows) of the R-point column transforms, 2^G elements per
// lane in registers.  A TWISTED transform X[k] = sum_i x_i d^i w_R^(ik) = P(d w_R^k) runs the same code: with P(t) = Pe(t^2) + t Po(t^2) the
// twist is absorbed by the twiddles -- stage s multiplies by d^(R/2^s) w_(2^s)^o (table entry 2^(s-1) - 1 + o) -- so evaluating on a coset
// costs no product beyond the plain transform's.  Input rows are in bit-reversed order (the loader permutes), output rows in natural order.
// tw: the stage-major table -- its copy in LDS for the groups up to stage lds_stages, the table itself (FROM_TABLE)
// for the later ones (lds_stages is even or the last stage, groups start at odd stages: a group never straddles).
template <int G, int NTT_THREADS, bool FROM_TABLE>
__device__ __forceinline__ void ntt_superstage29(uint2* data, uint32_t* d8, const f29_t* tw, uint32_t TILE, uint32_t logC, bool unit1, uint32_t s, uint32_t tid) {
    constexpr uint32_t M = 1u << G;
    const uint32_t C = 1u << logC;
    const uint32_t lh = s - 1, h = 1u << lh;
    const uint32_t ngroups = TILE >> G;
    for (uint32_t gid = tid; gid < ngroups; gid += NTT_THREADS) {
        const uint32_t c = gid & (C - 1), q = gid >> logC;
        const uint32_t o = q & (h - 1), blk = q >> lh;
        const uint32_t r0 = (blk << (lh + G)) + o;
        f29_t x[M];
#pragma unroll
        for (uint32_t i = 0; i < M; i++) x[i] = lds_get29(data, d8, TILE, ((r0 + i * h) << logC) + c);
#pragma unroll
        for (int t = 0; t < G; t++) {
            const uint32_t half = 1u << t, st = s + t;
            const uint32_t base = (1u << (st - 1)) - 1u;
#pragma unroll
            for (uint32_t i = 0; i < M; i++) {
                if (i & half) continue;
                if (unit1 && st == 1) {                              // w = 1; the partner is a loaded value, anything below 2^256 < 7p
                    const f29_t v = x[i + half];
                    x[i + half] = Fr29::sub<2>(x[i], v);
                    x[i] = Fr29::add(x[i], v);
                } else if (unit1 && st == 2 && (i & 1u) == 0) {      // stage 2 (always the second stage of the group s = 1: o = 0, h = 1): w_4^0 = 1
                    // every other butterfly of stage 2 multiplies by 1: ONE carry pass instead of the product (24 instructions for 206).  The
                    // partner came out of stage 1 (below 15p, loose limbs), so the subtraction borrows 32p; values then stay below
                    // 46p + 4p per later stage (82p after the 11 stages of the largest tile; tools/ntt29_model.py SKIP_UNIT2)
                    const f29_t v = Fr29::normalize(x[i + half]);
                    x[i + half] = Fr29::sub<4>(x[i], v);
                    x[i] = Fr29::add(x[i], v);
                } else {
                    const uint32_t off = o + (i & (half - 1)) * h;
                    const f29_t w = ld_f29(tw + base + off);
                    const f29_t v = Fr29::mul(x[i + half], w);       // limbs of x below 6 units of 2^29 (at most two stages since the last carry pass); v < 2p
                    x[i + half] = Fr29::sub<1>(x[i], v);
                    x[i] = Fr29::add(x[i], v);
                }
            }
        }
#pragma unroll
        for (uint32_t i = 0; i < M; i++) lds_put29(data, d8, TILE, ((r0 + i * h) << logC) + c, Fr29::normalize(x[i]));
    }
}

template <int NTT_THREADS>
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t TILE = 1u << a.log_tile;
    const uint32_t logC = a.log_tile - a.log_r, C = 1u << logC;
    const uint32_t log_s = a.log_m - a.log_r, S = 1u << log_s;
    uint2* data = reinterpret_cast<uint2*>(smem);
    uint32_t* d8 = reinterpret_cast<uint32_t*>(smem + 32u * TILE);
    f29_t* tloc = reinterpret_cast<f29_t*>(smem + 36u * TILE);
    const uint32_t tid = threadIdx.x, tile = blockIdx.x;
    const uint32_t cm_l = a.cm ? (blockIdx.y & ((1u << a.cm_log_cnt) - 1u)) : 0u, cm_col = a.cm ? (blockIdx.y >> a.cm_log_cnt) : blockIdx.y;
    const uint32_t cm_b = a.cm_first + cm_l;
    const bool twisted = a.cm && a.first;                         // coset-major first pass: coset b's own stage table and inter-pass table
    const fe_t* in = a.in + (size_t)(twisted ? cm_col : blockIdx.y) * a.in_stride;
    fe_t* out = (a.cm && a.last) ? a.out + (size_t)cm_col * a.out_stride + ((size_t)cm_l << a.log_n) : a.out + (size_t)blockIdx.y * a.out_stride;
    const fe_t* tw_inter = a.tw_inter ? a.tw_inter + (twisted ? ((size_t)cm_b << a.log_n) : 0) : nullptr;
    const f29_t* tws = twisted ? a.tw_pre29 + ((size_t)cm_b << a.log_r) : a.tw_stage29;
    for (uint32_t j = tid; j + 1 < (1u << a.lds_stages); j += NTT_THREADS) st_f29(tloc + j, ld_f29(tws + j));

    uint32_t n_blocks = 1u << (a.log_n - a.log_r);               // last pass only
    uint32_t sblk = a.npass >= 2 ? (n_blocks >> a.log_radix[0]) : 1u;
    auto col_base = [&](uint32_t c) -> size_t {
        if (!a.last) {
            uint32_t colid = tile * C + c;
            return ((size_t)(colid >> log_s) << a.log_m) + (colid & (S - 1));
        }
        uint32_t blk;
        if (a.k1_major) {
            uint32_t rest = tile % sblk, k10 = (tile / sblk) * C;
            blk = (k10 + c) * sblk + rest;
        } else {
            blk = tile * C + c;
        }
        return (size_t)blk << a.log_r;
    };

    // ---- load: row i1 of the tile goes to LDS row bitrev(i1), as limbs ----
    const size_t in_len = (size_t)1 << a.in_log_len;
    auto load_one = [&](uint32_t e, size_t& addr_out) -> fe_t {
        uint32_t c = e & (C - 1), i1 = e >> logC;
        size_t addr = col_base(c) + ((size_t)i1 << log_s);
        addr_out = addr;
        if (!a.first || addr < in_len) return ld_fe(in + addr);
        return Fr::zero();
    };
    auto put_one = [&](uint32_t e, size_t addr, const fe_t& raw) {
        f29_t x = Fr29::unpack(raw);
        if (a.first && a.coset_pre && addr < in_len) {
            uint32_t m3 = (uint32_t)(addr % 3);
            if (m3) x = Fr29::mul(x, Fr29::unpack(a.zeta[m3 - 1]));
        }
        e = (a.log_r ? ((__brev(e >> logC) >> (32 - a.log_r)) << logC) : 0u) | (e & (C - 1));
        lds_put29(data, d8, TILE, e, x);
    };
    if (TILE == 4 * NTT_THREADS) {
        fe_t x[4];
        size_t ad[4];
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = load_one(tid + i * NTT_THREADS, ad[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) put_one(tid + i * NTT_THREADS, ad[i], x[i]);
    } else {
        for (uint32_t e = tid; e < TILE; e += NTT_THREADS) {
            size_t ad;
            fe_t x = load_one(e, ad);
            put_one(e, ad, x);
        }
    }
    __syncthreads();

    // register groups of two stages (radix 4): 36 data VGPRs, 96 in all.  Three stages between carry passes are possible arithmetically (limbs
    // reach 7 units of 2^29, still 32 bits; tools/ntt29_model.py) but eight elements in registers take the kernel to 250 VGPRs, or to
    // 100-190 spills under a 128 / 168 cap.
    for (uint32_t s = 1; s <= a.log_r;) {
        const uint32_t g = a.log_r - s >= 1 ? 2u : 1u;
        if (s > a.lds_stages) {
            if (g == 2) ntt_superstage29<2, NTT_THREADS, true>(data, d8, tws, TILE, logC, !twisted, s, tid);
            else ntt_superstage29<1, NTT_THREADS, true>(data, d8, tws, TILE, logC, !twisted, s, tid);
        } else {
            if (g == 2) ntt_superstage29<2, NTT_THREADS, false>(data, d8, tloc, TILE, logC, !twisted, s, tid);
            else ntt_superstage29<1, NTT_THREADS, false>(data, d8, tloc, TILE, logC, !twisted, s, tid);
        }
        __syncthreads();
        s += g;
    }

    // ---- store: y[k1] sits at row k1 ----
    auto store_one = [&](uint32_t e, const fe_t* twp) {
        uint32_t c = e & (C - 1), k1 = e >> logC;
        f29_t x = lds_get29(data, d8, TILE, (k1 << logC) + c);
        if (!a.last) {
            uint32_t colid = tile * C + c;
            uint32_t pos = (k1 << log_s) + (colid & (S - 1));        // position inside the block
            x = Fr29::mul(x, Fr29::unpack(twp ? *twp : ld_fe(tw_inter + pos)));          // below 2p: the work buffer holds 256-bit words
            st_fe(out + (((size_t)(colid >> log_s) << a.log_m) + pos), Fr29::pack(x));
        } else {
            uint32_t blk = (uint32_t)(col_base(c) >> a.log_r);
            size_t oidx = 0;
            uint32_t rem = blk, lw = a.log_n - a.log_r, shift = 0;
            for (uint32_t p = 0; p + 1 < a.npass; p++) {
                lw -= a.log_radix[p];
                uint32_t kp = rem >> lw;
                rem &= (1u << lw) - 1;
                oidx += (size_t)kp << shift;
                shift += a.log_radix[p];
            }
            oidx += (size_t)k1 << shift;
            x = a.post ? Fr29::cond_sub<0>(Fr29::mul(x, Fr29::unpack(a.post_c[oidx % 3]))) : Fr29::canonical(x);
            st_fe(out + oidx, Fr29::pack(x));
        }
    };
    if (TILE == 4 * NTT_THREADS && !a.last) {
        fe_t tw[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t e = tid + i * NTT_THREADS, c = e & (C - 1), k1 = e >> logC;
            tw[i] = ld_fe(tw_inter + ((k1 << log_s) + ((tile * C + c) & (S - 1))));
        }
#pragma unroll
        for (int i = 0; i < 4; i++) store_one(tid + i * NTT_THREADS, &tw[i]);
    } else {
        for (uint32_t e = tid; e < TILE; e += NTT_THREADS) store_one(e, nullptr);
    }
}

// out[idx] = w^(e(idx)):  mode 0: e = idx * mult ;  mode 1: idx = k1*S + i2, e = i2 * k1 * mult  (mod 2^log_n)
// init: 1 in the domain the table is wanted in (Fr::one() = the files' 2^256 domain; 32 there = the 2^261 domain of the radix-2^29 product)
__global__ void ntt_twiddle_kernel(fe_t* out, uint32_t count, uint64_t mult, uint32_t log_n, uint32_t log_s,
                                   int mode, const fe_t* pow2tab, fe_t init) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    uint64_t e;
    if (mode == 0) e = (uint64_t)idx * mult;
    else e = (uint64_t)(idx & ((1u << log_s) - 1)) * (uint64_t)(idx >> log_s) * mult;
    e &= ((uint64_t)1 << log_n) - 1;
    fe_t acc = init;
    for (uint32_t b = 0; b < log_n; b++)
        if ((e >> b) & 1) acc = Fr::mul(acc, ld_fe(pow2tab + b));
    st_fe(out + idx, acc);
}


// coset-major tables: out[b * cnt + m] = w_ext^(e mod 2^log_ext) * zeta^(z mod 3) with
//   mode 0 (stage twiddles of the twisted DIT first pass, cnt = R1): m = 2^(s-1) - 1 + o (stage s, offset o), q = n / 2^s,
//                                                 e = q * (b + E * o),     z = q          -> c_b^q * w_(2^s)^o
//   mode 1 (first-pass inter-pass table, cnt = n): m = k1 * S + i2,     e = E * i2 * k1 + b * i2, z = i2        -> w_n^(i2 k1) * c_b^i2
// for the coset generators c_b = zeta * w_ext^b, b < E = 2^log_e
// z0, z1, z2: zeta^0, zeta^1, zeta^2 in the wanted domain (all three = 1 for the stage table of a plain transform: log_e = 0, w_ext = w_n);
// out29: the entries as unpacked radix-2^29 limbs instead of 256-bit words
__global__ void ntt_coset_table_kernel(fe_t* out, f29_t* out29, uint32_t cnt, uint32_t log_e, uint32_t log_ext, uint32_t log_s, int mode, const fe_t* pow2tab,
                                       fe_t z0, fe_t z1, fe_t z2) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ((size_t)cnt << log_e)) return;
    const uint64_t b = idx / cnt, m = idx % cnt;
    uint64_t e, z;
    if (mode == 0) {
        uint32_t st = 1;                                                  // stage of entry m: 2^(st-1) - 1 <= m < 2^st - 1
        while (((uint64_t)1 << st) - 1 <= m) st++;
        const uint64_t o = m - (((uint64_t)1 << (st - 1)) - 1), log_nn = log_ext - log_e;
        if (st > log_nn) {                                              // padding entry
            if (out29) st_f29(out29 + idx, Fr29::zero());
            else st_fe(out + idx, Fr::zero());
            return;
        }
        const uint64_t q = (uint64_t)1 << (log_nn - st);
        e = q * (b + (o << log_e));
        z = q % 3;
    } else {
        const uint64_t i2 = m & (((uint64_t)1 << log_s) - 1), k1 = m >> log_s;
        e = ((i2 * k1) << log_e) + b * i2;
        z = i2 % 3;
    }
    e &= ((uint64_t)1 << log_ext) - 1;
    fe_t acc = z == 0 ? z0 : (z == 1 ? z1 : z2);
    for (uint32_t q = 0; q < log_ext; q++)
        if ((e >> q) & 1) acc = Fr::mul(acc, ld_fe(pow2tab + q));
    if (out29) st_f29(out29 + idx, Fr29::unpack(acc));
    else st_fe(out + idx, acc);
}
// natural <-> coset-major order of an extended column: nat[E j + b] = cm[b n + j].  One thread per j moves E elements: the coset-major
// side is coalesced across the wave, the natural side is E * 32 contiguous bytes per thread.
__global__ __launch_bounds__(256) void ntt_cm_transpose_kernel(const fe_t* in, fe_t* out, uint32_t log_n, uint32_t log_e, int to_natural) {
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >> log_n) return;
    const uint32_t E = 1u << log_e;
    for (uint32_t b = 0; b < E; b++) {
        const size_t cm = ((size_t)b << log_n) + j, nat = (j << log_e) + b;
        if (to_natural) st_fe(out + nat, ld_fe(in + cm));
        else st_fe(out + cm, ld_fe(in + nat));
    }
}

struct NttPlan {
    uint32_t log_n = 0;
    int npass = 0;
    uint32_t log_radix[4] = {0, 0, 0, 0};
    fe_t* tw_inter[4] = {nullptr, nullptr, nullptr, nullptr};      // 256-bit words holding w 2^261 mod p: the product divides by 2^261
    f29_t* stage29 = nullptr;     // the stage-major table w_(2^s)^o = omega^(o n / 2^s), 2^(largest radix) entries: its first R - 1 serve a pass of radix R
    fe_t n_inv;
};

// per-context state (Ctx::ntt_state): the twiddle tables are device memory of the context's device
struct CosetTables {
    f29_t* pre29 = nullptr;   // E x R1: the stage twiddles of the twisted first pass, as limbs in the 2^261 domain
    fe_t* inter = nullptr;    // E x n, same domain (nullptr for single-pass transforms)
};
struct NttState {
    std::map<std::string, NttPlan*> plans;
    std::map<uint64_t, CosetTables> coset_tables;       // key log_n | log_ext << 8
    bool attrs_set = false;
};
static NttState& ntt_state() {
    Ctx* c = ctx();
    if (!c->ntt_state) c->ntt_state = new NttState();
    return *static_cast<NttState*>(c->ntt_state);
}
#define g_plans (ntt_state().plans)
#define g_coset_tables (ntt_state().coset_tables)

// stages whose twiddles are staged in LDS (2^6 - 1 = 63 entries of 36 B: a 1024-element tile + table = 38.3 KiB, four workgroups per CU)
static uint32_t ntt_lds_stages(uint32_t log_r) { return log_r < 6 ? log_r : 6; }
static size_t ntt_lds_bytes(uint32_t log_tile, uint32_t log_r) { return 36u * ((size_t)1 << log_tile) + 36u * ((size_t)1 << ntt_lds_stages(log_r)); }
// 1 in the domain of the twiddles: the data are x 2^256 (the files' Montgomery form), the radix-2^29 product divides by 2^261 = 32 * 2^256
static fe_t ntt_domain_one() { return Fr::from_u64(32); }

// Pass radices.  A pass of radix 2^r works on tiles of 2^r rows x 4 adjacent columns (128-byte row segments) held in LDS; r <= 8 is a
// 36 KiB tile for a workgroup of 256 threads (4 workgroups per CU), r = 9 / 10 a 72 / 144 KiB tile for 512 / 1024 threads (16 waves per
// CU in every case).  Radix 2^10 would make 2^17 .. 2^20 two passes instead of three (one inter-pass twiddle product per element and a
// third of the traffic less) and was MEASURED twice: slower with the round-3 pass (2^20 -> 2^22 cosets 0.519 vs 0.494 ms per column,
// profiles/r03j_ntt_ab.log), level with this one (0.475 vs 0.472; the k = 20 MLP proof 72.6 vs 71.6 ms, profiles/r04ak_ab_ntt29r.log):
// sixteen waves behind one barrier wait for each other where four independent 4-wave workgroups fill each other's stalls, and the pass is
// issue-bound, not traffic-bound.  Default 8; EZKL_NTT_MAXR=9 / 10 select the larger tiles.
static uint32_t ntt_max_radix() {
    static const uint32_t v = [] {
        const char* e = getenv("EZKL_NTT_MAXR");
        const int x = e ? atoi(e) : 0;
        return (uint32_t)(x >= 6 && x <= 11 ? x : 8);
    }();
    return v;
}
static void plan_radices(uint32_t log_n, NttPlan* p) {
    if (log_n <= NTT_LOG_SINGLE) {
        p->npass = 1;
        p->log_radix[0] = log_n;
        return;
    }
    const uint32_t maxr = ntt_max_radix();
    int np = (int)((log_n + maxr - 1) / maxr);
    if (np < 2) np = 2;
    if (np > 4) np = 4;
    p->npass = np;
    uint32_t base = log_n / np, extra = log_n % np;
    for (int i = 0; i < np; i++) p->log_radix[i] = base + ((uint32_t)i < extra ? 1 : 0);
}
// tile of a multi-pass plan's pass of radix 2^r: at least 1024 elements, 4 columns
// (radix 2^11 -- EZKL_NTT_MAXR=11, a 2^22 transform in TWO passes -- keeps the 72 KiB tile of the single-pass transforms: ONE column per tile,
//  32-byte row segments; measured in round 6, profiles/r06j_ntt_maxr.log)
static uint32_t pass_log_tile(uint32_t log_r) { return log_r >= 11 ? log_r : (log_r + 2 > NTT_LOG_TILE ? log_r + 2 : NTT_LOG_TILE); }
static void launch_pass(const PassArgs& a, uint32_t tiles, unsigned blocks_y, size_t lds, hipStream_t st) {
    if (a.npass > 1 && a.log_tile == 12) hipLaunchKernelGGL(ntt_pass_kernel<1024>, dim3(tiles, blocks_y), dim3(1024), lds, st, a);
    else if (a.npass > 1 && a.log_tile == 11) hipLaunchKernelGGL(ntt_pass_kernel<512>, dim3(tiles, blocks_y), dim3(512), lds, st, a);
    else hipLaunchKernelGGL(ntt_pass_kernel<256>, dim3(tiles, blocks_y), dim3(256), lds, st, a);
}
static int ntt_kernel_attrs() {
    bool& attr_set = ntt_state().attrs_set;               // function attributes are per device
    if (!attr_set) {
        EZ_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EZ_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        EZ_HIP(hipFuncSetAttribute((const void*)ntt_pass_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    return EZKL_OK;
}

static int plan_get(Ctx* c, hipStream_t st, uint32_t log_n, const fe_t& omega, NttPlan** out) {
    std::string key((const char*)&omega, sizeof(fe_t));
    key.push_back((char)log_n);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) { *out = it->second; return EZKL_OK; }
    NttPlan* p = new NttPlan();
    p->log_n = log_n;
    plan_radices(log_n, p);
    // host: omega^(2^b), n^-1
    std::vector<fe_t> pow2(log_n ? log_n : 1);
    fe_t w = omega;
    for (uint32_t b = 0; b < log_n; b++) { pow2[b] = w; w = Fr::sqr(w); }
    p->n_inv = Fr::inv(Fr::from_u64((uint64_t)1 << log_n));
    fe_t* d_pow2 = nullptr;
    EZ_HIP(hipMalloc(&d_pow2, sizeof(fe_t) * pow2.size()));
    EZ_HIP(hipMemcpyAsync(d_pow2, pow2.data(), sizeof(fe_t) * pow2.size(), hipMemcpyHostToDevice, st));
    uint32_t log_m = log_n;
    const fe_t one = ntt_domain_one();
    {
        uint32_t maxlr = 0;
        for (int i = 0; i < p->npass; i++) maxlr = p->log_radix[i] > maxlr ? p->log_radix[i] : maxlr;
        const uint32_t cnt = 1u << maxlr;                       // w_(2^s)^o = omega^(o n / 2^s) whatever the pass: its first R - 1 entries serve a pass of radix R
        EZ_HIP(hipMalloc(&p->stage29, sizeof(f29_t) * cnt));
        hipLaunchKernelGGL(ntt_coset_table_kernel, dim3(cdiv(cnt, 256)), dim3(256), 0, st, (fe_t*)nullptr, p->stage29, cnt, 0u, log_n, 0u, 0, d_pow2, one, one, one);
    }
    for (int i = 0; i < p->npass; i++) {
        uint32_t lr = p->log_radix[i];
        if (i + 1 < p->npass) {
            uint32_t cnt = 1u << log_m;
            EZ_HIP(hipMalloc(&p->tw_inter[i], sizeof(fe_t) * (size_t)cnt));
            hipLaunchKernelGGL(ntt_twiddle_kernel, dim3(cdiv(cnt, 256)), dim3(256), 0, st, p->tw_inter[i], cnt,
                               (uint64_t)1 << (log_n - log_m), log_n, log_m - lr, 1, d_pow2, one);
        }
        log_m -= lr;
    }
    EZ_HIP(hipGetLastError());
    EZ_HIP(hipStreamSynchronize(st));
    EZ_HIP(hipFree(d_pow2));
    g_plans[key] = p;
    *out = p;
    return EZKL_OK;
}

// coset_mode: 0 plain; 1 coeff_to_extended (pre-multiply zeta^i, zero-pad from 2^in_log_len);
//             2 extended_to_coeff (post-multiply zeta^-i; implies inverse_scale)
static int ntt_run_chunk(Ctx* c, hipStream_t st, const fe_t* in, fe_t* out, uint32_t log_n, const fe_t& omega, bool inverse_scale,
                         size_t batch, size_t in_stride, size_t out_stride, uint32_t in_log_len, int coset_mode);

// columns are transformed in groups so that the ping-pong work buffer stays <= 4 GiB and gridDim.y <= 65535
int ntt_run(Ctx* c, hipStream_t st, const fe_t* in, fe_t* out, uint32_t log_n, const fe_t& omega, bool inverse_scale,
            size_t batch, size_t in_stride, size_t out_stride, uint32_t in_log_len, int coset_mode) {
    if (log_n > 28 || in_log_len > log_n || batch == 0) return EZKL_ERR_INVALID;
    size_t group = ((size_t)4 << 30) / ((size_t)32 << log_n);
    if (group < 1) group = 1;
    if (group > 32768) group = 32768;
    for (size_t b0 = 0; b0 < batch; b0 += group) {
        size_t nb = batch - b0 < group ? batch - b0 : group;
        int rc = ntt_run_chunk(c, st, in + b0 * in_stride, out + b0 * out_stride, log_n, omega, inverse_scale, nb, in_stride, out_stride,
                               in_log_len, coset_mode);
        if (rc) return rc;
    }
    return EZKL_OK;
}

static int ntt_run_chunk(Ctx* c, hipStream_t st, const fe_t* in, fe_t* out, uint32_t log_n, const fe_t& omega, bool inverse_scale,
                         size_t batch, size_t in_stride, size_t out_stride, uint32_t in_log_len, int coset_mode) {
    NttPlan* p = nullptr;
    int rc = plan_get(c, st, log_n, omega, &p);
    if (rc) return rc;
    const size_t n = (size_t)1 << log_n;
    if ((rc = ntt_kernel_attrs())) return rc;
    fe_t* work = nullptr;
    if (p->npass > 1) {
        rc = arena_reserve(scratch_arena(c, st), batch * n * sizeof(fe_t), st, (void**)&work);
        if (rc) return rc;
    }
    const fe_t dom = ntt_domain_one();                      // constants that enter a product carry the product's domain
    const fe_t zeta = Fr::mul(fr_const(FrConst::ZETA), dom), zeta2 = Fr::mul(fr_const(FrConst::ZETA2), dom);
    uint32_t log_m = log_n;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const char* tmode = getenv("EZKL_HIP_TIMING");                      // "none": no event pair around the passes (include/ezkl_hip.h)
    const bool timed = !(tmode && !strcmp(tmode, "none"));
    if (timed) {
        rc = ev_pair(c, coset_mode ? "coset_ntt" : "ntt", &e0, &e1);
        if (rc) return rc;
        EZ_HIP(hipEventRecord(e0, st));
    }
    for (int i = 0; i < p->npass; i++) {
        PassArgs a;
        memset(&a, 0, sizeof a);
        const bool first = (i == 0), last = (i + 1 == p->npass);
        a.in = first ? in : work;
        a.in_stride = first ? in_stride : n;
        a.out = last ? out : work;
        a.out_stride = last ? out_stride : n;
        a.tw_inter = p->tw_inter[i];
        a.tw_stage29 = p->stage29;
        a.log_n = log_n;
        a.log_r = p->log_radix[i];
        a.lds_stages = ntt_lds_stages(a.log_r);
        a.log_m = log_m;
        a.log_tile = p->npass == 1 ? log_n : pass_log_tile(a.log_r);
        if (a.log_tile < a.log_r) a.log_tile = a.log_r;
        a.first = first;
        a.last = last;
        a.in_log_len = first ? in_log_len : log_n;
        a.coset_pre = (first && coset_mode == 1);
        a.npass = (uint32_t)p->npass;
        for (int q = 0; q < 4; q++) a.log_radix[q] = p->log_radix[q];
        a.zeta[0] = zeta;
        a.zeta[1] = zeta2;
        if (last) {
            uint32_t logC = a.log_tile - a.log_r;
            a.k1_major = (p->npass >= 2 && p->log_radix[0] >= logC) ? 1u : 0u;
            if (inverse_scale || coset_mode == 2) {
                a.post = 1;
                fe_t s = Fr::mul(inverse_scale || coset_mode == 2 ? p->n_inv : Fr::one(), dom);
                a.post_c[0] = s;
                a.post_c[1] = coset_mode == 2 ? Fr::mul(s, fr_const(FrConst::ZETA2)) : s;   // zeta^-1 = zeta^2
                a.post_c[2] = coset_mode == 2 ? Fr::mul(s, fr_const(FrConst::ZETA)) : s;    // zeta^-2 = zeta
            }
        }
        const uint32_t tiles = 1u << (log_n - a.log_tile);
        launch_pass(a, tiles, (unsigned)batch, ntt_lds_bytes(a.log_tile, a.log_r), st);
        log_m -= a.log_r;
    }
    EZ_HIP(hipGetLastError());
    if (timed) EZ_HIP(hipEventRecord(e1, st));
    if (work) return arena_done(scratch_arena(c, st), st);
    return EZKL_OK;
}


// ---- coset-major extended evaluation: EvaluationDomain::coeff_to_extended as E = 2^(ext_k - k) transforms of n points -------------
// The extended domain {zeta w_ext^i} is the union of the E cosets c_b H, c_b = zeta w_ext^b, H = <w_n>, w_n = w_ext^E; natural index
// i = E j + b.  Evaluating p (degree < n) on coset b is an n-point NTT of (a_j c_b^j): the 2^(ext_k - k) zero-padding stages of the
// 4n-point transform never run, and a rotation by r rows is a shift by r INSIDE a coset, so the quotient sweep of coset b touches
// nothing but coset b of every column.  Output layout: out[b n + j] = p(c_b w_n^j).  The scaling c_b^j with j = i1 S + i2 is split:
// c_b^(S i1) multiplies the rows at the first pass's load, c_b^i2 is folded into that pass's (per-coset) inter-pass twiddle table.

static int coset_tables_get(Ctx* c, hipStream_t st, NttPlan* p, uint32_t log_n, uint32_t log_ext, const fe_t& w_ext, CosetTables* out) {
    const uint64_t key = (uint64_t)log_n | ((uint64_t)log_ext << 8);
    auto it = g_coset_tables.find(key);
    if (it != g_coset_tables.end()) { *out = it->second; return EZKL_OK; }
    const uint32_t log_e = log_ext - log_n, lr = p->log_radix[0], log_s = log_n - lr;
    std::vector<fe_t> pow2(log_ext ? log_ext : 1);
    fe_t w = w_ext;
    for (uint32_t b = 0; b < log_ext; b++) { pow2[b] = w; w = Fr::sqr(w); }
    fe_t* d_pow2 = nullptr;
    EZ_HIP(hipMalloc(&d_pow2, sizeof(fe_t) * pow2.size()));
    EZ_HIP(hipMemcpyAsync(d_pow2, pow2.data(), sizeof(fe_t) * pow2.size(), hipMemcpyHostToDevice, st));
    CosetTables t;
    const fe_t one = ntt_domain_one(), zeta = Fr::mul(fr_const(FrConst::ZETA), one), zeta2 = Fr::mul(fr_const(FrConst::ZETA2), one);
    const size_t n_pre = (size_t)1 << (lr + log_e);
    EZ_HIP(hipMalloc(&t.pre29, sizeof(f29_t) * n_pre));
    hipLaunchKernelGGL(ntt_coset_table_kernel, dim3(cdiv(n_pre, 256)), dim3(256), 0, st, (fe_t*)nullptr, t.pre29, 1u << lr, log_e, log_ext, log_s, 0, d_pow2, one, zeta, zeta2);
    if (p->npass > 1) {
        const size_t n_int = (size_t)1 << log_ext;
        EZ_HIP(hipMalloc(&t.inter, sizeof(fe_t) * n_int));
        hipLaunchKernelGGL(ntt_coset_table_kernel, dim3(cdiv(n_int, 256)), dim3(256), 0, st, t.inter, (f29_t*)nullptr, 1u << log_n, log_e, log_ext, log_s, 1, d_pow2, one, zeta, zeta2);
    }
    EZ_HIP(hipGetLastError());
    EZ_HIP(hipStreamSynchronize(st));
    EZ_HIP(hipFree(d_pow2));
    g_coset_tables[key] = t;
    *out = t;
    return EZKL_OK;
}

static int coset_cm_chunk(Ctx* c, hipStream_t st, const fe_t* in, fe_t* out, uint32_t log_n, uint32_t log_ext, const fe_t& w_n, const fe_t& w_ext, size_t batch,
                          size_t in_stride, size_t out_stride, uint32_t first_coset, uint32_t log_count) {
    NttPlan* p = nullptr;
    int rc = plan_get(c, st, log_n, w_n, &p);
    if (rc) return rc;
    CosetTables ct;
    if ((rc = coset_tables_get(c, st, p, log_n, log_ext, w_ext, &ct))) return rc;
    const uint32_t log_e = log_ext - log_n;
    const size_t n = (size_t)1 << log_n, blocks = batch << log_count;
    if ((rc = ntt_kernel_attrs())) return rc;
    fe_t* work = nullptr;
    if (p->npass > 1) {
        rc = arena_reserve(scratch_arena(c, st), blocks * n * sizeof(fe_t), st, (void**)&work);
        if (rc) return rc;
    }
    uint32_t log_m = log_n;
    hipEvent_t e0, e1;
    if ((rc = ev_pair(c, "coset_ntt", &e0, &e1))) return rc;
    EZ_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < p->npass; i++) {
        PassArgs a;
        memset(&a, 0, sizeof a);
        const bool first = (i == 0), last = (i + 1 == p->npass);
        a.in = first ? in : work;
        a.in_stride = first ? in_stride : n;
        a.out = last ? out : work;
        a.out_stride = last ? out_stride : n;
        a.tw_inter = first ? ct.inter : p->tw_inter[i];
        a.tw_stage29 = p->stage29;
        a.tw_pre29 = ct.pre29;
        a.log_n = log_n;
        a.log_r = p->log_radix[i];
        a.lds_stages = ntt_lds_stages(a.log_r);
        a.log_m = log_m;
        a.log_tile = p->npass == 1 ? log_n : pass_log_tile(a.log_r);
        if (a.log_tile < a.log_r) a.log_tile = a.log_r;
        a.first = first;
        a.last = last;
        a.in_log_len = log_n;
        a.npass = (uint32_t)p->npass;
        for (int q = 0; q < 4; q++) a.log_radix[q] = p->log_radix[q];
        a.cm = 1;
        a.cm_log_e = log_e;
        a.cm_first = first_coset;
        a.cm_log_cnt = log_count;
        if (last) {
            const uint32_t logC = a.log_tile - a.log_r;
            a.k1_major = (p->npass >= 2 && p->log_radix[0] >= logC) ? 1u : 0u;
        }
        const uint32_t tiles = 1u << (log_n - a.log_tile);
        launch_pass(a, tiles, (unsigned)blocks, ntt_lds_bytes(a.log_tile, a.log_r), st);
        log_m -= a.log_r;
    }
    EZ_HIP(hipGetLastError());
    EZ_HIP(hipEventRecord(e1, st));
    if (work) return arena_done(scratch_arena(c, st), st);
    return EZKL_OK;
}
// `batch` coefficient columns (2^log_n each, in_stride apart) -> their coset-major extended forms (2^log_ext each, out_stride apart)
int coset_cm_run(Ctx* c, hipStream_t st, const fe_t* in, fe_t* out, uint32_t log_n, uint32_t log_ext, const fe_t& w_n, const fe_t& w_ext, size_t batch,
                 size_t in_stride, size_t out_stride, uint32_t first_coset, uint32_t n_cosets) {
    if (log_ext > 28 || log_n > log_ext || log_ext - log_n > 6 || batch == 0) return EZKL_ERR_INVALID;
    const uint32_t E = 1u << (log_ext - log_n);
    if (n_cosets == 0) { first_coset = 0; n_cosets = E; }              // all of them
    if ((n_cosets & (n_cosets - 1)) || first_coset + n_cosets > E) return EZKL_ERR_INVALID;
    uint32_t log_count = 0;
    while ((1u << log_count) < n_cosets) log_count++;
    size_t group = ((size_t)4 << 30) / ((size_t)32 << (log_n + log_count));
    if (group < 1) group = 1;
    if ((group << log_count) > 32768) group = 32768 >> log_count;
    for (size_t b0 = 0; b0 < batch; b0 += group) {
        const size_t nb = batch - b0 < group ? batch - b0 : group;
        int rc = coset_cm_chunk(c, st, in + b0 * in_stride, out + b0 * out_stride, log_n, log_ext, w_n, w_ext, nb, in_stride, out_stride, first_coset, log_count);
        if (rc) return rc;
    }
    return EZKL_OK;
}
int cm_transpose(Ctx* c, hipStream_t st, const fe_t* in, fe_t* out, uint32_t log_n, uint32_t log_e, bool to_natural) {
    (void)c;
    if (in == out || log_n + log_e > 28) return EZKL_ERR_INVALID;
    hipLaunchKernelGGL(ntt_cm_transpose_kernel, dim3(cdiv((size_t)1 << log_n, 256)), dim3(256), 0, st, in, out, log_n, log_e, to_natural ? 1 : 0);
    EZ_HIP(hipGetLastError());
    return EZKL_OK;
}

}  // namespace ezkl
