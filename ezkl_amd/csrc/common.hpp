// common.hpp -- library context, error plumbing, launch helpers (host side of libezkl_hip.so)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <rocprofiler-sdk-roctx/roctx.h>
#include "../../include/ezkl_hip.h"
#include "field.hpp"

namespace ezkl {

struct Ctx {
    int device = -1;
    hipStream_t stream = nullptr;     // library stream (used when the caller passes NULL)
    hipEvent_t order_event = nullptr;  // ezkl_hip_stream_wait_library
    int num_cus = 256;
    std::recursive_mutex mu;          // serialises calls on this device (halo2 calls from rayon workers)
    // HIP event pairs recorded on the stream the kernels run on, per measured region ("ntt", "coset_ntt", "msm", "msm_accumulate", ...).
    // A RING of pairs per region: every acquisition (ev_pair) gets a pair of its own, so regions recorded back to back -- or from two
    // streams -- never re-record a pair that has not been read yet; a pair is harvested (elapsed time added to sum_ms / count) when the ring
    // comes round to it or when the statistics are read (ezkl_hip_kernel_ms_stats).  ezkl_hip_last_kernel_ms reads the newest pair.
    struct EvRing {
        static constexpr unsigned N = 64;
        hipEvent_t e0[N], e1[N];
        bool made[N];
        uint64_t head = 0, tail = 0;      // pairs [tail, head) are recorded and not harvested yet
        double sum_ms = 0;
        uint64_t count = 0, skipped = 0;   // skipped: pairs whose events could not be read (taken by a call that failed before recording them)
        EvRing() { for (unsigned i = 0; i < N; i++) { e0[i] = e1[i] = nullptr; made[i] = false; } }
    };
    std::map<std::string, EvRing> events;
    // Device arenas reused across calls (grown on demand, never shrunk).  `scratch` backs the NTT ping-pong buffer and
    // the sweep's spill area; `aux` backs the small helpers.  An arena remembers the stream and an event of its last
    // user: the next user on a DIFFERENT stream waits for that event first, so stream-ordered calls on caller streams
    // never race on shared scratch.
    struct Arena {
        void* ptr = nullptr;
        size_t bytes = 0;
        hipStream_t last_stream = nullptr;
        hipEvent_t last_event = nullptr;
        bool in_use = false;
    };
    Arena scratch, aux;
    // `scratch` as seen from a stream other than the library stream: an arena of its own per stream.  The prover transforms every finished
    // column on the context's side stream while the helper programs of the next argument run on the library stream; with ONE arena the
    // rule above made each side wait for everything the other had queued -- a gate program of a few kilobytes of arguments waited for a
    // 10 ms burst of NTTs whose work buffer it never touches (round 4: the multiplicity phase of the k = 20 MLP, 12 -> 6 ms).
    std::map<hipStream_t, Arena> scratch_by_stream;
    // Everything a module keeps between calls lives in the context it was made for (MSM window tables, slots and streams; NTT plans and
    // coset tables; JIT modules; the column pool): several contexts -- one per device of a single-process multi-GPU prover, or several
    // on one device -- never share device state.  The modules own these (created on first use, see msm.hip / ntt.hip / evalh.hip / capi.hip).
    void* msm_state = nullptr;
    void* ntt_state = nullptr;
    void* jit_state = nullptr;
    void* pool_state = nullptr;
    int index = 0;                    // position in the context table (ezkl_hip_set_context)
    void* staging_state = nullptr;    // pinned staging ring for small host -> device copies (capi.hip)
    hipStream_t side_stream = nullptr;  // ezkl_hip_context_stream: a second stream that lives as long as the context
};

Ctx* ctx();                 // the context the CALLING THREAD is bound to (ezkl_hip_set_context; default 0), lazily initialised (nullptr + last error set if no device)
int ctx_init(int device);
int ctx_count();
int ctx_bind(int idx);
int ctx_device_of(int idx);
int set_hip_error(hipError_t e, const char* what, const char* file, int line);
int arena_reserve(Ctx::Arena& a, size_t bytes, hipStream_t st, void** out);   // acquire for work on stream st
int arena_done(Ctx::Arena& a, hipStream_t st);                               // mark the end of that work
int ev_pair(Ctx* c, const char* key, hipEvent_t* e0, hipEvent_t* e1);
int ev_harvest(Ctx::EvRing& r, bool all);
// Pinned staging for small host arrays a call borrows: acquire a block (nullptr: none to be had -- copy from the caller's memory and
// synchronise instead), fill it, queue the copy out of it on `st`, release it on `st` (the block is reused once that copy has run).
uint8_t* staging_acquire(Ctx* c, size_t bytes, void** token);
int staging_release(void* token, hipStream_t st);

#define EZ_HIP(call)                                                              \
    do {                                                                          \
        hipError_t _e = (call);                                                   \
        if (_e != hipSuccess) return ::ezkl::set_hip_error(_e, #call, __FILE__, __LINE__); \
    } while (0)

// a roctx range per C-ABI call: `rocprofv3 --marker-trace` shows ezkl_hip_msm_g1_dev, ezkl_hip_ntt_dev ... as named
// host ranges above the kernels they launch; a push / pop pair costs nanoseconds when no tool is attached
struct RoctxRange {
    explicit RoctxRange(const char* name) { roctxRangePushA(name); }
    ~RoctxRange() { roctxRangePop(); }
};

#define EZ_CTX(c)                                   \
    ::ezkl::RoctxRange _roctx(__func__);            \
    ::ezkl::Ctx* c = ::ezkl::ctx();                 \
    if (!c) return EZKL_ERR_NO_DEVICE;              \
    std::lock_guard<std::recursive_mutex> _lk(c->mu); \
    EZ_HIP(hipSetDevice(c->device))

// Stream priorities (EZKL_HIP_PRIO_<LIB|AUX|MSM> = -1 high / 0 normal / 1 low, clamped to the device's range): the library stream carries
// the helper chains every commitment of a proof waits for, the context's side stream the NTT forms that are only needed by the quotient
// sweep, the MSM slot streams the commitments themselves.
static inline hipError_t stream_create_prio(hipStream_t* st, const char* env_name, int dflt) {
    int prio = dflt;
    if (const char* e = getenv(env_name)) prio = atoi(e);
    int lo = 0, hi = 0;                                   // lo = least priority (largest number), hi = greatest
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); return hipStreamCreateWithFlags(st, hipStreamNonBlocking); }
    if (prio > lo) prio = lo;
    if (prio < hi) prio = hi;
    if (prio == 0) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio);
}
static inline Ctx::Arena& scratch_arena(Ctx* c, hipStream_t st) { return st == c->stream ? c->scratch : c->scratch_by_stream[st]; }
static inline hipStream_t pick_stream(Ctx* c, void* s) { return s ? (hipStream_t)s : c->stream; }
static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// -------- module entry points implemented in the .hip files --------
struct Bases {
    fe_t* pts = nullptr;   // n x (x,y) affine
    size_t n = 0;
};

int ntt_run(Ctx* c, hipStream_t st, const fe_t* in, fe_t* out, uint32_t log_n, const fe_t& omega, bool inverse_scale,
            size_t batch, size_t in_stride, size_t out_stride, uint32_t in_log_len, int coset_mode);
int coset_cm_run(Ctx* c, hipStream_t st, const fe_t* in, fe_t* out, uint32_t log_n, uint32_t log_ext, const fe_t& w_n, const fe_t& w_ext, size_t batch,
                 size_t in_stride, size_t out_stride, uint32_t first_coset = 0, uint32_t n_cosets = 0);
int cm_transpose(Ctx* c, hipStream_t st, const fe_t* in, fe_t* out, uint32_t log_n, uint32_t log_e, bool to_natural);
int msm_run(Ctx* c, hipStream_t st, const Bases* b, size_t base_offset, const fe_t* scalars_dev, size_t n,
            void* out_affine_host);
int msm_run_batch(Ctx* c, hipStream_t st, const Bases* b, size_t base_offset, const fe_t* const* scalars, size_t batch, size_t n,
                  void* out_host, bool small_scalars = false);
struct MsmUpload;
int msm_upload_begin(Ctx* c, const fe_t* const* host_cols, fe_t* const* dev_cols, size_t batch, size_t n, const fe_t* const* tails, size_t tail_start,
                     size_t tail_count, MsmUpload** out, const uint8_t* formats = nullptr);
int msm_upload_wait(MsmUpload* u, size_t j, hipStream_t st);
int msm_upload_commit(Ctx* c, MsmUpload* u, const Bases* b, size_t commit_first, size_t commit_count, void* out_host);
int msm_upload_end(MsmUpload* u);
struct MsmBatch;
int msm_batch_begin(Ctx* c, hipStream_t st, const Bases* b, size_t base_offset, size_t n, MsmBatch** out);
int msm_batch_push(Ctx* c, MsmBatch* mb, const fe_t* scalars_dev);
int msm_batch_push_many(Ctx* c, MsmBatch* mb, const fe_t* const* cols, size_t count, hipStream_t after);
int msm_batch_finish(Ctx* c, MsmBatch* mb, void* out_host, size_t capacity);
int vec_op(Ctx* c, hipStream_t st, int op, const fe_t* a, const fe_t* b, fe_t* o, size_t n);
int vec_fill(Ctx* c, hipStream_t st, fe_t* o, const fe_t& v, size_t n);
int vec_scale(Ctx* c, hipStream_t st, const fe_t* a, const fe_t& s, fe_t* o, size_t n);
int perm_sigma(Ctx* c, hipStream_t st, const uint32_t* next, const fe_t* omega_col, const fe_t* delta_pows, uint32_t log_n, uint32_t m, fe_t* out);
int divide_by_vanishing(Ctx* c, hipStream_t st, fe_t* a, uint32_t k, uint32_t ext_k);
int batch_invert(Ctx* c, hipStream_t st, fe_t* a, size_t n);
int lookup_multiplicity_batch(Ctx* c, hipStream_t st, const fe_t* const* inputs, const uint32_t* which, uint32_t n_items, const fe_t* const* tables,
                              uint32_t n_lookups, uint32_t n_rows, uint32_t usable, fe_t* const* m_outs, uint32_t* missing_host, uint32_t* missing_dev);
int lookup_multiplicity(Ctx* c, hipStream_t st, const fe_t* const* inputs, uint32_t n_inputs, const fe_t* table, uint32_t n_rows,
                        uint32_t usable, fe_t* m_out, uint32_t* missing_host, uint32_t* missing_dev);
int eval_poly(Ctx* c, hipStream_t st, const fe_t* coeffs, size_t n, const fe_t& x, void* out_host);
int prefix_scan(Ctx* c, hipStream_t st, int op, int exclusive, const fe_t* in, fe_t* out, size_t n);
int eval_poly_batch(Ctx* c, hipStream_t st, const fe_t* const* coeffs, const fe_t* xs, uint32_t m, size_t n, void* out_host);
int lincomb(Ctx* c, hipStream_t st, const fe_t* const* in, const fe_t* coeffs, uint32_t m, fe_t* out, size_t n, int accumulate);
int kate_division(Ctx* c, hipStream_t st, const fe_t* a, const fe_t& z, fe_t* out, size_t n);
int chacha20_fr(Ctx* c, hipStream_t st, const uint32_t key[8], uint64_t stream, size_t first, fe_t* out, size_t n);
bool msm_upload_is_open();
int msm_run_concurrent(Ctx* c, std::unique_lock<std::recursive_mutex>& lk, const Bases* b, size_t base_offset, const fe_t* scalars, size_t n, void* out_host);
int msm_call_start(Ctx* c, std::unique_lock<std::recursive_mutex>& lk, const Bases* b, size_t base_offset, const fe_t* scalars, size_t n, int* slot_out);
int msm_call_finish(Ctx* c, std::unique_lock<std::recursive_mutex>& lk, int slot, void* out_host);
int eval_program(Ctx* c, hipStream_t st, const ezkl_program_t* p, fe_t* out, bool ordered);
int eval_jit_compile_only(const ezkl_program_t* p);
int eval_prepare(Ctx* c, const ezkl_program_t* p);
int eval_schedule_only(const ezkl_program_t* p, uint32_t* out_code);
void eval_jit_stats(uint64_t* compiled, uint64_t* from_disk, uint64_t* hits);
int ubench(Ctx* c, const char* which, double* out);
void msm_table_drop(const Bases* b);
int msm_table_prepare(Ctx* c, const Bases* b);
int g1_mul_fixed(Ctx* c, hipStream_t st, const void* base_host, const fe_t* scalars, size_t n, void* out_dev);
int gen_bases(Ctx* c, hipStream_t st, uint64_t seed, size_t first, size_t n, void* out_dev);
int g1_to_lagrange(Ctx* c, hipStream_t st, const void* g_dev, uint32_t log_n, fe_t* tw_mont, const fe_t& ninv_mont, void* out_dev);
void g1_add_affine_host(const void* a, const void* b, void* out);
int g2_msm(Ctx* c, hipStream_t st, const void* pts_host, const void* scalars_host, size_t n, void* out_host);

}  // namespace ezkl
