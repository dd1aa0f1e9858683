/*
 * ezkl_hip.h -- C ABI of libezkl_hip.so, the MI355X (gfx950) backend for the `ezkl prove` hot path.
 *
 * This is the drop-in boundary of SURVEY.md §8(b): the functions below are what the halo2 fork's
 * `halo2_proofs::icicle` glue asks of the icicle runtime today (gate: cargo feature `gpu-accelerated`,
 * /root/reference/Cargo.toml:259; runtime gate ENABLE_ICICLE_GPU / ICICLE_SMALL_K,
 * /root/reference/README.md:106-122; device selection /root/reference/src/execute.rs:84-97).
 * Plain pointers and sizes only; no C++/torch types; never throws or unwinds across the boundary.
 *
 * Representation contract (fixture-verified, SURVEY.md §8(b,c)):
 *   Fr / Fq element : 32 bytes, 4 x u64 little-endian limbs, Montgomery form (R = 2^256), fully reduced
 *   G1 affine point : 64 bytes, x || y (Fq), identity encoded as (0,0)  -- the raw-bytes SRS layout
 * Results are canonical (fully reduced, affine), so equality with the CPU prover is byte equality.
 *
 * Threading: every entry point is callable from any host thread (halo2 calls from rayon workers);
 * the library lazily initialises on first use (the Python bindings never call set_device(),
 * /root/reference/src/bindings/python.rs:1056-1075) and serialises per device internally.
 */
#ifndef EZKL_HIP_H
#define EZKL_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (0 = ok, negative = error; see ezkl_hip_strerror) ---- */
#define EZKL_OK 0
#define EZKL_ERR_NO_DEVICE (-1)   /* no HIP device visible: the product path fails loudly, no CPU fallback */
#define EZKL_ERR_HIP (-2)         /* a HIP runtime call failed; ezkl_hip_last_hip_error() has the code */
#define EZKL_ERR_INVALID (-3)     /* bad argument (null pointer, size out of range, bad program) */
#define EZKL_ERR_NOMEM (-4)
#define EZKL_ERR_UNSUPPORTED (-5)
#define EZKL_ERR_TIMEOUT (-7)     /* a collective did not complete within EZKL_COMM_TIMEOUT_S seconds (comm.hip's watchdog): the communicator is aborted */
#define EZKL_ERR_BUSY (-6)        /* the calling thread already holds every slot of a bounded resource (ezkl_hip_msm_g1_start_dev: four per context) */


/* ---- environment: every variable libezkl_hip.so and libezkl_prover.so read, in ONE place (round 6: the A/B switches of rounds 3-5 --
 *      EZKL_MSM_PREFETCH / LEAN / FETCH_ALWAYS / UNPACK_FIRST / RESET_ZZ / COOP / ZEROCOPY / ORDER_ALWAYS / NO_TAPER / FIXUP_TREE /
 *      DEBUG_PLANES, EZKL_NTT_LOAD_ALWAYS / SKIP_UNIT2, EZKL_PROVER_MERGED_COMMITS -- are gone from the sources; their measurements are in profiles/).  All are read
 *      in-process (getenv); unset = the default.  A fork exposes the first group, the rest is for whoever works on the library.
 *
 *  gate and modes (what a deployment sets)
 *   ENABLE_HIP_GPU            unset   the runtime gate, as ENABLE_ICICLE_GPU: ezkl_hip_enabled(k) is 0 without it
 *   HIP_SMALL_K               8       circuits with k <= this stay on the CPU prover, as ICICLE_SMALL_K
 *   LOCAL_RANK                unset   ezkl_hip_init(-1): one context on this device instead of one per visible device
 *   EZKL_KEY_COSETS           auto    auto | resident | recompute: how a proving key is held (ezkl_prover.h: the streamed / degraded mode)
 *   EZKL_PROVER_SKIP_FIT_CHECK unset  1: no up-front HBM fit check at keygen / key load
 *   EZKL_COMM_TIMEOUT_S       120     watchdog of every collective (0: none): EZKL_ERR_TIMEOUT instead of a hang
 *   EZKL_COMM_SELFTEST        1       0: ezkl_hip_comm_init does not run ezkl_hip_comm_selftest (same value on every rank)
 *   EZKL_COMM_SLAB_MB         128     bytes per peer and round of the packed all-to-all (same value on every rank)
 *   EZKL_HIP_CACHE_DIR        ~/.cache/ezkl_hip   where compiled sweep kernels are kept ("off": nowhere)
 *   EZKL_HIP_POOL_CAP_GB      25 % of the device, at least what is live x 1.5   floor of the column pool's footprint bound
 *   EZKL_PK_READ_THREADS      4       reader threads of ezkl_prover_pk_read_file
 *   EZKL_HIP_LIB / EZKL_PROVER_LIB    (Python bindings) another build of the two libraries
 *  measurement
 *   EZKL_HIP_TIMING           all     all | kernel | none: which HIP event pairs a timed call records (ezkl_hip_kernel_ms_stats)
 *   EZKL_MSM_HOST_TIMING, EZKL_PROVER_KEYGEN_TIMING     host-side stage times on stderr
 *   EZKL_HIP_DEBUG, EZKL_MSM_DEBUG, EZKL_HIP_JIT_DEBUG, EZKL_HIP_JIT_DUMP=<file>     diagnostics on stderr / the generated sweep source
 *  tuning knobs whose defaults are the measured optimum (NOTEBOOK.md §4; tools/ab.sh sweeps them)
 *   EZKL_MSM_SLOTS 4 | EZKL_MSM_GROUP / _SMALL 6 / _BIG 4 | EZKL_MSM_L (device-chosen) | EZKL_MSM_LMIN 8 | EZKL_MSM_SPAN 16 | EZKL_MSM_E 8
 *   EZKL_NTT_MAXR 8 (9 / 10: 2048- / 4096-element tiles) | EZKL_PROVER_SWEEP_TERMS 0 | EZKL_PROVER_SWEEP_INSTRS 640
 *  comparison switches kept because tests or the multi-rank fallbacks use them
 *   EZKL_EVALH_MODE=interp (the sweep interpreter instead of the JIT kernel), EZKL_EVALH_NO_SCHEDULE, EZKL_COMM_UNPACKED=1 (one send / recv per
 *   segment), EZKL_COMM_SELF_VIA_RCCL=1 (world 1 through the wire format), EZKL_GATHER_HOST, EZKL_MSM_SERIAL_CALLS, EZKL_HIP_POOL_SYNC,
 *   EZKL_PROVER_SYNC_CALLS, EZKL_PROVER_NO_EARLY_RANDOM, EZKL_PROVER_NO_SUM_SCATTER,
 *   EZKL_PROVER_ASSUME_FREE_GIB=<x> (test hook of the fit check) ---- */

typedef struct ezkl_bases_s* ezkl_bases_t;     /* device-resident G1 base set (SRS g or g_lagrange) */

/* ---- device management: replaces icicle try_load_and_set_backend_device("CUDA") + warmup(),
 *      /root/reference/src/execute.rs:88-95 ---- */
/* ezkl_hip_init(device >= 0): one context on that device.  ezkl_hip_init(-1): ALL visible devices, one context per device (context i
 * on device i) -- `ezkl prove` is one process (src/execute.rs:1575-1627) -- unless LOCAL_RANK is set (a launcher started one process
 * per GPU): then one context on device LOCAL_RANK.  Idempotent, thread-safe. */
int ezkl_hip_init(int device);
/* ---- contexts: single-process multi-GPU ----
 * A context = one device + everything the library keeps for it (streams, scratch arenas, MSM window tables and slots, NTT plans,
 * JIT modules, the column pool), behind its own lock.  Every host thread works on the context it bound itself to
 * (ezkl_hip_set_context; new threads start on context 0), so N threads drive N GPUs concurrently and handles (bases, device pointers,
 * upload phases, batches) belong to the context they were made on.  ezkl_hip_contexts_configure(n, devices) sets the table
 * explicitly BEFORE first use -- several contexts may name the same device (how the multi-device prover is tested on a one-GPU box).
 * ezkl_hip_memcpy_peer: device-to-device copy INTO the calling thread's context (dst_context must be it) from another context
 * (hipMemcpyPeerAsync across devices), ordered on the calling context's library stream; the source may be reused once this context has
 * synchronised (ezkl_hip_synchronize). */
int ezkl_hip_contexts_configure(int n_contexts, const int* devices);
int ezkl_hip_context_count(void);
int ezkl_hip_set_context(int index);
int ezkl_hip_context_device(int index);                /* device ordinal of a context, -1 if out of range */
int ezkl_hip_memcpy_peer(void* dst_dev, int dst_context, const void* src_dev, int src_context, size_t bytes);
int ezkl_hip_warmup(void);
int ezkl_hip_device_count(void);
int ezkl_hip_mem_info(size_t* free_bytes, size_t* total_bytes);   /* hipMemGetInfo of the calling context's device */
/* The column pool of the calling context (ezkl_hip_malloc / ezkl_hip_free recycle blocks by exact size; icicle's DeviceVec allocations
 * behind the reference's wrappers play this role).  live + parked bytes are held to max(25 % of the device / contexts sharing it,
 * 1.5 x the high-water mark of the live bytes); least recently used size classes are released first; an out-of-memory hipMalloc
 * releases what every idle context of the device has parked and retries once.  EZKL_HIP_POOL_CAP_GB overrides the 25 %.
 * ezkl_hip_pool_stats: out[0] = bytes handed out now, out[1] = their high-water mark, out[2] = bytes parked, out[3] = the bound.
 * ezkl_hip_pool_trim: release every parked block of the calling context and forget the high-water mark. */
int ezkl_hip_pool_stats(size_t out[4]);
int ezkl_hip_pool_trim(void);
int ezkl_hip_synchronize(void);
/* caller streams for the `stream` arguments below (a hipStream_t created by the caller works just as well; these exist so
 * that a client of this header alone can use the stream-ordered mode): work queued on one is asynchronous */
int ezkl_hip_stream_create(void** out_stream);
int ezkl_hip_stream_synchronize(void* stream);
int ezkl_hip_stream_destroy(void* stream);
/* A second stream owned by the calling thread's context, created on first use and alive as long as the context: creating and
 * destroying a HIP stream costs ~2 ms each (measured, profiles/r03v_hosttrace.txt), which a per-proof auxiliary stream would pay in
 * every proof.  Not to be destroyed by the caller; one per context, so two concurrent users of one context must agree on it. */
int ezkl_hip_context_stream(void** out_stream);
/* Asynchronous library stream.  By default a call with stream == NULL returns when its work is done.  A host that issues hundreds of
 * small device-only calls per proof (vec ops, scans, NTTs, inversions: the lookup / permutation helper chains) pays a host round trip
 * for each; ezkl_hip_set_async(1, &prev) makes those calls return as soon as they are queued on the library stream -- still in order
 * with each other.  Entry points that return host data (MSMs, eval_poly*, lookup_multiplicity, memcpy_d2h) or borrow host memory
 * (memcpy_h2d, eval_h_dev, divide_by_vanishing) keep synchronising by themselves.  ezkl_hip_set_async(0, ..) drains the stream.
 * The mode is a property of the CALLING THREAD: concurrent provers on other threads, and unrelated callers, keep their own setting.
 * ezkl_hip_stream_wait_library(s): work queued on the caller stream s from now on waits for everything queued on the library
 * stream so far (a column produced by library-stream calls, consumed on s). */
int ezkl_hip_set_async(int on, int* previous);
int ezkl_hip_stream_wait_library(void* stream);
const char* ezkl_hip_strerror(int code);
int ezkl_hip_last_hip_error(void);
const char* ezkl_hip_version(void);
/* The runtime gate, in one place for every host language: 1 if the environment variable ENABLE_HIP_GPU is set (to anything:
 * like ENABLE_ICICLE_GPU it is disabled by unsetting it, /root/reference/README.md:106-122) and k > HIP_SMALL_K (default 8, the
 * role of ICICLE_SMALL_K: below the cutoff the caller keeps its CPU path), else 0.  Touches no device. */
int ezkl_hip_enabled(uint32_t k);

/* ---- raw device memory for resident columns (library-owned until freed) ----
 * Freed blocks are recycled by size without returning to the driver (no implicit device synchronisation, unlike hipFree):
 * a block that was used on a CALLER stream must not be freed before that stream has been synchronised.  Blocks only
 * touched through stream = NULL calls are always safe to free. */
int ezkl_hip_malloc(void** dptr, size_t bytes);
int ezkl_hip_free(void* dptr);
int ezkl_hip_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
/* page-locked host memory for columns that will be uploaded (the witness: A x 2^k x 32 B per proof).  Copies from it run
 * without the runtime's pageable staging step; a fork would allocate the advice / instance vectors that synthesize() fills
 * from here.  (On the measured box the blocking pageable path was already as fast: 14 x 32 MiB in ~10 ms either way.) */
int ezkl_hip_host_malloc(void** host_ptr, size_t bytes);
int ezkl_hip_host_free(void* host_ptr);
int ezkl_hip_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);

/* ---- MSM: replaces ParamsKZG::commit / commit_lagrange -> halo2curves::msm (CPU) / icicle msm (GPU);
 *      in-tree call site /root/reference/src/circuit/modules/polycommit.rs:71 ---- */
/* bases: n affine points, host pointer, copied to HBM once (SRS load time, src/pfsys/srs.rs:40-47) */
int ezkl_hip_bases_upload(const void* affine_pts, size_t n, ezkl_bases_t* out_handle);
/* start the per-base-set precompute (the window tables of msm.hip, ~55 ms for 2^20 points) on a library stream WITHOUT waiting: a
 * one-shot `prove` (src/execute.rs:1575-1627) uploads the SRS, calls this, and reads the proving key while the tables are built.
 * Optional: the first MSM on a base set builds them itself (blocking) when this was not called.  The icicle build does the analogous
 * work at ParamsKZG::read time. */
int ezkl_hip_bases_prepare(ezkl_bases_t h);
int ezkl_hip_bases_free(ezkl_bases_t h);
size_t ezkl_hip_bases_len(ezkl_bases_t h);
/* synthetic base set for benchmarks/tests (no public SRS without network, src/pfsys/srs.rs:10-11):
 * n deterministic curve points by try-and-increment, generated on the device */
int ezkl_hip_bases_generate(uint64_t seed, size_t first, size_t n, ezkl_bases_t* out_handle);
int ezkl_hip_bases_download(ezkl_bases_t h, void* out_host /* n x 64 B */);
/* base set bases[i] = scalars[i] * P for one point P (64 B affine, host) and n resident Montgomery scalars: SRS
 * generation for tests (gen_srs -> ParamsKZG::setup, /root/reference/src/pfsys/srs.rs:14-16): g[i] = s^i G,
 * g_lagrange[i] = L_i(s) G */
int ezkl_hip_bases_from_scalars(const void* base_point, const void* scalars_dev, size_t n, ezkl_bases_t* out_handle);
/* ParamsKZG::downsize(new_k) -- halo2; called by load_params_prover (/root/reference/src/execute.rs:1739-1750) whenever the SRS file is
 * larger than the circuit, the normal case with a shared kzg22.srs: `g` (at least 2^new_k points of the coefficient basis s^i G) ->
 * out_g = its first 2^new_k points (may be NULL), out_g_lagrange = the Lagrange basis of the 2^new_k-point domain, L_i(s) G, computed as
 * halo2's g_to_lagrange does: an inverse NTT over G1 (omega^-1 butterflies on group elements, scaled by 1 / 2^new_k).  New handles,
 * freed with ezkl_hip_bases_free. */
int ezkl_hip_bases_downsize(ezkl_bases_t g, uint32_t new_k, ezkl_bases_t* out_g, ezkl_bases_t* out_g_lagrange);
/* sum_i scalars[i] * points[i] over G2 (the twist over Fq2).  points: n x 128 B affine -- x.c0, x.c1, y.c0, y.c1 as 32-byte little-endian
 * Montgomery Fq, the layout of the g2 / s_g2 tail of an SRS file, (0, 0) = identity; scalars: n x 32 B Montgomery Fr; out: 128 B canonical
 * affine.  Host pointers, borrowed.  On the reference's prove path G2 is data (the verifier's pairing consumes g2 and s_g2); the one G2
 * computation is gen_srs's s_g2 = [s] g2 (/root/reference/src/pfsys/srs.rs:14-16 -> ParamsKZG::setup), i.e. n = 1.  A plain
 * double-and-add per pair with a tree fold: not a Pippenger pipeline, not a hot path. */
int ezkl_hip_msm_g2(const void* points_affine, const void* scalars, size_t n, void* out_affine);
/* sum_i scalars[i] * bases[offset + i], i < n.  scalars: n x 32 B Montgomery Fr (host pointer, borrowed).
 * out_affine: 64 B, caller-allocated, canonical affine ((0,0) if the sum is the identity). */
int ezkl_hip_msm_g1(ezkl_bases_t h, const void* scalars, size_t n, void* out_affine);
/* same with scalars already resident in HBM; `stream` is a hipStream_t or NULL (library stream).
 * The result is written to host memory and the call returns after the stream has drained. */
int ezkl_hip_msm_g1_dev(ezkl_bases_t h, size_t base_offset, const void* scalars_dev, size_t n,
                        void* out_affine, void* stream);
/* batch of `batch` scalar vectors against the same bases (one commit per advice column) */
int ezkl_hip_msm_g1_batch(ezkl_bases_t h, const void* const* scalars, size_t batch, size_t n, void* out_affine);
/* One MSM in two halves: start queues it behind what the library stream has been asked to do so far (the scalar column may still be in
 * the making there) and returns a token; finish waits and writes the 64-byte affine point.  In between the caller may issue any other
 * call of this header -- commit batches and upload phases included: the MSM runs on a call slot with its own stream and scratch.  At most
 * four may be in flight per context: a fifth start WAITS while other threads hold the slots, and returns EZKL_ERR_BUSY when the calling
 * thread itself holds all four (it could never finish one).  finish takes a token exactly once, from any thread; a token that is not in
 * flight (never started, already finished, being finished by another thread) is EZKL_ERR_INVALID.  (The prover commits the vanishing argument's random polynomial, which depends on nothing but the
 * randomness, under the upload of the witness: halo2 draws it at the same place of the RNG stream either way.) */
int ezkl_hip_msm_g1_start_dev(ezkl_bases_t h, size_t base_offset, const void* scalars_dev, size_t n, int* token);
int ezkl_hip_msm_g1_finish(int token, void* out_affine);
/* the same with every scalar vector resident in HBM (array of DEVICE pointers held in host memory); the MSMs are
 * pipelined over several streams so the short reduce/sort kernels of one overlap the accumulation of the next */
int ezkl_hip_msm_g1_batch_dev(ezkl_bases_t h, size_t base_offset, const void* const* scalars_dev, size_t batch, size_t n,
                              void* out_affine, void* stream);
/* the same for WITNESS-SHAPED columns -- the caller knows that most scalars are small (advice columns, lookup multiplicities: one or
 * two non-zero window digits each).  Such an MSM is a chain of ~17 latency-bound launches whatever its length, so the batch runs as
 * fused groups of four columns per chain.  Any scalars are accepted (the result is the same); uniform ones just run a little slower. */
int ezkl_hip_msm_g1_batch_small_dev(ezkl_bases_t h, size_t base_offset, const void* const* scalars_dev, size_t batch, size_t n,
                                    void* out_affine, void* stream);
/* the same pipeline fed one column at a time, for callers that PRODUCE columns one by one on the device (compressed
 * lookup columns, grand products): the MSM of column j runs while column j+1 is being computed.  (It does not help for
 * columns arriving through blocking copies from pageable host memory: those queue behind the kernels in flight --
 * measured 36 vs 22 ms for 14 columns at k = 20 -- so the prover uploads first and commits in one batch.)  push returns
 * once the column's kernels are queued (it may first retire the oldest in-flight MSM); finish drains, writes `pushed`
 * affine points (EZKL_ERR_INVALID if capacity is smaller) and closes the batch, also on error.  One batch may be open at a
 * time; the other MSM entry points return EZKL_ERR_INVALID while it is. */
/* One prover phase in one call (create_proof: "assign advice, blind, commit"): host_cols[j] (n x 32 B) is copied into the
 * caller's device column dev_cols[j], rows [tail_start, tail_start + tail_count) are overwritten with tail_rows[j] (the blinding
 * values; tail_rows may be NULL) and rows [commit_first, commit_first + commit_count) of the column are committed against
 * bases [0, commit_count) (the whole column: 0, n; a rank holding one slice of the SRS: its slice); out_affine gets `batch` points.
 * Every copy is queued at once on a dedicated copy stream and the MSM of column j only waits for ITS copy, so the PCIe
 * traffic of the later columns runs under the kernels of the earlier ones.  Page-locked host columns
 * (ezkl_hip_host_malloc) make the copies asynchronous; pageable ones work but are staged by the runtime. */
int ezkl_hip_upload_commit_batch(ezkl_bases_t h, const void* const* host_cols, void* const* dev_cols, size_t batch, size_t n,
                                 const void* const* tail_rows, size_t tail_start, size_t tail_count, size_t commit_first, size_t commit_count,
                                 void* out_affine);
/* The same phase in steps, for callers that want other work to start as each column lands (the prover queues the iNTT and the
 * coset NTT of an advice column on its own stream while later columns are still crossing PCIe and earlier ones are being
 * committed): begin queues every copy and returns at once (the host columns must stay valid until end); wait makes `stream` (a
 * caller stream, not NULL) wait for column j; commit runs the MSMs (each waits for its own column) and returns the `batch`
 * points; end drains the copy stream and closes the phase, also after an error.  One phase may be open at a time. */
typedef struct ezkl_upload_s* ezkl_upload_t;
int ezkl_hip_upload_begin(const void* const* host_cols, void* const* dev_cols, size_t batch, size_t n, const void* const* tail_rows,
                          size_t tail_start, size_t tail_count, ezkl_upload_t* out_upload);
/* begin with a FORMAT per host column: what crosses PCIe need not be 32 bytes per cell.  Every cell of an ezkl advice column is
 * integer_rep_to_felt of an IntegerRep (i128: the quantised tensor value; /root/reference/src/fieldutils.rs:6-17) before halo2 sees it as an
 * Fp, so a caller that still has the integers hands THOSE over -- 16 bytes per cell, or 8 when every value of the column fits an int64
 * -- and the column is expanded on the device (x >= 0 -> x, x < 0 -> r - |x|, then the Montgomery form) as its copy lands; the device
 * column, the blinding rows (tail_rows: always 32-byte Montgomery words) and the commitment are bit for bit those of the 32-byte form.
 * formats == NULL: every column EZKL_COLUMN_FP (= ezkl_hip_upload_begin). */
#define EZKL_COLUMN_FP 0        /* 32-byte Montgomery words, halo2curves' in-memory Fr */
#define EZKL_COLUMN_INT64 1     /* int64_t per cell */
#define EZKL_COLUMN_INT128 2    /* little-endian two's-complement 128-bit integer per cell (Rust's i128 = ezkl's IntegerRep) */
int ezkl_hip_upload_begin_fmt(const void* const* host_cols, const uint8_t* formats, void* const* dev_cols, size_t batch, size_t n,
                              const void* const* tail_rows, size_t tail_start, size_t tail_count, ezkl_upload_t* out_upload);
int ezkl_hip_upload_wait(ezkl_upload_t upload, size_t column, void* stream);
int ezkl_hip_upload_commit(ezkl_upload_t upload, ezkl_bases_t h, size_t commit_first, size_t commit_count, void* out_affine);
int ezkl_hip_upload_end(ezkl_upload_t upload);
/* A commit batch fed as its columns become final: begin; push (one column / several: fused into groups as the one-call batch does) any
 * number of times; finish returns the points in push order and closes the batch (also after an error).  A push returns once the MSMs
 * are QUEUED: they start behind everything queued on the library stream so far (an event, no host synchronisation), on the MSM slot
 * streams, so the caller can go on queuing unrelated library-stream work while they run -- the prover commits the permutation
 * products z while the lookup arguments' running sums are still being computed.  The pushed columns must stay untouched until finish.
 * One batch may be open per context; the other MSM entry points refuse (EZKL_ERR_INVALID) until it is finished. */
typedef struct ezkl_msm_batch_s* ezkl_msm_batch_t;
int ezkl_hip_msm_batch_begin(ezkl_bases_t h, size_t base_offset, size_t n, ezkl_msm_batch_t* out_batch);
int ezkl_hip_msm_batch_push_dev(ezkl_msm_batch_t batch, const void* scalars_dev);
int ezkl_hip_msm_batch_push_many_dev(ezkl_msm_batch_t batch, const void* const* scalars_dev, size_t count);
int ezkl_hip_msm_batch_finish(ezkl_msm_batch_t batch, void* out_affine, size_t capacity);
/* out = a + b on affine points (host, used to fold per-GPU partial sums after the all-gather) */
int ezkl_hip_g1_add_affine(const void* a, const void* b, void* out);

/* ---- NTT: replaces halo2curves::fft::best_fft / EvaluationDomain::{ifft, coeff_to_extended,
 *      extended_to_coeff} (type used in-tree at /root/reference/src/circuit/modules/polycommit.rs:52) ---- */
/* best_fft(data, omega, log_n): natural order in and out; if `inverse`, additionally scales by 1/n
 * (omega is then the inverse root, as EvaluationDomain::ifft passes it).  Host buffer, in place. */
int ezkl_hip_ntt(void* data, uint32_t log_n, const void* omega, int inverse);
/* device-resident batch: `batch` columns of 2^log_n elements, column b at data_dev + b*stride_elems*32 */
int ezkl_hip_ntt_dev(void* data_dev, uint32_t log_n, const void* omega, int inverse,
                     size_t batch, size_t stride_elems, void* stream);
/* coeff_to_extended (inverse = 0): in[b] has 2^log_n coefficients, out[b] gets 2^log_n_ext evaluations on
 * the zeta-coset.  extended_to_coeff (inverse = 1): in[b] has 2^log_n_ext evaluations, out[b] receives
 * 2^log_n_ext coefficients (caller truncates).  Host pointers.  in == out allowed when sizes match. */
int ezkl_hip_coset_ntt_batch(const void* const* in, void* const* out, size_t batch,
                             uint32_t log_n, uint32_t log_n_ext, int inverse);
int ezkl_hip_coset_ntt_dev(const void* in_dev, void* out_dev, size_t batch, size_t in_stride_elems,
                           size_t out_stride_elems, uint32_t log_n, uint32_t log_n_ext, int inverse, void* stream);

/* coeff_to_extended with the result in COSET-MAJOR order: the extended domain {zeta w_ext^i} is the union of E = 2^(log_n_ext - log_n)
 * cosets c_b H (c_b = zeta w_ext^b, H = <omega>), natural index i = E j + b; out[b 2^log_n + j] = p(c_b omega^j).  Computed as E
 * transforms of 2^log_n points (the zero-padding stages of the 2^log_n_ext-point transform never run).  A rotation by r rows of the
 * 2^log_n domain is a shift by r inside a coset, so ezkl_hip_eval_h_dev run with k = ext_k = log_n on the b-th cosets of its columns IS
 * the quotient sweep of coset b (and a coset is the unit a multi-GPU prover hands to a rank).  in != out.
 * ezkl_hip_cosets_transpose_dev converts one extended column between the two orders (to_natural = 1: coset-major -> natural, the order
 * of halo2's pk.key `*_cosets` sections and of ezkl_hip_coset_ntt_dev). */
int ezkl_hip_coeff_to_cosets_dev(const void* in_dev, void* out_dev, size_t batch, size_t in_stride_elems, size_t out_stride_elems,
                                 uint32_t log_n, uint32_t log_n_ext, void* stream);
int ezkl_hip_cosets_transpose_dev(const void* in_dev, void* out_dev, uint32_t log_n, uint32_t log_n_ext, int to_natural, void* stream);
/* the same for a RANGE of cosets: out[(b - first_coset) 2^log_n + j] = p(c_b omega^j) for b in [first_coset, first_coset + n_cosets),
 * n_cosets a power of two (out_stride_elems >= n_cosets 2^log_n).  A rank of a sharded prover sweeps only some cosets of the extended
 * domain and needs only those cosets of the key columns (fixed, sigma, l0 / l_last / l_active): 1 / world of the bytes and of the work. */
int ezkl_hip_coeff_to_cosets_range_dev(const void* in_dev, void* out_dev, size_t batch, size_t in_stride_elems, size_t out_stride_elems,
                                       uint32_t log_n, uint32_t log_n_ext, uint32_t first_coset, uint32_t n_cosets, void* stream);

/* ---- element-wise Fr vector ops (icicle vec-ops surface) on device-resident data ---- */
#define EZKL_VEC_ADD 0
#define EZKL_VEC_SUB 1
#define EZKL_VEC_MUL 2
int ezkl_hip_vec_op_dev(int op, const void* a_dev, const void* b_dev, void* out_dev, size_t n, void* stream);
int ezkl_hip_vec_scale_dev(const void* a_dev, const void* scalar_host, void* out_dev, size_t n, void* stream);
int ezkl_hip_vec_fill_dev(void* out_dev, const void* value_host, size_t n, void* stream);   /* out[i] = value */
/* One sigma column of the permutation argument (halo2 permutation::keygen::Assembly::build_pk: sigma_c[r] = delta^c' omega^r' where
 * (c', r') is the cycle successor of cell (c, r)): cells are numbered column position * 2^log_n + row; next_dev = the successors of the
 * 2^log_n cells of this column (u32, device), omega_col_dev[r] = omega^r, delta_pows_dev[c] = delta^c (n_columns entries).
 * out_dev[r] = delta_pows[t >> log_n] * omega_col[t mod 2^log_n], t = next_dev[r] (zero for a successor outside the n_columns). */
int ezkl_hip_permutation_sigma_dev(const void* next_dev, const void* omega_col_dev, const void* delta_pows_dev, uint32_t n_columns, uint32_t log_n,
                                   void* out_dev, void* stream);
/* a[i] *= t[i mod 2^(ext_k-k)], t = 1/((zeta*omega_ext^j)^n - 1): EvaluationDomain::divide_by_vanishing_poly */
int ezkl_hip_divide_by_vanishing_dev(void* a_dev, uint32_t k, uint32_t ext_k, void* stream);
/* running sum (EZKL_VEC_ADD) / running product (EZKL_VEC_MUL): out[i] = in[0] o ... o in[i] (inclusive) or
 * o in[i-1] with out[0] = identity (exclusive): the grand sum of mv-lookup::commit_grand_sum and the grand
 * product z(X) of permutation::commit.  in == out allowed. */
int ezkl_hip_prefix_scan_dev(int op, int exclusive, const void* in_dev, void* out_dev, size_t n, void* stream);
/* mv-lookup multiplicities m(X) (mv_lookup::prover::prepare): for every usable row r of every (theta-compressed) input
 * column, the FIRST usable table row holding the same value gets +1; m_out is an n_rows Fr column (rows >= usable
 * are 0).  *out_missing (may be NULL) = number of input values absent from the table (a failing witness). */
int ezkl_hip_lookup_multiplicity_dev(const void* const* inputs_dev, uint32_t n_inputs, const void* table_dev, uint32_t n_rows,
                                     uint32_t usable_rows, void* m_out_dev, uint32_t* out_missing, void* stream);
/* the same with the count of absent input values ADDED to a device u32 (missing_dev, zeroed by the caller): nothing returns to the
 * host, so the call is stream-ordered like the other helpers; a prover queues all its lookup arguments and reads the counter once */
int ezkl_hip_lookup_multiplicity_acc_dev(const void* const* inputs_dev, uint32_t n_inputs, const void* table_dev, uint32_t n_rows,
                                         uint32_t usable_rows, void* m_out_dev, void* missing_dev, void* stream);
/* every lookup argument of a proof in ONE call (mv_lookup::prover::prepare runs once per argument, on independent data): argument l has
 * the table tables_dev[l] and the output column m_outs_dev[l]; input column j (n_inputs of them in all, any number per argument)
 * belongs to argument input_lookup[j].  Three launches for the whole batch instead of three per argument -- the passes are random
 * 4- and 32-byte reads, and one argument alone leaves most of the machine waiting on them.  missing_dev as in _acc_dev. */
int ezkl_hip_lookup_multiplicity_batch_dev(const void* const* inputs_dev, const uint32_t* input_lookup, uint32_t n_inputs, const void* const* tables_dev,
                                           uint32_t n_lookups, uint32_t n_rows, uint32_t usable_rows, void* const* m_outs_dev, void* missing_dev, void* stream);
/* halo2 eval_polynomial(poly, x): sum_i coeffs[i] * x^i for a resident coefficient vector; x and the 32-byte result
 * are host memory (create_proof evaluates every queried (column, rotation) this way before SHPLONK) */
int ezkl_hip_eval_poly_dev(const void* coeffs_dev, size_t n, const void* x_host, void* out_host, void* stream);
/* m evaluations in one call: polynomial j (n coefficients, resident) at xs[j] (host, m x 32 B) -> out_host[j]; one upload,
 * one download, one synchronisation for the whole batch of create_proof's step 10 */
int ezkl_hip_eval_poly_batch_dev(const void* const* coeffs_dev, const void* xs_host, uint32_t m, size_t n, void* out_host, void* stream);
/* out[i] = (accumulate ? out[i] : 0) + sum_j coeffs[j] * inputs[j][i]: the linear combinations of SHPLONK
 * (ProverSHPLONK::create_proof) and the x^n-Horner over the quotient pieces, each input read once.  inputs: m DEVICE
 * pointers in host memory; coeffs: m x 32 B host */
/* q(X) = a(X) / (X - z), remainder dropped: out[n-1] = 0, out[i-1] = a[i] + z out[i] (halo2_proofs::arithmetic::kate_division,
 * the quotients of the KZG / SHPLONK openings).  n coefficients resident, z host (32 B Montgomery); out may alias a. */
int ezkl_hip_kate_division_dev(const void* a_dev, const void* z_host, void* out_dev, size_t n, void* stream);
int ezkl_hip_lincomb_dev(const void* const* inputs_dev, const void* coeffs_host, uint32_t m, void* out_dev, size_t n, int accumulate,
                         void* stream);
/* out[i] = uniform element of Fr, i < n, expanded from a 256-bit key with ChaCha20 (64-bit block counter, 64-bit stream
 * id): element (first + i) owns blocks 16(first+i) .. +15; its 32 candidates are the 8-word halves of those blocks with
 * the top two bits cleared; the first candidate < r is taken (rejection sampling => uniform on [0, r)).  Replaces the
 * host-side OsRng loop of halo2 (blinding rows; vanishing::Argument::commit's random polynomial, 2^k elements) by a
 * keystream expanded where the column lives; the caller provides the key (OS entropy, or a seed for det-prove,
 * /root/reference/src/pfsys/mod.rs:436-439). */
int ezkl_hip_chacha20_fr_dev(const void* key32, uint64_t stream_id, size_t first, void* out_dev, size_t n, void* stream);
/* Montgomery batch inversion (zeros stay zero), in place */
int ezkl_hip_batch_invert_dev(void* a_dev, size_t n, void* stream);

/* ---- quotient numerator: replaces plonk::evaluation::Evaluator::evaluate_h's row sweep
 *      (GraphEvaluator::evaluate; icicle "gate_eval" program on the GPU build) ---- */
/* instruction = 8 x u32: [op, target, s0.kind, s0.idx, s0.rot, s1.kind, s1.idx, s1.rot] */
enum { EZKL_OP_ADD = 0, EZKL_OP_SUB, EZKL_OP_MUL, EZKL_OP_SQUARE, EZKL_OP_DOUBLE, EZKL_OP_NEGATE,
       EZKL_OP_STORE, EZKL_OP_HORNER_STEP /* target = target*s1 + s0 */ };
enum { EZKL_SRC_CONST = 0, EZKL_SRC_INTERMEDIATE, EZKL_SRC_COLUMN, EZKL_SRC_CHALLENGE, EZKL_SRC_PREVIOUS };
typedef struct {
    const uint32_t* code;       uint32_t n_instr;  uint32_t n_intermediates;
    const void* constants;      uint32_t n_constants;     /* n x 32 B Fr, host */
    const int32_t* rotations;   uint32_t n_rotations;     /* in rows of the 2^k domain */
    const void* const* columns; uint32_t n_columns;       /* DEVICE pointers, each 2^ext_k x 32 B */
    const void* challenges;     uint32_t n_challenges;    /* n x 32 B Fr, host */
    uint32_t k, ext_k;
} ezkl_program_t;
/* out_dev[r] = program(row r) with ValueSource::PreviousValue = old out_dev[r]; 2^ext_k rows.  The host arrays of `prog` are borrowed
 * for the call only (packed into pinned staging the library owns); on a caller stream, or on the library stream in asynchronous mode
 * (ezkl_hip_set_async), the call returns once the launch is queued -- the columns and out_dev must stay valid until it has run. */
int ezkl_hip_eval_h_dev(const ezkl_program_t* prog, void* out_dev, void* stream);
/* the sweep is JIT-compiled (hiprtc) into straight-line gfx950 code, once per program; this host-only call
 * checks that a program lowers and compiles (column pointers are not dereferenced; no GPU needed) */
int ezkl_hip_eval_h_check(const ezkl_program_t* prog);
/* host-only: the instruction order the sweep will execute (the library re-orders a program for short live ranges: every term is
 * computed right before the Horner step that consumes it; dependencies per intermediate are kept, so the value is the program's);
 * out_code receives n_instr x 8 words in the layout of prog->code */
int ezkl_hip_eval_h_schedule(const ezkl_program_t* prog, uint32_t* out_code);
/* Compile the program's kernel now (or load it from the on-disk cache: $EZKL_HIP_CACHE_DIR, ~/.cache/ezkl_hip) without running it:
 * columns / constants / challenges are not read.  A key generator calls it for the circuit's quotient program, so that the first
 * prove of a new circuit -- in this or a later process -- does not wait for hiprtc (seconds for an ezkl-sized program). */
int ezkl_hip_eval_h_prepare(const ezkl_program_t* prog);
/* how this process obtained its sweep kernels so far: compiled by hiprtc, loaded from the on-disk cache, found in memory */
int ezkl_hip_eval_h_jit_stats(uint64_t* compiled, uint64_t* from_disk, uint64_t* memory_hits);

/* ---- multi-GPU: the collectives of the sharded prove path, RCCL over xGMI on the library's device pointers (csrc/comm.hip) ----
 * One process per GPU (ezkl_hip_init picks the device).  The reference has no multi-GPU path (icicle is single-GPU), so
 * these replace nothing: a launcher creates the id on rank 0 (ezkl_hip_comm_unique_id), hands the 128 bytes to every rank by whatever
 * means it has (env, file, MPI, torch.distributed), and every rank calls ezkl_hip_comm_init.  librccl is dlopen'ed on first use.
 * EZKL_ERR_UNSUPPORTED: RCCL cannot be loaded.  MSM partials: gather-then-add (RCCL has no elliptic-curve reduce op). */
int ezkl_hip_comm_unique_id(void* out128);
int ezkl_hip_comm_init(const void* id128, int world, int rank);
int ezkl_hip_comm_info(int* world, int* rank);            /* world = 0: no communicator */
int ezkl_hip_comm_destroy(void);
/* in-place all_gather of a device buffer made of `world` equal slices (rank r wrote slice r): the row-sharded quotient sweep's h */
int ezkl_hip_comm_allgather_dev(void* buf_dev, size_t total_bytes);
/* one commit batch: `count` 64-byte affine Montgomery partial sums (host) -> their sums over all ranks, in place, on every rank */
int ezkl_hip_comm_fold_points(void* points_host, uint32_t count);
/* broadcast of a small host buffer (<= 4096 B) from rank `root`: the OS-entropy ChaCha key of a sharded proof */
int ezkl_hip_comm_broadcast_host(void* buf_host, size_t bytes, int root);
/* all-to-all on device pointers (per-peer byte offsets / lengths): columns transformed by their owner -> the row shards of the sweep */
int ezkl_hip_comm_alltoall_dev(const void* send_dev, const size_t* send_off, const size_t* send_len, void* recv_dev, const size_t* recv_off,
                               const size_t* recv_len);

/* all_gather of small HOST buffers: buf_host holds world x bytes_per_rank bytes, rank r has filled slice r; on return every slice is
 * filled on every rank (the evaluations a rank computed for the polynomials it owns; the chaining values of the permutation chunks) */
int ezkl_hip_comm_allgather_host(void* buf_host, size_t bytes_per_rank);
/* all-to-all with ANY NUMBER of contiguous device segments per peer: sends[i] goes to sends[i].peer, recvs[j] arrives from recvs[j].peer.
 * Per (sender, receiver) pair the segments form ONE byte stream in list order -- the k-th byte this rank sends to peer p is the k-th byte
 * p receives from this rank -- so only the per-pair totals must agree.  A column-sharded prover moves the cosets (or halo-extended row
 * windows) of the columns it transformed to the ranks that sweep those rows with ONE such call.  Wire format: a device gather kernel packs
 * each peer's stream into a slab (rounds of at most EZKL_COMM_SLAB_MB, default 128 MiB, per peer and direction; the same on every rank),
 * ONE ncclSend + ONE ncclRecv per peer and round inside one group (all xGMI links at once), a scatter kernel unpacks.  Bytes to the own
 * rank are copied by the same kernel.  Segments may have any address and length: 16-byte aligned pieces (the prover's field columns) move
 * as vectors, others byte by byte, in the SAME wire format -- which format is used never depends on what one rank sees.  EZKL_COMM_UNPACKED=1
 * (set it on every rank or on none): one ncclSend / ncclRecv per segment (round 3; sizes must then agree pairwise).
 * ezkl_hip_comm_stats: out[0] = exchanges, out[1] / out[2] = bytes sent to / received from other ranks, out[3] = microseconds of host wall
 * time inside the exchanges, out[4] / out[5] = ncclSend / ncclRecv operations issued, out[6] = rounds; reset != 0 clears them. */
/* EZKL_OK when librccl can be loaded (from the directory of the HIP runtime this library is bound to) and exports every entry point the
 * communicator binds; EZKL_ERR_UNSUPPORTED otherwise.  Touches no device and calls nothing in RCCL. */
int ezkl_hip_comm_available(void);
typedef struct { int peer; void* ptr; size_t bytes; } ezkl_comm_seg_t;
int ezkl_hip_comm_alltoallv_dev(const ezkl_comm_seg_t* sends, size_t n_sends, const ezkl_comm_seg_t* recvs, size_t n_recvs);
int ezkl_hip_comm_stats(uint64_t out[8], int reset);
/* The communicator tries itself out -- an all_gather of rank ids, ONE packed all-to-all of odd-sized segments, one fold of partial points,
 * each checked -- and fails with a message on stderr instead of hanging or proving wrongly later.  A collective: every rank calls it.
 * ezkl_hip_comm_init runs it by default (EZKL_COMM_SELFTEST=0 on every rank skips).  Every collective of this section is waited for under a
 * watchdog: past EZKL_COMM_TIMEOUT_S seconds (default 120, 0 = none) the call returns EZKL_ERR_TIMEOUT and the communicator is aborted. */
int ezkl_hip_comm_selftest(void);

/* ---- measurement hooks (used by bench.py; HIP events on the stream the kernels run on) ---- */
/* after an msm/ntt call: average device milliseconds of the dominant kernel of the last call */
int ezkl_hip_last_kernel_ms(const char* which, float* out_ms);
/* sum and count of the device milliseconds of EVERY region `which` recorded since the last reset (each call records into an event pair
 * of its own, a ring of 64 per region): a caller can queue K steps back to back and read their kernel times afterwards, without a host
 * synchronisation inside its timed loop.  Waits for the pairs still in flight.  A region that never ran: sum 0, count 0. */
int ezkl_hip_kernel_ms_stats(const char* which, double* sum_ms, uint64_t* count, int reset);
/* Which pairs a synchronous MSM records is the caller's choice through the environment variable EZKL_HIP_TIMING, read at every call:
 * "all" (default: the chain, "msm", and its dominant kernel, "msm_accumulate"), "kernel" (the dominant kernel only), "none" (which also
 * drops the pair around the passes of ezkl_hip_ntt_dev).  An event record
 * is a barrier packet with a timestamp on the stream: the four of one call cost about 15 us of a 1.31 ms 2^20-point MSM. */
/* microbenchmarks: which = "modmul" (Montgomery products/s), "mad64" (v_mad_u64_u32/s),
 * "copy" (HBM float4 copy bytes/s); result in *out (per second) */
int ezkl_hip_ubench(const char* which, double* out);

#ifdef __cplusplus
}
#endif
#endif
