// capi.hip -- the extern "C" surface of libezkl_hip.so (include/ezkl_hip.h) and the library context.
#include "common.hpp"
#include <string.h>
#include <atomic>
#include <algorithm>
#include <iterator>

namespace ezkl {

static std::mutex g_init_mu;
// The context table.  One context = one device + everything the library keeps for it (streams, arenas, MSM tables, NTT plans, JIT
// modules, the column pool) behind its own mutex.  A process launched per GPU (torchrun: LOCAL_RANK) has ONE context; a single-process
// multi-GPU prover (ezkl_hip_init(-1) without LOCAL_RANK: /root/reference/src/execute.rs:1575-1627 is one process) gets one per visible
// device, and every host thread works on the context it bound itself to (ezkl_hip_set_context; threads start on context 0).  Several
// contexts may name the same device (ezkl_hip_contexts_configure): that is how the multi-device prover is tested on a one-GPU box.
static constexpr int MAX_CTX = 64;
// (entries are published with release / read with acquire: N group threads call ctx() concurrently while another creates its context)
static std::atomic<Ctx*> g_ctxs[MAX_CTX];
static int g_ctx_device[MAX_CTX];
static std::atomic<int> g_n_ctx{0};   // 0: table not configured yet
static thread_local int t_ctx = 0;
static std::atomic<int> g_last_hip_err{0};

int set_hip_error(hipError_t e, const char* what, const char* file, int line) {
    g_last_hip_err.store((int)e);
    fprintf(stderr, "[ezkl_hip] HIP error %d (%s) at %s:%d in %s\n", (int)e, hipGetErrorString(e), file, line, what);
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? EZKL_ERR_NOMEM : EZKL_ERR_HIP;
}

static int device_count_checked(int* n) {
    // The prover keeps ~10 streams busy (6 MSM slots, copy, NTT aux, table, caller); the HIP runtime multiplexes streams onto
    // GPU_MAX_HW_QUEUES hardware queues (default 4), and the latency-bound MSM tails of different slots then wait for each other:
    // 8 queues took the k = 17 MLP proof from 44.5 to 38.3 ms (more slots did not help).  Only effective if this is the first HIP call
    // of the process; a user setting wins.
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    *n = 0;
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess || *n <= 0) {
        g_last_hip_err.store((int)e);
        (void)hipGetLastError();
        return EZKL_ERR_NO_DEVICE;
    }
    return EZKL_OK;
}
// explicit table: context i on device devices[i]; only before any context exists
int ctx_configure(int n_ctx, const int* devices) {
    std::lock_guard<std::mutex> lk(g_init_mu);
    if (n_ctx < 1 || n_ctx > MAX_CTX || !devices) return EZKL_ERR_INVALID;
    for (int i = 0; i < MAX_CTX; i++)
        if (g_ctxs[i].load(std::memory_order_acquire)) return EZKL_ERR_INVALID;
    int n = 0;
    int rc = device_count_checked(&n);
    if (rc) return rc;
    for (int i = 0; i < n_ctx; i++)
        if (devices[i] < 0 || devices[i] >= n) return EZKL_ERR_INVALID;
    for (int i = 0; i < n_ctx; i++) g_ctx_device[i] = devices[i];
    g_n_ctx.store(n_ctx, std::memory_order_release);
    return EZKL_OK;
}
// ezkl_hip_init(device): device >= 0 -> ONE context on that device.  device < 0 -> LOCAL_RANK set (one process per GPU): one context on
// device LOCAL_RANK; otherwise ALL visible devices, context i on device i.  Idempotent; a second call may not contradict the first.
int ctx_init(int device) {
    std::lock_guard<std::mutex> lk(g_init_mu);
    if (const int have = g_n_ctx.load(std::memory_order_acquire)) {     // already configured: fine unless the caller names a device no context is on
        if (device < 0) return EZKL_OK;
        for (int i = 0; i < have; i++)
            if (g_ctx_device[i] == device) return EZKL_OK;
        return EZKL_ERR_INVALID;
    }
    int n = 0;
    int rc = device_count_checked(&n);
    if (rc) return rc;
    if (device >= n) return EZKL_ERR_INVALID;
    if (device >= 0) {
        g_ctx_device[0] = device;
        g_n_ctx.store(1, std::memory_order_release);
    } else if (const char* lr = getenv("LOCAL_RANK")) {
        g_ctx_device[0] = atoi(lr) % n;
        g_n_ctx.store(1, std::memory_order_release);
    } else {
        const int m = n < MAX_CTX ? n : MAX_CTX;
        for (int i = 0; i < m; i++) g_ctx_device[i] = i;
        g_n_ctx.store(m, std::memory_order_release);
    }
    return EZKL_OK;
}
static int ctx_create(int idx) {
    const int device = g_ctx_device[idx];
    EZ_HIP(hipSetDevice(device));
    Ctx* c = new Ctx();
    c->device = device;
    c->index = idx;
    EZ_HIP(stream_create_prio(&c->stream, "EZKL_HIP_PRIO_LIB", 0));
    hipDeviceProp_t prop;
    EZ_HIP(hipGetDeviceProperties(&prop, device));
    c->num_cus = prop.multiProcessorCount;
    // one process driving several GPUs (the prover group): let this device read the other contexts' devices directly, so that
    // ezkl_hip_memcpy_peer is one xGMI transfer instead of a copy staged through host memory.  Best effort: without peer access the
    // copies still work.
    for (int j = 0; j < g_n_ctx.load(std::memory_order_acquire); j++) {
        const int other = g_ctx_device[j];
        if (other == device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device, other) == hipSuccess && can) {
            const hipError_t e = hipDeviceEnablePeerAccess(other, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled && getenv("EZKL_HIP_DEBUG"))
                fprintf(stderr, "[ezkl_hip] no peer access %d -> %d: %s\n", device, other, hipGetErrorString(e));
        }
        (void)hipGetLastError();
    }
    g_ctxs[idx].store(c, std::memory_order_release);
    return EZKL_OK;
}

Ctx* ctx() {
    if (!g_n_ctx.load(std::memory_order_acquire) && ctx_init(-1) != EZKL_OK) return nullptr;
    const int idx = t_ctx;
    if (idx < 0 || idx >= g_n_ctx.load(std::memory_order_acquire)) return nullptr;
    Ctx* c = g_ctxs[idx].load(std::memory_order_acquire);
    if (!c) {
        std::lock_guard<std::mutex> lk(g_init_mu);
        c = g_ctxs[idx].load(std::memory_order_acquire);
        if (!c) {
            if (ctx_create(idx) != EZKL_OK) return nullptr;
            c = g_ctxs[idx].load(std::memory_order_acquire);
        }
    }
    return c;
}
int ctx_count() { return g_n_ctx.load(std::memory_order_acquire); }
int ctx_bind(int idx) {
    if (!ctx_count() && ctx_init(-1) != EZKL_OK) return EZKL_ERR_NO_DEVICE;
    if (idx < 0 || idx >= ctx_count()) return EZKL_ERR_INVALID;
    t_ctx = idx;
    return EZKL_OK;
}
int ctx_device_of(int idx) { return idx >= 0 && idx < ctx_count() ? g_ctx_device[idx] : -1; }

int arena_reserve(Ctx::Arena& a, size_t bytes, hipStream_t st, void** out) {
    if (!a.last_event) EZ_HIP(hipEventCreateWithFlags(&a.last_event, hipEventDisableTiming));
    if (bytes > a.bytes) {
        if (a.ptr) {
            EZ_HIP(hipDeviceSynchronize());
            EZ_HIP(hipFree(a.ptr));
            a.ptr = nullptr;
            a.bytes = 0;
            a.in_use = false;
        }
        size_t want = bytes + (bytes >> 3) + 256;
        EZ_HIP(hipMalloc(&a.ptr, want));
        a.bytes = want;
    }
    if (a.in_use && a.last_stream != st) EZ_HIP(hipStreamWaitEvent(st, a.last_event, 0));
    *out = a.ptr;
    return EZKL_OK;
}
int arena_done(Ctx::Arena& a, hipStream_t st) {
    EZ_HIP(hipEventRecord(a.last_event, st));
    a.last_stream = st;
    a.in_use = true;
    return EZKL_OK;
}

// adds the elapsed time of the oldest pending pair(s) to the ring's statistics: `all` = every pending pair (waits for them), otherwise
// only as many as it takes to free the slot the next acquisition needs
int ev_harvest(Ctx::EvRing& r, bool all) {
    while (r.tail < r.head && (all || r.head - r.tail >= Ctx::EvRing::N)) {
        const unsigned i = (unsigned)(r.tail % Ctx::EvRing::N);
        float ms = 0;
        // a pair whose second event was never recorded (the call that took it returned on an error in between) is SKIPPED, not fatal: one
        // transient failure must not leave the ring -- and every later timed call -- broken
        if (hipEventSynchronize(r.e1[i]) == hipSuccess && hipEventElapsedTime(&ms, r.e0[i], r.e1[i]) == hipSuccess) {
            r.sum_ms += ms;
            r.count++;
        } else {
            (void)hipGetLastError();
            r.skipped++;
        }
        r.tail++;
    }
    return EZKL_OK;
}
int ev_pair(Ctx* c, const char* key, hipEvent_t* e0, hipEvent_t* e1) {
    Ctx::EvRing& r = c->events[key];
    int rc = ev_harvest(r, false);
    if (rc) return rc;
    const unsigned i = (unsigned)(r.head % Ctx::EvRing::N);
    if (!r.made[i]) {
        EZ_HIP(hipEventCreate(&r.e0[i]));
        EZ_HIP(hipEventCreate(&r.e1[i]));
        r.made[i] = true;
    }
    *e0 = r.e0[i];
    *e1 = r.e1[i];
    r.head++;
    return EZKL_OK;
}

struct StagingBlock {
    uint8_t* host = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;
    bool pending = false;
};
struct StagingState {
    static constexpr unsigned N = 32;
    StagingBlock ring[N];
    unsigned next = 0;
};
uint8_t* staging_acquire(Ctx* c, size_t bytes, void** token) {
    if (!c->staging_state) c->staging_state = new StagingState();
    StagingState& ss = *static_cast<StagingState*>(c->staging_state);
    StagingBlock& s = ss.ring[ss.next++ % StagingState::N];
    if (s.pending) {
        if (hipEventSynchronize(s.done) != hipSuccess) return nullptr;
        s.pending = false;
    }
    if (s.bytes < bytes) {
        if (s.host) (void)hipHostFree(s.host);
        s.host = nullptr;
        s.bytes = 0;
        const size_t want = bytes < (64u << 10) ? (64u << 10) : bytes;
        if (hipHostMalloc((void**)&s.host, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); s.host = nullptr; return nullptr; }
        s.bytes = want;
    }
    if (!s.done && hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) return nullptr;
    *token = &s;
    return s.host;
}
int staging_release(void* token, hipStream_t st) {
    StagingBlock* s = static_cast<StagingBlock*>(token);
    EZ_HIP(hipEventRecord(s->done, st));
    s->pending = true;
    return EZKL_OK;
}

// calls made on the library stream are synchronous (unless ezkl_hip_set_async(1): then they are ordered on that one stream and the
// caller synchronises where it needs to -- every entry point that returns host data or borrows host memory still does by itself);
// calls on a caller stream are stream-ordered
// The mode belongs to the CALLING THREAD (a process-wide flag saved / restored by overlapping create_proof calls on
// different threads could stay on, and unrelated threads lost the synchronous semantics while any proof ran).
static thread_local bool t_async_library_stream = false;
static int finish(Ctx* c, hipStream_t st, void* user_stream) {
    (void)c;
    if (!user_stream && !t_async_library_stream) EZ_HIP(hipStreamSynchronize(st));
    return EZKL_OK;
}

}  // namespace ezkl

using namespace ezkl;

extern "C" {

int ezkl_hip_init(int device) { return ctx_init(device); }
int ezkl_hip_contexts_configure(int n_contexts, const int* devices) { return ctx_configure(n_contexts, devices); }
int ezkl_hip_context_count(void) {
    if (!ctx_count() && ctx_init(-1) != EZKL_OK) return 0;
    return ctx_count();
}
int ezkl_hip_set_context(int index) { return ctx_bind(index); }
int ezkl_hip_context_device(int index) { return ctx_device_of(index); }
// device-to-device copy INTO the calling thread's context from another context (their devices may differ: the peer copy of a
// single-process multi-GPU prover's exchange).  Stream-ordered on the calling context's library stream: later library-stream work of this
// context sees the data; the SOURCE may be reused only after this context has synchronised (ezkl_hip_synchronize).  A plain hipMemcpy
// device-to-device is asynchronous with respect to the host and unordered with the library's non-blocking streams: an earlier version
// raced at k = 20 (32 MiB slabs) and passed at k = 17.
int ezkl_hip_memcpy_peer(void* dst_dev, int dst_context, const void* src_dev, int src_context, size_t bytes) {
    if ((!dst_dev || !src_dev) && bytes) return EZKL_ERR_INVALID;
    const int dd = ctx_device_of(dst_context), sd = ctx_device_of(src_context);
    if (dd < 0 || sd < 0) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (c->index != dst_context) return EZKL_ERR_INVALID;          // the destination is the caller's own context
    if (!bytes) return EZKL_OK;
    if (dd == sd) EZ_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, c->stream));
    else EZ_HIP(hipMemcpyPeerAsync(dst_dev, dd, src_dev, sd, bytes, c->stream));
    return EZKL_OK;
}
// free / total bytes of the calling context's device (hipMemGetInfo): the HBM high-water mark of a proof = total - free after it,
// the column pool included
int ezkl_hip_mem_info(size_t* free_bytes, size_t* total_bytes) {
    if (!free_bytes || !total_bytes) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    EZ_HIP(hipMemGetInfo(free_bytes, total_bytes));
    return EZKL_OK;
}
int ezkl_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int ezkl_hip_synchronize(void) {
    EZ_CTX(c);
    EZ_HIP(hipDeviceSynchronize());
    return EZKL_OK;
}

int ezkl_hip_set_async(int on, int* previous) {
    EZ_CTX(c);
    if (previous) *previous = t_async_library_stream ? 1 : 0;
    if (!on && t_async_library_stream) EZ_HIP(hipStreamSynchronize(c->stream));     // leaving the mode: everything queued has run
    t_async_library_stream = on != 0;
    return EZKL_OK;
}
int ezkl_hip_stream_wait_library(void* stream) {
    if (!stream) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (!c->order_event) EZ_HIP(hipEventCreateWithFlags(&c->order_event, hipEventDisableTiming));
    EZ_HIP(hipEventRecord(c->order_event, c->stream));
    EZ_HIP(hipStreamWaitEvent((hipStream_t)stream, c->order_event, 0));
    return EZKL_OK;
}
int ezkl_hip_warmup(void) {
    EZ_CTX(c);
    // touch the stream + allocator once, as icicle's warmup(stream) does (src/execute.rs:89)
    void* p = nullptr;
    EZ_HIP(hipMalloc(&p, 1 << 20));
    EZ_HIP(hipMemsetAsync(p, 0, 1 << 20, c->stream));
    EZ_HIP(hipStreamSynchronize(c->stream));
    EZ_HIP(hipFree(p));
    return EZKL_OK;
}

const char* ezkl_hip_strerror(int code) {
    switch (code) {
    case EZKL_OK: return "ok";
    case EZKL_ERR_NO_DEVICE: return "no HIP device visible (libezkl_hip has no CPU fallback)";
    case EZKL_ERR_HIP: return "HIP runtime error (see ezkl_hip_last_hip_error)";
    case EZKL_ERR_INVALID: return "invalid argument";
    case EZKL_ERR_NOMEM: return "out of device memory";
    case EZKL_ERR_UNSUPPORTED: return "unsupported";
    case EZKL_ERR_BUSY: return "every slot is held by the calling thread";
    default: return "unknown error";
    }
}
int ezkl_hip_last_hip_error(void) { return g_last_hip_err.load(); }
const char* ezkl_hip_version(void) { return "ezkl_hip 0.1 (gfx950)"; }
int ezkl_hip_enabled(uint32_t k) {
    if (!getenv("ENABLE_HIP_GPU")) return 0;
    unsigned long small_k = 8;
    if (const char* e = getenv("HIP_SMALL_K")) {
        char* end = nullptr;
        const unsigned long v = strtoul(e, &end, 10);
        if (end != e && *end == 0) small_k = v;          // a malformed value keeps the default
    }
    return k > small_k ? 1 : 0;
}

// Column buffers are recycled: hipMalloc / hipFree cost ~0.2 ms each (hipFree synchronises the device) and a prover
// allocates and drops hundreds of same-sized columns per proof.  Freed blocks are parked per exact size (up to
// POOL_CAP bytes) and handed out again.  Ordering contract (the one of every caching allocator): calls on the library
// stream are complete when they return, so a block the caller only ever used through stream = NULL calls is idle when it
// is freed; a caller that used it on ITS OWN stream synchronises that stream before freeing it.  (hipFree's implicit
// device-wide sync on every reuse was measured at 0.3 ms per allocation with the library's ten streams alive, and it turned
// every allocation into a barrier for work in flight on other streams; EZKL_HIP_POOL_SYNC=1 brings it back for debugging.)
extern "C++" {
namespace {
struct PoolState {
    std::map<size_t, std::vector<void*>> pool;        // guarded by the context's mutex
    std::map<void*, size_t> sizes;
    size_t pool_bytes = 0;                            // parked
    size_t live_bytes = 0, live_peak = 0;             // handed out and not yet freed; its high-water mark
    size_t floor_bytes = 0;                           // the share of the device this context may always keep parked
    std::map<size_t, uint64_t> last_use;              // per size class: tick of the last allocation (eviction is least-recently-used class first)
    uint64_t tick = 0;
    size_t bound() const { return std::max(floor_bytes, live_peak + live_peak / 2); }
};
PoolState& pool_state(Ctx* c) {
    if (!c->pool_state) c->pool_state = new PoolState();
    return *static_cast<PoolState*>(c->pool_state);
}
// How much freed column memory a context keeps parked.  The context's FOOTPRINT (live + parked) is held to
//     max(floor, 1.5 x high-water mark of its live bytes)
// (1.5: the phases of a proof use different size classes -- 2^k-row columns first, extended ones later -- and a bound of exactly the
// peak would make each phase evict the other's blocks, every proof)
// where floor = 25 % of the device divided by the number of contexts that share the device (EZKL_HIP_POOL_CAP_GB overrides the floor).
// A prover that runs the same proof again finds every block of the last one parked (the footprint of a proof is its own peak: the
// k = 22 / 30-column proof cycles through > 100 GB of columns, and freeing / reallocating its 512 MiB extended columns between proofs
// cost 0.65 s per proof, profiles/r03p_k22_gantt_poolcap48.txt), while nothing is retained beyond what some call actually needed at
// once: after a k = 20 proof the pool holds that proof's columns, not 60 % of HBM (round 3's fixed cap).
size_t pool_floor(Ctx* c) {
    PoolState& ps = pool_state(c);
    if (!ps.floor_bytes) {
        if (const char* e = getenv("EZKL_HIP_POOL_CAP_GB")) ps.floor_bytes = (size_t)strtoull(e, nullptr, 10) << 30;
        if (!ps.floor_bytes) {
            size_t fr = 0, tot = 0;
            int sharing = 0;
            for (int i = 0; i < ctx_count(); i++) sharing += ctx_device_of(i) == c->device ? 1 : 0;
            if (sharing < 1) sharing = 1;
            ps.floor_bytes = (hipMemGetInfo(&fr, &tot) == hipSuccess && tot ? tot / 4 : (size_t)48 << 30) / (size_t)sharing;
        }
    }
    return ps.floor_bytes;
}
void pool_drop_parked(PoolState& ps) {
    for (auto& kv : ps.pool)
        for (void* p : kv.second) { (void)hipFree(p); ps.sizes.erase(p); }
    ps.pool.clear();
    ps.pool_bytes = 0;
}
// out of memory: give back what EVERY context on this device has parked (one context could fail while another sat on idle
// GBs).  The caller holds its own context's lock; the others are taken one at a time with try_lock -- a context that is busy right now
// keeps its blocks (no lock order to get wrong, no deadlock), an idle one gives them up.
void pool_trim_device(Ctx* self) {
    pool_drop_parked(pool_state(self));
    for (int i = 0; i < ctx_count(); i++) {
        Ctx* o = g_ctxs[i].load(std::memory_order_acquire);
        if (!o || o == self || o->device != self->device || !o->pool_state) continue;
        if (!o->mu.try_lock()) continue;
        pool_drop_parked(*static_cast<PoolState*>(o->pool_state));
        o->mu.unlock();
    }
}
}
}
int ezkl_hip_malloc(void** dptr, size_t bytes) {
    if (!dptr) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (!bytes) bytes = 1;
    PoolState& ps = pool_state(c);
    auto it = ps.pool.find(bytes);
    if (it != ps.pool.end() && !it->second.empty()) {
        *dptr = it->second.back();
        it->second.pop_back();
        ps.pool_bytes -= bytes;
        ps.live_bytes += bytes;
        ps.last_use[bytes] = ++ps.tick;
        if (ps.live_bytes > ps.live_peak) ps.live_peak = ps.live_bytes;
        static const bool pool_sync = getenv("EZKL_HIP_POOL_SYNC") != nullptr;
        if (pool_sync) EZ_HIP(hipDeviceSynchronize());
        return EZKL_OK;
    }
    // a new block would take the footprint past its bound: parked blocks of the least recently used size classes go first, so that a
    // process that moves from one problem size to another does not keep both sets
    ps.last_use[bytes] = ++ps.tick;
    (void)pool_floor(c);
    while (ps.pool_bytes && ps.live_bytes + ps.pool_bytes + bytes > ps.bound()) {
        auto victim = ps.pool.end();
        uint64_t oldest = UINT64_MAX;
        for (auto k = ps.pool.begin(); k != ps.pool.end();) {
            if (k->second.empty()) { k = ps.pool.erase(k); continue; }
            const uint64_t t = ps.last_use[k->first];
            if (t < oldest) { oldest = t; victim = k; }
            ++k;
        }
        if (victim == ps.pool.end()) break;
        void* p = victim->second.back();
        victim->second.pop_back();
        ps.pool_bytes -= victim->first;
        ps.sizes.erase(p);
        (void)hipFree(p);
    }
    hipError_t e = hipMalloc(dptr, bytes);
    if (e == hipErrorOutOfMemory) {                      // give the parked blocks of this device back and retry once
        (void)hipGetLastError();
        pool_trim_device(c);
        e = hipMalloc(dptr, bytes);
    }
    if (e != hipSuccess) return set_hip_error(e, "hipMalloc", __FILE__, __LINE__);
    ps.sizes[*dptr] = bytes;
    ps.live_bytes += bytes;
    if (ps.live_bytes > ps.live_peak) ps.live_peak = ps.live_bytes;
    return EZKL_OK;
}
int ezkl_hip_free(void* dptr) {
    if (!dptr) return EZKL_OK;
    EZ_CTX(c);
    PoolState& ps = pool_state(c);
    auto it = ps.sizes.find(dptr);
    if (it == ps.sizes.end()) { EZ_HIP(hipFree(dptr)); return EZKL_OK; }
    const size_t bytes = it->second;
    ps.live_bytes -= std::min(ps.live_bytes, bytes);
    (void)pool_floor(c);
    if (ps.live_bytes + ps.pool_bytes + bytes <= ps.bound()) {
        ps.pool[bytes].push_back(dptr);
        ps.pool_bytes += bytes;
        return EZKL_OK;
    }
    ps.sizes.erase(it);
    EZ_HIP(hipFree(dptr));
    return EZKL_OK;
}
// the column pool of the calling context: out[0] = bytes handed out now, out[1] = their high-water mark since the context was made
// (or since the last ezkl_hip_pool_trim), out[2] = bytes parked, out[3] = the bound on live + parked; with hipMemGetInfo (ezkl_hip_mem_info) this is the
// HBM high-water report of a proof or a test run
int ezkl_hip_pool_stats(size_t out[4]) {
    if (!out) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    PoolState& ps = pool_state(c);
    (void)pool_floor(c);
    out[0] = ps.live_bytes; out[1] = ps.live_peak; out[2] = ps.pool_bytes; out[3] = ps.bound();
    return EZKL_OK;
}
// hipFree every parked block of the calling context and forget the high-water mark (a long-lived process that has finished with its
// large circuit; between the size classes of a test run)
int ezkl_hip_pool_trim(void) {
    EZ_CTX(c);
    PoolState& ps = pool_state(c);
    EZ_HIP(hipDeviceSynchronize());
    pool_drop_parked(ps);
    ps.live_peak = ps.live_bytes;
    return EZKL_OK;
}
int ezkl_hip_host_malloc(void** p, size_t bytes) {
    if (!p) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    hipError_t e = hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e == hipErrorOutOfMemory) return EZKL_ERR_NOMEM;
    if (e != hipSuccess) return set_hip_error(e, "hipHostMalloc", __FILE__, __LINE__);
    return EZKL_OK;
}
int ezkl_hip_host_free(void* p) {
    if (!p) return EZKL_OK;
    EZ_CTX(c);
    EZ_HIP(hipHostFree(p));
    return EZKL_OK;
}
// Blocking copies go through the library stream, not the legacy null stream: hipMemcpy's implicit null-stream
// synchronisation was measured at up to 2x the copy time of a 32 MiB column with the library's ten streams alive
// (448 MB: 8.4 ms after a device sync, 15-21 ms without one), and at ~0.3 ms for a 32-byte row.
static hipError_t copy_sync(Ctx* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    return e;
}
int ezkl_hip_memcpy_h2d(void* dst, const void* src, size_t bytes) {
    EZ_CTX(c);
    // asynchronous mode, a few rows (blinding factors, boundary values): through pinned staging, ordered on the library stream
    if (t_async_library_stream && bytes && bytes <= (64u << 10)) {
        void* stg = nullptr;
        if (uint8_t* H = staging_acquire(c, bytes, &stg)) {
            memcpy(H, src, bytes);
            EZ_HIP(hipMemcpyAsync(dst, H, bytes, hipMemcpyHostToDevice, c->stream));
            return staging_release(stg, c->stream);
        }
    }
    EZ_HIP(copy_sync(c, dst, src, bytes, hipMemcpyHostToDevice));
    return EZKL_OK;
}
int ezkl_hip_memcpy_d2h(void* dst, const void* src, size_t bytes) {
    EZ_CTX(c);
    EZ_HIP(copy_sync(c, dst, src, bytes, hipMemcpyDeviceToHost));
    return EZKL_OK;
}

/* ------------------------------------------------------------------ MSM -------------------- */
int ezkl_hip_bases_upload(const void* pts, size_t n, ezkl_bases_t* out) {
    if (!pts || !out || n == 0) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    Bases* b = new Bases();
    b->n = n;
    hipError_t e = hipMalloc(&b->pts, n * 64);
    if (e == hipSuccess) e = copy_sync(c, b->pts, pts, n * 64, hipMemcpyHostToDevice);
    if (e != hipSuccess) {                     // nothing leaks on failure
        if (b->pts) (void)hipFree(b->pts);
        delete b;
        return set_hip_error(e, "ezkl_hip_bases_upload", __FILE__, __LINE__);
    }
    *out = reinterpret_cast<ezkl_bases_t>(b);
    return EZKL_OK;
}
int ezkl_hip_bases_prepare(ezkl_bases_t h) {
    if (!h) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return msm_table_prepare(c, reinterpret_cast<Bases*>(h));
}
int ezkl_hip_bases_free(ezkl_bases_t h) {
    if (!h) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    Bases* b = reinterpret_cast<Bases*>(h);
    msm_table_drop(b);
    EZ_HIP(hipFree(b->pts));
    delete b;
    return EZKL_OK;
}
int ezkl_hip_bases_generate(uint64_t seed, size_t first, size_t n, ezkl_bases_t* out) {
    if (!out || n == 0) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    Bases* b = new Bases();
    b->n = n;
    EZ_HIP(hipMalloc(&b->pts, n * 64));
    int rc = gen_bases(c, c->stream, seed, first, n, b->pts);
    if (rc) { (void)hipFree(b->pts); delete b; return rc; }
    *out = reinterpret_cast<ezkl_bases_t>(b);
    return EZKL_OK;
}
int ezkl_hip_bases_download(ezkl_bases_t h, void* out_host) {
    if (!h || !out_host) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    Bases* b = reinterpret_cast<Bases*>(h);
    EZ_HIP(copy_sync(c, out_host, b->pts, b->n * 64, hipMemcpyDeviceToHost));
    return EZKL_OK;
}
int ezkl_hip_bases_from_scalars(const void* base_point, const void* scalars_dev, size_t n, ezkl_bases_t* out) {
    if (!base_point || !scalars_dev || !out || n == 0) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    Bases* b = new Bases();
    b->n = n;
    EZ_HIP(hipMalloc(&b->pts, n * 64));
    int rc = g1_mul_fixed(c, c->stream, base_point, (const fe_t*)scalars_dev, n, b->pts);
    if (!rc) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = set_hip_error(e, "g1_mul_fixed", __FILE__, __LINE__);
    }
    if (rc) { (void)hipFree(b->pts); delete b; return rc; }
    *out = reinterpret_cast<ezkl_bases_t>(b);
    return EZKL_OK;
}
// ParamsKZG::downsize (see g1_to_lagrange in msm.hip): out_g = the first 2^new_k points of g (a device copy), out_g_lagrange = the
// Lagrange basis of the 2^new_k-point domain = the inverse NTT over G1 of those points
static fe_t domain_omega(uint32_t k, bool inverse);
int ezkl_hip_bases_downsize(ezkl_bases_t g, uint32_t new_k, ezkl_bases_t* out_g, ezkl_bases_t* out_g_lagrange) {
    if (!g || !out_g_lagrange || new_k > 28) return EZKL_ERR_INVALID;
    Bases* src = reinterpret_cast<Bases*>(g);
    const size_t n = (size_t)1 << new_k;
    if (src->n < n) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    Bases *bg = nullptr, *bl = new Bases();
    bl->n = n;
    fe_t* tw = nullptr;
    int rc = EZKL_OK;
    hipError_t e = hipMalloc(&bl->pts, n * 64);
    if (e == hipSuccess && n > 1) e = hipMalloc(&tw, (n / 2) * sizeof(fe_t));
    if (e != hipSuccess) rc = set_hip_error(e, "ezkl_hip_bases_downsize", __FILE__, __LINE__);
    if (!rc && n > 1) {                                    // tw[e] = omega^-e: fill with omega^-1, exclusive product scan
        rc = vec_fill(c, c->stream, tw, domain_omega(new_k, true), n / 2);
        if (!rc) rc = prefix_scan(c, c->stream, EZKL_VEC_MUL, 1, tw, tw, n / 2);
    }
    if (!rc) {
        fe_t ninv = Fr::one();
        const fe_t half = Fr::inv(Fr::add(Fr::one(), Fr::one()));
        for (uint32_t i = 0; i < new_k; i++) ninv = Fr::mul(ninv, half);
        rc = g1_to_lagrange(c, c->stream, src->pts, new_k, tw, ninv, bl->pts);
    }
    if (!rc && out_g) {
        bg = new Bases();
        bg->n = n;
        e = hipMalloc(&bg->pts, n * 64);
        if (e == hipSuccess) e = hipMemcpyAsync(bg->pts, src->pts, n * 64, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = set_hip_error(e, "ezkl_hip_bases_downsize", __FILE__, __LINE__);
    }
    if (tw) (void)hipFree(tw);
    if (rc) {
        if (bl->pts) (void)hipFree(bl->pts);
        delete bl;
        if (bg) { if (bg->pts) (void)hipFree(bg->pts); delete bg; }
        return rc;
    }
    if (out_g) *out_g = reinterpret_cast<ezkl_bases_t>(bg);
    *out_g_lagrange = reinterpret_cast<ezkl_bases_t>(bl);
    return EZKL_OK;
}
// sum_i scalars[i] * points[i] over G2 (host buffers; csrc/g2.hip): the G2 side of the SRS (gen_srs: s_g2 = [s] g2 is n = 1)
int ezkl_hip_msm_g2(const void* points_affine, const void* scalars, size_t n, void* out_affine) {
    if (!out_affine || (n && (!points_affine || !scalars))) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return g2_msm(c, c->stream, points_affine, scalars, n, out_affine);
}
size_t ezkl_hip_bases_len(ezkl_bases_t h) { return h ? reinterpret_cast<Bases*>(h)->n : 0; }

int ezkl_hip_msm_g1_dev(ezkl_bases_t h, size_t base_offset, const void* scalars_dev, size_t n, void* out, void* stream) {
    if (!h || !out || (!scalars_dev && n)) return EZKL_ERR_INVALID;
    Bases* b = reinterpret_cast<Bases*>(h);
    if (base_offset + n > b->n) return EZKL_ERR_INVALID;
    if (!stream && !getenv("EZKL_MSM_SERIAL_CALLS")) {     // on the library stream: concurrent callers overlap (msm_run_concurrent)
        ::ezkl::RoctxRange _roctx(__func__);
        Ctx* c = ctx();
        if (!c) return EZKL_ERR_NO_DEVICE;
        std::unique_lock<std::recursive_mutex> lk(c->mu);
        EZ_HIP(hipSetDevice(c->device));
        if (!msm_upload_is_open()) return msm_run_concurrent(c, lk, b, base_offset, (const fe_t*)scalars_dev, n, out);
        return msm_run(c, c->stream, b, base_offset, (const fe_t*)scalars_dev, n, out);
    }
    EZ_CTX(c);
    return msm_run(c, pick_stream(c, stream), b, base_offset, (const fe_t*)scalars_dev, n, out);
}
// one MSM in two halves (msm_call_start / msm_call_finish): the launches are queued behind the library stream's work so far and the call
// returns; finish waits for them and writes the point.  In between the caller may issue any other call, batches and upload phases included
// (the MSM runs on a call slot of its own).  `token` is the slot (0 .. 3).
int ezkl_hip_msm_g1_start_dev(ezkl_bases_t h, size_t base_offset, const void* scalars_dev, size_t n, int* token) {
    if (!h || !scalars_dev || !token || n == 0) return EZKL_ERR_INVALID;
    Bases* b = reinterpret_cast<Bases*>(h);
    if (base_offset + n > b->n) return EZKL_ERR_INVALID;
    ::ezkl::RoctxRange _roctx(__func__);
    Ctx* c = ctx();
    if (!c) return EZKL_ERR_NO_DEVICE;
    std::unique_lock<std::recursive_mutex> lk(c->mu);
    EZ_HIP(hipSetDevice(c->device));
    if (msm_upload_is_open()) return EZKL_ERR_INVALID;
    return msm_call_start(c, lk, b, base_offset, (const fe_t*)scalars_dev, n, token);
}
int ezkl_hip_msm_g1_finish(int token, void* out) {
    if (!out) return EZKL_ERR_INVALID;
    ::ezkl::RoctxRange _roctx(__func__);
    Ctx* c = ctx();
    if (!c) return EZKL_ERR_NO_DEVICE;
    std::unique_lock<std::recursive_mutex> lk(c->mu);
    EZ_HIP(hipSetDevice(c->device));
    return msm_call_finish(c, lk, token, out);
}
int ezkl_hip_msm_g1(ezkl_bases_t h, const void* scalars, size_t n, void* out) {
    if (!h || !out || (!scalars && n)) return EZKL_ERR_INVALID;
    Bases* b = reinterpret_cast<Bases*>(h);
    if (n > b->n) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (n == 0) { memset(out, 0, 64); return EZKL_OK; }
    fe_t* d = nullptr;
    EZ_HIP(hipMalloc(&d, n * 32));
    hipError_t e = hipMemcpyAsync(d, scalars, n * 32, hipMemcpyHostToDevice, c->stream);
    int rc = e == hipSuccess ? msm_run(c, c->stream, b, 0, d, n, out) : set_hip_error(e, "memcpy", __FILE__, __LINE__);
    (void)hipFree(d);
    return rc;
}
int ezkl_hip_msm_g1_batch_dev(ezkl_bases_t h, size_t base_offset, const void* const* scalars_dev, size_t batch, size_t n,
                              void* out, void* stream) {
    if (!h || !scalars_dev || !out) return EZKL_ERR_INVALID;
    Bases* b = reinterpret_cast<Bases*>(h);
    if (base_offset + n > b->n) return EZKL_ERR_INVALID;
    for (size_t i = 0; i < batch; i++)
        if (!scalars_dev[i] && n) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return msm_run_batch(c, pick_stream(c, stream), b, base_offset, (const fe_t* const*)scalars_dev, batch, n, out);
}
int ezkl_hip_msm_g1_batch_small_dev(ezkl_bases_t h, size_t base_offset, const void* const* scalars_dev, size_t batch, size_t n,
                                    void* out, void* stream) {
    if (!h || !scalars_dev || !out) return EZKL_ERR_INVALID;
    Bases* b = reinterpret_cast<Bases*>(h);
    if (base_offset + n > b->n) return EZKL_ERR_INVALID;
    for (size_t i = 0; i < batch; i++)
        if (!scalars_dev[i] && n) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return msm_run_batch(c, pick_stream(c, stream), b, base_offset, (const fe_t* const*)scalars_dev, batch, n, out, true);
}
int ezkl_hip_msm_g1_batch(ezkl_bases_t h, const void* const* scalars, size_t batch, size_t n, void* out) {
    if (!h || !scalars || !out) return EZKL_ERR_INVALID;
    Bases* b = reinterpret_cast<Bases*>(h);
    if (n > b->n) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (n == 0 || batch == 0) { memset(out, 0, 64 * batch); return EZKL_OK; }
    // stage the host columns in HBM (one allocation), then run the pipelined device batch
    fe_t* d = nullptr;
    EZ_HIP(hipMalloc(&d, batch * n * 32));
    std::vector<const void*> ptrs(batch);
    int rc = EZKL_OK;
    for (size_t i = 0; i < batch && !rc; i++) {
        if (!scalars[i]) { rc = EZKL_ERR_INVALID; break; }
        hipError_t e = hipMemcpyAsync(d + i * n, scalars[i], n * 32, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) rc = set_hip_error(e, "h2d", __FILE__, __LINE__);
        ptrs[i] = d + i * n;
    }
    if (!rc) rc = msm_run_batch(c, c->stream, b, 0, (const fe_t* const*)ptrs.data(), batch, n, out);
    (void)hipFree(d);
    return rc;
}
static int upload_args_ok(const void* const* host_cols, void* const* dev_cols, size_t batch, const void* const* tail_rows, size_t tail_count) {
    if (batch && (!host_cols || !dev_cols)) return 0;
    for (size_t j = 0; j < batch; j++)
        if (!host_cols[j] || !dev_cols[j] || (tail_rows && tail_count && !tail_rows[j])) return 0;
    return 1;
}
int ezkl_hip_upload_begin(const void* const* host_cols, void* const* dev_cols, size_t batch, size_t n, const void* const* tail_rows, size_t tail_start,
                          size_t tail_count, ezkl_upload_t* out) {
    if (!out || !upload_args_ok(host_cols, dev_cols, batch, tail_rows, tail_count)) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    MsmUpload* u = nullptr;
    int rc = msm_upload_begin(c, (const fe_t* const*)host_cols, (fe_t* const*)dev_cols, batch, n, (const fe_t* const*)tail_rows, tail_start, tail_count, &u);
    if (!rc) *out = reinterpret_cast<ezkl_upload_t>(u);
    return rc;
}
int ezkl_hip_upload_begin_fmt(const void* const* host_cols, const uint8_t* formats, void* const* dev_cols, size_t batch, size_t n, const void* const* tail_rows,
                              size_t tail_start, size_t tail_count, ezkl_upload_t* out) {
    if (!out || !upload_args_ok(host_cols, dev_cols, batch, tail_rows, tail_count)) return EZKL_ERR_INVALID;
    for (size_t j = 0; formats && j < batch; j++)
        if (formats[j] > EZKL_COLUMN_INT128) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    MsmUpload* u = nullptr;
    int rc = msm_upload_begin(c, (const fe_t* const*)host_cols, (fe_t* const*)dev_cols, batch, n, (const fe_t* const*)tail_rows, tail_start, tail_count, &u, formats);
    if (!rc) *out = reinterpret_cast<ezkl_upload_t>(u);
    return rc;
}
int ezkl_hip_upload_wait(ezkl_upload_t u, size_t column, void* stream) {
    if (!u || !stream) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return msm_upload_wait(reinterpret_cast<MsmUpload*>(u), column, (hipStream_t)stream);
}
int ezkl_hip_upload_commit(ezkl_upload_t u, ezkl_bases_t h, size_t commit_first, size_t commit_count, void* out) {
    if (!u || !h || !out) return EZKL_ERR_INVALID;
    Bases* b = reinterpret_cast<Bases*>(h);
    if (commit_count > b->n) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return msm_upload_commit(c, reinterpret_cast<MsmUpload*>(u), b, commit_first, commit_count, out);
}
int ezkl_hip_upload_end(ezkl_upload_t u) {
    if (!u) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return msm_upload_end(reinterpret_cast<MsmUpload*>(u));
}
int ezkl_hip_upload_commit_batch(ezkl_bases_t h, const void* const* host_cols, void* const* dev_cols, size_t batch, size_t n,
                                 const void* const* tail_rows, size_t tail_start, size_t tail_count, size_t commit_first, size_t commit_count,
                                 void* out) {
    if (!h || (batch && !out) || !upload_args_ok(host_cols, dev_cols, batch, tail_rows, tail_count)) return EZKL_ERR_INVALID;
    Bases* b = reinterpret_cast<Bases*>(h);
    if (commit_count > b->n || commit_first > n || commit_count > n - commit_first) return EZKL_ERR_INVALID;
    if (batch == 0) return EZKL_OK;
    EZ_CTX(c);
    MsmUpload* u = nullptr;
    int rc = msm_upload_begin(c, (const fe_t* const*)host_cols, (fe_t* const*)dev_cols, batch, n, (const fe_t* const*)tail_rows, tail_start, tail_count, &u);
    if (rc) return rc;
    rc = msm_upload_commit(c, u, b, commit_first, commit_count, out);
    int rc2 = msm_upload_end(u);
    return rc ? rc : rc2;
}
int ezkl_hip_stream_create(void** out) {
    if (!out) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    hipStream_t st;
    EZ_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *out = st;
    return EZKL_OK;
}
int ezkl_hip_context_stream(void** out) {
    if (!out) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (!c->side_stream) EZ_HIP(stream_create_prio(&c->side_stream, "EZKL_HIP_PRIO_AUX", 0));
    *out = c->side_stream;
    return EZKL_OK;
}
int ezkl_hip_stream_synchronize(void* stream) {
    if (!stream) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    EZ_HIP(hipStreamSynchronize((hipStream_t)stream));
    return EZKL_OK;
}
int ezkl_hip_stream_destroy(void* stream) {
    if (!stream) return EZKL_OK;
    EZ_CTX(c);
    auto it = c->scratch_by_stream.find((hipStream_t)stream);        // the stream's scratch arena goes with it
    if (it != c->scratch_by_stream.end()) {
        EZ_HIP(hipStreamSynchronize((hipStream_t)stream));
        if (it->second.ptr) (void)hipFree(it->second.ptr);
        if (it->second.last_event) (void)hipEventDestroy(it->second.last_event);
        c->scratch_by_stream.erase(it);
    }
    EZ_HIP(hipStreamDestroy((hipStream_t)stream));
    return EZKL_OK;
}
int ezkl_hip_msm_batch_begin(ezkl_bases_t h, size_t base_offset, size_t n, ezkl_msm_batch_t* out) {
    if (!h || !out) return EZKL_ERR_INVALID;
    Bases* b = reinterpret_cast<Bases*>(h);
    if (base_offset + n > b->n) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    MsmBatch* mb = nullptr;
    int rc = msm_batch_begin(c, c->stream, b, base_offset, n, &mb);
    if (!rc) *out = reinterpret_cast<ezkl_msm_batch_t>(mb);
    return rc;
}
int ezkl_hip_msm_batch_push_dev(ezkl_msm_batch_t batch, const void* scalars_dev) {
    if (!batch || !scalars_dev) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return msm_batch_push(c, reinterpret_cast<MsmBatch*>(batch), (const fe_t*)scalars_dev);   // ordered behind the library stream by an event
}
int ezkl_hip_msm_batch_push_many_dev(ezkl_msm_batch_t batch, const void* const* scalars_dev, size_t count) {
    if (!batch || (count && !scalars_dev)) return EZKL_ERR_INVALID;
    for (size_t i = 0; i < count; i++)
        if (!scalars_dev[i]) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return msm_batch_push_many(c, reinterpret_cast<MsmBatch*>(batch), (const fe_t* const*)scalars_dev, count, c->stream);
}
int ezkl_hip_msm_batch_finish(ezkl_msm_batch_t batch, void* out, size_t capacity) {
    if (!batch || (!out && capacity)) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return msm_batch_finish(c, reinterpret_cast<MsmBatch*>(batch), out, capacity);
}
int ezkl_hip_g1_add_affine(const void* a, const void* b, void* out) {
    if (!a || !b || !out) return EZKL_ERR_INVALID;
    g1_add_affine_host(a, b, out);
    return EZKL_OK;
}

/* ------------------------------------------------------------------ NTT -------------------- */
int ezkl_hip_ntt_dev(void* data, uint32_t log_n, const void* omega, int inverse, size_t batch, size_t stride, void* stream) {
    if (!data || !omega || batch == 0) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    fe_t w;
    memcpy(&w, omega, 32);
    hipStream_t st = pick_stream(c, stream);
    int rc = ntt_run(c, st, (const fe_t*)data, (fe_t*)data, log_n, w, inverse != 0, batch, stride, stride, log_n, 0);
    if (rc) return rc;
    return finish(c, st, stream);
}
int ezkl_hip_ntt(void* data, uint32_t log_n, const void* omega, int inverse) {
    if (!data || !omega || log_n > 28) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    size_t bytes = (size_t)32 << log_n;
    fe_t* d = nullptr;
    EZ_HIP(hipMalloc(&d, bytes));
    int rc = EZKL_OK;
    hipError_t e = hipMemcpyAsync(d, data, bytes, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) rc = set_hip_error(e, "h2d", __FILE__, __LINE__);
    if (!rc) rc = ezkl_hip_ntt_dev(d, log_n, omega, inverse, 1, (size_t)1 << log_n, nullptr);
    if (!rc) {
        e = copy_sync(c, data, d, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = set_hip_error(e, "d2h", __FILE__, __LINE__);
    }
    (void)hipFree(d);
    return rc;
}

static fe_t domain_omega(uint32_t k, bool inverse) {
    fe_t w = fr_const(FrConst::ROOT);
    for (uint32_t i = k; i < 28; i++) w = Fr::sqr(w);
    return inverse ? Fr::inv(w) : w;
}

int ezkl_hip_coset_ntt_dev(const void* in, void* out, size_t batch, size_t in_stride, size_t out_stride,
                           uint32_t log_n, uint32_t log_n_ext, int inverse, void* stream) {
    if (!in || !out || batch == 0 || log_n > log_n_ext || log_n_ext > 28) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    hipStream_t st = pick_stream(c, stream);
    fe_t w = domain_omega(log_n_ext, inverse != 0);
    int rc;
    if (!inverse)
        rc = ntt_run(c, st, (const fe_t*)in, (fe_t*)out, log_n_ext, w, false, batch, in_stride, out_stride, log_n, 1);
    else
        rc = ntt_run(c, st, (const fe_t*)in, (fe_t*)out, log_n_ext, w, true, batch, in_stride, out_stride, log_n_ext, 2);
    if (rc) return rc;
    return finish(c, st, stream);
}
int ezkl_hip_coeff_to_cosets_dev(const void* in, void* out, size_t batch, size_t in_stride, size_t out_stride, uint32_t log_n, uint32_t log_n_ext, void* stream) {
    if (!in || !out || in == out || batch == 0 || log_n > log_n_ext || log_n_ext > 28 || log_n_ext - log_n > 6) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    hipStream_t st = pick_stream(c, stream);
    int rc = coset_cm_run(c, st, (const fe_t*)in, (fe_t*)out, log_n, log_n_ext, domain_omega(log_n, false), domain_omega(log_n_ext, false), batch, in_stride, out_stride);
    if (rc) return rc;
    return finish(c, st, stream);
}
int ezkl_hip_coeff_to_cosets_range_dev(const void* in, void* out, size_t batch, size_t in_stride, size_t out_stride, uint32_t log_n, uint32_t log_n_ext,
                                       uint32_t first_coset, uint32_t n_cosets, void* stream) {
    if (!in || !out || in == out || batch == 0 || log_n > log_n_ext || log_n_ext > 28 || log_n_ext - log_n > 6) return EZKL_ERR_INVALID;
    if (n_cosets == 0 || (n_cosets & (n_cosets - 1)) || (uint64_t)first_coset + n_cosets > (1ull << (log_n_ext - log_n))) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    hipStream_t st = pick_stream(c, stream);
    int rc = coset_cm_run(c, st, (const fe_t*)in, (fe_t*)out, log_n, log_n_ext, domain_omega(log_n, false), domain_omega(log_n_ext, false), batch, in_stride, out_stride,
                          first_coset, n_cosets);
    if (rc) return rc;
    return finish(c, st, stream);
}
int ezkl_hip_cosets_transpose_dev(const void* in, void* out, uint32_t log_n, uint32_t log_n_ext, int to_natural, void* stream) {
    if (!in || !out || in == out || log_n > log_n_ext || log_n_ext > 28) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    hipStream_t st = pick_stream(c, stream);
    int rc = cm_transpose(c, st, (const fe_t*)in, (fe_t*)out, log_n, log_n_ext - log_n, to_natural != 0);
    if (rc) return rc;
    return finish(c, st, stream);
}
int ezkl_hip_coset_ntt_batch(const void* const* in, void* const* out, size_t batch, uint32_t log_n, uint32_t log_n_ext, int inverse) {
    if (!in || !out || log_n > log_n_ext || log_n_ext > 28) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    const size_t ne = (size_t)1 << log_n_ext, nin = inverse ? ne : ((size_t)1 << log_n);
    fe_t *din = nullptr, *dout = nullptr;
    EZ_HIP(hipMalloc(&din, nin * 32));
    EZ_HIP(hipMalloc(&dout, ne * 32));
    int rc = EZKL_OK;
    for (size_t b = 0; b < batch && !rc; b++) {
        hipError_t e = hipMemcpyAsync(din, in[b], nin * 32, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { rc = set_hip_error(e, "h2d", __FILE__, __LINE__); break; }
        rc = ezkl_hip_coset_ntt_dev(din, dout, 1, nin, ne, log_n, log_n_ext, inverse, nullptr);
        if (rc) break;
        e = copy_sync(c, out[b], dout, ne * 32, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = set_hip_error(e, "d2h", __FILE__, __LINE__);
    }
    (void)hipFree(din);
    (void)hipFree(dout);
    return rc;
}

/* ------------------------------------------------------------------ vec ops ---------------- */
int ezkl_hip_vec_op_dev(int op, const void* a, const void* b, void* o, size_t n, void* stream) {
    if (!a || !b || !o || op < 0 || op > 2) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    hipStream_t st = pick_stream(c, stream);
    int rc = vec_op(c, st, op, (const fe_t*)a, (const fe_t*)b, (fe_t*)o, n);
    return rc ? rc : finish(c, st, stream);
}
int ezkl_hip_vec_fill_dev(void* o, const void* v, size_t n, void* stream) {
    if (!o || !v) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    fe_t val;
    memcpy(&val, v, 32);
    hipStream_t st = pick_stream(c, stream);
    int rc = vec_fill(c, st, (fe_t*)o, val, n);
    return rc ? rc : finish(c, st, stream);
}
int ezkl_hip_permutation_sigma_dev(const void* next_dev, const void* omega_col_dev, const void* delta_pows_dev, uint32_t n_columns, uint32_t log_n,
                                   void* out_dev, void* stream) {
    if (!next_dev || !omega_col_dev || !delta_pows_dev || !out_dev || n_columns == 0 || log_n > 28) return EZKL_ERR_INVALID;
    if (((uint64_t)n_columns << log_n) > ((uint64_t)1 << 32)) return EZKL_ERR_INVALID;         // cells are numbered in 32 bits
    EZ_CTX(c);
    hipStream_t st = pick_stream(c, stream);
    int rc = perm_sigma(c, st, (const uint32_t*)next_dev, (const fe_t*)omega_col_dev, (const fe_t*)delta_pows_dev, log_n, n_columns, (fe_t*)out_dev);
    return rc ? rc : finish(c, st, stream);
}
int ezkl_hip_vec_scale_dev(const void* a, const void* s, void* o, size_t n, void* stream) {
    if (!a || !s || !o) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    fe_t sc;
    memcpy(&sc, s, 32);
    hipStream_t st = pick_stream(c, stream);
    int rc = vec_scale(c, st, (const fe_t*)a, sc, (fe_t*)o, n);
    return rc ? rc : finish(c, st, stream);
}
int ezkl_hip_divide_by_vanishing_dev(void* a, uint32_t k, uint32_t ext_k, void* stream) {
    if (!a || k > ext_k || ext_k > 28) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return divide_by_vanishing(c, pick_stream(c, stream), (fe_t*)a, k, ext_k);
}
int ezkl_hip_batch_invert_dev(void* a, size_t n, void* stream) {
    if (!a) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    hipStream_t st = pick_stream(c, stream);
    int rc = batch_invert(c, st, (fe_t*)a, n);
    return rc ? rc : finish(c, st, stream);
}

int ezkl_hip_lookup_multiplicity_dev(const void* const* inputs_dev, uint32_t n_inputs, const void* table_dev, uint32_t n_rows,
                                     uint32_t usable_rows, void* m_out_dev, uint32_t* out_missing, void* stream) {
    if (!table_dev || !m_out_dev || (n_inputs && !inputs_dev)) return EZKL_ERR_INVALID;
    for (uint32_t j = 0; j < n_inputs; j++)
        if (!inputs_dev[j]) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return lookup_multiplicity(c, pick_stream(c, stream), (const fe_t* const*)inputs_dev, n_inputs, (const fe_t*)table_dev, n_rows,
                               usable_rows, (fe_t*)m_out_dev, out_missing, nullptr);
}
int ezkl_hip_lookup_multiplicity_acc_dev(const void* const* inputs_dev, uint32_t n_inputs, const void* table_dev, uint32_t n_rows,
                                         uint32_t usable_rows, void* m_out_dev, void* missing_dev, void* stream) {
    if (!table_dev || !m_out_dev || !missing_dev || (n_inputs && !inputs_dev)) return EZKL_ERR_INVALID;
    for (uint32_t j = 0; j < n_inputs; j++)
        if (!inputs_dev[j]) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    hipStream_t st = pick_stream(c, stream);
    int rc = lookup_multiplicity(c, st, (const fe_t* const*)inputs_dev, n_inputs, (const fe_t*)table_dev, n_rows, usable_rows,
                                 (fe_t*)m_out_dev, nullptr, (uint32_t*)missing_dev);
    return rc ? rc : finish(c, st, stream);
}

int ezkl_hip_lookup_multiplicity_batch_dev(const void* const* inputs_dev, const uint32_t* input_lookup, uint32_t n_inputs, const void* const* tables_dev,
                                           uint32_t n_lookups, uint32_t n_rows, uint32_t usable_rows, void* const* m_outs_dev, void* missing_dev, void* stream) {
    if (!tables_dev || !m_outs_dev || !missing_dev || n_lookups == 0 || (n_inputs && (!inputs_dev || !input_lookup))) return EZKL_ERR_INVALID;
    for (uint32_t j = 0; j < n_inputs; j++)
        if (!inputs_dev[j]) return EZKL_ERR_INVALID;
    for (uint32_t l = 0; l < n_lookups; l++)
        if (!tables_dev[l] || !m_outs_dev[l]) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    hipStream_t st = pick_stream(c, stream);
    int rc = lookup_multiplicity_batch(c, st, (const fe_t* const*)inputs_dev, input_lookup, n_inputs, (const fe_t* const*)tables_dev, n_lookups, n_rows,
                                       usable_rows, (fe_t* const*)m_outs_dev, nullptr, (uint32_t*)missing_dev);
    return rc ? rc : finish(c, st, stream);
}

int ezkl_hip_eval_poly_dev(const void* coeffs, size_t n, const void* x, void* out, void* stream) {
    if ((!coeffs && n) || !x || !out) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    fe_t xx;
    memcpy(&xx, x, 32);
    return eval_poly(c, pick_stream(c, stream), (const fe_t*)coeffs, n, xx, out);
}

int ezkl_hip_eval_poly_batch_dev(const void* const* coeffs, const void* xs, uint32_t m, size_t n, void* out, void* stream) {
    if (m && (!coeffs || !xs || !out)) return EZKL_ERR_INVALID;
    for (uint32_t j = 0; j < m; j++)
        if (!coeffs[j] && n) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return eval_poly_batch(c, pick_stream(c, stream), (const fe_t* const*)coeffs, (const fe_t*)xs, m, n, out);
}
int ezkl_hip_lincomb_dev(const void* const* inputs, const void* coeffs, uint32_t m, void* out, size_t n, int accumulate, void* stream) {
    if (!out || (m && (!inputs || !coeffs))) return EZKL_ERR_INVALID;
    for (uint32_t j = 0; j < m; j++)
        if (!inputs[j]) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    hipStream_t st = pick_stream(c, stream);
    int rc = lincomb(c, st, (const fe_t* const*)inputs, (const fe_t*)coeffs, m, (fe_t*)out, n, accumulate);
    return rc ? rc : finish(c, st, stream);
}
int ezkl_hip_kate_division_dev(const void* a, const void* z, void* out, size_t n, void* stream) {
    if (!z || ((!a || !out) && n)) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    fe_t zz;
    memcpy(&zz, z, 32);
    hipStream_t st = pick_stream(c, stream);
    int rc = kate_division(c, st, (const fe_t*)a, zz, (fe_t*)out, n);
    return rc ? rc : finish(c, st, stream);
}
int ezkl_hip_chacha20_fr_dev(const void* key32, uint64_t stream_id, size_t first, void* out, size_t n, void* stream) {
    if (!key32 || (!out && n)) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    uint32_t key[8];
    memcpy(key, key32, 32);
    hipStream_t st = pick_stream(c, stream);
    int rc = chacha20_fr(c, st, key, stream_id, first, (fe_t*)out, n);
    return rc ? rc : finish(c, st, stream);
}

int ezkl_hip_prefix_scan_dev(int op, int exclusive, const void* in, void* out, size_t n, void* stream) {
    if (!in || !out || (op != EZKL_VEC_ADD && op != EZKL_VEC_MUL)) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    hipStream_t st = pick_stream(c, stream);
    int rc = prefix_scan(c, st, op, exclusive, (const fe_t*)in, (fe_t*)out, n);
    return rc ? rc : finish(c, st, stream);
}

int ezkl_hip_eval_h_dev(const ezkl_program_t* prog, void* out, void* stream) {
    if (!prog || !out) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return eval_program(c, pick_stream(c, stream), prog, (fe_t*)out, stream != nullptr || t_async_library_stream);
}

int ezkl_hip_eval_h_check(const ezkl_program_t* prog) {
    if (!prog || !prog->code || prog->n_instr == 0) return EZKL_ERR_INVALID;
    return eval_jit_compile_only(prog);        // host-only: hiprtc cross-compiles for gfx950 without a GPU
}

int ezkl_hip_eval_h_schedule(const ezkl_program_t* prog, uint32_t* out_code) {
    if (!prog || !prog->code || !out_code || prog->n_instr == 0) return EZKL_ERR_INVALID;
    return eval_schedule_only(prog, out_code);       // host-only: no device needed
}
int ezkl_hip_eval_h_prepare(const ezkl_program_t* prog) {
    if (!prog || !prog->code || prog->n_instr == 0) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return eval_prepare(c, prog);
}

int ezkl_hip_eval_h_jit_stats(uint64_t* compiled, uint64_t* from_disk, uint64_t* memory_hits) {
    if (!compiled || !from_disk || !memory_hits) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    eval_jit_stats(compiled, from_disk, memory_hits);
    return EZKL_OK;
}

int ezkl_hip_last_kernel_ms(const char* which, float* out_ms) {
    if (!which || !out_ms) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    auto it = c->events.find(which);
    if (it == c->events.end() || it->second.head == 0) return EZKL_ERR_INVALID;
    const unsigned i = (unsigned)((it->second.head - 1) % Ctx::EvRing::N);
    EZ_HIP(hipEventSynchronize(it->second.e1[i]));
    EZ_HIP(hipEventElapsedTime(out_ms, it->second.e0[i], it->second.e1[i]));
    return EZKL_OK;
}
int ezkl_hip_kernel_ms_stats(const char* which, double* sum_ms, uint64_t* count, int reset) {
    if (!which) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    auto it = c->events.find(which);
    if (it == c->events.end()) {                        // a region that has not run yet: empty statistics, not an error
        if (sum_ms) *sum_ms = 0;
        if (count) *count = 0;
        return EZKL_OK;
    }
    int rc = ev_harvest(it->second, true);
    if (rc) return rc;
    if (sum_ms) *sum_ms = it->second.sum_ms;
    if (count) *count = it->second.count;
    if (reset) { it->second.sum_ms = 0; it->second.count = 0; }
    return EZKL_OK;
}
int ezkl_hip_ubench(const char* which, double* out) {
    if (!which || !out) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    return ubench(c, which, out);
}

}  // extern "C"
