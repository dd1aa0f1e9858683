// comm.hip -- the collectives of the sharded prove path INSIDE the library: RCCL over xGMI on the library's own device pointers.
//
// one process per GPU; MSM shards by POINTS (every rank commits its slice of the SRS, the 64-byte partial sums are
// all_gathered and folded with the group law -- RCCL has no elliptic-curve reduction op, so this is gather-then-add, not
// ncclReduce(sum)); the quotient sweep shards by ROWS (h is all_gathered in place); columns transformed by their owner reach the
// row shards through ONE all-to-all (grouped ncclSend / ncclRecv to all peers at once: xGMI is point-to-point, 7 links per GPU, so
// a direct exchange uses every link concurrently where a ring would be bound by one).  The reference has no multi-GPU path at all
// (SURVEY.md §2 "Parallelism strategies": none; icicle is single-GPU), so there is no interface to replace: the entry points
// below are what a fork's launcher calls once (unique id exchange + init) and what libezkl_prover.so uses afterwards.
// librccl is loaded lazily (dlopen) on the first comm call: a single-GPU `ezkl prove` never pays for it.
#include "common.hpp"
#include <dlfcn.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <chrono>
#include <thread>
#include <rccl/rccl.h>

namespace ezkl {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;       // optional: the watchdog uses it to release a stuck communicator
};
static RcclApi g_rccl;
struct Comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    hipStream_t st = nullptr;
    void* stage = nullptr;          // device staging for the partial points of one commit batch
    size_t stage_bytes = 0;
    // the packed all-to-all: one slab per peer and direction, the copy descriptors of one round, and what the exchanges have moved
    char* slab_send = nullptr;
    char* slab_recv = nullptr;
    size_t slab_bytes = 0;          // per peer and direction
    void* desc_dev = nullptr;
    void* desc_host = nullptr;      // pinned
    size_t desc_cap = 0;            // descriptors
    uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipEvent_t wd_ev = nullptr;     // the watchdog's event (comm_wait)
    bool broken = false;            // a collective timed out: every later call fails at once instead of hanging behind it
};
static Comm g_comm;

// ---- watchdog: every collective is WAITED FOR with a deadline ---------------------------------------------------------------------------
// A peer that died, took another branch or never arrived leaves the others inside a collective for ever: hipStreamSynchronize on the
// communicator's stream does not come back, the job hangs until a driver's outer timeout kills it and nobody learns which call it was.
// comm_wait records an event behind the queued work and polls it: past EZKL_COMM_TIMEOUT_S seconds (default 120; 0 = wait for ever) the
// call names the collective on stderr, aborts the communicator (ncclCommAbort, so that the process can exit) and returns
// EZKL_ERR_TIMEOUT -- a status code a caller can act on (libezkl_prover.so fails the proof on every rank: failures are collective).
static double comm_timeout_s() {
    const char* e = getenv("EZKL_COMM_TIMEOUT_S");
    if (!e || !*e) return 120.0;
    const double v = atof(e);
    return v < 0 ? 120.0 : v;
}
static int comm_wait(const char* what) {
    if (g_comm.broken) return EZKL_ERR_TIMEOUT;
    const double limit = comm_timeout_s();
    if (limit == 0.0) { EZ_HIP(hipStreamSynchronize(g_comm.st)); return EZKL_OK; }
    if (!g_comm.wd_ev) EZ_HIP(hipEventCreateWithFlags(&g_comm.wd_ev, hipEventDisableTiming));
    EZ_HIP(hipEventRecord(g_comm.wd_ev, g_comm.st));
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;) {
        const hipError_t q = hipEventQuery(g_comm.wd_ev);
        if (q == hipSuccess) return EZKL_OK;
        if (q != hipErrorNotReady) { EZ_HIP(q); }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > limit) break;
        if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(waited < 0.01 ? 20 : 500));   // spin for the common case, then back off
    }
    (void)hipGetLastError();
    fprintf(stderr, "[ezkl_hip] rank %d of %d: %s did not complete within %.0f s (EZKL_COMM_TIMEOUT_S): a peer is missing or took another "
                    "path; aborting the communicator\n", g_comm.rank, g_comm.world, what, limit);
    g_comm.broken = true;
    if (g_rccl.CommAbort && g_comm.comm) { (void)g_rccl.CommAbort(g_comm.comm); g_comm.comm = nullptr; }
    return EZKL_ERR_TIMEOUT;
}
#define EZ_WAIT(what)                                                   \
    do {                                                                \
        int _w = comm_wait(what);                                       \
        if (_w) return _w;                                              \
    } while (0)

static int rccl_load() {
    if (g_rccl.lib) return EZKL_OK;
    // RCCL must run on the SAME HIP runtime as this library.  A process can hold two (PyTorch wheels bundle libamdhip64 / librccl with
    // the system sonames; whichever is loaded first satisfies later NEEDED entries), and a second runtime finds "no ROCm-capable
    // device": take librccl from the directory of the libamdhip64 this library is bound to, by path, before falling back to the soname.
    void* h = nullptr;
    Dl_info di;
    if (dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) {
        std::string dir(di.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash + 1);
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                h = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
                if (h) break;
            }
        }
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        fprintf(stderr, "[ezkl_hip] RCCL not available: %s\n", dlerror());
        return EZKL_ERR_UNSUPPORTED;
    }
    RcclApi a;
    a.lib = h;
#define EZ_SYM(field, name)                                             \
    *(void**)(&a.field) = dlsym(h, name);                               \
    if (!a.field) { dlclose(h); return EZKL_ERR_UNSUPPORTED; }
    EZ_SYM(GetUniqueId, "ncclGetUniqueId")
    EZ_SYM(CommInitRank, "ncclCommInitRank")
    EZ_SYM(CommDestroy, "ncclCommDestroy")
    EZ_SYM(AllGather, "ncclAllGather")
    EZ_SYM(Send, "ncclSend")
    EZ_SYM(Recv, "ncclRecv")
    EZ_SYM(GroupStart, "ncclGroupStart")
    EZ_SYM(GroupEnd, "ncclGroupEnd")
    EZ_SYM(GetErrorString, "ncclGetErrorString")
#undef EZ_SYM
    *(void**)(&a.CommAbort) = dlsym(h, "ncclCommAbort");
    g_rccl = a;
    return EZKL_OK;
}
static int rccl_fail(ncclResult_t r, const char* what) {
    fprintf(stderr, "[ezkl_hip] %s failed: %s\n", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return EZKL_ERR_HIP;
}
#define EZ_RCCL(call)                                                   \
    do {                                                                \
        ncclResult_t _r = (call);                                       \
        if (_r != ncclSuccess) return rccl_fail(_r, #call);             \
    } while (0)


// ---- the packed all-to-all ------------------------------------------------------------------------------------------------------------
// xGMI wants a few large messages: the column-sharded prover's exchange is hundreds (k = 20) to thousands (k = 22) of row slabs per peer,
// and one ncclSend / ncclRecv each would leave their scheduling to RCCL's point-to-point engine, which has never been shown to like
// that (missing #1).  Wire format: per peer ONE contiguous byte stream = the peer's segments in list order, cut into rounds
// of at most `slab` bytes; a gather kernel packs every peer's share of the round into its slab, ONE ncclSend + ONE ncclRecv per peer
// move the slabs (7 + 7 per rank at world 8), a scatter kernel unpacks.  The segment lists stay the description; both sides derive the
// same round boundaries from the same totals.
struct CopyDesc {
    const char* src;
    char* dst;
    uint64_t bytes;
    uint64_t start;       // offset of this piece in the dense byte space of the launch (prefix sum of the pieces before it)
};
static constexpr uint32_t PACK_CHUNK = 32768;       // bytes per workgroup
// workgroup b copies dense bytes [b * PACK_CHUNK, (b + 1) * PACK_CHUNK): it finds its first piece by bisection and walks on while the chunk
// spans several.  The prover's pieces are multiples of 16 bytes at 16-byte aligned addresses (field elements: 32 B) and move as 16-byte
// vectors; a piece (or the part of it inside this chunk) that is NOT so aligned moves byte by byte -- the wire format never depends on the
// alignment a rank happens to see (a rank-local fallback to another format was a hang, not an error)
__global__ __launch_bounds__(256) void comm_pack_kernel(const CopyDesc* desc, uint32_t n_desc, uint64_t total) {
    const uint64_t lo = (uint64_t)blockIdx.x * PACK_CHUNK, hi = lo + PACK_CHUNK < total ? lo + PACK_CHUNK : total;
    uint32_t a = 0, b = n_desc;                       // last piece with start <= lo
    while (b - a > 1) {
        const uint32_t m = (a + b) >> 1;
        if (desc[m].start <= lo) a = m; else b = m;
    }
    for (uint32_t d = a; d < n_desc; d++) {
        const CopyDesc ds = desc[d];
        if (ds.start >= hi) break;
        const uint64_t from = lo > ds.start ? lo - ds.start : 0, to = (hi - ds.start) < ds.bytes ? (hi - ds.start) : ds.bytes;
        if ((((uintptr_t)(ds.src + from)) | ((uintptr_t)(ds.dst + from)) | (uintptr_t)(to - from)) & 15) {
            const char* src = ds.src + from;
            char* dst = ds.dst + from;
            for (uint64_t i = threadIdx.x; i < to - from; i += 256) dst[i] = src[i];
            continue;
        }
        const uint4* src = (const uint4*)(ds.src + from);
        uint4* dst = (uint4*)(ds.dst + from);
        const uint32_t n16 = (uint32_t)((to - from) >> 4);
        for (uint32_t i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
    }
}
// a byte stream over a list of segments to / from one peer
struct SegCursor {
    const ezkl_comm_seg_t* segs;
    size_t n, idx = 0, off = 0;
    int peer;
    SegCursor(const ezkl_comm_seg_t* s, size_t n_, int p) : segs(s), n(n_), peer(p) { skip(); }
    void skip() {
        while (idx < n && (segs[idx].peer != peer || off >= segs[idx].bytes)) { idx++; off = 0; }
    }
    bool done() const { return idx >= n; }
    // the next contiguous piece, at most `want` bytes
    size_t take(size_t want, char** ptr) {
        const size_t m = std::min(want, segs[idx].bytes - off);
        *ptr = (char*)segs[idx].ptr + off;
        off += m;
        skip();
        return m;
    }
};
static int comm_descs_reserve(size_t count) {
    if (count <= g_comm.desc_cap) return EZKL_OK;
    if (g_comm.desc_dev) { EZ_HIP(hipStreamSynchronize(g_comm.st)); EZ_HIP(hipFree(g_comm.desc_dev)); g_comm.desc_dev = nullptr; }
    if (g_comm.desc_host) { EZ_HIP(hipHostFree(g_comm.desc_host)); g_comm.desc_host = nullptr; }
    const size_t cap = count + count / 2 + 64;
    EZ_HIP(hipMalloc(&g_comm.desc_dev, 2 * cap * sizeof(CopyDesc)));            // [0, cap): pack, [cap, 2 cap): unpack of the same round
    EZ_HIP(hipHostMalloc(&g_comm.desc_host, 2 * cap * sizeof(CopyDesc), hipHostMallocDefault));
    g_comm.desc_cap = cap;
    return EZKL_OK;
}
static int comm_launch_copies(const std::vector<CopyDesc>& d, uint64_t total, size_t slot) {
    if (d.empty() || !total) return EZKL_OK;
    CopyDesc* h = (CopyDesc*)g_comm.desc_host + slot * g_comm.desc_cap;
    CopyDesc* dv = (CopyDesc*)g_comm.desc_dev + slot * g_comm.desc_cap;
    memcpy(h, d.data(), d.size() * sizeof(CopyDesc));
    EZ_HIP(hipMemcpyAsync(dv, h, d.size() * sizeof(CopyDesc), hipMemcpyHostToDevice, g_comm.st));
    hipLaunchKernelGGL(comm_pack_kernel, dim3((unsigned)((total + PACK_CHUNK - 1) / PACK_CHUNK)), dim3(256), 0, g_comm.st, dv, (uint32_t)d.size(), total);
    EZ_HIP(hipGetLastError());
    return EZKL_OK;
}

}  // namespace ezkl

using namespace ezkl;
extern "C" {

// does librccl load next to this library's HIP runtime, with every entry point comm.hip binds (ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclAllGather, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString)?  No device and no RCCL call is
// made: a dry check for a build box / CI, and the first thing an 8-rank job can ask before it commits to the library communicator.
int ezkl_hip_comm_available(void) { return rccl_load(); }
int ezkl_hip_comm_selftest(void);

int ezkl_hip_comm_unique_id(void* out128) {
    if (!out128) return EZKL_ERR_INVALID;
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    EZ_RCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out128, &id, 128);
    return EZKL_OK;
}

int ezkl_hip_comm_init(const void* id128, int world, int rank) {
    if (!id128 || world < 1 || rank < 0 || rank >= world) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (g_comm.comm) return EZKL_ERR_INVALID;           // one communicator per process (= per GPU)
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    EZ_HIP(hipStreamCreateWithFlags(&g_comm.st, hipStreamNonBlocking));
    EZ_HIP(hipDeviceSynchronize());
    (void)hipGetLastError();                            // RCCL reads the thread's last HIP error during init: an earlier, already reported one must not fail it
    EZ_RCCL(g_rccl.CommInitRank(&g_comm.comm, world, id, rank));
    g_comm.world = world;
    g_comm.rank = rank;
    // every rank tries the communicator out before anything depends on it (EZKL_COMM_SELFTEST=0 skips; the setting must be the same
    // on every rank -- the self-test is a collective)
    const char* stest = getenv("EZKL_COMM_SELFTEST");
    if (!stest || *stest != '0') {
        rc = ezkl_hip_comm_selftest();
        if (rc) fprintf(stderr, "[ezkl_hip] rank %d of %d: communicator self-test failed (%d); the communicator is not usable\n", rank, world, rc);
        return rc;
    }
    return EZKL_OK;
}

int ezkl_hip_comm_info(int* world, int* rank) {
    if (world) *world = g_comm.comm ? g_comm.world : 0;
    if (rank) *rank = g_comm.comm ? g_comm.rank : 0;
    return EZKL_OK;
}

int ezkl_hip_comm_destroy(void) {
    EZ_CTX(c);
    if (!g_comm.comm && !g_comm.broken) return EZKL_OK;
    if (g_comm.comm) {
        EZ_HIP(hipStreamSynchronize(g_comm.st));
        EZ_RCCL(g_rccl.CommDestroy(g_comm.comm));
    }
    if (g_comm.wd_ev) (void)hipEventDestroy(g_comm.wd_ev);
    if (g_comm.stage) (void)hipFree(g_comm.stage);
    if (g_comm.slab_send) (void)hipFree(g_comm.slab_send);
    if (g_comm.slab_recv) (void)hipFree(g_comm.slab_recv);
    if (g_comm.desc_dev) (void)hipFree(g_comm.desc_dev);
    if (g_comm.desc_host) (void)hipHostFree(g_comm.desc_host);
    (void)hipStreamDestroy(g_comm.st);
    g_comm = Comm();
    return EZKL_OK;
}

// In-place all_gather of a device buffer of world equal slices: rank r wrote slice r.  The library stream's earlier work on the
// buffer is ordered before the collective, and the call returns when the gathered data is in place.
int ezkl_hip_comm_allgather_dev(void* buf_dev, size_t total_bytes) {
    if (!buf_dev) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (g_comm.broken) return EZKL_ERR_TIMEOUT;     // a collective timed out earlier: nothing runs behind it
    if (!g_comm.comm || total_bytes % (size_t)g_comm.world) return EZKL_ERR_INVALID;
    const size_t slice = total_bytes / (size_t)g_comm.world;
    EZ_HIP(hipStreamSynchronize(c->stream));
    EZ_RCCL(g_rccl.AllGather((const char*)buf_dev + slice * (size_t)g_comm.rank, buf_dev, slice, ncclUint8, g_comm.comm, g_comm.st));
    EZ_WAIT("all_gather (device)");
    return EZKL_OK;
}

// The fold of one commit batch: `count` 64-byte affine Montgomery partial sums (host) in, their sums over all ranks out (in place).
// all_gather of count * 64 B per rank + the group law on the host (host64.hpp): the RCCL "reduce" of the bucket sums.
int ezkl_hip_comm_fold_points(void* points_host, uint32_t count) {
    if (!points_host || count == 0) return count ? EZKL_ERR_INVALID : EZKL_OK;
    EZ_CTX(c);
    if (g_comm.broken) return EZKL_ERR_TIMEOUT;     // a collective timed out earlier: nothing runs behind it
    if (!g_comm.comm) return EZKL_ERR_INVALID;
    const size_t slice = (size_t)count * 64, total = slice * (size_t)g_comm.world;
    if (g_comm.stage_bytes < total) {
        if (g_comm.stage) EZ_HIP(hipFree(g_comm.stage));
        EZ_HIP(hipMalloc(&g_comm.stage, total));
        g_comm.stage_bytes = total;
    }
    char* st = (char*)g_comm.stage;
    EZ_HIP(hipMemcpyAsync(st + slice * (size_t)g_comm.rank, points_host, slice, hipMemcpyHostToDevice, g_comm.st));
    EZ_RCCL(g_rccl.AllGather(st + slice * (size_t)g_comm.rank, st, slice, ncclUint8, g_comm.comm, g_comm.st));
    std::vector<uint8_t> all(total);
    EZ_HIP(hipMemcpyAsync(all.data(), st, total, hipMemcpyDeviceToHost, g_comm.st));
    EZ_WAIT("fold_points (all_gather of partial sums)");
    uint8_t* out = (uint8_t*)points_host;
    for (uint32_t j = 0; j < count; j++) {
        uint8_t acc[64];
        memcpy(acc, all.data() + (size_t)j * 64, 64);
        for (int r = 1; r < g_comm.world; r++) g1_add_affine_host(acc, all.data() + (size_t)r * slice + (size_t)j * 64, acc);
        memcpy(out + (size_t)j * 64, acc, 64);
    }
    return EZKL_OK;
}

// broadcast of a small host buffer from `root` (an all_gather of everyone's copy, root's slice kept): the 256-bit ChaCha key of a sharded
// proof drawn from OS entropy on rank 0 (every rank must blind with the same randomness to emit the same proof)
int ezkl_hip_comm_broadcast_host(void* buf_host, size_t bytes, int root) {
    if (!buf_host || bytes == 0 || bytes > 4096) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (g_comm.broken) return EZKL_ERR_TIMEOUT;     // a collective timed out earlier: nothing runs behind it
    if (!g_comm.comm || root < 0 || root >= g_comm.world) return EZKL_ERR_INVALID;
    const size_t total = bytes * (size_t)g_comm.world;
    if (g_comm.stage_bytes < total) {
        if (g_comm.stage) EZ_HIP(hipFree(g_comm.stage));
        EZ_HIP(hipMalloc(&g_comm.stage, total));
        g_comm.stage_bytes = total;
    }
    char* st = (char*)g_comm.stage;
    EZ_HIP(hipMemcpyAsync(st + bytes * (size_t)g_comm.rank, buf_host, bytes, hipMemcpyHostToDevice, g_comm.st));
    EZ_RCCL(g_rccl.AllGather(st + bytes * (size_t)g_comm.rank, st, bytes, ncclUint8, g_comm.comm, g_comm.st));
    EZ_HIP(hipMemcpyAsync(buf_host, st + bytes * (size_t)root, bytes, hipMemcpyDeviceToHost, g_comm.st));
    EZ_WAIT("broadcast (host)");
    return EZKL_OK;
}

int ezkl_hip_comm_allgather_host(void* buf_host, size_t bytes) {
    if (!buf_host || bytes == 0 || bytes > ((size_t)1 << 24)) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (g_comm.broken) return EZKL_ERR_TIMEOUT;     // a collective timed out earlier: nothing runs behind it
    if (!g_comm.comm) return EZKL_ERR_INVALID;
    const size_t total = bytes * (size_t)g_comm.world;
    if (g_comm.stage_bytes < total) {
        if (g_comm.stage) EZ_HIP(hipFree(g_comm.stage));
        EZ_HIP(hipMalloc(&g_comm.stage, total));
        g_comm.stage_bytes = total;
    }
    char* st = (char*)g_comm.stage;
    char* h = (char*)buf_host;
    EZ_HIP(hipMemcpyAsync(st + bytes * (size_t)g_comm.rank, h + bytes * (size_t)g_comm.rank, bytes, hipMemcpyHostToDevice, g_comm.st));
    EZ_RCCL(g_rccl.AllGather(st + bytes * (size_t)g_comm.rank, st, bytes, ncclUint8, g_comm.comm, g_comm.st));
    EZ_HIP(hipMemcpyAsync(h, st, total, hipMemcpyDeviceToHost, g_comm.st));
    EZ_WAIT("all_gather (host)");
    return EZKL_OK;
}

// all-to-all with any number of segments per peer (see include/ezkl_hip.h): the k-th BYTE this rank sends to peer p is the k-th byte p
// receives from this rank -- per peer the segments form one stream, and only the totals have to agree.  Packed wire format (above);
// EZKL_COMM_UNPACKED=1 keeps round 3's one ncclSend / ncclRecv per segment (then the segment SIZES must agree pairwise too).
static int alltoallv_unpacked(Ctx* c, const ezkl_comm_seg_t* sends, size_t n_sends, const ezkl_comm_seg_t* recvs, size_t n_recvs, int skip) {
    const int me = g_comm.rank;
    if (skip == me) {
        size_t j = 0;                                 // self segments, in order
        for (size_t i = 0; i < n_sends; i++) {
            if (sends[i].peer != me) continue;
            while (j < n_recvs && recvs[j].peer != me) j++;
            if (j == n_recvs || recvs[j].bytes != sends[i].bytes) return EZKL_ERR_INVALID;
            if (sends[i].bytes) EZ_HIP(hipMemcpyAsync(recvs[j].ptr, sends[i].ptr, sends[i].bytes, hipMemcpyDeviceToDevice, g_comm.st));
            j++;
        }
    }
    EZ_RCCL(g_rccl.GroupStart());
    for (size_t i = 0; i < n_sends; i++)
        if (sends[i].peer != skip && sends[i].bytes) { EZ_RCCL(g_rccl.Send(sends[i].ptr, sends[i].bytes, ncclUint8, sends[i].peer, g_comm.comm, g_comm.st)); g_comm.stats[4]++; }
    for (size_t i = 0; i < n_recvs; i++)
        if (recvs[i].peer != skip && recvs[i].bytes) { EZ_RCCL(g_rccl.Recv(recvs[i].ptr, recvs[i].bytes, ncclUint8, recvs[i].peer, g_comm.comm, g_comm.st)); g_comm.stats[5]++; }
    EZ_RCCL(g_rccl.GroupEnd());
    EZ_WAIT("all-to-all (one send / recv per segment)");
    return EZKL_OK;
}
int ezkl_hip_comm_alltoallv_dev(const ezkl_comm_seg_t* sends, size_t n_sends, const ezkl_comm_seg_t* recvs, size_t n_recvs) {
    if ((n_sends && !sends) || (n_recvs && !recvs)) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (g_comm.broken) return EZKL_ERR_TIMEOUT;     // a collective timed out earlier: nothing runs behind it
    if (!g_comm.comm) return EZKL_ERR_INVALID;
    const int me = g_comm.rank, world = g_comm.world;
    std::vector<uint64_t> tot_s(world, 0), tot_r(world, 0);
    for (size_t i = 0; i < n_sends; i++) {
        if (sends[i].peer < 0 || sends[i].peer >= world || (!sends[i].ptr && sends[i].bytes)) return EZKL_ERR_INVALID;
        tot_s[sends[i].peer] += sends[i].bytes;
    }
    for (size_t i = 0; i < n_recvs; i++) {
        if (recvs[i].peer < 0 || recvs[i].peer >= world || (!recvs[i].ptr && recvs[i].bytes)) return EZKL_ERR_INVALID;
        tot_r[recvs[i].peer] += recvs[i].bytes;
    }
    if (tot_s[me] != tot_r[me]) return EZKL_ERR_INVALID;
    const auto t_begin = std::chrono::steady_clock::now();
    EZ_HIP(hipStreamSynchronize(c->stream));          // what the library stream produced is what gets sent
    // EZKL_COMM_SELF_VIA_RCCL=1 (testing): the bytes to self take the slab + ncclSend / ncclRecv path too, so that a one-GPU box exercises
    // the wire format, the rounds and RCCL's matching (tests/test_gpu_comm.py)
    const char* via = getenv("EZKL_COMM_SELF_VIA_RCCL");
    const int skip = via && *via == '1' ? -1 : me;    // the peer whose bytes are plain copies
    const char* unp = getenv("EZKL_COMM_UNPACKED");
    int rc = EZKL_OK;
    if (unp && *unp == '1') {                         // the environment is the same on every rank: the format is never a rank-local decision
        rc = alltoallv_unpacked(c, sends, n_sends, recvs, n_recvs, skip);
    } else {
        size_t slab = (size_t)128 << 20;              // per peer and direction; every rank must use the same value
        if (const char* e = getenv("EZKL_COMM_SLAB_MB")) { const size_t v = (size_t)strtoull(e, nullptr, 10); if (v) slab = v << 20; }
        slab &= ~(size_t)15;
        uint64_t most = 0;
        for (int p = 0; p < world; p++) if (p != skip) most = std::max(most, std::max(tot_s[p], tot_r[p]));
        const size_t need = (size_t)std::min<uint64_t>(most, slab);
        if (need > g_comm.slab_bytes) {               // grown on demand, kept for the next proof
            if (g_comm.slab_send) { EZ_HIP(hipStreamSynchronize(g_comm.st)); EZ_HIP(hipFree(g_comm.slab_send)); EZ_HIP(hipFree(g_comm.slab_recv)); g_comm.slab_send = g_comm.slab_recv = nullptr; g_comm.slab_bytes = 0; }
            EZ_HIP(hipMalloc((void**)&g_comm.slab_send, need * (size_t)world));
            EZ_HIP(hipMalloc((void**)&g_comm.slab_recv, need * (size_t)world));
            g_comm.slab_bytes = need;
        }
        rc = comm_descs_reserve(n_sends + n_recvs + 2 * (size_t)world);
        if (rc) return rc;
        std::vector<SegCursor> cs, cr;
        for (int p = 0; p < world; p++) { cs.emplace_back(sends, n_sends, p); cr.emplace_back(recvs, n_recvs, p); }
        std::vector<uint64_t> left_s = tot_s, left_r = tot_r;
        std::vector<CopyDesc> pack, unpack;
        bool first_round = true;
        for (;;) {
            bool any = false;
            for (int p = 0; p < world; p++) any |= left_s[p] || left_r[p];
            if (!any) break;
            pack.clear(); unpack.clear();
            uint64_t pack_total = 0, unpack_total = 0;
            std::vector<size_t> rs(world, 0), rr(world, 0);       // bytes of this round per peer
            for (int p = 0; p < world; p++) {
                if (p == skip) {                       // self: straight from the send segments into the receive segments, all of it in the first round
                    while (left_s[p]) {
                        char *sp, *dp;
                        const size_t want = std::min(cs[p].segs[cs[p].idx].bytes - cs[p].off, cr[p].segs[cr[p].idx].bytes - cr[p].off);
                        const size_t m = cs[p].take(want, &sp);
                        cr[p].take(m, &dp);
                        pack.push_back({sp, dp, m, pack_total});
                        pack_total += m; left_s[p] -= m; left_r[p] -= m;
                    }
                    continue;
                }
                rs[p] = (size_t)std::min<uint64_t>(left_s[p], slab);      // the ROUND size is the agreed slab size, not what happens to be allocated here
                rr[p] = (size_t)std::min<uint64_t>(left_r[p], slab);
                for (size_t done = 0; done < rs[p];) {
                    char* sp;
                    const size_t m = cs[p].take(rs[p] - done, &sp);
                    pack.push_back({sp, g_comm.slab_send + (size_t)p * g_comm.slab_bytes + done, m, pack_total});
                    pack_total += m; done += m;
                }
                for (size_t done = 0; done < rr[p];) {
                    char* dp;
                    const size_t m = cr[p].take(rr[p] - done, &dp);
                    unpack.push_back({g_comm.slab_recv + (size_t)p * g_comm.slab_bytes + done, dp, m, unpack_total});
                    unpack_total += m; done += m;
                }
                left_s[p] -= rs[p]; left_r[p] -= rr[p];
            }
            if (pack.size() > g_comm.desc_cap || unpack.size() > g_comm.desc_cap) {       // segments cut by round boundaries: at most one more per peer and round
                EZ_HIP(hipStreamSynchronize(g_comm.st));
                rc = comm_descs_reserve(std::max(pack.size(), unpack.size()));
                if (rc) return rc;
            }
            if (!first_round) EZ_WAIT("packed all-to-all (a round)");      // the pinned descriptor block of the last round has been read
            first_round = false;
            rc = comm_launch_copies(pack, pack_total, 0);
            if (rc) return rc;
            EZ_RCCL(g_rccl.GroupStart());
            for (int p = 0; p < world; p++) {
                if (p == skip) continue;
                if (rs[p]) { EZ_RCCL(g_rccl.Send(g_comm.slab_send + (size_t)p * g_comm.slab_bytes, rs[p], ncclUint8, p, g_comm.comm, g_comm.st)); g_comm.stats[4]++; }
                if (rr[p]) { EZ_RCCL(g_rccl.Recv(g_comm.slab_recv + (size_t)p * g_comm.slab_bytes, rr[p], ncclUint8, p, g_comm.comm, g_comm.st)); g_comm.stats[5]++; }
            }
            EZ_RCCL(g_rccl.GroupEnd());
            rc = comm_launch_copies(unpack, unpack_total, 1);
            if (rc) return rc;
            g_comm.stats[6]++;
        }
        EZ_WAIT("packed all-to-all");
    }
    if (rc) return rc;
    g_comm.stats[0]++;
    for (int p = 0; p < world; p++) if (p != me) { g_comm.stats[1] += tot_s[p]; g_comm.stats[2] += tot_r[p]; }
    g_comm.stats[3] += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_begin).count();
    return EZKL_OK;
}

// what the exchanges of this process have moved: out[0] = calls, out[1] / out[2] = bytes sent to / received from OTHER ranks, out[3] =
// microseconds of host wall time inside the calls, out[4] / out[5] = ncclSend / ncclRecv operations issued, out[6] = rounds, out[7] = 0
int ezkl_hip_comm_stats(uint64_t out[8], int reset) {
    if (!out) return EZKL_ERR_INVALID;
    for (int i = 0; i < 8; i++) out[i] = g_comm.stats[i];
    if (reset) for (auto& x : g_comm.stats) x = 0;
    return EZKL_OK;
}

// all-to-all on device pointers: for every peer p, send_len[p] bytes at send_dev + send_off[p] go to p, recv_len[p] bytes from p land
// at recv_dev + recv_off[p].  All peers at once (grouped ncclSend / ncclRecv); the slice to self is a device-to-device copy.
int ezkl_hip_comm_alltoall_dev(const void* send_dev, const size_t* send_off, const size_t* send_len, void* recv_dev, const size_t* recv_off,
                               const size_t* recv_len) {
    if (!send_off || !send_len || !recv_off || !recv_len) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (g_comm.broken) return EZKL_ERR_TIMEOUT;     // a collective timed out earlier: nothing runs behind it
    if (!g_comm.comm) return EZKL_ERR_INVALID;
    EZ_HIP(hipStreamSynchronize(c->stream));
    const int me = g_comm.rank;
    if (send_len[me] != recv_len[me]) return EZKL_ERR_INVALID;
    if (send_len[me])
        EZ_HIP(hipMemcpyAsync((char*)recv_dev + recv_off[me], (const char*)send_dev + send_off[me], send_len[me], hipMemcpyDeviceToDevice, g_comm.st));
    EZ_RCCL(g_rccl.GroupStart());
    for (int p = 0; p < g_comm.world; p++) {
        if (p == me) continue;
        if (send_len[p]) EZ_RCCL(g_rccl.Send((const char*)send_dev + send_off[p], send_len[p], ncclUint8, p, g_comm.comm, g_comm.st));
        if (recv_len[p]) EZ_RCCL(g_rccl.Recv((char*)recv_dev + recv_off[p], recv_len[p], ncclUint8, p, g_comm.comm, g_comm.st));
    }
    EZ_RCCL(g_rccl.GroupEnd());
    EZ_WAIT("all-to-all (device)");
    return EZKL_OK;
}

// The communicator tries itself out: every collective shape the prover uses, once, with contents that can be checked -- run by every
// rank right after ezkl_hip_comm_init (which calls it unless EZKL_COMM_SELFTEST=0), so that a job learns in its first second, with a
// message, what it would otherwise learn as a hang or a wrong proof minutes in:
//   1. all_gather (device) of the rank ids: slice r must hold r on every rank;
//   2. ONE packed all-to-all of odd-sized segments (17 + 13 r + 7 p bytes from r to p, two segments each, byte pattern f(r, p, i)) --
//      sizes that are no multiple of 16 and differ per pair: the byte path of the pack kernel and the per-peer totals;
//   3. fold_points: every rank contributes (r + 1) G as an affine point; the fold must be (world (world + 1) / 2) G -- the group law on the
//      host after an all_gather of 64-byte partials, i.e. the "RCCL reduce of the bucket sums" of every sharded commitment.
// Everything under the watchdog.  EZKL_OK, or the failing step on stderr and EZKL_ERR_INVALID (wrong data) / EZKL_ERR_TIMEOUT / EZKL_ERR_HIP.
int ezkl_hip_comm_selftest(void) {
    EZ_CTX(c);
    if (g_comm.broken) return EZKL_ERR_TIMEOUT;     // a collective timed out earlier: nothing runs behind it
    if (!g_comm.comm) return EZKL_ERR_INVALID;
    const int world = g_comm.world, me = g_comm.rank;
    auto fail = [&](const char* what) {
        fprintf(stderr, "[ezkl_hip] rank %d of %d: communicator self-test FAILED at: %s\n", me, world, what);
        return EZKL_ERR_INVALID;
    };
    // 1. all_gather of rank ids
    {
        std::vector<uint32_t> ids((size_t)world * 4, 0xffffffffu);
        for (int i = 0; i < 4; i++) ids[(size_t)me * 4 + i] = (uint32_t)me;
        uint32_t* d = nullptr;
        EZ_HIP(hipMalloc((void**)&d, ids.size() * 4));
        EZ_HIP(hipMemcpy(d, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
        int rc = ezkl_hip_comm_allgather_dev(d, ids.size() * 4);
        if (!rc) EZ_HIP(hipMemcpy(ids.data(), d, ids.size() * 4, hipMemcpyDeviceToHost));
        (void)hipFree(d);
        if (rc) return rc;
        for (int r = 0; r < world; r++)
            for (int i = 0; i < 4; i++)
                if (ids[(size_t)r * 4 + i] != (uint32_t)r) return fail("all_gather of rank ids");
    }
    // 2. packed all-to-all of odd-sized segments
    {
        auto seg_len = [](int from, int to) { return (size_t)(17 + 13 * from + 7 * to); };
        auto pat = [](int from, int to, size_t i) { return (uint8_t)(31 * from + 17 * to + 7 * i + 3); };
        size_t tot_s = 0, tot_r = 0;
        for (int p = 0; p < world; p++) { tot_s += seg_len(me, p); tot_r += seg_len(p, me); }
        std::vector<uint8_t> hs(tot_s), hr(tot_r, 0);
        char *ds = nullptr, *dr = nullptr;
        EZ_HIP(hipMalloc((void**)&ds, tot_s));
        EZ_HIP(hipMalloc((void**)&dr, tot_r));
        std::vector<ezkl_comm_seg_t> sends, recvs;
        size_t os = 0, orr = 0;
        for (int p = 0; p < world; p++) {
            const size_t ls = seg_len(me, p), lr = seg_len(p, me);
            for (size_t i = 0; i < ls; i++) hs[os + i] = pat(me, p, i);
            sends.push_back({p, ds + os, 5});                    // two segments per peer: the stream, not the segment, is the unit
            sends.push_back({p, ds + os + 5, ls - 5});
            recvs.push_back({p, dr + orr, lr - 9});              // ... cut elsewhere on the receiving side
            recvs.push_back({p, dr + orr + (lr - 9), 9});
            os += ls; orr += lr;
        }
        EZ_HIP(hipMemcpy(ds, hs.data(), tot_s, hipMemcpyHostToDevice));
        EZ_HIP(hipMemset(dr, 0, tot_r));
        int rc = ezkl_hip_comm_alltoallv_dev(sends.data(), sends.size(), recvs.data(), recvs.size());
        if (!rc) EZ_HIP(hipMemcpy(hr.data(), dr, tot_r, hipMemcpyDeviceToHost));
        (void)hipFree(ds); (void)hipFree(dr);
        if (rc) return rc;
        orr = 0;
        for (int p = 0; p < world; p++) {
            const size_t lr = seg_len(p, me);
            for (size_t i = 0; i < lr; i++)
                if (hr[orr + i] != pat(p, me, i)) return fail("packed all-to-all of odd-sized segments");
            orr += lr;
        }
    }
    // 3. fold of partial points
    {
        uint8_t G[64], mine[64], want[64], zero[64];
        memset(zero, 0, 64);
        static const uint64_t g_xy[8] = {0xd35d438dc58f0d9dull, 0x0a78eb28f5c70b3dull, 0x666ea36f7879462cull, 0x0e0a77c19a07df2full,      // 1 * 2^256 mod q
                                         0xa6ba871b8b1e1b3aull, 0x14f1d651eb8e167bull, 0xccdd46def0f28c58ull, 0x1c14ef83340fbe5eull};     // 2 * 2^256 mod q
        memcpy(G, g_xy, 64);                                     // the generator (1, 2), Montgomery form
        auto mul_small = [&](unsigned m, uint8_t* out) {          // m G by repeated addition (m <= a few dozen)
            memcpy(out, zero, 64);
            for (unsigned i = 0; i < m; i++) g1_add_affine_host(out, G, out);
        };
        mul_small((unsigned)me + 1, mine);
        mul_small((unsigned)(world * (world + 1) / 2), want);
        int rc = ezkl_hip_comm_fold_points(mine, 1);
        if (rc) return rc;
        if (memcmp(mine, want, 64)) return fail("fold_points (all_gather of 64-byte partials + group law)");
    }
    return EZKL_OK;
}

}  // extern "C"
