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
};
static RcclApi g_rccl;
struct Comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    hipStream_t st = nullptr;
    void* stage = nullptr;          // device staging for the partial points of one commit batch
    size_t stage_bytes = 0;
};
static Comm g_comm;

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

}  // namespace ezkl

using namespace ezkl;
extern "C" {

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
    return EZKL_OK;
}

int ezkl_hip_comm_info(int* world, int* rank) {
    if (world) *world = g_comm.comm ? g_comm.world : 0;
    if (rank) *rank = g_comm.comm ? g_comm.rank : 0;
    return EZKL_OK;
}

int ezkl_hip_comm_destroy(void) {
    EZ_CTX(c);
    if (!g_comm.comm) return EZKL_OK;
    EZ_HIP(hipStreamSynchronize(g_comm.st));
    EZ_RCCL(g_rccl.CommDestroy(g_comm.comm));
    if (g_comm.stage) (void)hipFree(g_comm.stage);
    (void)hipStreamDestroy(g_comm.st);
    g_comm = Comm();
    return EZKL_OK;
}

// In-place all_gather of a device buffer of world equal slices: rank r wrote slice r.  The library stream's earlier work on the
// buffer is ordered before the collective, and the call returns when the gathered data is in place.
int ezkl_hip_comm_allgather_dev(void* buf_dev, size_t total_bytes) {
    if (!buf_dev) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (!g_comm.comm || total_bytes % (size_t)g_comm.world) return EZKL_ERR_INVALID;
    const size_t slice = total_bytes / (size_t)g_comm.world;
    EZ_HIP(hipStreamSynchronize(c->stream));
    EZ_RCCL(g_rccl.AllGather((const char*)buf_dev + slice * (size_t)g_comm.rank, buf_dev, slice, ncclUint8, g_comm.comm, g_comm.st));
    EZ_HIP(hipStreamSynchronize(g_comm.st));
    return EZKL_OK;
}

// The fold of one commit batch: `count` 64-byte affine Montgomery partial sums (host) in, their sums over all ranks out (in place).
// all_gather of count * 64 B per rank + the group law on the host (host64.hpp): the RCCL "reduce" of the bucket sums.
int ezkl_hip_comm_fold_points(void* points_host, uint32_t count) {
    if (!points_host || count == 0) return count ? EZKL_ERR_INVALID : EZKL_OK;
    EZ_CTX(c);
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
    EZ_HIP(hipStreamSynchronize(g_comm.st));
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
    EZ_HIP(hipStreamSynchronize(g_comm.st));
    return EZKL_OK;
}

int ezkl_hip_comm_allgather_host(void* buf_host, size_t bytes) {
    if (!buf_host || bytes == 0 || bytes > ((size_t)1 << 24)) return EZKL_ERR_INVALID;
    EZ_CTX(c);
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
    EZ_HIP(hipStreamSynchronize(g_comm.st));
    return EZKL_OK;
}

// all-to-all with any number of segments per peer (see include/ezkl_hip.h): matching is by order per (sender, receiver) pair, which is
// how ncclSend / ncclRecv inside one group match; the segments to self are matched the same way and copied device-to-device.
int ezkl_hip_comm_alltoallv_dev(const ezkl_comm_seg_t* sends, size_t n_sends, const ezkl_comm_seg_t* recvs, size_t n_recvs) {
    if ((n_sends && !sends) || (n_recvs && !recvs)) return EZKL_ERR_INVALID;
    EZ_CTX(c);
    if (!g_comm.comm) return EZKL_ERR_INVALID;
    const int me = g_comm.rank;
    for (size_t i = 0; i < n_sends; i++)
        if (sends[i].peer < 0 || sends[i].peer >= g_comm.world || (!sends[i].ptr && sends[i].bytes)) return EZKL_ERR_INVALID;
    for (size_t i = 0; i < n_recvs; i++)
        if (recvs[i].peer < 0 || recvs[i].peer >= g_comm.world || (!recvs[i].ptr && recvs[i].bytes)) return EZKL_ERR_INVALID;
    EZ_HIP(hipStreamSynchronize(c->stream));          // what the library stream produced is what gets sent
    // EZKL_COMM_SELF_VIA_RCCL=1 (testing): the segments to self take the grouped ncclSend / ncclRecv path too, so that a one-GPU box
    // exercises RCCL's matching of MANY sends and receives per peer inside one group (tests/test_gpu_comm.py)
    const char* via = getenv("EZKL_COMM_SELF_VIA_RCCL");
    const int skip = via && *via == '1' ? -1 : me;    // the peer whose segments are plain copies
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
        if (sends[i].peer != skip && sends[i].bytes) EZ_RCCL(g_rccl.Send(sends[i].ptr, sends[i].bytes, ncclUint8, sends[i].peer, g_comm.comm, g_comm.st));
    for (size_t i = 0; i < n_recvs; i++)
        if (recvs[i].peer != skip && recvs[i].bytes) EZ_RCCL(g_rccl.Recv(recvs[i].ptr, recvs[i].bytes, ncclUint8, recvs[i].peer, g_comm.comm, g_comm.st));
    EZ_RCCL(g_rccl.GroupEnd());
    EZ_HIP(hipStreamSynchronize(g_comm.st));
    return EZKL_OK;
}

// all-to-all on device pointers: for every peer p, send_len[p] bytes at send_dev + send_off[p] go to p, recv_len[p] bytes from p land
// at recv_dev + recv_off[p].  All peers at once (grouped ncclSend / ncclRecv); the slice to self is a device-to-device copy.
int ezkl_hip_comm_alltoall_dev(const void* send_dev, const size_t* send_off, const size_t* send_len, void* recv_dev, const size_t* recv_off,
                               const size_t* recv_len) {
    if (!send_off || !send_len || !recv_off || !recv_len) return EZKL_ERR_INVALID;
    EZ_CTX(c);
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
    EZ_HIP(hipStreamSynchronize(g_comm.st));
    return EZKL_OK;
}

}  // extern "C"
