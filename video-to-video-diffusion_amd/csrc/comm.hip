// Depth-sharding collectives behind the C ABI: one process per GPU, RCCL over xGMI.
//
// A volume's depth D is cut into `world` slabs; every non-pointwise conv of the U-Net / VAE has k_D = 3, s_D = 1, so a
// slab needs one boundary slice of each depth neighbour (models/unet3d.py:56,96,204-207,218-221; models/vae.py:27,65-69,
// 86-90), GroupNorm needs (sum, sumsq) over the whole depth (unet3d.py:59,97,151,329), and TemporalAttention needs the
// depth sum (csrc/attention.hip).  These entry points put exactly those exchanges on the caller's HIP stream as RCCL
// calls -- ONE ncclGroup per sync point (neighbour send/recv of both boundary slices + the fp64 statistics all-reduce
// + an optional fp32 all-reduce travel together), so a sharded U-Net evaluation is ~64 sync points instead of one per
// tensor and statistic, and the whole step can be stream-captured into a hipGraph like the single-GPU step.
//
// RCCL is bound at run time (dlopen): the process usually already holds a librccl (PyTorch links one), and a second
// copy must not be loaded; a host without PyTorch gets /opt/rocm/lib/librccl.so.1.  Nothing here allocates device
// memory or synchronises the stream.
#include "ctsi_internal.h"
#include <dlfcn.h>
#include <string.h>

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess_ = 0 };
enum { ncclUint8_ = 1, ncclFloat32_ = 7, ncclFloat64_ = 8 };
enum { ncclSum_ = 0 };

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl g_rccl;

bool rccl_load() {
    if (g_rccl.ok) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    // CTSI_RCCL_LIB names the collective library explicitly: another RCCL build, or the recording stub of
    // tests/test_rccl_stub.py (which checks peers, byte counts and grouping of every sync point without a GPU)
    const char* forced = getenv("CTSI_RCCL_LIB");
    if (forced && *forced) {
        h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            ctsi_set_error("CTSI_RCCL_LIB=%s cannot be loaded: %s", forced, dlerror());
            return false;
        }
    }
    for (int i = 0; !h && i < 3; ++i)   // a copy already in the process (PyTorch's) first
        h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        ctsi_set_error("RCCL is not available: %s", dlerror());
        return false;
    }
    g_rccl.handle = h;
#define CTSI_SYM(field, name)                                             \
    *(void**)(&g_rccl.field) = dlsym(h, name);                            \
    if (!g_rccl.field) {                                                  \
        ctsi_set_error("librccl does not export %s", name);               \
        return false;                                                     \
    }
    CTSI_SYM(GetUniqueId, "ncclGetUniqueId")
    CTSI_SYM(CommInitRank, "ncclCommInitRank")
    CTSI_SYM(CommDestroy, "ncclCommDestroy")
    CTSI_SYM(GroupStart, "ncclGroupStart")
    CTSI_SYM(GroupEnd, "ncclGroupEnd")
    CTSI_SYM(Send, "ncclSend")
    CTSI_SYM(Recv, "ncclRecv")
    CTSI_SYM(AllReduce, "ncclAllReduce")
    CTSI_SYM(AllGather, "ncclAllGather")
    CTSI_SYM(GetErrorString, "ncclGetErrorString")
#undef CTSI_SYM
    g_rccl.ok = true;
    return true;
}

}  // namespace

struct ctsi_comm {
    ncclComm_t nccl;   // null when world == 1
    int rank, world;
};

#define CTSI_NCCL(call)                                                                            \
    do {                                                                                           \
        const int r_ = (call);                                                                     \
        if (r_ != ncclSuccess_) {                                                                  \
            ctsi_set_error("%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
            return CTSI_ERR_HIP;                                                                   \
        }                                                                                          \
    } while (0)

// Inside ncclGroupStart .. ncclGroupEnd: a failed call must not leave the thread's RCCL group open (every later RCCL call
// of the thread -- torch.distributed's too -- would be queued and never issued: a hang instead of the reported error).
#define CTSI_NCCL_IN_GROUP(call)                                                                   \
    do {                                                                                           \
        const int r_ = (call);                                                                     \
        if (r_ != ncclSuccess_) {                                                                  \
            ctsi_set_error("%s failed: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?"); \
            g_rccl.GroupEnd();                                                                     \
            return CTSI_ERR_HIP;                                                                   \
        }                                                                                          \
    } while (0)

extern "C" int ctsi_comm_unique_id(void* id128) {
    CTSI_CHECK_ARG(id128, "ctsi_comm_unique_id: null argument");
    if (!rccl_load()) return CTSI_ERR_UNSUPPORTED;
    ncclUniqueId id;
    CTSI_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return CTSI_OK;
}

extern "C" int ctsi_comm_init(ctsi_comm** out, const void* id128, int rank, int world) {
    CTSI_CHECK_ARG(out, "ctsi_comm_init: null argument");
    *out = nullptr;   // stays null on every failure path
    CTSI_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "ctsi_comm_init: bad rank %d / world %d", rank, world);
    ctsi_comm* c = (ctsi_comm*)calloc(1, sizeof(ctsi_comm));
    if (!c) {
        ctsi_set_error("ctsi_comm_init: out of host memory");
        return CTSI_ERR_INVALID;
    }
    c->rank = rank;
    c->world = world;
    if (world > 1 || id128) {   // (world 1 with an id: a one-rank RCCL communicator -- exercises the binding on one GPU)
        if (!id128 || !rccl_load()) {
            if (!id128) ctsi_set_error("ctsi_comm_init: a unique id is required for world > 1");
            free(c);
            return CTSI_ERR_UNSUPPORTED;
        }
        ncclUniqueId id;
        memcpy(id.internal, id128, 128);
        const int r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
        if (r != ncclSuccess_) {
            ctsi_set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
            free(c);
            return CTSI_ERR_HIP;
        }
    }
    *out = c;
    return CTSI_OK;
}

extern "C" void ctsi_comm_destroy(ctsi_comm* c) {
    if (!c) return;
    if (c->nccl && g_rccl.ok) g_rccl.CommDestroy(c->nccl);
    free(c);
}

extern "C" int ctsi_comm_rank(const ctsi_comm* c) { return c ? c->rank : -1; }
extern "C" int ctsi_comm_world(const ctsi_comm* c) { return c ? c->world : 0; }

// lo_halo <- rank-1's hi_own, hi_halo <- rank+1's lo_own (`bytes` each; zeros at the volume's two ends = the conv's zero
// padding along depth), and -- in the same RCCL group, i.e. the same sync point -- sums[nsums] (fp64) and f32[nf32]
// summed over all ranks in place.  Any of the three parts may be absent (null pointer / zero count).
extern "C" int ctsi_halo_exchange_reduce(ctsi_comm* c, const void* lo_own, const void* hi_own, void* lo_halo, void* hi_halo,
                                         size_t bytes, double* sums, int nsums, float* f32, long long nf32, void* stream) {
    CTSI_CHECK_ARG(c, "ctsi_halo_exchange_reduce: null communicator");
    CTSI_CHECK_ARG(bytes == 0 || (lo_own && hi_own && lo_halo && hi_halo), "ctsi_halo_exchange_reduce: null slice pointer");
    hipStream_t st = (hipStream_t)stream;
    if (bytes) {
        if (c->rank == 0) CTSI_HIP(hipMemsetAsync(lo_halo, 0, bytes, st));
        if (c->rank == c->world - 1) CTSI_HIP(hipMemsetAsync(hi_halo, 0, bytes, st));
    }
    if (!c->nccl) return CTSI_OK;   // single rank without a communicator: nothing to exchange
    CTSI_NCCL(g_rccl.GroupStart());
    if (bytes && c->rank > 0) {
        CTSI_NCCL_IN_GROUP(g_rccl.Send(lo_own, bytes, ncclUint8_, c->rank - 1, c->nccl, st));
        CTSI_NCCL_IN_GROUP(g_rccl.Recv(lo_halo, bytes, ncclUint8_, c->rank - 1, c->nccl, st));
    }
    if (bytes && c->rank < c->world - 1) {
        CTSI_NCCL_IN_GROUP(g_rccl.Send(hi_own, bytes, ncclUint8_, c->rank + 1, c->nccl, st));
        CTSI_NCCL_IN_GROUP(g_rccl.Recv(hi_halo, bytes, ncclUint8_, c->rank + 1, c->nccl, st));
    }
    if (sums && nsums > 0) CTSI_NCCL_IN_GROUP(g_rccl.AllReduce(sums, sums, (size_t)nsums, ncclFloat64_, ncclSum_, c->nccl, st));
    if (f32 && nf32 > 0) CTSI_NCCL_IN_GROUP(g_rccl.AllReduce(f32, f32, (size_t)nf32, ncclFloat32_, ncclSum_, c->nccl, st));
    CTSI_NCCL(g_rccl.GroupEnd());
    return CTSI_OK;
}

extern "C" int ctsi_halo_exchange(ctsi_comm* c, const void* lo_own, const void* hi_own, void* lo_halo, void* hi_halo,
                                  size_t bytes, void* stream) {
    return ctsi_halo_exchange_reduce(c, lo_own, hi_own, lo_halo, hi_halo, bytes, nullptr, 0, nullptr, 0, stream);
}

// GroupNorm statistics (fp64 (sum, sumsq) per (sample, group)) and, optionally, an fp32 buffer (the TemporalAttention
// depth sum) summed over all ranks in place: one sync point.
extern "C" int ctsi_gn_allreduce(ctsi_comm* c, double* sums, int nsums, float* f32, long long nf32, void* stream) {
    return ctsi_halo_exchange_reduce(c, nullptr, nullptr, nullptr, nullptr, 0, sums, nsums, f32, nf32, stream);
}

// every rank contributes `bytes` from `send`; recv holds world x bytes in rank order (result gather along depth)
extern "C" int ctsi_comm_allgather(ctsi_comm* c, const void* send, void* recv, size_t bytes, void* stream) {
    CTSI_CHECK_ARG(c && send && recv, "ctsi_comm_allgather: null argument");
    hipStream_t st = (hipStream_t)stream;
    if (!c->nccl) {
        if (send != recv) CTSI_HIP(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, st));
        return CTSI_OK;
    }
    CTSI_NCCL(g_rccl.AllGather(send, recv, bytes, ncclUint8_, c->nccl, st));
    return CTSI_OK;
}
