/* Recording stand-in for librccl (test infrastructure; loaded through CTSI_RCCL_LIB, and LD_PRELOADed so that the two HIP
 * calls csrc/comm.hip makes -- hipMemsetAsync / hipMemcpyAsync -- are recorded too instead of reaching a GPU).  Every
 * call appends one line to the file named by RCCL_STUB_LOG; nothing is dereferenced, so the test passes fake device
 * pointers.  Used by tests/test_rccl_stub.py to check, without any GPU, what each rank of a depth-sharded run would ask
 * RCCL to do: peers r-1 / r+1 only, slice byte counts, one group per sync point, memsets only at the volume ends. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct stub_comm { int rank, world; } stub_comm;
static int g_depth = 0, g_fail_send = -1, g_sends = 0;

static void logf_(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
#include <stdarg.h>
static void logf_(const char* fmt, ...) {
    const char* path = getenv("RCCL_STUB_LOG");
    if (!path) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    va_list ap;
    va_start(ap, fmt);
    vfprintf(f, fmt, ap);
    va_end(ap);
    fputc('\n', f);
    fclose(f);
}

int ncclGetUniqueId(ncclUniqueId* id) { memset(id, 0x5a, sizeof(*id)); logf_("uid"); return 0; }
int ncclCommInitRank(stub_comm** c, int world, ncclUniqueId id, int rank) {
    (void)id;
    *c = (stub_comm*)malloc(sizeof(stub_comm));
    (*c)->rank = rank;
    (*c)->world = world;
    const char* fs = getenv("RCCL_STUB_FAIL_SEND");   /* the n-th ncclSend fails (error-path test) */
    g_fail_send = fs ? atoi(fs) : -1;
    logf_("init rank=%d world=%d", rank, world);
    return 0;
}
int ncclCommDestroy(stub_comm* c) { logf_("destroy"); free(c); return 0; }
int ncclGroupStart(void) { logf_("group_start depth=%d", g_depth); ++g_depth; return 0; }
int ncclGroupEnd(void) { --g_depth; logf_("group_end depth=%d", g_depth); return 0; }
int ncclSend(const void* p, size_t count, int dtype, int peer, stub_comm* c, void* st) {
    (void)c;
    if (g_sends++ == g_fail_send) { logf_("send FAIL"); return 1; }
    logf_("send ptr=%p count=%zu dtype=%d peer=%d stream=%p grouped=%d", p, count, dtype, peer, st, g_depth > 0);
    return 0;
}
int ncclRecv(void* p, size_t count, int dtype, int peer, stub_comm* c, void* st) {
    (void)c;
    logf_("recv ptr=%p count=%zu dtype=%d peer=%d stream=%p grouped=%d", p, count, dtype, peer, st, g_depth > 0);
    return 0;
}
int ncclAllReduce(const void* s, void* r, size_t count, int dtype, int op, stub_comm* c, void* st) {
    (void)c;
    logf_("allreduce src=%p dst=%p count=%zu dtype=%d op=%d stream=%p grouped=%d", s, r, count, dtype, op, st, g_depth > 0);
    return 0;
}
int ncclAllGather(const void* s, void* r, size_t count, int dtype, stub_comm* c, void* st) {
    (void)c;
    logf_("allgather src=%p dst=%p count=%zu dtype=%d stream=%p grouped=%d", s, r, count, dtype, st, g_depth > 0);
    return 0;
}
const char* ncclGetErrorString(int r) { return r ? "stub failure" : "ok"; }

/* interposed HIP runtime calls of csrc/comm.hip (LD_PRELOAD) */
int hipMemsetAsync(void* p, int v, size_t n, void* st) { logf_("memset ptr=%p value=%d bytes=%zu stream=%p", p, v, n, st); return 0; }
int hipMemcpyAsync(void* d, const void* s, size_t n, int kind, void* st) {
    logf_("memcpy dst=%p src=%p bytes=%zu kind=%d stream=%p", d, s, n, kind, st);
    return 0;
}
