// Library plumbing: error text, device probe, hipGraph capture helpers, HIP event timers.
#include "ctsi_internal.h"
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void ctsi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Timing-only ablation switches: see ctsi_internal.h.  The release library honours bit 4096 (in-kernel timeline) only.
int ctsi_debug_flags() {
    const char* f = getenv("CTSI_DEBUG_FLAGS");   // read per call: tools/ab_variants.py alternates values in one process
    return f ? (atoi(f) & CTSI_DBG_MASK) : 0;
}
int ctsi_debug_ksteps(int n) {
#ifdef CTSI_ABLATE
    const char* k = getenv("CTSI_DEBUG_KSTEPS");
    if (k && atoi(k) < n) return atoi(k);
#endif
    return n;
}
// 1 when the library was built with -DCTSI_ABLATE (libctsi_ablate.so): bench.py refuses to time such a build
extern "C" int ctsi_ablation_build(void) {
#ifdef CTSI_ABLATE
    return 1;
#else
    return 0;
#endif
}

extern "C" int ctsi_version(void) { return 100; }
extern "C" const char* ctsi_last_error(void) { return g_err; }

extern "C" int ctsi_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n > 0 ? 1 : 0;
}

// ---- hipGraph: one captured graph per denoising step -------------------------------------------------
struct ctsi_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

extern "C" int ctsi_graph_begin_capture(void* stream) {
    CTSI_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return CTSI_OK;
}

extern "C" int ctsi_graph_end_capture(void* stream, ctsi_graph** out) {
    CTSI_CHECK_ARG(out, "ctsi_graph_end_capture: null argument");
    hipGraph_t g = nullptr;
    CTSI_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        hipGraphDestroy(g);
        ctsi_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return CTSI_ERR_HIP;
    }
    ctsi_graph* cg = (ctsi_graph*)calloc(1, sizeof(ctsi_graph));
    cg->graph = g;
    cg->exec = ex;
    *out = cg;
    return CTSI_OK;
}

extern "C" int ctsi_graph_launch(ctsi_graph* g, void* stream) {
    CTSI_CHECK_ARG(g, "ctsi_graph_launch: null graph");
    CTSI_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return CTSI_OK;
}

extern "C" void ctsi_graph_destroy(ctsi_graph* g) {
    if (!g) return;
    hipGraphExecDestroy(g->exec);
    hipGraphDestroy(g->graph);
    free(g);
}

// ---- events ---------------------------------------------------------------------------------------------
struct ctsi_event {
    hipEvent_t ev;
};

extern "C" int ctsi_event_create(ctsi_event** out) {
    CTSI_CHECK_ARG(out, "ctsi_event_create: null argument");
    hipEvent_t e;
    CTSI_HIP(hipEventCreate(&e));
    ctsi_event* ce = (ctsi_event*)calloc(1, sizeof(ctsi_event));
    ce->ev = e;
    *out = ce;
    return CTSI_OK;
}
extern "C" int ctsi_event_record(ctsi_event* ev, void* stream) {
    CTSI_CHECK_ARG(ev, "ctsi_event_record: null event");
    CTSI_HIP(hipEventRecord(ev->ev, (hipStream_t)stream));
    return CTSI_OK;
}
// `stream` waits (on the device) for everything recorded into `ev`: cross-stream ordering without a host sync; inside a stream
// capture it forks / joins the capture (the depth-sharded programs run the halo transfer on a second stream this way)
extern "C" int ctsi_stream_wait_event(void* stream, ctsi_event* ev) {
    CTSI_CHECK_ARG(ev, "ctsi_stream_wait_event: null event");
    CTSI_HIP(hipStreamWaitEvent((hipStream_t)stream, ev->ev, 0));
    return CTSI_OK;
}
extern "C" int ctsi_event_elapsed_ms(ctsi_event* a, ctsi_event* b, float* ms) {
    CTSI_CHECK_ARG(a && b && ms, "ctsi_event_elapsed_ms: null argument");
    CTSI_HIP(hipEventSynchronize(b->ev));
    CTSI_HIP(hipEventElapsedTime(ms, a->ev, b->ev));
    return CTSI_OK;
}
extern "C" void ctsi_event_destroy(ctsi_event* ev) {
    if (!ev) return;
    hipEventDestroy(ev->ev);
    free(ev);
}

extern "C" int ctsi_memset_async(void* ptr, int value, size_t bytes, void* stream) {
    CTSI_CHECK_ARG(ptr || bytes == 0, "ctsi_memset_async: null pointer");
    if (bytes == 0) return CTSI_OK;
    CTSI_HIP(hipMemsetAsync(ptr, value, bytes, (hipStream_t)stream));
    return CTSI_OK;
}
