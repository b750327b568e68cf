// Internal helpers shared by the HIP translation units of libctsi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ctsi.h"

typedef uint16_t bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---- error plumbing -------------------------------------------------------------------------
void ctsi_set_error(const char* fmt, ...);

#define CTSI_CHECK_ARG(cond, ...)            \
    do {                                     \
        if (!(cond)) {                       \
            ctsi_set_error(__VA_ARGS__);     \
            return CTSI_ERR_INVALID;         \
        }                                    \
    } while (0)

// ">64 KB of dynamic LDS" is a per-device function attribute: set it once per (call site, device) -- a process may drive
// several GPUs, one engine context each.  (Benign race: two threads may both set the same attribute.)
struct CtsiPerDeviceOnce {
    bool done[64] = {};
    bool first() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};

#define CTSI_HIP(call)                                                                  \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            ctsi_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),       \
                           __FILE__, __LINE__);                                         \
            return CTSI_ERR_HIP;                                                        \
        }                                                                               \
    } while (0)

#define CTSI_LAUNCH_CHECK()                                                             \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            ctsi_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_),   \
                           __FILE__, __LINE__);                                         \
            return CTSI_ERR_HIP;                                                        \
        }                                                                               \
    } while (0)

// ---- timing-only ablation switches ---------------------------------------------------------------
// CTSI_DEBUG_FLAGS bits that make a kernel skip work (DMAs, barriers, stores, the epilogue: wrong results by design) and
// CTSI_DEBUG_KSTEPS exist only in builds with -DCTSI_ABLATE (`make ablate` -> libctsi_ablate.so, which tools/ load through
// CTSI_LIB).  In the release library every such test is a compile-time 0 and the environment is not consulted for them;
// only bit 4096 (conv3_halo_k32_kernel's in-kernel timeline: correct results) survives.
#ifdef CTSI_ABLATE
#define CTSI_DBG(flags, bit) ((flags) & (bit))
#define CTSI_DBG_MASK (~0)
#else
#define CTSI_DBG(flags, bit) (((bit) == 4096) ? ((flags) & 4096) : 0)
#define CTSI_DBG_MASK 4096
#endif
// conv_gather_mfma_kernel keeps its four tests as RUN-TIME tests on a flags word that the host can only set in ablation builds
// (ctsi_debug_flags() masks it to bit 4096 in the release library, and this kernel has no 4096 use): the branches are never taken
// there, but compiling them out changed hipcc's register allocation of the 256 x 256 instantiation -- 380 -> 1616 bytes of
// scratch per lane in the small-Cin form, the VAE decoder's first conv 0.44 -> 1.54 ms (round 4, caught by the per-op table).
#define CTSI_DBG_RT(flags, bit) ((flags) & (bit))
int ctsi_debug_flags();        // CTSI_DEBUG_FLAGS & CTSI_DBG_MASK, read per call (runtime.hip)
int ctsi_debug_ksteps(int n);  // min(n, CTSI_DEBUG_KSTEPS) in ablation builds, n otherwise

// ---- device helpers ---------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even; a plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
// x * sigmoid(x) with the hardware exp2 / rcp (1 ulp each; results are rounded to bf16 afterwards)
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// the same on a channel pair (one bf16x2 dword): elementwise identical to silu_f / pack_bf16x2
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t silu2_f(f32x2_t x) {
    f32x2_t t = x * -1.4426950408889634f;
    t.x = __builtin_amdgcn_exp2f(t.x);
    t.y = __builtin_amdgcn_exp2f(t.y);
    t = 1.0f + t;
    t.x = __builtin_amdgcn_rcpf(t.x);
    t.y = __builtin_amdgcn_rcpf(t.y);
    return x * t;
}
__device__ __forceinline__ uint32_t pack_bf16x2_v(f32x2_t v) {
    const bf16x2_t b = __builtin_convertvector(v, bf16x2_t);
    return *reinterpret_cast<const uint32_t*>(&b);
}

__device__ __forceinline__ float nan_to_num_f(float v) {
    // torch.nan_to_num(v, nan=0.0, posinf=1.0, neginf=-1.0)
    if (v != v) return 0.0f;
    if (v == __builtin_inff()) return 1.0f;
    if (v == -__builtin_inff()) return -1.0f;
    return v;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- conv3_halo.hip (3x3x3 halo-tile kernel), driven through the conv plan in conv_mfma.hip ------------
struct Conv3HaloParams {
    const bf16_t* x1;
    const bf16_t* x2;
    const bf16_t* w;      // [chunk][27][cout_pad][32], rows pre-swizzled
    const float* bias;
    void* y;
    float* colsum;
    int C1, C2;
    int Di, Hi, Wi;       // input dims (Di includes depth halo slices when dshift = 1)
    int Do, Ho, Wo;       // output dims (== own input dims)
    int dshift;
    int tilesD, tilesH, tilesW, tps, mtiles, ntiles_n;
    int nchunks;          // (C1 + C2) / 32
    int Cout, CoutPad;
    int cout_stride, c_off;
    int n_major;          // block order inside an XCD: 1 = all m-tiles of one n-tile first (see h3_decode_tile)
    int dbg;              // ablation bits (compiled out of libctsi.so, see CTSI_DBG): 128 (k32) no loads of a unit's second B half; 1 no halo DMA after chunk 0, 2 no weight DMA after step 1, 4 no
                          // global stores, 8 no epilogue (1-8: wrong results, timing only); 16 / 32: DMA pieces issued right behind
                          // the step's barrier / at the end of the step instead of behind the first MFMA phase (correct results)
    // normalise-on-load (experiments/conv3_halo_m512.hip only; no kernel of libctsi.so reads these): the input is the RAW output of the previous conv; the kernel
    // applies y = silu?(x * gamma * rstd + (beta - mean * gamma * rstd)) + tbias to every halo-tile element in LDS right after
    // its DMA has landed, i.e. the GroupNorm + SiLU + time-bias pass between two convs never touches HBM.
    const double* nin_sums;   // [n][groups][2] fp64 (sum, sumsq) of the input; NULL = plain input
    const float* nin_gamma;
    const float* nin_beta;
    const float* nin_tbias;   // [rows][tb_stride] or NULL; row = (*nin_step_ptr) * n_total + sample
    const int* nin_step_ptr;
    int nin_tb_stride, nin_groups, nin_silu, nin_n_total;
    int nin_pad_lo, nin_pad_hi;   // depth-sharded input: the first / last halo slice is a volume end (reads as zero padding)
    float nin_eps;
    double nin_count;
    int ksplit;           // conv3_halo_k32_kernel<SK>: 2 = two blocks per (tile, n-tile), each half of the chunks
    float* sk_ws;         //   fp32 partial accumulators [tile][register][512]
    int* sk_sync;         //   [tile][2]: ticket, ready flag (zero before the first launch; the kernel resets them)
    int tr;               // conv3_halo_k32_kernel: ConvTranspose3d (3,4,4) / (1,2,2): Do/Ho/Wo are the OUTPUT dims, tiles walk the input grid
    int ds;               // conv3_halo_k32_kernel: strided Conv3d (3,4,4) / (1,2,2): tiles walk the OUTPUT grid, nchunks = 4 * Cin / 16
    int tile_order;       // conv3_halo_k32_kernel: 0 = (tD, tH, tW); 1 = (tH, tD, tW); 2 = 8 x 4 super-tiles inside a depth band (see its tile decode)
    int nt_store;         // conv3_halo_k32_kernel: non-temporal output stores (large outputs: keep the L2 for halos and weights)
    int dbg_epi_barrier;  // conv3_halo_k32_kernel, direct epilogue: keep the workgroup barrier between loop and epilogue (CTSI_CONV_EPI_BARRIER=1: A/B timing)
};

extern "C" int ctsi_conv3_halo_pack(const float* w, void* packed, int cout, int cout_pad, int cin, int cin_w,
                                    void* stream);
extern "C" int ctsi_conv3_halo_launch(const Conv3HaloParams* hp, int wide, void* stream);
extern "C" int ctsi_conv3_head_launch(const Conv3HaloParams* hp, int rows, int out_mode, int act, long long sn, long long sc,
                                      long long sd, long long sh, long long sw, void* stream);
extern "C" int ctsi_conv3_head2_supported(int cin, int cout);
extern "C" size_t ctsi_conv3_head2_weight_bytes(int cout);
extern "C" int ctsi_conv3_head2_pack(const float* w, void* packed, int cout, int cin, int cin_w, void* stream);
extern "C" int ctsi_conv3_head2_launch(const Conv3HaloParams* hp, int n, const void* packed, int out_mode, int act, long long sn,
                                       long long sc, long long sd, long long sh, long long sw, void* stream);
// ---- conv3_stem.hip (3x3x3 conv of a one-channel volume: the VAE encoder's first layer), driven through the conv plan -------
struct StemParams {
    const bf16_t* x;          // bf16 NDHWC, cx channels per voxel, channel 0 is the volume
    const bf16_t* w;          // packed [n-tile][cout tile j][lane][8]
    const float* bias;
    bf16_t* y;
    float* colsum;            // [2][tiles][cout_pad] or NULL
    int cx, D, H, W;
    int Cout, CoutPad, cout_stride, c_off;
    int tilesD, tilesH, tilesW, tps, mtiles;
};
extern "C" void ctsi_conv3_stem_tile(int* td, int* th, int* tw);
extern "C" size_t ctsi_conv3_stem_weight_bytes(int cout_pad);
extern "C" int ctsi_conv3_stem_pack(const float* w, void* packed, int cout, int cout_pad, int cin_w, void* stream);
extern "C" int ctsi_conv3_stem_launch(const StemParams* q, void* stream);

// ---- conv1_stream.hip (1x1x1 conv + fused GroupNorm tail, streaming), driven through the conv plan ---------
struct Conv1StreamParams {
    const bf16_t* x1;
    const bf16_t* x2;
    const bf16_t* w;          // [n-tile][k-step][cout tile t][lane][8 bf16]
    const float* bias;
    const bf16_t* h;          // tensor whose GroupNorm is added (may alias y), NULL: plain conv + bias
    bf16_t* y;
    const double* gn_sums;
    const float* gn_gamma;
    const float* gn_beta;
    int gn_groups, gn_silu;
    float gn_eps;
    double gn_count;
    int C1, C2, Cout, cout_stride, c_off;
    long long V;              // voxels per sample
    int tiles;                // 16-voxel tiles per sample
    int P;                    // blocks per (sample, n-tile)
};
extern "C" int ctsi_conv1_stream_nt(int c1, int c2, int cout);
extern "C" int ctsi_conv1_stream_pack(const float* w, void* packed, int cout, int cin, int cin_w, int nt, void* stream);
extern "C" int ctsi_conv1_stream_launch(Conv1StreamParams* q, int n, int nt, void* stream);
extern "C" size_t ctsi_conv3_halo_k32_weight_bytes(int cin, int cout_pad, int bn, int transposed);
extern "C" int ctsi_conv3_halo_k32_pack(const float* w, void* packed, int cout, int cout_pad, int cin, int cin_w, int bn,
                                        int transposed, int direct, void* stream);
extern "C" int ctsi_conv3_halo_k32_direct(int tile, int ksplit, int ds);   // 1: the form stores straight from the accumulators (cout-permuted image)
extern "C" size_t ctsi_conv3_halo_k32_splitk_bytes(int tiles);
extern "C" int ctsi_conv3_halo_k32_launch(const Conv3HaloParams* hp, int tile, int bn, void* stream);

