// Weight gradient of Conv3d / ConvTranspose3d on NDHWC bf16 tensors (gfx950 / MI355X).
//
//   dW[t][cr][cg] = sum over row voxels v of  R[v][cr] * G[map(v, t)][cg]
//
// R is the tensor that lives on the layer's *small/output-side* grid and G the one that is gathered
// with the tap shift (and the H/W stride):
//   Conv3d          : R = dL/dy (output grid),       G = layer input x,     dW -> weight[cout][cin][t]
//   ConvTranspose3d : R = layer input (small grid),  G = dL/dy (big grid),  dW -> weight[cin][cout][t]
// map(v, t) = (n, d - pd + kd, h*sh - ph + kh, w*sw - pw + kw); out-of-volume taps contribute zero.
//
// GEMM view: M = R channels, N = G channels, K = voxels.  Both operands sit in memory as
// [voxel][channel] (k-major), i.e. transposed with respect to what the MFMA operand registers want
// (8 consecutive k per lane).  The slabs are therefore staged as they are ([32 voxels][128 ch], 256-byte
// rows, LDS-DMA with the tap shift / zero fill done by the per-lane source address) and read with
// gfx950's transposing LDS read ds_read_b64_tr_b16: a 16-lane group fetches a 4-voxel x 16-channel block
// and every lane receives its channel's 4 consecutive voxels.  Row image: chunk16 ^= ((row&3)<<2)|((row>>2)&3)
// (conflict-free for the 32x32x16 transposed reads), applied on the DMA source side.
//
// Block = 4 waves, tile 128 (R ch) x 128 (G ch) for ONE tap over a slice of the voxel range (split-K);
// partial tiles go to a workspace [slice][tap][CRpad][CGpad] fp32 and a second kernel sums the slices in a
// fixed order into the PyTorch weight-gradient layout (deterministic, no atomics).
#include "conv3_halo_common.h"
#include <stdlib.h>

namespace wgk {
constexpr int BR = 128, BG = 128, BK = 32, NTH = 256;
constexpr int SLAB = BK * 256;     // 8 KB: [32 voxels][128 ch] bf16
constexpr int STAGE = 2 * SLAB;    // R slab, G slab
constexpr int LDS_BYTES = 2 * STAGE;
}  // namespace wgk

struct WgradParams {
    const bf16_t* R;
    const bf16_t* G;
    float* part;
    unsigned r_bytes, g_bytes;
    int CR, CRs, CG, CGs;
    int N, Dr, Hr, Wr, Dg, Hg, Wg;
    int KH, KW, sh, sw, pd, ph, pw;
    int T, tiles_r, tiles_g, S, ksteps, kps;
    int V;  // row voxels (< 2^31)
    int dbg;  // timing-only ablations (CTSI_DEBUG_FLAGS; wrong results): 1 no LDS-DMA after the prologue, 2 no validity masks, 64 no barrier
    unsigned mW, mH, mD;
    int shW, shH, shD;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

__device__ __forceinline__ unsigned wg_div(unsigned v, unsigned m, int sh) {
    return (unsigned)(((unsigned long long)v * m) >> sh);
}

__device__ __forceinline__ s16x4 wg_tr_read(unsigned lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(unsigned long long)lds_addr);
}

// SHAPE16 = true: the same tile on v_mfma_f32_16x16x32_bf16 (4 x 4 tiles of 16 x 16 per wave, one MFMA K = the whole 32-voxel
// K-step): the operand fragments are the same transposed reads (a 16-lane group = 4 voxels x 16 channels), addressed per
// k-group lane >> 4 instead of per k-half lane >> 5; same LDS bytes per MFMA FLOP.  (The chip holds a higher clock on this
// shape under load: MI355X_MICROARCH.md, DVFS give-back item 7.)
template <bool SHAPE16>
__global__ void __launch_bounds__(256)
conv_wgrad_kernel(const WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace wgk;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr3_t)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- block decode: tap fastest, then tiles, then slice.  Hardware deals consecutive block ids round-robin to the 8
    // XCDs; the remap gives every XCD a contiguous range of logical ids, so the 27 taps of one voxel slice (which read the
    // same R slab and neighbouring G rows) meet in ONE L2 instead of being fetched by eight ------------------------------------
    int bid = xcd_remap_h((int)blockIdx.x, (int)gridDim.x);
    const int t = bid % p.T;
    bid /= p.T;
    const int tg = bid % p.tiles_g;
    bid /= p.tiles_g;
    const int tr = bid % p.tiles_r;
    const int sl = bid / p.tiles_r;
    const int kd = t / (p.KH * p.KW);
    const int kh = (t / p.KW) % p.KH;
    const int kw = t % p.KW;
    const int k_begin = sl * p.kps;
    const int k_end = min(k_begin + p.kps, p.ksteps);

    const v4i_t rsR = h3_make_rsrc(p.R, p.r_bytes);
    const v4i_t rsG = h3_make_rsrc(p.G, p.g_bytes);

    // ---- per-lane staging constants: this lane fills LDS rows rl[j] = 8*wv + 4*j + lane/16, slot lane%16 ----
    unsigned r_col[2], g_col[2];
    bool r_ok[2], g_ok[2];
    int rl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        rl[j] = 8 * wv + 4 * j + (lane >> 4);
        const int f = ((rl[j] & 3) << 2) | ((rl[j] >> 2) & 3);
        const int ch = (lane & 15) ^ f;
        const int cr = tr * BR + ch * 8, cg = tg * BG + ch * 8;
        r_ok[j] = cr < p.CR;
        g_ok[j] = cg < p.CG;
        r_col[j] = (unsigned)cr * 2u;
        g_col[j] = (unsigned)cg * 2u;
    }

    auto issue = [&](int ks, int stage) {
        const unsigned base = lds0 + stage * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned v = (unsigned)ks * BK + rl[j];
            const bool vin = v < (unsigned)p.V;
            // row tensor: linear voxel index
            const unsigned roff = (vin && r_ok[j]) ? v * (unsigned)(p.CRs * 2) + r_col[j] : 0xffffffffu;
            h3_dma16(rsR, base + (8 * wv + 4 * j) * 256, roff, 0);
            // gathered tensor: decode (n, d, h, w), apply the tap
            const unsigned q1 = wg_div(v, p.mW, p.shW);
            const int w = (int)(v - q1 * p.Wr);
            const unsigned q2 = wg_div(q1, p.mH, p.shH);
            const int h = (int)(q1 - q2 * p.Hr);
            const unsigned nn = wg_div(q2, p.mD, p.shD);
            const int d = (int)(q2 - nn * p.Dr);
            const int gd = d - p.pd + kd, gh = h * p.sh - p.ph + kh, gw = w * p.sw - p.pw + kw;
            const bool ok = vin && g_ok[j] && (unsigned)gd < (unsigned)p.Dg && (unsigned)gh < (unsigned)p.Hg &&
                            (unsigned)gw < (unsigned)p.Wg;
            const unsigned gvox = ((nn * p.Dg + gd) * p.Hg + gh) * p.Wg + gw;
            const unsigned goff = ok ? gvox * (unsigned)(p.CGs * 2) + g_col[j] : 0xffffffffu;
            h3_dma16(rsG, base + SLAB + (8 * wv + 4 * j) * 256, goff, 0);
        }
    };

    // ---- per-lane transposed-read offsets (bytes inside a slab, k-substep 0) ----------------------------------
    // operand tile with channel base 32*ct: lane reads rows 8*(lane>>5) + 4e + q (q = (lane&15)>>2),
    // 16-byte chunk 4*ct + 2*((lane>>4)&1) + (p>>1) (p = lane&3), + 8*(p&1) bytes.
    const int wr = wv >> 1, wgc = wv & 1;
    const int q = (lane & 15) >> 2, pp = lane & 3;
    constexpr int NT = SHAPE16 ? 4 : 2;                      // operand tiles per wave and side
    unsigned a_off[NT][2], b_off[NT][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        // 32x32x16: rows 8 (lane >> 5) + 4 e + q of a 16-row sub-step, channel half (lane >> 4) & 1 of the 32-channel tile;
        // 16x16x32: rows 8 (lane >> 4) + 4 e + q of the 32-row step, the tile's 16 channels
        const int row = SHAPE16 ? 8 * (lane >> 4) + 4 * e + q : 8 * (lane >> 5) + 4 * e + q;
        const int f = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int cha = SHAPE16 ? 2 * (4 * wr + i) + (pp >> 1) : 4 * (2 * wr + i) + 2 * ((lane >> 4) & 1) + (pp >> 1);
            const int chb = SHAPE16 ? 2 * (4 * wgc + i) + (pp >> 1) : 4 * (2 * wgc + i) + 2 * ((lane >> 4) & 1) + (pp >> 1);
            a_off[i][e] = (unsigned)(256 * row + 16 * (cha ^ f) + 8 * (pp & 1));
            b_off[i][e] = (unsigned)(SLAB + 256 * row + 16 * (chb ^ f) + 8 * (pp & 1));
        }
    }

    f32x16 acc[2][2];
    f32x4 acc16[4][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.0f;

    if (k_begin < k_end) {
        issue(k_begin, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int ks = k_begin; ks < k_end; ++ks) {
            const int stage = (ks - k_begin) & 1;
            if (ks + 1 < k_end) issue(ks + 1, stage ^ 1);
            const unsigned sb = lds0 + stage * STAGE;
            if (SHAPE16) {
                bf16x8 af[4], bfr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const s16x4 a0 = wg_tr_read(sb + a_off[i][0]);
                    const s16x4 a1 = wg_tr_read(sb + a_off[i][1]);
                    const s16x4 b0 = wg_tr_read(sb + b_off[i][0]);
                    const s16x4 b1 = wg_tr_read(sb + b_off[i][1]);
                    af[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    bfr[i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc16[i][j], 0, 0, 0);
            } else
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const unsigned so = sb + sub * (16 * 256);   // rows 16*sub ...: the swizzle only sees row % 16
                bf16x8 af[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const s16x4 a0 = wg_tr_read(so + a_off[i][0]);
                    const s16x4 a1 = wg_tr_read(so + a_off[i][1]);
                    const s16x4 b0 = wg_tr_read(so + b_off[i][0]);
                    const s16x4 b1 = wg_tr_read(so + b_off[i][1]);
                    af[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                    bfr[i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---- partial tile -> workspace [slice][tap][CRpad][CGpad] ------------------------------------------------
    const int CRp = p.tiles_r * BR, CGp = p.tiles_g * BG;
    float* dst = p.part + ((size_t)sl * p.T + t) * (size_t)CRp * CGp;
    if (SHAPE16) {   // accumulator (i, j)[r]: R channel 64 wr + 16 i + 4 (lane >> 4) + r, G channel 64 wgc + 16 j + (lane & 15)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cg = tg * BG + 64 * wgc + 16 * j + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cr = tr * BR + 64 * wr + 16 * i + 4 * (lane >> 4) + r;
                    dst[(size_t)cr * CGp + cg] = acc16[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cg = tg * BG + 64 * wgc + 32 * j + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = tr * BR + 64 * wr + 32 * i + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                dst[(size_t)cr * CGp + cg] = acc[i][j][r];
            }
        }
#endif
}

// ---- stride-1 'same' convolutions: one block serves the TG taps of a kw row ------------------------------------------
// For stride 1 and equal R / G grids the gathered voxel of (row voxel v, tap (kd,kh,kw)) is simply
//   g = v + (kd-pd)*H*W + (kh-ph)*W + (kw-pw)                      (linear voxel indices)
// so the TG = KW taps of one (kd, kh) read the SAME [32 + TG - 1] consecutive G rows at row shifts 0..TG-1 and the same
// R slab: 16.5 KB of LDS fill feed 3 taps (190 flop/B instead of 64), and the G slab needs no per-row address decode.
// Pairs (v, tap) whose true neighbour lies outside the volume (the linear shift wrapped into the next row / slice /
// sample) are removed by zeroing the k-entries of the B fragment: validity depends on the voxel only, so one wave
// ballot per tap and K-step yields the masks (skipped when all 32 voxels are valid).
// Block = 8 waves (2 along R x 4 along G), tile 128 x 128 x TG taps, 96 accumulators per lane for TG = 3.
namespace wg3 {
constexpr int BR = 128, BG = 128, NTH = 512;
// <BK> voxels per K-step: 32 (4 LDS stages of 17 KB, one barrier per 12 MFMAs and wave) or 64 (3 stages of 33 KB, one barrier
// per 24 MFMAs: the step length of the forward halo-tile kernels)
template <int BK_>
struct Cfg {
    static constexpr int BK = BK_;
    static constexpr int RSLAB = BK * 256;           // [BK voxels][128 ch] bf16
    static constexpr int GROWS = BK + 4;             // + up to 3 halo rows, rounded to whole 4-row DMA groups
    static constexpr int GSLAB = GROWS * 256;
    static constexpr int STAGE = RSLAB + GSLAB;
    static constexpr int NS = BK == 64 ? 3 : 4;      // LDS stages: the DMA of step s + NS - 1 is issued while step s is on the matrix cores
    static constexpr int LDS_BYTES = NS * STAGE;
    static constexpr int GPW = BK / 32;              // 4-row DMA groups per wave, slab and step (8 waves x GPW x 4 rows = BK)
    static_assert((BK == 32 || BK == 64) && LDS_BYTES <= 160 * 1024, "unsupported K-step");
};
}  // namespace wg3

__device__ __forceinline__ unsigned wg_pair_mask(unsigned b2) {   // 2 validity bits -> dword mask over 2 bf16
    return ((b2 & 1u) ? 0x0000ffffu : 0u) | ((b2 & 2u) ? 0xffff0000u : 0u);
}

template <int N>
__device__ __forceinline__ void wg_wait_vm() {   // all but the wave's N youngest LDS-DMA pieces landed, all LDS reads returned
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

template <int TG, int BK>
__global__ void __launch_bounds__(512)
conv_wgrad_s1_kernel(const WgradParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C3 = wg3::Cfg<BK>;
    constexpr int BR = wg3::BR, BG = wg3::BG, RSLAB = C3::RSLAB, STAGE = C3::STAGE, NS = C3::NS, GPW = C3::GPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr3_t)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 2, wgc = wv & 3;

    // ---- block decode: tap group fastest, then tiles, then slice ---------------------------------------------------
    const int ngroups = p.T / TG;
    int bid = xcd_remap_h((int)blockIdx.x, (int)gridDim.x);
    const int grp = bid % ngroups;
    bid /= ngroups;
    const int tg = bid % p.tiles_g;
    bid /= p.tiles_g;
    const int tr = bid % p.tiles_r;
    const int sl = bid / p.tiles_r;
    const int kd = grp / p.KH, kh = grp % p.KH;
    const int k_begin = sl * p.kps;
    const int k_end = min(k_begin + p.kps, p.ksteps);
    // linear shift of the group's first tap (kw = 0)
    const int delta = ((kd - p.pd) * p.Hg + (kh - p.ph)) * p.Wg - p.pw;

    const v4i_t rsR = h3_make_rsrc(p.R, p.r_bytes);
    const v4i_t rsG = h3_make_rsrc(p.G, p.g_bytes);

    // ---- staging: BK/4 (R) + BK/4 + 1 (G) wave-DMAs of 4 rows each per K-step; wave w issues R and G groups w (+ 8), and
    //      wave 0 also the G halo group (rows BK .. BK + 3).  The row swizzle only sees row % 16, so a wave's groups share
    //      their column offsets ------------------------------------------------------------------------------------------
    const int lrow = lane >> 4, slot = lane & 15;
    auto row_col = [&](int row, int tile, int cmax, bool& ok) -> unsigned {
        const int f = ((row & 3) << 2) | ((row >> 2) & 3);
        const int ch = slot ^ f;
        const int c = tile * 128 + ch * 8;
        ok = c < cmax;
        return (unsigned)c * 2u;
    };
    bool r_ok, g_ok0, g_ok1;
    const unsigned r_col = row_col(4 * wv + lrow, tr, p.CR, r_ok);
    const unsigned g_col0 = row_col(4 * wv + lrow, tg, p.CG, g_ok0);
    const unsigned g_col1 = row_col(BK + lrow, tg, p.CG, g_ok1);
    const long long g_total = (long long)p.N * p.Dg * p.Hg * p.Wg;

    auto issue = [&](int ks, int stage) {
        const unsigned base = lds0 + stage * STAGE;
        const long long v0 = (long long)ks * BK;
#pragma unroll
        for (int k = 0; k < GPW; ++k) {
            const int row0 = 4 * (wv + 8 * k);
            {
                const long long v = v0 + row0 + lrow;
                const unsigned off = (r_ok && v < p.V) ? (unsigned)v * (unsigned)(p.CRs * 2) + r_col : 0xffffffffu;
                h3_dma16(rsR, base + row0 * 256, off, 0);
            }
            {
                const long long g = v0 + delta + row0 + lrow;
                const unsigned off = (g_ok0 && g >= 0 && g < g_total) ? (unsigned)g * (unsigned)(p.CGs * 2) + g_col0 : 0xffffffffu;
                h3_dma16(rsG, base + RSLAB + row0 * 256, off, 0);
            }
        }
        if (wv == 0 && TG > 1) {
            const long long g = v0 + delta + BK + lrow;
            const unsigned off = (g_ok1 && g >= 0 && g < g_total) ? (unsigned)g * (unsigned)(p.CGs * 2) + g_col1 : 0xffffffffu;
            h3_dma16(rsG, base + RSLAB + BK * 256, off, 0);
        }
    };

    // ---- transposed-read offsets ---------------------------------------------------------------------------------------
    const int q = (lane & 15) >> 2, pp = lane & 3;
    unsigned a_off[2][2], b_off[TG][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = 8 * (lane >> 5) + 4 * e + q;
        const int f = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cha = 4 * (2 * wr + i) + 2 * ((lane >> 4) & 1) + (pp >> 1);
            a_off[i][e] = (unsigned)(256 * row + 16 * (cha ^ f) + 8 * (pp & 1));
        }
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            const int rowb = row + t;    // the tap's row shift
            const int fb = ((rowb & 3) << 2) | ((rowb >> 2) & 3);
            const int chb = 4 * wgc + 2 * ((lane >> 4) & 1) + (pp >> 1);
            b_off[t][e] = (unsigned)(RSLAB + 256 * rowb + 16 * (chb ^ fb) + 8 * (pp & 1));
        }
    }

    f32x16 acc[TG][2];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.0f;

    // ---- main loop -----------------------------------------------------------------------------------------------------------
    // A K-step = NSUB sub-steps of 16 voxels (2 + TG MFMA operand fragments, 2 TG MFMAs each).  The fragments of a sub-step are
    // loaded during the previous one (two register sets), also across steps: the barrier that publishes step ks + 1 stands
    // before the LAST sub-step of step ks, whose MFMAs run beside the loads of step ks + 1's first sub-step.  By then every
    // wave has read all of stage(ks) into registers, so that stage is refilled with step ks + NS right behind the barrier:
    // NS - 1 steps of flight time.  SIMD partners (waves w, w + 4) issue their pieces one sub-step apart (one's MFMAs cover
    // the other's ~60-100 issue cycles per piece, conv3_halo_k32.hip).  Pieces per wave and step: 2 GPW (+ 1 for wave 0).
    constexpr int NSUB = BK / 16;
    constexpr int PPW = 2 * GPW;
    bf16x8 af[2][2], bfr[2][TG];
#define WG_LOAD(SET, SB, SUB)                                                                                  \
    {                                                                                                          \
        const unsigned so_ = (SB) + (SUB) * (16 * 256);                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                     \
            const s16x4 a0_ = wg_tr_read(so_ + a_off[i_][0]);                                                  \
            const s16x4 a1_ = wg_tr_read(so_ + a_off[i_][1]);                                                  \
            af[SET][i_] = __builtin_shufflevector(a0_, a1_, 0, 1, 2, 3, 4, 5, 6, 7);                           \
        }                                                                                                      \
        _Pragma("unroll") for (int t_ = 0; t_ < TG; ++t_) {                                                    \
            const s16x4 b0_ = wg_tr_read(so_ + b_off[t_][0]);                                                  \
            const s16x4 b1_ = wg_tr_read(so_ + b_off[t_][1]);                                                  \
            bfr[SET][t_] = __builtin_shufflevector(b0_, b1_, 0, 1, 2, 3, 4, 5, 6, 7);                          \
        }                                                                                                      \
    }
#define WG_MFMA(SET, SUB)                                                                                      \
    {                                                                                                          \
        _Pragma("unroll") for (int t_ = 0; t_ < TG; ++t_) {                                                    \
            bf16x8 b_ = bfr[SET][t_];                                                                          \
            /* wave-uniform: only sub-steps whose 16 voxels hold an invalid (voxel, tap) pair pay for the masking (with   \
               W = 48 a 64-voxel step always holds a row end, one 16-voxel sub-step in three does) */                  \
            if (((unsigned)(vmask[t_] >> (16 * (SUB))) & 0xffffu) != 0xffffu) {                                \
                const unsigned byte_ = (unsigned)(vmask[t_] >> (16 * (SUB) + 8 * (lane >> 5))) & 0xffu;        \
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));                                    \
                u32x4 m_;                                                                                      \
                m_.x = wg_pair_mask(byte_);                                                                    \
                m_.y = wg_pair_mask(byte_ >> 2);                                                               \
                m_.z = wg_pair_mask(byte_ >> 4);                                                               \
                m_.w = wg_pair_mask(byte_ >> 6);                                                               \
                u32x4 bu_ = __builtin_bit_cast(u32x4, b_);                                                     \
                bu_ &= m_;                                                                                     \
                b_ = __builtin_bit_cast(bf16x8, bu_);                                                          \
            }                                                                                                  \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                                   \
                acc[t_][i_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[SET][i_], b_, acc[t_][i_], 0, 0, 0);  \
        }                                                                                                      \
    }
    if (k_begin < k_end) {
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (k_begin + i < k_end) issue(k_begin + i, i);
        // step k_begin has landed (NS - 1 later ones may be in flight)
        {
            const int later = min(k_end - k_begin, NS) - 1;          // wave-uniform
            const bool extra = TG > 1 && wv == 0;
            if (later >= 3) {
                if (extra) wg_wait_vm<3 * (PPW + 1)>(); else wg_wait_vm<3 * PPW>();
            } else if (later == 2) {
                if (extra) wg_wait_vm<2 * (PPW + 1)>(); else wg_wait_vm<2 * PPW>();
            } else if (later == 1) {
                if (extra) wg_wait_vm<PPW + 1>(); else wg_wait_vm<PPW>();
            } else {
                wg_wait_vm<0>();
            }
        }
        __builtin_amdgcn_s_barrier();
        WG_LOAD(0, lds0, 0);
        int stage = 0;
        for (int ks = k_begin; ks < k_end; ++ks) {
            // validity of (voxel, tap) pairs of this step: lane l decodes voxel BK*ks + l
            unsigned long long vmask[TG];
            {
                const unsigned v = (unsigned)ks * BK + (BK == 64 ? lane : (lane & 31));
                const unsigned q1 = wg_div(v, p.mW, p.shW);
                const int w = (int)(v - q1 * p.Wr);
                const unsigned q2 = wg_div(q1, p.mH, p.shH);
                const int h = (int)(q1 - q2 * p.Hr);
                const unsigned nn = wg_div(q2, p.mD, p.shD);
                const int d = (int)(q2 - nn * p.Dr);
                const bool dh = v < (unsigned)p.V && (unsigned)(d - p.pd + kd) < (unsigned)p.Dg &&
                                (unsigned)(h - p.ph + kh) < (unsigned)p.Hg;
#pragma unroll
                for (int t = 0; t < TG; ++t) {
                    vmask[t] = __builtin_amdgcn_ballot_w64(dh && (unsigned)(w - p.pw + t) < (unsigned)p.Wg);
                    if (BK == 32) vmask[t] |= 0xffffffff00000000ull;
                    if (CTSI_DBG(p.dbg, 2)) vmask[t] = ~0ull;
                }
            }
            const unsigned sb = lds0 + stage * STAGE;
            const int nstage = stage + 1 == NS ? 0 : stage + 1;
            const bool refill = ks + NS < k_end;                       // step ks + NS goes into this step's stage
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                if (sub == NSUB - 1) {
                    // step ks + 1 must have landed: the steps issued after it (ks + 2 .. ks + NS - 1) may stay in flight
                    const int later = min(k_end - 1 - (ks + 1), NS - 2);   // wave-uniform, may be negative at the very end
                    const bool extra = TG > 1 && wv == 0;
                    if (NS == 4 && later >= 2) {
                        if (extra) wg_wait_vm<2 * (PPW + 1)>(); else wg_wait_vm<2 * PPW>();
                    } else if (later >= 1) {
                        if (extra) wg_wait_vm<PPW + 1>(); else wg_wait_vm<PPW>();
                    } else {
                        wg_wait_vm<0>();
                    }
                    if (!(CTSI_DBG(p.dbg, 64))) __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 1 < k_end) WG_LOAD(0, lds0 + nstage * STAGE, 0);   // (NSUB is even: the last sub-step computes on set 1)
                } else {
                    if ((sub & 1) == 0) { WG_LOAD(1, sb, sub + 1); } else { WG_LOAD(0, sb, sub + 1); }
                }
                if ((sub & 1) == 0) { WG_MFMA(0, sub); } else { WG_MFMA(1, sub); }
                __builtin_amdgcn_sched_barrier(0);
                // refill of this step's stage (free since the barrier above): waves 0-3 behind the last sub-step, waves 4-7 behind
                // the next step's first one
                if (sub == NSUB - 1 && refill && wv < 4 && !(CTSI_DBG(p.dbg, 1))) issue(ks + NS, stage);
                if (sub == 0 && wv >= 4 && ks > k_begin && ks - 1 + NS < k_end && !(CTSI_DBG(p.dbg, 1)))
                    issue(ks - 1 + NS, stage == 0 ? NS - 1 : stage - 1);
            }
            stage = nstage;
        }
    }
#undef WG_LOAD
#undef WG_MFMA

    const int CRp = p.tiles_r * BR, CGp = p.tiles_g * BG;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        float* dst = p.part + ((size_t)sl * p.T + grp * TG + t) * (size_t)CRp * CGp;
        const int cg = tg * BG + 32 * wgc + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = tr * BR + 64 * wr + 32 * i + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                dst[(size_t)cr * CGp + cg] = acc[t][i][r];
            }
    }
#endif
}

// dw[cr*sr + cg*sg + t*st] = scale * sum_s part[s][t][cr][cg]        (fixed summation order; one thread per output)
__global__ void __launch_bounds__(256)
conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int T, int CRp, int CGp,
                         int CR, int CG, long long sr, long long sg, long long st, float scale) {
    const long long total = (long long)T * CR * CG;
    const size_t slice = (size_t)T * CRp * CGp;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int cg = (int)(e % CG);
        const long long r = e / CG;
        const int cr = (int)(r % CR);
        const int t = (int)(r / CR);
        const float* src = part + ((size_t)t * CRp + cr) * CGp + cg;
        float s0 = 0.0f, s1 = 0.0f;
        int k = 0;
        for (; k + 1 < S; k += 2) {
            s0 += src[(size_t)k * slice];
            s1 += src[(size_t)(k + 1) * slice];
        }
        if (k < S) s0 += src[(size_t)k * slice];
        dw[cr * sr + cg * sg + t * st] = (s0 + s1) * scale;
    }
}

// The same reduction for the PyTorch weight layout (st == 1, sg == T: the T taps of one (cr, cg) pair are contiguous in dw),
// transposed through LDS: the kernel above reads coalesced but writes every 4-byte result into its own 108-byte-strided line
// and ran at ~1 TB/s (53.8 us per launch, 88 launches = 4.7 ms of a config-3 micro-step, profiles/r03_train_kernel_stats.csv).
// One block = one cr x 64 cg x all T taps: 16-byte loads down the slices (4 independent accumulators, fixed order: slices
// k = 0, 4, 8, .. / 1, 5, .. / 2, .. / 3, .. then (a0 + a1) + (a2 + a3)), the T x 64 sums staged in LDS, written out as one
// contiguous run of 64 T floats.
template <int TMAX>
__global__ void __launch_bounds__(256)
conv_wgrad_reduce_t_kernel(const float* __restrict__ part, float* __restrict__ dw, int S, int T, int CRp, int CGp, int CR,
                           int CG, long long sr, float scale) {
    __shared__ float s_t[TMAX][64 + 1];
    const int tid = threadIdx.x;
    const int cgt = blockIdx.x, cr = blockIdx.y;
    const int cg0 = cgt * 64;
    const size_t slice = (size_t)T * CRp * CGp;
    const int c4 = (tid & 15) * 4;                                     // 4 consecutive cg of this thread
    for (int t = tid >> 4; t < T; t += 16) {
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        if (cg0 + c4 < CGp) {                                          // (CGp is a multiple of 16; rows beyond CG hold zeros / are not written)
            const float* src = part + ((size_t)t * CRp + cr) * CGp + cg0 + c4;
            int k = 0;
            for (; k + 3 < S; k += 4) {
                const float4 v0 = *reinterpret_cast<const float4*>(src + (size_t)k * slice);
                const float4 v1 = *reinterpret_cast<const float4*>(src + (size_t)(k + 1) * slice);
                const float4 v2 = *reinterpret_cast<const float4*>(src + (size_t)(k + 2) * slice);
                const float4 v3 = *reinterpret_cast<const float4*>(src + (size_t)(k + 3) * slice);
                a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
                a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
                a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
                a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
            }
            for (int u = 0; k < S; ++k, ++u) {
                const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * slice);
                float4& a = u == 0 ? a0 : (u == 1 ? a1 : a2);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        }
        s_t[t][c4 + 0] = ((a0.x + a1.x) + (a2.x + a3.x)) * scale;
        s_t[t][c4 + 1] = ((a0.y + a1.y) + (a2.y + a3.y)) * scale;
        s_t[t][c4 + 2] = ((a0.z + a1.z) + (a2.z + a3.z)) * scale;
        s_t[t][c4 + 3] = ((a0.w + a1.w) + (a2.w + a3.w)) * scale;
    }
    __syncthreads();
    const int ncg = min(64, CG - cg0);
    float* dst = dw + cr * sr + (long long)cg0 * T;
    for (int o = tid; o < ncg * T; o += 256) {
        const int cgl = o / T, t = o - cgl * T;
        dst[o] = s_t[t][cgl];
    }
}

// reduce pass of ctsi_wgrad (both kernel families): picks the transposing form for the PyTorch layout
static void wg_reduce(const float* part, float* dw, int S, int T, int CRp, int CGp, int CR, int CG, long long sr, long long sg,
                      long long st, float scale, hipStream_t stream) {
    const char* e = getenv("CTSI_WGRAD_REDUCE_T");   // "0": the one-thread-per-output form (A/B timing, tests)
    if (st == 1 && sg == T && (CGp % 16) == 0 && T <= 48 && !(e && atoi(e) == 0)) {
        const dim3 grid((CG + 63) / 64, CR);
        if (T <= 27)
            hipLaunchKernelGGL(conv_wgrad_reduce_t_kernel<27>, grid, dim3(256), 0, stream, part, dw, S, T, CRp, CGp, CR, CG, sr, scale);
        else
            hipLaunchKernelGGL(conv_wgrad_reduce_t_kernel<48>, grid, dim3(256), 0, stream, part, dw, S, T, CRp, CGp, CR, CG, sr, scale);
        return;
    }
    const long long total = (long long)CR * CG * T;
    long long rb = (total + 255) / 256;
    if (rb > 8192) rb = 8192;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, stream, part, dw, S, T, CRp, CGp, CR, CG, sr, sg,
                       st, scale);
}

// ---- host side ---------------------------------------------------------------------------------------------
static void wg_magic(unsigned d, unsigned* m, int* sh) {
    int s = 0;
    while ((1ull << s) < d) ++s;
    *m = (unsigned)(((1ull << (31 + s)) / d) + 1ull);
    *sh = 31 + s;
}

struct WgradGeom {
    int T, tiles_r, tiles_g, S, ksteps, kps;
    int tg;   // taps per block: KW for stride-1 'same' layers (conv_wgrad_s1_kernel), 0 = one tap per block (gather kernel)
    int bk;   // voxels per K-step: 32 (gather kernel, conv_wgrad_s1_kernel<.., 32>) or 64
    long long V;
};

static int wg_geometry(const ctsi_wgrad_desc* d, WgradGeom* g) {
    CTSI_CHECK_ARG(d && d->kd > 0 && d->kh > 0 && d->kw > 0 && d->sh > 0 && d->sw > 0, "ctsi_wgrad: bad kernel/stride");
    CTSI_CHECK_ARG(d->n > 0 && d->dr > 0 && d->hr > 0 && d->wr > 0 && d->dg > 0 && d->hg > 0 && d->wg > 0,
                   "ctsi_wgrad: bad spatial sizes");
    CTSI_CHECK_ARG(d->cr > 0 && d->cg > 0 && (d->cr % 8) == 0 && (d->cg % 8) == 0 && (d->cr_stride % 8) == 0 &&
                       (d->cg_stride % 8) == 0 && d->cr <= d->cr_stride && d->cg <= d->cg_stride,
                   "ctsi_wgrad: channel counts must be multiples of 8 (cr=%d/%d cg=%d/%d)", d->cr, d->cr_stride, d->cg,
                   d->cg_stride);
    g->V = (long long)d->n * d->dr * d->hr * d->wr;
    const long long rb = g->V * d->cr_stride * 2, gb = (long long)d->n * d->dg * d->hg * d->wg * d->cg_stride * 2;
    CTSI_CHECK_ARG(g->V < (1ll << 31) - 64 && rb < 0xffffff00ll && gb < 0xffffff00ll,
                   "ctsi_wgrad: tensor larger than one 4 GiB buffer descriptor");
    g->T = d->kd * d->kh * d->kw;
    g->tiles_r = (d->cr + wgk::BR - 1) / wgk::BR;
    g->tiles_g = (d->cg + wgk::BG - 1) / wgk::BG;
    g->tg = 0;
    g->bk = wgk::BK;
    // Tap-sharing kernel (conv_wgrad_s1_kernel: one R slab + one halo'd G slab serve the 3 kw taps, 190 instead of 64 flop per
    // byte of LDS fill, 8-wave blocks, one per CU).  Measured against the one-tap kernel (profiles/r02_notes.md): K-steps of
    // 64 voxels + software-pipelined fragments + staggered DMA issue bring it from 10-20 % behind to par on the config-3
    // shapes (653 / 715 / 606 / 767 vs 656 / 747 / 691 / 766 TFLOP/s); masking only the 16-voxel sub-steps that hold an invalid
    // (voxel, tap) pair (the masks cost 12-18 %: knock-out) puts it ahead on single-tile layers (128->128 @4x48^3: 694 vs 674,
    // 128-channel slice of 384->128: 843 vs 768, 128->128 @48x128^2: 852-882 vs 766-790) and leaves it behind where several
    // channel tiles share the grid (256->256: 737 vs 757, 512->512: 619 vs 699): selected for one-tile layers of >= 300 k voxels;
    // CTSI_WGRAD_S1 = "0" | "64" overrides (off / on wherever it applies).
    // (Its 32-voxel K-step form, 10-20 % behind, is no longer instantiated.)
    const char* s1 = getenv("CTSI_WGRAD_S1");
    const bool s1_ok = d->sh == 1 && d->sw == 1 && d->dr == d->dg && d->hr == d->hg && d->wr == d->wg && (d->kw == 3 || d->kw == 1);
    if (s1_ok && s1 && atoi(s1) == 64) {
        g->tg = d->kw;
        g->bk = 64;
    } else if (s1_ok && !s1 && d->kw == 3 && g->tiles_r * g->tiles_g == 1 && g->V >= 300000) {
        g->tg = 3;
        g->bk = 64;
    }
    g->ksteps = (int)((g->V + g->bk - 1) / g->bk);
    const int combos = (g->tg ? g->T / g->tg : g->T) * g->tiles_r * g->tiles_g;
    const char* tgt = getenv("CTSI_WGRAD_TARGET");   // blocks to aim for (tuning aid)
    const int target = tgt && atoi(tgt) > 0 ? atoi(tgt) : (g->tg ? 768 : 2048);         // blocks: 256 CUs x 3 (8-wave blocks) / x 8 (4-wave blocks).  Measured: 1024
                                                   // blocks (half the partial-sum traffic) is 10 % slower end to end (tail balance)
    // 8-wave blocks run one per CU: the grid must not spill a few blocks into another round (774 blocks = 4 rounds for the
    // work of 3.02), so S rounds DOWN there; the 4-wave kernel (4-5 blocks per CU) keeps the round-1 rule
    int S = g->tg ? target / combos : (target + combos - 1) / combos;
    const int smax = (g->ksteps * (g->bk / 32) + 15) / 16;   // at least 512 voxels per slice
    if (S > smax) S = smax;
    if (S < 1) S = 1;
    g->kps = (g->ksteps + S - 1) / S;
    g->S = (g->ksteps + g->kps - 1) / g->kps;
    return CTSI_OK;
}

// conv_wgrad_halo.hip: the halo-tile weight-gradient kernel for 3x3x3 stride-1 'same' Conv3d layers
extern "C" int ctsi_wgrad_halo_plan(int n, int d, int h, int w, int cx, int cy, int* S, size_t* ws_bytes, int* CRp, int* CGp);
extern "C" int ctsi_wgrad_halo_launch(const void* X, const void* dY, void* part, int n, int d, int h, int w, int cx, int cx_stride,
                                      int cy, int cy_stride, void* stream);

// R = dL/dy (cr = cout), G = layer input (cg = cin) of a 3x3x3 / stride 1 / pad 1 Conv3d on equal grids: the halo-tile kernel
// (CTSI_WGRAD_HALO=0 keeps conv_wgrad_kernel / conv_wgrad_s1_kernel: A/B timing, tests)
static int wg_use_halo(const ctsi_wgrad_desc* d, int* S, size_t* ws, int* CRp, int* CGp) {
    const char* e = getenv("CTSI_WGRAD_HALO");
    if (e && atoi(e) == 0) return 0;
    if (!(d->kd == 3 && d->kh == 3 && d->kw == 3 && d->sh == 1 && d->sw == 1 && d->pd == 1 && d->ph == 1 && d->pw == 1 &&
          d->dr == d->dg && d->hr == d->hg && d->wr == d->wg))
        return 0;
    const long long xb = (long long)d->n * d->dg * d->hg * d->wg * d->cg_stride * 2, yb = (long long)d->n * d->dr * d->hr * d->wr * d->cr_stride * 2;
    if (xb >= 0x7fffffffll || yb >= 0x7fffffffll) return 0;
    return ctsi_wgrad_halo_plan(d->n, d->dg, d->hg, d->wg, d->cg, d->cr, S, ws, CRp, CGp);
}

extern "C" size_t ctsi_wgrad_workspace_bytes(const ctsi_wgrad_desc* d) {
    WgradGeom g;
    if (wg_geometry(d, &g) != CTSI_OK) return 0;
    {
        size_t ws = 0;
        if (wg_use_halo(d, nullptr, &ws, nullptr, nullptr)) return ws;
    }
    return (size_t)g.S * g.T * (size_t)(g.tiles_r * wgk::BR) * (g.tiles_g * wgk::BG) * sizeof(float);
}

extern "C" double ctsi_wgrad_flops(const ctsi_wgrad_desc* d) {
    WgradGeom g;
    if (wg_geometry(d, &g) != CTSI_OK) return 0.0;
    return 2.0 * (double)g.V * d->cr * d->cg * g.T;
}

extern "C" int ctsi_wgrad(const ctsi_wgrad_desc* d, const void* r, const void* gt, void* workspace, size_t workspace_bytes,
                          float* dw, long long stride_r, long long stride_g, long long stride_t, float scale, void* stream) {
    WgradGeom g;
    int rc = wg_geometry(d, &g);
    if (rc != CTSI_OK) return rc;
    CTSI_CHECK_ARG(r && gt && workspace && dw, "ctsi_wgrad: null pointer");
    {   // the kernel family (and with it the partial-sum layout) is chosen at launch from the tuning switches of the moment
        // (CTSI_WGRAD_HALO / _S1 / _TARGET): a workspace sized under other settings must fail here, not be written past its end
        const size_t need = ctsi_wgrad_workspace_bytes(d);
        CTSI_CHECK_ARG(workspace_bytes >= need, "ctsi_wgrad: workspace of %zu bytes, this launch needs %zu (sized under other "
                       "CTSI_WGRAD_* settings?)", workspace_bytes, need);
    }
    {
        int hS = 0, hCRp = 0, hCGp = 0;
        if (wg_use_halo(d, &hS, nullptr, &hCRp, &hCGp)) {
            rc = ctsi_wgrad_halo_launch(gt, r, workspace, d->n, d->dg, d->hg, d->wg, d->cg, d->cg_stride, d->cr, d->cr_stride, stream);
            if (rc != CTSI_OK) return rc;
            wg_reduce((const float*)workspace, dw, hS, 27, hCRp, hCGp, d->cr, d->cg, stride_r, stride_g, stride_t, scale,
                      (hipStream_t)stream);
            CTSI_LAUNCH_CHECK();
            return CTSI_OK;
        }
    }
    WgradParams p;
    p.R = (const bf16_t*)r;
    p.G = (const bf16_t*)gt;
    p.part = (float*)workspace;
    p.r_bytes = (unsigned)(g.V * d->cr_stride * 2);
    p.g_bytes = (unsigned)((long long)d->n * d->dg * d->hg * d->wg * d->cg_stride * 2);
    p.CR = d->cr; p.CRs = d->cr_stride; p.CG = d->cg; p.CGs = d->cg_stride;
    p.N = d->n; p.Dr = d->dr; p.Hr = d->hr; p.Wr = d->wr; p.Dg = d->dg; p.Hg = d->hg; p.Wg = d->wg;
    p.KH = d->kh; p.KW = d->kw; p.sh = d->sh; p.sw = d->sw; p.pd = d->pd; p.ph = d->ph; p.pw = d->pw;
    p.T = g.T; p.tiles_r = g.tiles_r; p.tiles_g = g.tiles_g; p.S = g.S; p.ksteps = g.ksteps; p.kps = g.kps;
    p.V = (int)g.V;
    p.dbg = ctsi_debug_flags();
    wg_magic((unsigned)d->wr, &p.mW, &p.shW);
    wg_magic((unsigned)d->hr, &p.mH, &p.shH);
    wg_magic((unsigned)d->dr, &p.mD, &p.shD);
    const long long blocks = (long long)(g.tg ? g.T / g.tg : g.T) * g.tiles_r * g.tiles_g * g.S;
    CTSI_CHECK_ARG(blocks < (1ll << 31), "ctsi_wgrad: grid too large");
    if (g.tg) {
        auto k = g.tg == 3 ? conv_wgrad_s1_kernel<3, 64> : conv_wgrad_s1_kernel<1, 64>;
        const int lds = wg3::Cfg<64>::LDS_BYTES;
        static CtsiPerDeviceOnce attr_once;
        if (attr_once.first()) {
            hipFuncSetAttribute((const void*)conv_wgrad_s1_kernel<3, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute((const void*)conv_wgrad_s1_kernel<1, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(wg3::NTH), lds, (hipStream_t)stream, p);
    } else
    {
        // 16x16x32 MFMAs by default: +1-3 % on every shape over the 32x32x16 form (interleaved runs, profiles/r02_notes.md):
        // equal cycles, higher clock under load.  CTSI_WGRAD_SHAPE16=0 selects the 32x32x16 form (A/B timing)
        const char* sh = getenv("CTSI_WGRAD_SHAPE16");
        if (!(sh && atoi(sh) == 0))
            hipLaunchKernelGGL(conv_wgrad_kernel<true>, dim3((unsigned)blocks), dim3(wgk::NTH), wgk::LDS_BYTES,
                               (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL(conv_wgrad_kernel<false>, dim3((unsigned)blocks), dim3(wgk::NTH), wgk::LDS_BYTES,
                               (hipStream_t)stream, p);
    }
    CTSI_LAUNCH_CHECK();
    wg_reduce((const float*)workspace, dw, g.S, g.T, g.tiles_r * wgk::BR, g.tiles_g * wgk::BG, d->cr, d->cg, stride_r, stride_g,
              stride_t, scale, (hipStream_t)stream);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
