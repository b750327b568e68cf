// Gather-GEMM convolution for gfx950 (MI355X): Conv3d / ConvTranspose3d as an implicit GEMM
//   rows  M = output voxels of one (TD x TH x TW) tile,
//   cols  N = output channels,
//   K     = taps x input channels, walked as 64-wide K-steps,
// bf16 operands, fp32 accumulation on v_mfma_f32_32x32x16_bf16.
//
// Replaces (reference): nn.Conv3d / nn.ConvTranspose3d call sites models/unet3d.py:56,96,102,152,
// 153,204-207,218-221,257,331 and models/vae.py:27,45,65-68,86-89,134,137,161,188.
//
// No im2col buffer exists anywhere: each K-step stages a [BM rows][64 ch] slab of the NDHWC
// input straight from global memory into LDS with global_load_lds (16 B per lane, the per-lane
// SOURCE address does the tap shift, the stride and the zero padding -> out-of-range taps read
// a zero page), plus the matching [BN couts][64] slab of the pre-packed weights.  Both slabs use
// the same XOR swizzle (16-B chunk ^= (row>>1)&7) applied on the source side of the DMA and on
// the ds_read_b128 fragment reads, which makes the 32x32x16 operand reads conflict-free.
// Two LDS stages: the DMA of step s+1 is in flight while step s is on the matrix cores.
//
// Epilogue: + bias, optional tanh, bf16 tile transposed through LDS for full-line 16-B stores
// (or strided fp32 stores for the few-channel output layers), and per-tile column sums
// (sum, sum of squares) for the GroupNorm that follows every large conv on this path.
#include <type_traits>
#include "conv3_halo_common.h"
#include <string.h>
#include <stdlib.h>
#include <math.h>

#define CTSI_MAX_TAPS 48
#define CTSI_BK 64

__device__ __attribute__((aligned(256))) uint32_t g_ctsi_zero_page[64];  // 256 B of zeros

struct ConvKParams {
    const bf16_t* x1;
    const bf16_t* x2;
    const bf16_t* w;
    const float* bias;
    void* y;
    float* colsum;
    int C1, C2, Cin;
    int Di, Hi, Wi;
    int Dr, Hr, Wr;
    int sH, sW;
    int lTH, lTW;
    int linear;   // 1: a tile = BM consecutive voxels of the (d, h, w) row grid (no power-of-two box, no padded rows)
    int tilesD, tilesH, tilesW, tps, mtiles;
    int ntiles_n;
    int T, ksteps, kc_per_tap, lcpt;  // kc_per_tap: 64-chunks per tap (big mode); lcpt: log2(chunks per tap) small
    int Cout, CoutPad, Ktot;
    int Do, Ho, Wo, uH, uW;
    int nclass;
    int out_mode, cout_stride, c_off, act;
    long long osn, osc, osd, osh, osw;
    int tapdelta[CTSI_MAX_TAPS];
    // taps are separable: t = (a*NB + b)*NC + c with input offsets (ad[a], bh[b], cw[c]) per class
    int NA, NB, NC;
    int ad[4][4], bh[4][4], cw[4][4];
    int pH[4], pW[4];
    int dshift;         // rows start at input depth `dshift` (1 when the input carries depth halo slices)
    int tap_margin[4];  // -min(tapdelta) per class (>= 0): makes every buffer soffset non-negative
    int ad_min[4];      // min depth tap offset per class
    int dbg;  // timing-only ablation bits (0 in production)
    // fused ResBlock tail (ctsi_conv_out.gn_x): y = silu?(gn(gn_x) + conv result)
    const bf16_t* gn_x;
    const double* gn_sums;
    const float* gn_gamma;
    const float* gn_beta;
    int gn_groups, gn_silu;
    float gn_eps;
    double gn_count;
    // S-way split-K (launches of a few dozen blocks with a deep K loop: the 6 x 6 / 12 x 12 levels of a single 192^2 patch):
    // every (m-tile, n-tile) runs as `ksplit` blocks that each walk 1 / ksplit of the K-steps and park their fp32 accumulators;
    // the block that takes the last ticket sums the parked tiles in split order and runs the epilogue
    int ksplit;
    float* sk_ws;       // [tile][split][register][thread] fp32
    int* sk_sync;       // [tile] ticket counters (zero before the first launch; the last block resets its tile's)
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

// bijective XCD-aware remap of a 1-D block id: blocks that share an XCD get a contiguous range
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (orig >> 3);
}

template <int WM, int WN, int TM, int TN, int MODE>
__global__ void __launch_bounds__(WM* WN * 64)
conv_gather_mfma_kernel(const ConvKParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (buffer-resource builtins are device-only)
    constexpr bool SMALL = (MODE == 1);
    constexpr bool FAST = (MODE == 2 || MODE == 3);
    // MODE 3: the buffer-addressed path with a deep LDS ring and inline-asm DMA (counted vmcnt waits): for launches with
    // at most ~2 blocks per CU, where a 2-stage pipeline exposes the full DMA latency in every K-step
    constexpr bool RING = (MODE == 3);
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NW = WM * WN, NTH = NW * 64;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;
    static_assert(A_INSTR >= 1 && B_INSTR >= 1, "tile too small for the wave count");
    static_assert(BM * BN * 2 <= 2 * STAGE, "epilogue tile must fit the stages");
    constexpr int NSTG = RING ? (STAGE <= 32768 ? 4 : (STAGE <= 49152 ? 3 : 2)) : 2;
    constexpr int DPS = A_INSTR + B_INSTR;   // DMA instructions per wave and K-step

    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long* s_rowoff = reinterpret_cast<long long*>(smem + NSTG * STAGE);

    if (CTSI_DBG_RT(p.dbg, 8)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- block decode ---------------------------------------------------------------------
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nsplit = p.ksplit > 1 ? p.ksplit : 1;
    const int split = bid % nsplit;                        // (the splits of a tile are neighbours in the grid)
    bid /= nsplit;
    const int per_class = p.mtiles * p.ntiles_n;
    const int cls = bid / per_class;
    const int rem = bid - cls * per_class;
    const int mt = rem / p.ntiles_n;
    const int nt = rem - mt * p.ntiles_n;
    const int n0 = nt * BN;
    const int nb = mt / p.tps;
    int r0 = mt - nb * p.tps;
    const int TWm = (1 << p.lTW) - 1, THm = (1 << p.lTH) - 1;
    const int lTHW = p.lTH + p.lTW;
    int d0, h0 = 0, w0 = 0;
    const int HWr = p.Hr * p.Wr;
    if (p.linear) {          // small planes (6x6, 12x12, ...): box tiles would be mostly padding
        d0 = (r0 * BM) / HWr;
    } else {
        const int tD = r0 / (p.tilesH * p.tilesW);
        r0 -= tD * p.tilesH * p.tilesW;
        const int tH = r0 / p.tilesW;
        const int tW = r0 - tH * p.tilesW;
        d0 = tD * (BM >> lTHW);
        h0 = tH << p.lTH;
        w0 = tW << p.lTW;
    }
    const int vlin0 = r0 * BM;   // linear mode: first voxel of the tile inside the sample
    // row r of the tile -> (d, h, w) of the row grid
    auto row_dhw = [&](int r, int& d, int& h, int& w) {
        if (p.linear) {
            const int v = vlin0 + r;
            d = v / HWr;
            const int rem = v - d * HWr;
            h = rem / p.Wr;
            w = rem - h * p.Wr;
        } else {
            d = d0 + (r >> lTHW);
            h = h0 + ((r >> p.lTW) & THm);
            w = w0 + (r & TWm);
        }
    };
    const int tapbase = cls * p.T;

    // ---- per-row bookkeeping ----------------------------------------------------------------
    if (tid < BM) {
        const int r = tid;
        int d, h, w;
        row_dhw(r, d, h, w);
        long long off = -1;
        if (d < p.Dr && h < p.Hr && w < p.Wr) {
            const int ho = h * p.uH + p.pH[cls], wo = w * p.uW + p.pW[cls];
            if (p.out_mode == 0)
                off = ((((long long)nb * p.Do + d) * p.Ho + ho) * p.Wo + wo) * p.cout_stride + p.c_off;
            else
                off = (long long)nb * p.osn + (long long)d * p.osd + (long long)ho * p.osh +
                      (long long)wo * p.osw;
        }
        s_rowoff[r] = off;
    }

    const int NA = p.NA, NB = p.NB, NC = p.NC;
    int ax_d[4], ax_h[4], ax_w[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        ax_d[a] = p.ad[cls][a];
        ax_h[a] = p.bh[cls][a];
        ax_w[a] = p.cw[cls][a];
    }
    // rows this lane feeds with DMA: instruction j = wave*A_INSTR + i covers rows 8j..8j+7
    int a_iv0[A_INSTR];
    unsigned long long a_mask[A_INSTR];
    int a_q8[A_INSTR];  // logical 16-B chunk (x8 channels) this lane fetches for LDS slot lane&7
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int j = wave * A_INSTR + i;
        const int r = j * 8 + (lane >> 3);
        a_q8[i] = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        int d, h, w;
        row_dhw(r, d, h, w);
        const bool rv = (d < p.Dr) && (h < p.Hr) && (w < p.Wr);
        const int hi = h * p.sH, wi = w * p.sW;
        a_iv0[i] = ((nb * p.Di + d + p.dshift) * p.Hi + hi) * p.Wi + wi;
        // separable validity: bit a of md says tap-depth a is inside the input, etc.
        unsigned md = 0, mh = 0, mw = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int dd = d + p.dshift + ax_d[a], hh = hi + ax_h[a], ww = wi + ax_w[a];
            md |= (unsigned)(a < NA && dd >= 0 && dd < p.Di) << a;
            mh |= (unsigned)(a < NB && hh >= 0 && hh < p.Hi) << a;
            mw |= (unsigned)(a < NC && ww >= 0 && ww < p.Wi) << a;
        }
        unsigned long long row_hw = 0, m = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) row_hw |= ((mh >> b) & 1u) ? ((unsigned long long)mw << (b * NC)) : 0ull;
#pragma unroll
        for (int a = 0; a < 3; ++a) m |= ((md >> a) & 1u) ? (row_hw << (a * NB * NC)) : 0ull;
        m = rv ? m : 0ull;
        a_mask[i] = m;
    }
    // weight rows this lane feeds: one buffer descriptor per block, per-lane 32-bit offsets,
    // the K-step advance (128 B) rides in the scalar offset -> no VALU work per step
    const char* wbase = reinterpret_cast<const char*>(p.w) + ((long long)(cls * p.CoutPad + n0)) * p.Ktot * 2;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(wbase), 0, 0x7fffffff, 0x00020000);
    unsigned b_voff[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int j = wave * B_INSTR + i;
        const int r = j * 8 + (lane >> 3);
        const int q = (lane & 7) ^ ((r >> 1) & 7);
        b_voff[i] = (unsigned)r * (unsigned)(p.Ktot * 2) + q * 16;
    }
    __syncthreads();  // s_rowoff visible

    // All sources are addressed relative to x1 with 64-bit byte offsets so that the per-lane
    // select between source 1, source 2 and the zero page is plain arithmetic (no pointer loads).
    const char* x1c = reinterpret_cast<const char*>(p.x1);
    const long long x2delta = reinterpret_cast<const char*>(p.x2) - x1c;
    const long long zdelta = reinterpret_cast<const char*>(g_ctsi_zero_page) - x1c;
    const int C1 = p.C1, C2 = p.C2, Cin = p.Cin, T = p.T, lcpt = p.lcpt;
    // K range of this block (split-K: big mode only; the host never splits SMALL plans)
    const int s_per = (p.ksteps + nsplit - 1) / nsplit;
    const int s_begin = split * s_per;
    const int s_end = (s_begin + s_per < p.ksteps) ? s_begin + s_per : p.ksteps;
    int st_tap = SMALL ? 0 : s_begin % p.T, st_cc = SMALL ? 0 : s_begin / p.T;  // staging cursor (big mode): tap fastest, then 64-channel chunk
    // small mode looks tap deltas up per lane: keep the table spread over the wave's lanes and
    // fetch with a lane permute (no LDS memory access, so it never orders against the DMA).
    const int td_tab = (lane < T) ? p.tapdelta[tapbase + lane] : 0;

    // fast path: per-block buffer descriptors for both sources, 32-bit per-lane offsets relative to the
    // first input plane this tile can touch; padding / masked rows use an out-of-range offset, for
    // which the buffer load returns zeros (no zero page, no 64-bit address arithmetic per step).
    __amdgpu_buffer_rsrc_t a_rsrc1, a_rsrc2;
    unsigned a_voff1[A_INSTR], a_voff2[A_INSTR];
    int margin = 0;
    if (FAST) {
        int dlo = d0 + p.dshift + p.ad_min[cls];
        dlo = dlo < 0 ? 0 : dlo;
        const long long basevox = ((long long)(nb * p.Di + dlo) * p.Hi) * p.Wi;
        margin = p.tap_margin[cls];
        const char* b1 = x1c + (basevox - margin) * C1 * 2;
        const char* b2 = reinterpret_cast<const char*>(p.x2) + (basevox - margin) * C2 * 2;
        a_rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(b1), 0, 0x7fffffff, 0x00020000);
        a_rsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(b2), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            const unsigned rel = (unsigned)((long long)a_iv0[i] - basevox);
            a_voff1[i] = rel * (unsigned)(C1 * 2) + a_q8[i] * 2;
            a_voff2[i] = rel * (unsigned)(C2 * 2) + a_q8[i] * 2;
        }
    }

    v4i_t rr1, rr2, rrw;   // MODE 3: the same descriptors as SGPR quads for the inline-asm DMA
    unsigned lds0 = 0;
    if (RING) {
        int dlo = d0 + p.dshift + p.ad_min[cls];
        dlo = dlo < 0 ? 0 : dlo;
        const long long basevox = ((long long)(nb * p.Di + dlo) * p.Hi) * p.Wi;
        rr1 = h3_make_rsrc(x1c + (basevox - margin) * C1 * 2, 0x7fffffffu);
        rr2 = h3_make_rsrc(reinterpret_cast<const char*>(p.x2) + (basevox - margin) * C2 * 2, 0x7fffffffu);
        rrw = h3_make_rsrc(wbase, 0x7fffffffu);
        lds0 = (unsigned)(unsigned long long)(lptr3_t)smem;
    }

    auto stage = [&](int s, char* buf) {
        char* a_dst = buf + wave * (A_INSTR * 1024);
        if (RING) {
            const int ch0 = st_cc * 64;
            const bool second = ch0 >= C1;
            const int td = p.tapdelta[tapbase + st_tap] + margin;
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(
                (int)(second ? (unsigned)td * (unsigned)(C2 * 2) + (unsigned)(ch0 - C1) * 2
                             : (unsigned)td * (unsigned)(C1 * 2) + (unsigned)ch0 * 2));
            const unsigned abase = lds0 + (unsigned)(a_dst - smem);
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i) {
                const bool ok = ((a_mask[i] >> st_tap) & 1ull) != 0ull;
                const unsigned voff = ok ? (second ? a_voff2[i] : a_voff1[i]) : 0x80000000u;
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(abase + i * 1024));
                if (second)
                    h3_dma16(rr2, dst, voff, soff);
                else
                    h3_dma16(rr1, dst, voff, soff);
            }
            const unsigned bbase = lds0 + (unsigned)(buf - smem) + A_BYTES + wave * (B_INSTR * 1024);
            const unsigned wsoff = (unsigned)__builtin_amdgcn_readfirstlane(s * 128);
#pragma unroll
            for (int i = 0; i < B_INSTR; ++i) {
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(bbase + i * 1024));
                h3_dma16(rrw, dst, b_voff[i], wsoff);
            }
            if (++st_tap == p.T) {
                st_tap = 0;
                ++st_cc;
            }
            return;
        }
        if (FAST) {
            const int ch0 = st_cc * 64;
            const bool second = ch0 >= C1;
            const int td = p.tapdelta[tapbase + st_tap] + margin;
            const unsigned soff = second ? (unsigned)td * (unsigned)(C2 * 2) + (unsigned)(ch0 - C1) * 2
                                         : (unsigned)td * (unsigned)(C1 * 2) + (unsigned)ch0 * 2;
#pragma unroll
            for (int i = 0; i < A_INSTR; ++i) {
                const bool ok = ((a_mask[i] >> st_tap) & 1ull) != 0ull;
                const unsigned voff = ok ? (second ? a_voff2[i] : a_voff1[i]) : 0x80000000u;
                if (second)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc2, (lptr_t)(a_dst + i * 1024), 16, voff, soff, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc1, (lptr_t)(a_dst + i * 1024), 16, voff, soff, 0, 0);
            }
        }
        int td_u = 0;
        if (!SMALL && !FAST) td_u = p.tapdelta[tapbase + st_tap];
#pragma unroll
        for (int i = 0; i < (FAST ? 0 : A_INSTR); ++i) {
            int tap, ch, td;
            if (SMALL) {
                const int g = s * 8 + (a_q8[i] >> 3);
                tap = g >> lcpt;
                ch = (g & ((1 << lcpt) - 1)) << 3;
                td = __shfl(td_tab, tap < T ? tap : 0);
            } else {
                tap = st_tap;
                ch = st_cc * 64 + a_q8[i];
                td = td_u;
            }
            const bool ok = (tap < T) & (ch < Cin) & (((a_mask[i] >> tap) & 1ull) != 0ull);
            const bool second = ch >= C1;
            const int C = second ? C2 : C1;
            const int chs = second ? ch - C1 : ch;
            long long boff = (((long long)a_iv0[i] + td) * C + chs) * 2 + (second ? x2delta : 0ll);
            boff = ok ? boff : zdelta;
            glds16(x1c + boff, a_dst + i * 1024);
        }
        char* b_dst = buf + A_BYTES + wave * (B_INSTR * 1024);
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lptr_t)(b_dst + i * 1024), 16, b_voff[i], s * 128, 0, 0);
        if (!SMALL) {
            if (++st_tap == p.T) {
                st_tap = 0;
                ++st_cc;
            }
        }
    };

    // ---- accumulators -------------------------------------------------------------------------
    // start value = the bias of the lane's cout (split-K: in the first split only; the splits are summed in split order)
    f32x16 acc[TM][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n0 + wn * TN * 32 + j * 32 + (lane & 31);
        const float bv = (p.bias != nullptr && co < p.Cout && split == 0) ? p.bias[co] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
    }

    // fragment read offsets: row = tile_base + (lane&31); (row>>1)&7 == (lane>>1)&7 for 32-aligned bases
    const int fsw = (lane >> 1) & 7;
    const int frow = (lane & 31) * 128;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = (((kk * 2 + (lane >> 5)) ^ fsw) << 4) + frow;

    const int S = s_end - s_begin;                       // K-steps this block walks: absolute steps s_begin + i
    auto compute = [&](const char* cur) {
        const char* a_base = cur + (wm * TM * 32) * 128;
        const char* b_base = cur + A_BYTES + (wn * TN * 32) * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(a_base + i * 32 * 128 + koff[kk]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8*>(b_base + j * 32 * 128 + koff[kk]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };
    if (RING) {
        // ring of NSTG stages: the DMAs of step s+NSTG-1 are issued while step s is on the matrix cores; the wait only
        // requires step s to have landed ((NSTG-2) * DPS younger DMA instructions of this wave may stay in flight)
#pragma unroll
        for (int i = 0; i < NSTG - 1; ++i)
            if (i < S) stage(s_begin + i, smem + i * STAGE);
        int slot = 0;
        for (int s = 0; s < S; ++s) {
            if (s + NSTG - 2 < S)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTG - 2) * DPS) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + NSTG - 1 < S) stage(s_begin + s + NSTG - 1, smem + (slot == 0 ? NSTG - 1 : slot - 1) * STAGE);
            compute(smem + slot * STAGE);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            slot = (slot + 1 == NSTG) ? 0 : slot + 1;
        }
    } else {
        if (!(CTSI_DBG_RT(p.dbg, 4)) && S > 0) stage(s_begin, smem);
        for (int s = 0; s < S; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            char* cur = smem + (s & 1) * STAGE;
            if (s + 1 < S) stage(s_begin + s + 1, smem + ((s + 1) & 1) * STAGE);
            compute(cur);
        }
    }
    __syncthreads();  // every wave is done with the stage buffers
    if (CTSI_DBG_RT(p.dbg, 2)) return;
    if constexpr (TM == 2 && TN == 2)          // (the host splits 128 x 128-tile plans only: keep the other forms' code lean)
    if (nsplit > 1) {
        // Split-K hand-off, S-way (conv3_halo_k32.hip has the 2-way form): every block parks its accumulators with agent-scope
        // (sc1) stores, drains them, and only then takes a ticket; the block that draws the LAST ticket therefore finds all
        // parked tiles complete -- no spinning -- and sums them IN SPLIT ORDER (its own included, read back like the others):
        // the result does not depend on which block came last.  No cache maintenance: every access to the parked bytes is sc1.
        constexpr int NREG = TM * TN * 16;
        const int tile = bid;                                 // (class, m-tile, n-tile) index
        float* wst = p.sk_ws + ((size_t)tile * nsplit) * NREG * NTH + tid;
        float* mine = wst + (size_t)split * NREG * NTH;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __hip_atomic_store(mine + (size_t)((i * TN + j) * 16 + r) * NTH, acc[i][j][r], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* s_tk = reinterpret_cast<int*>(smem);
        if (tid == 0) *s_tk = __hip_atomic_fetch_add(p.sk_sync + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int ticket = *s_tk;
        __syncthreads();
        if (ticket != nsplit - 1) return;
        if (tid == 0) __hip_atomic_store(p.sk_sync + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        // (split loop OUTSIDE the register loops: the NREG loads of one parked tile are independent and go out back to back;
        //  with the loops the other way round every register waited for its own S dependent round trips: 123 us per launch)
        for (int k = 0; k < nsplit; ++k) {
            float tmp[TM][TN][16];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        tmp[i][j][r] = __hip_atomic_load(wst + ((size_t)k * NREG + (i * TN + j) * 16 + r) * NTH, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = k == 0 ? tmp[i][j][r] : acc[i][j][r] + tmp[i][j][r];
        }
    }

    // ---- epilogue --------------------------------------------------------------------------------
    bf16_t* s_tile = reinterpret_cast<bf16_t*>(smem);                          // [BM][BN]
    float* s_cs = reinterpret_cast<float*>(smem + NSTG * STAGE + BM * 8);      // [WM][BN][2]
    const int lhi = lane >> 5, lcol = lane & 31;
    const bool want_sums = p.colsum != nullptr;
    // validity of the 16*TM rows this lane's accumulators belong to (bit r of vbits[i])
    unsigned vbits[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        unsigned vb = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            vb |= (unsigned)(s_rowoff[row] >= 0) << r;
        }
        vbits[i] = vb;
    }
    if (p.out_mode == 0 && p.act == 0) {
        // fast path: bf16 NDHWC output staged through LDS, optional column sums.  Per accumulator: half a v_cvt_pk_bf16_f32
        // (rows r, r + 1 share it), one 2-byte LDS write, one add + one fma for the sums; the validity select only in waves
        // that own rows outside the volume.
        auto tile_out = [&](auto masked_tag, auto sums_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value, SUMS = decltype(sums_tag)::value;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = wn * TN * 32 + j * 32 + lcol;
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    bf16_t* trow = s_tile + (wm * TM * 32 + i * 32 + 4 * lhi) * BN + col;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float v0 = acc[i][j][r], v1 = acc[i][j][r + 1];
                        const uint32_t pk = pack_bf16x2_v(f32x2_t{v0, v1});
                        const int rr = (r & 3) + 8 * (r >> 2);
                        trow[rr * BN] = (bf16_t)(pk & 0xffffu);
                        trow[(rr + 1) * BN] = (bf16_t)(pk >> 16);
                        if (SUMS) {
                            const float m0 = (!MASKED || ((vbits[i] >> r) & 1u)) ? v0 : 0.0f;
                            const float m1 = (!MASKED || ((vbits[i] >> (r + 1)) & 1u)) ? v1 : 0.0f;
                            s1 += m0;
                            s2 = __builtin_fmaf(m0, m0, s2);
                            s1 += m1;
                            s2 = __builtin_fmaf(m1, m1, s2);
                        }
                    }
                }
                if (SUMS) {
                    s1 += __shfl_xor(s1, 32);
                    s2 += __shfl_xor(s2, 32);
                    if (lhi == 0) {
                        s_cs[(wm * BN + col) * 2 + 0] = s1;
                        s_cs[(wm * BN + col) * 2 + 1] = s2;
                    }
                }
            }
        };
        using T = std::true_type;
        using F = std::false_type;
        bool rg = false;
#pragma unroll
        for (int i = 0; i < TM; ++i) rg = rg || vbits[i] != 0xffffu;
        const bool ragged = __builtin_amdgcn_ballot_w64(rg) != 0ull;   // wave-uniform
        if (!want_sums) tile_out(F{}, F{});
        else if (ragged) tile_out(T{}, T{});
        else tile_out(F{}, T{});
    } else {
        // generic path: activation and/or strided fp32 output (few-channel output layers)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * TN * 32 + j * 32 + lcol;
            const int co = n0 + col;
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    float v = acc[i][j][r];
                    if (p.act == 1) v = tanhf(v);
                    const bool valid = (vbits[i] >> r) & 1u;
                    if (valid) {
                        s1 += v;
                        s2 += v * v;
                    }
                    if (p.out_mode == 0) {
                        s_tile[row * BN + col] = f32_to_bf16(v);
                    } else if (valid && co < p.Cout) {
                        reinterpret_cast<float*>(p.y)[s_rowoff[row] + (long long)co * p.osc] = v;
                    }
                }
            }
            if (want_sums) {
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (lhi == 0) {
                    s_cs[(wm * BN + col) * 2 + 0] = s1;
                    s_cs[(wm * BN + col) * 2 + 1] = s2;
                }
            }
        }
    }
    if (p.gn_x != nullptr && tid < BN) {
        // fused ResBlock tail: per-channel scale / shift of this n-tile from the fp64 group statistics (as gn_apply_kernel
        // computes them); the column-sum scratch is free (fused launches never emit column sums)
        const int co = n0 + tid;
        float sc = 0.0f, sh = 0.0f;
        if (co < p.Cout) {
            const int cpg = p.Cout / p.gn_groups, g = co / cpg;
            const double* sm = p.gn_sums + ((long long)nb * p.gn_groups + g) * 2;
            const double mean = sm[0] / p.gn_count;
            double var = sm[1] / p.gn_count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
            sc = p.gn_gamma[co] * rstd;
            sh = p.gn_beta[co] - (float)mean * sc;
        }
        s_cs[tid] = sc;
        s_cs[BN + tid] = sh;
    }
    __syncthreads();
    if (p.colsum != nullptr && tid < BN) {
        float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            t1 += s_cs[(m * BN + tid) * 2 + 0];
            t2 += s_cs[(m * BN + tid) * 2 + 1];
        }
        const long long tg = (long long)cls * p.mtiles + mt;
        const long long slab = (long long)p.nclass * p.mtiles * p.CoutPad;
        p.colsum[tg * p.CoutPad + n0 + tid] = t1;
        p.colsum[slab + tg * p.CoutPad + n0 + tid] = t2;
    }
    if (p.out_mode == 0) {
        constexpr int CPR = BN / 8;  // 16-B chunks per row
        constexpr int ITER = (BM * CPR) / NTH;
        static_assert((BM * CPR) % NTH == 0, "store loop assumes whole iterations");
        bf16_t* y = reinterpret_cast<bf16_t*>(p.y);
        if (p.gn_x != nullptr) {
            // fused ResBlock tail: all of this thread's 16-byte pieces of the normalised tensor are requested up front
            // (ITER independent loads in flight), then combined with the staged conv result
            uint4 hreg[ITER];
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int c = tid + it * NTH;
                const int row = c / CPR, cc = c - row * CPR;
                const long long off = s_rowoff[row];
                const int co = n0 + cc * 8;
                hreg[it] = (off >= 0 && co < p.Cout) ? *reinterpret_cast<const uint4*>(p.gn_x + off + co)
                                                     : make_uint4(0, 0, 0, 0);
            }
            // NTH is a multiple of the row's chunk count: a thread meets the same 8 couts in every iteration and keeps their scale /
            // shift in registers (4 16-byte LDS reads per thread instead of 16 scalar ones per piece, whose 32-byte lane stride put
            // two lanes on every bank: most of the 34-48 % LDS bank conflicts the counters showed for the 1x1x1 tails)
            static_assert(NTH % CPR == 0, "a thread's chunk must not depend on the iteration");
            const int cc0 = tid % CPR;
            float gsc[8], gsh[8];
            *reinterpret_cast<float4*>(gsc) = *reinterpret_cast<const float4*>(s_cs + cc0 * 8);
            *reinterpret_cast<float4*>(gsc + 4) = *reinterpret_cast<const float4*>(s_cs + cc0 * 8 + 4);
            *reinterpret_cast<float4*>(gsh) = *reinterpret_cast<const float4*>(s_cs + BN + cc0 * 8);
            *reinterpret_cast<float4*>(gsh + 4) = *reinterpret_cast<const float4*>(s_cs + BN + cc0 * 8 + 4);
            auto tail = [&](auto silu_tag) {
                constexpr bool SILU = decltype(silu_tag)::value;
#pragma unroll
                for (int it = 0; it < ITER; ++it) {
                    const int c = tid + it * NTH;
                    const int row = c / CPR, cc = c - row * CPR;
                    const long long off = s_rowoff[row];
                    const int co = n0 + cc * 8;
                    if (off >= 0 && co < p.Cout) {
                        const uint4 v = *reinterpret_cast<const uint4*>(s_tile + row * BN + cc * 8);
                        const uint4 h = hreg[it];
                        const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, rw[4] = {v.x, v.y, v.z, v.w};
                        uint32_t ow[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            f32x2_t a = {__uint_as_float(hw[k] << 16), __uint_as_float(hw[k] & 0xffff0000u)};
                            a = a * f32x2_t{gsc[2 * k], gsc[2 * k + 1]} + f32x2_t{gsh[2 * k], gsh[2 * k + 1]} +
                                f32x2_t{__uint_as_float(rw[k] << 16), __uint_as_float(rw[k] & 0xffff0000u)};
                            if (SILU) a = silu2_f(a);
                            ow[k] = pack_bf16x2_v(a);
                        }
                        *reinterpret_cast<uint4*>(y + off + co) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                    }
                }
            };
            if (p.gn_silu) tail(std::true_type{});
            else tail(std::false_type{});
        } else {
            for (int c = tid; c < BM * CPR; c += NTH) {
                const int row = c / CPR, cc = c - row * CPR;
                const long long off = s_rowoff[row];
                const int co = n0 + cc * 8;
                if (off >= 0 && co < p.Cout && !(CTSI_DBG_RT(p.dbg, 1))) {
                    const uint4 v = *reinterpret_cast<const uint4*>(s_tile + row * BN + cc * 8);
                    *reinterpret_cast<uint4*>(y + off + co) = v;
                }
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- weight packing -----------------------------------------------------------------------------
struct PackParams {
    const float* w;
    bf16_t* out;
    int nclass, T, Cout, CoutPad, Cin, CinW, Ktot, small, kc_per_tap, lcpt;
    long long s_co, s_ci;  // element strides of the fp32 weight tensor (tap stride is 1)
    int tapk[CTSI_MAX_TAPS];
};

__global__ void conv_pack_weights_kernel(const PackParams q) {
    const long long total = (long long)q.nclass * q.CoutPad * q.Ktot;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % q.Ktot);
        const long long rc = idx / q.Ktot;
        const int co = (int)(rc % q.CoutPad);
        const int cls = (int)(rc / q.CoutPad);
        int tap, ci;
        if (q.small) {
            const int g = k >> 3;
            tap = g >> q.lcpt;
            ci = ((g & ((1 << q.lcpt) - 1)) << 3) + (k & 7);
        } else {
            const int s = k >> 6;
            tap = s % q.T;
            ci = (s / q.T) * 64 + (k & 63);
        }
        float v = 0.0f;
        if (tap < q.T && ci < q.CinW && co < q.Cout)
            v = q.w[(long long)co * q.s_co + (long long)ci * q.s_ci + q.tapk[cls * q.T + tap]];
        q.out[idx] = f32_to_bf16(v);
    }
}

// ---- host side --------------------------------------------------------------------------------------
struct ctsi_conv_plan {
    ctsi_conv_desc d;
    int Cin, CinW;
    int Do, Ho, Wo;
    int Dr, Hr, Wr, sH, sW, uH, uW;
    int nclass, T;
    int small, lcpt, kc_per_tap, ksteps, Ktot;
    int BM, BN, CoutPad, ntiles_n;
    int lTH, lTW, TD, TH, TW, tilesD, tilesH, tilesW, tps, mtiles;
    int linear;   // gather kernel: tiles are runs of BM consecutive row-grid voxels
    int tapk[CTSI_MAX_TAPS];
    int tapdelta[CTSI_MAX_TAPS];
    int8_t od[CTSI_MAX_TAPS], oh[CTSI_MAX_TAPS], ow[CTSI_MAX_TAPS];
    int NA, NB, NC;
    int ad[4][4], bh[4][4], cw[4][4];
    int8_t pH[4], pW[4];
    int tap_margin[4], ad_min[4];
    int fast, dshift;
    int h32_w16;    // halo3 == 2 only: 1 = 4x4x16 tile (two W-lines of 16 per A tile), 2 = 3x4x16 tile, instead of 4x2x32
    int m512_w16;   // halo3 == 7 only: tile of the k32 kernel: 0 = 4x4x32, 2 = 4x8x16, 3 = 3x4x32, 5 = 3x8x16, 6 = 4x4x24, 7 = 8x4x12 (384 voxels)
    int gsplit;     // gather kernel (halo3 == 0): S-way split-K for launches of a few dozen blocks with a deep K loop (needs a workspace)
    int ds;         // halo3 == 7: the strided (3,4,4)/(1,2,2) Downsample form of the k32 kernel (conv3_halo_k32.hip, DS)
    int head2;      // halo3 == 6: conv3_head2_kernel (taps as the GEMM's N dimension) serves the launches that ask for no column sums
    int ksplit;     // halo3 == 7, tile 5: 2 = two blocks per (tile, n-tile), each half of the input-channel chunks (needs a workspace)
    int stem;       // 1: conv3_stem_kernel (conv3_stem.hip: 3x3x3 conv of a one-channel volume stored with 8 channels); chosen by
                    // ctsi_conv_plan_set_weight_cin(plan, 1)
    int stream1;    // > 0: conv1_stream_kernel (conv1_stream.hip: 1x1x1 conv + fused GroupNorm tail as a streaming pass), value = 16-cout
                    // tiles per n-tile; chosen by ctsi_conv_plan_set_stream_tail, never by ctsi_conv_plan_create
    int halo3;  // 3x3x3 halo-tile kernels: 2 = conv3_halo32_kernel (conv3_halo.hip: 4x2x32 / 4x4x16 / 3x4x16 tiles), 6 = few-cout
                // heads (conv3_head.hip), 7 = conv3_halo_k32_kernel (conv3_halo_k32.hip: 512- / 384-voxel tiles, ConvTranspose).
                // 1 (16x16x32 form of the 4x4x16 tile), 3 / 4 (persistent and half-size blocks) and 5 (32x32x16 form of the
                // 512-voxel tile) were measured slower and live under csrc/experiments/, outside libctsi.so.
    double flops;
};

static int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// pick the power-of-two (TD,TH,TW) with TD*TH*TW == bm that covers the row grid with the fewest
// padded rows; ties go to the most cube-like tile (smallest halo for the L2).
static void choose_tile(int bm, int Dr, int Hr, int Wr, int* TD, int* TH, int* TW) {
    long long best = -1;
    int bsurf = 0;
    const int lb = ilog2(bm);
    for (int lw = 0; lw <= lb; ++lw)
        for (int lh = 0; lh + lw <= lb; ++lh) {
            const int tw = 1 << lw, th = 1 << lh, td = bm >> (lw + lh);
            const long long tiles = (long long)ceil_div(Dr, td) * ceil_div(Hr, th) * ceil_div(Wr, tw);
            const int surf = (td + 2) * (th + 2) * (tw + 2);
            if (best < 0 || tiles < best || (tiles == best && surf < bsurf)) {
                best = tiles;
                bsurf = surf;
                *TD = td;
                *TH = th;
                *TW = tw;
            }
        }
}

extern "C" int ctsi_conv_plan_create(ctsi_conv_plan** out, const ctsi_conv_desc* desc) {
    CTSI_CHECK_ARG(out && desc, "ctsi_conv_plan_create: null argument");
    const ctsi_conv_desc& d = *desc;
    CTSI_CHECK_ARG(d.n > 0 && d.c1 > 0 && d.c2 >= 0 && d.cout > 0 && d.di > 0 && d.hi > 0 && d.wi > 0,
                   "ctsi_conv_plan_create: bad sizes n=%d c1=%d c2=%d cout=%d in=%dx%dx%d", d.n, d.c1,
                   d.c2, d.cout, d.di, d.hi, d.wi);
    CTSI_CHECK_ARG(d.c1 % 8 == 0 && d.c2 % 8 == 0,
                   "ctsi_conv_plan_create: source channel counts must be multiples of 8 (got %d, %d); "
                   "pad the tensor at the layout boundary", d.c1, d.c2);
    CTSI_CHECK_ARG(d.kd >= 1 && d.kh >= 1 && d.kw >= 1 && d.sh >= 1 && d.sw >= 1,
                   "ctsi_conv_plan_create: bad kernel/stride");
    ctsi_conv_plan* p = (ctsi_conv_plan*)calloc(1, sizeof(ctsi_conv_plan));
    if (!p) {
        ctsi_set_error("ctsi_conv_plan_create: out of host memory");
        return CTSI_ERR_INVALID;
    }
    p->d = d;
    p->dshift = d.halo_d ? 1 : 0;
    const int di_own = d.di - 2 * p->dshift;   // depth of the slab the rows cover
    if (d.halo_d && (d.kd != 3 || d.pd != 1 || di_own < 1)) {
        free(p);
        ctsi_set_error("ctsi_conv_plan_create: halo_d needs kd=3, pd=1 and di >= 3");
        return CTSI_ERR_UNSUPPORTED;
    }
    p->Cin = d.c1 + d.c2;
    p->CinW = p->Cin;
    const int KK = d.kd * d.kh * d.kw;
    if (!d.transposed) {
        p->Do = di_own + 2 * d.pd - d.kd + 1;
        p->Ho = (d.hi + 2 * d.ph - d.kh) / d.sh + 1;
        p->Wo = (d.wi + 2 * d.pw - d.kw) / d.sw + 1;
        if (KK > CTSI_MAX_TAPS || p->Do <= 0 || p->Ho <= 0 || p->Wo <= 0) {
            free(p);
            ctsi_set_error("ctsi_conv_plan_create: unsupported Conv3d geometry k=(%d,%d,%d)", d.kd, d.kh, d.kw);
            return CTSI_ERR_UNSUPPORTED;
        }
        p->nclass = 1;
        p->T = KK;
        p->Dr = p->Do; p->Hr = p->Ho; p->Wr = p->Wo;
        p->sH = d.sh; p->sW = d.sw; p->uH = 1; p->uW = 1;
        p->pH[0] = 0; p->pW[0] = 0;
        p->NA = d.kd; p->NB = d.kh; p->NC = d.kw;
        if (d.kd > 3 || d.kh > 4 || d.kw > 4) {
            free(p);
            ctsi_set_error("ctsi_conv_plan_create: unsupported Conv3d geometry k=(%d,%d,%d)", d.kd, d.kh, d.kw);
            return CTSI_ERR_UNSUPPORTED;
        }
        for (int a = 0; a < d.kd; ++a) p->ad[0][a] = a - d.pd;
        for (int b = 0; b < d.kh; ++b) p->bh[0][b] = b - d.ph;
        for (int c = 0; c < d.kw; ++c) p->cw[0][c] = c - d.pw;
        int t = 0;
        for (int a = 0; a < d.kd; ++a)
            for (int b = 0; b < d.kh; ++b)
                for (int c = 0; c < d.kw; ++c, ++t) {
                    p->od[t] = (int8_t)(a - d.pd);
                    p->oh[t] = (int8_t)(b - d.ph);
                    p->ow[t] = (int8_t)(c - d.pw);
                    p->tapk[t] = (a * d.kh + b) * d.kw + c;
                }
    } else {
        // ConvTranspose3d, depth stride 1: o_d = i_d - pd + k_d; o_h = i_h*sh - ph + k_h.
        // One parity class per (o_h % sh, o_w % sw); each class is a stride-1 gather conv on the
        // input grid with kd * (kh/sh) * (kw/sw) taps.
        const bool ok = d.sh == 2 && d.sw == 2 && d.kh == 4 && d.kw == 4 && d.ph == 1 && d.pw == 1 &&
                        d.kd == 3 && d.pd == 1;
        if (!ok) {
            free(p);
            ctsi_set_error("ctsi_conv_plan_create: ConvTranspose3d supports k=(3,4,4) s=(1,2,2) p=1 only");
            return CTSI_ERR_UNSUPPORTED;
        }
        p->Do = di_own; p->Ho = d.hi * 2; p->Wo = d.wi * 2;
        p->nclass = 4;
        p->T = 12;
        p->Dr = di_own; p->Hr = d.hi; p->Wr = d.wi;
        p->sH = 1; p->sW = 1; p->uH = 2; p->uW = 2;
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
            p->pH[cls] = (int8_t)py;
            p->pW[cls] = (int8_t)px;
            // output o = 2m+py receives (k, i): py=0 -> (1,m),(3,m-1); py=1 -> (2,m),(0,m+1)
            const int ky[2] = {py == 0 ? 1 : 2, py == 0 ? 3 : 0};
            const int oy[2] = {0, py == 0 ? -1 : 1};
            const int kx[2] = {px == 0 ? 1 : 2, px == 0 ? 3 : 0};
            const int ox[2] = {0, px == 0 ? -1 : 1};
            p->NA = 3; p->NB = 2; p->NC = 2;
            for (int a = 0; a < 3; ++a) p->ad[cls][a] = 1 - a;
            for (int b = 0; b < 2; ++b) { p->bh[cls][b] = oy[b]; p->cw[cls][b] = ox[b]; }
            int t = cls * 12;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 2; ++b)
                    for (int c = 0; c < 2; ++c, ++t) {
                        p->od[t] = (int8_t)(1 - a);  // i_d = o_d + pd - k_d
                        p->oh[t] = (int8_t)oy[b];
                        p->ow[t] = (int8_t)ox[c];
                        p->tapk[t] = (a * 4 + ky[b]) * 4 + kx[c];
                    }
        }
    }
    for (int t = 0; t < p->nclass * p->T; ++t)
        p->tapdelta[t] = (p->od[t] * d.hi + p->oh[t]) * d.wi + p->ow[t];
    for (int c = 0; c < p->nclass; ++c) {
        int mn = 0, dmin = 0;
        for (int t = 0; t < p->T; ++t) {
            if (p->tapdelta[c * p->T + t] < mn) mn = p->tapdelta[c * p->T + t];
            if (p->od[c * p->T + t] < dmin) dmin = p->od[c * p->T + t];
        }
        p->tap_margin[c] = -mn;
        p->ad_min[c] = dmin;
    }

    // K walk
    if (p->Cin <= 32 && (p->Cin & (p->Cin - 1)) == 0) {
        p->small = 1;
        p->lcpt = ilog2(p->Cin / 8);
        const int chunks = p->T << p->lcpt;
        p->ksteps = ceil_div(chunks, 8);
        p->kc_per_tap = 0;
    } else {
        p->small = 0;
        p->kc_per_tap = ceil_div(p->Cin, 64);
        p->ksteps = p->T * p->kc_per_tap;
    }
    p->Ktot = p->ksteps * CTSI_BK;
    // tile selection: the bigger the tile the fewer L2->LDS bytes per flop (128x128: 64 flop/B,
    // 256x128: 85, 256x256: 128), but the grid must still fill 256 CUs about twice over.
    {
        const long long rows = (long long)d.n * p->Dr * p->Hr * p->Wr * p->nclass;
        const char* force = getenv("CTSI_CONV_TILE");  // "128x128" | "256x128" | "256x256" (tuning aid)
        p->BM = 128;
        p->BN = d.cout <= 32 ? 32 : 128;
        if (d.cout > 32) {
            const long long wg_256x256 = (rows / 256) * ceil_div(d.cout, 256);
            const long long wg_256x128 = (rows / 256) * ceil_div(d.cout, 128);
            if (d.cout >= 256 && d.cout % 256 == 0 && wg_256x256 >= 640) {
                p->BM = 256; p->BN = 256;
            } else if (wg_256x128 >= 640) {
                p->BM = 256; p->BN = 128;
            }
            if (force) {
                if (!strcmp(force, "128x128")) { p->BM = 128; p->BN = 128; }
                if (!strcmp(force, "256x128")) { p->BM = 256; p->BN = 128; }
                if (!strcmp(force, "256x256") && d.cout % 256 == 0) { p->BM = 256; p->BN = 256; }
            }
        }
    }
    {   // 3x3x3 / stride 1 / pad 1 with whole 32-channel chunks per source: LDS halo-tile kernel, provided its
        // fixed 4x4x16 tile does not waste more than ~30 % of the rows and the per-tile halo fits 2^31 bytes
        const bool k3 = !d.transposed && d.kd == 3 && d.kh == 3 && d.kw == 3 && d.sh == 1 && d.sw == 1 &&
                        d.pd == 1 && d.ph == 1 && d.pw == 1;
        const long long rows = (long long)p->Dr * p->Hr * p->Wr;
        const long long padded = (long long)ceil_div(p->Dr, 4) * ceil_div(p->Hr, 4) * ceil_div(p->Wr, 16) * 256;
        const int cmax = d.c1 > d.c2 ? d.c1 : d.c2;
        const double extent = 8.0 * d.hi * d.wi * cmax * 2.0;
        // whole 32-channel chunks per source for the 4x4x16 / 4x2x32 kernels; the 512-voxel kernel walks 16-channel chunks,
        // so sources of 16 channels (the U-Net stem: [z | cond] = 2 x latent_dim = 16) can use it too
        const bool c32 = !p->small && d.c1 % 32 == 0 && d.c2 % 32 == 0;
        const bool c16 = d.c1 % 16 == 0 && d.c2 % 16 == 0 && !getenv("CTSI_CONV_NO_C16");
        p->halo3 = k3 && (c32 || c16) && d.cout >= 64 && d.cout % 8 == 0 &&
                   p->CinW == p->Cin && (rows * 10 >= padded * 7 || getenv("CTSI_CONV_FORCE_HALO3")) && extent < 2.0e9 &&
                   !getenv("CTSI_CONV_NO_HALO3");
        if (p->halo3) {
            p->BM = 256;
            p->BN = 128;
            // Which of the three halo-tile kernels: score = useful fraction of the tile rows x fill of the 256 CUs (blocks /
            // whole rounds) x the kernel's measured relative efficiency on full grids (16-wide 0.85, 4x2x32 1.0, 4x4x32 / 512
            // voxels 1.1).  Reproduces every interleaved A/B measurement of profiles/r01_notes.md: 48x128^2 and 48x64^2 -> 512-voxel
            // tile, 48x32^2 x 512 couts -> 4x2x32 (384 big blocks would idle a quarter of the CUs), 48x16^2 and 48x48^2 ->
            // 16-wide, a lone 48x24^2 level (144 blocks either way) -> 512-voxel tile.
            auto score = [&](int td, int th, int tw, double eff) {
                const long long t = (long long)d.n * ceil_div(p->Dr, td) * ceil_div(p->Hr, th) * ceil_div(p->Wr, tw);
                const long long b = t * ceil_div(d.cout, 128);
                const double useful = (double)rows * d.n / ((double)t * td * th * tw);
                return useful * (double)b / (double)(((b + 255) / 256) * 256) * eff;
            };
            const double s16 = score(4, 4, 16, 0.85), s32 = score(4, 2, 32, 1.0), s512 = score(4, 4, 32, 1.1);
            const char* w16 = getenv("CTSI_CONV_M512W16");    // "0" | "1": the 4x8x16 form of the 512-voxel kernel
            const double s512w = (w16 && !strcmp(w16, "0")) ? 0.0 : score(4, 8, 16, 1.0);
            int pick = s16 >= s32 && s16 >= s512 ? 1 : (s32 >= s512 ? 2 : 5);
            bool use_w16 = s512w > s16 && s512w > s32 && s512w > s512;
            if (w16 && !strcmp(w16, "1")) use_w16 = true;
            const char* hv = getenv("CTSI_CONV_HALO_TILE");   // "16" | "32" (tuning aids)
            const char* m5 = getenv("CTSI_CONV_M512");        // "0" | "1"
            if (hv && !strcmp(hv, "16")) pick = 1;
            if (hv && !strcmp(hv, "32") && pick == 1) pick = s32 >= s512 ? 2 : 5;
            if (m5 && !strcmp(m5, "0") && pick == 5) pick = 2;
            if (m5 && !strcmp(m5, "1") && pick != 1) pick = 5;
            if (!c32 && pick != 5) {   // 16-channel sources: only the 512-voxel kernel applies
                pick = 5;
                use_w16 = s512w > s512;
            }
            if (use_w16 && (!c32 || (!(hv && !strcmp(hv, "16")) && !(m5 && !strcmp(m5, "0"))))) {
                pick = 5;
                p->m512_w16 = 2;
            }
            {   // 16-wide levels: tile 4x4x16 (256 voxels; an A tile of 32 rows = two W-lines of 16) or 3x4x16 (192 voxels; 3x1
                // MFMA tiles per wave: 0.95 of the 2x2 form's efficiency) of conv3_halo32_kernel -- whichever fills the 256
                // CUs better (48x16x16 x 512 couts: 192 vs 256 blocks)
                const char* hw = getenv("CTSI_CONV_H32W16");   // "1" | "2" (tuning / test aid)
                if (pick == 1) {
                    pick = 2;
                    p->h32_w16 = score(3, 4, 16, 0.95) > score(4, 4, 16, 1.0) ? 2 : 1;
                    if (hw && !strcmp(hw, "1")) p->h32_w16 = 1;
                    if (hw && !strcmp(hw, "2")) p->h32_w16 = 2;
                    if (p->h32_w16 == 2) p->BM = 192;
                }
            }
            p->halo3 = pick;
            if (pick == 5) {
                p->BM = 512;
                // the 512-voxel tile runs on v_mfma_f32_16x16x32_bf16 over tap pairs (conv3_halo_k32.hip: +10-12 % on real data
                // over the 32x32x16 form conv3_halo32m_kernel, which -- with its normalise-on-load experiment -- now lives
                // under csrc/experiments/, outside libctsi.so; profiles/r02_notes.md)
                p->halo3 = 7;
            }
            {   // 384-voxel tile (3x4x32, 48 voxels per wave) of the k32 kernel where 512-voxel tiles fill the CUs badly: the
                // 48x32x32 x 512-cout layers are 384 blocks of 512 x 128 (1.5 rounds of the 256 CUs) or 512 blocks of 384 x 128
                // (2 rounds).  Relative efficiencies on full grids: 4x2x32 / 32x32x16 MFMAs 1.0, k32 512-voxel 1.15, k32 384-voxel 1.1
                const char* t384 = getenv("CTSI_CONV_K32_384");   // "0" | "1" (tuning / test aid)
                const double cur = p->halo3 == 7 ? (p->m512_w16 == 2 ? score(4, 8, 16, 1.15) : score(4, 4, 32, 1.15))
                                   : p->halo3 == 2 ? (p->h32_w16 == 2 ? score(3, 4, 16, 0.95) : p->h32_w16 == 1 ? score(4, 4, 16, 1.0) : s32)
                                                   : s16;
                bool use384 = score(3, 4, 32, 1.1) > cur;
                if (t384 && !strcmp(t384, "0")) use384 = false;
                if (t384 && !strcmp(t384, "1")) use384 = true;
                if (use384) {
                    p->halo3 = 7;
                    p->BM = 384;
                    p->h32_w16 = 0;
                    p->m512_w16 = 3;
                }
                // 16-wide levels with deep K and few voxels (48x16x16, 512 -> 512: 192 / 256 / 128 blocks with 512- / 192- / 384-
                // voxel tiles): 3x8x16 = 384 voxels with 2-way split-K = 256 blocks of half the channel chunks each (one round,
                // half the weight bytes per block).  Measured 0.141 ms per half-K block against 0.18 ms for the 192-voxel tile.
                const char* sk = getenv("CTSI_CONV_K32_SPLITK");   // "0" | "1" | "plain" (tuning / test aid)
                auto score_sk = [&]() {
                    const long long t = (long long)d.n * ceil_div(p->Dr, 3) * ceil_div(p->Hr, 8) * ceil_div(p->Wr, 16);
                    const long long b = 2 * t * ceil_div(d.cout, 128);
                    const double useful = (double)rows * d.n / ((double)t * 384);
                    return useful * (double)b / (double)(((b + 255) / 256) * 256) * 1.1;
                };
                // (round 4: not below 256 input channels -- a half-K block of a 128-channel layer walks 4 chunks, and its prologue,
                //  hand-off and epilogue cost more than the better fill returns: config-3 128 -> 128 @ 4 x 48^3 0.408 ms as
                //  2304 half-K blocks of 3x8x16, 0.331 ms as 864 blocks of 4x8x16; 256 -> 256 equal either way; profiles/r04_notes.md.
                //  CTSI_CONV_K32_SK384_MIN: that threshold, for A/B timing)
                const char* skc = getenv("CTSI_CONV_K32_SK384_MIN");
                const bool sk_ok = p->Cin % 128 == 0 && d.c1 % 16 == 0 && d.c2 % 16 == 0;
                const bool sk_deep = p->Cin >= (skc ? atoi(skc) : 256);
                const double cur2 = use384 ? score(3, 4, 32, 1.1) : cur;
                bool use_sk = sk_ok && sk_deep && score_sk() > cur2;
                if (sk && !strcmp(sk, "0")) use_sk = false;
                if (sk && !strcmp(sk, "1") && sk_ok) use_sk = true;
                // 2-way split-K on the 4x4x32 tile: pays where the 512-voxel grid fills the CUs badly AND K is deep enough to
                // amortise the parked accumulators (256 KB per tile): 1024 -> 512 @48x32x32 1325 -> 1400 TFLOP/s against the
                // 384-voxel tile, 512 -> 512 +0.5 %, 256 -> 512 -7 % (profiles/r02_notes.md).  Round 4: with the direct-store
                // epilogue (which the split-K form takes and the plain 384-voxel tile does not) 512 -> 512 is +5.6 % (1260 ->
                // 1330 TFLOP/s, profiles/r04_notes.md), 256 -> 512 still -3 %: the threshold is 512 input channels now
                auto fill = [&](long long b) { return (double)b / (double)(((b + 255) / 256) * 256); };
                const long long b512 = (long long)d.n * ceil_div(p->Dr, 4) * ceil_div(p->Hr, 4) * ceil_div(p->Wr, 32) * ceil_div(d.cout, 128);
                const char* skm = getenv("CTSI_CONV_K32_SK512_MIN");   // tuning aid: least input channels for this form (A/B timing)
                const bool sk512 = sk_ok && p->Cin >= (skm ? atoi(skm) : 512) && p->Wr % 32 == 0 && p->Hr % 4 == 0 && p->Dr % 4 == 0 &&
                                   fill(b512) < 0.8 && fill(2 * b512) >= 0.95 && !(sk && !strcmp(sk, "0"));
                if ((sk && !strcmp(sk, "512") && sk_ok) || (sk512 && !(sk && (!strcmp(sk, "1") || !strcmp(sk, "plain"))))) {
                    p->halo3 = 7;
                    p->BM = 512;
                    p->h32_w16 = 0;
                    p->m512_w16 = 0;
                    p->ksplit = 2;
                } else
                if (use_sk || (sk && !strcmp(sk, "plain"))) {
                    p->halo3 = 7;
                    p->BM = 384;
                    p->h32_w16 = 0;
                    p->m512_w16 = 5;
                    p->ksplit = use_sk ? 2 : 1;
                }
            }
        }
        if (k3 && (c32 || c16) && d.cout >= 64 && d.cout % 8 == 0 && p->CinW == p->Cin && extent < 2.0e9 && !getenv("CTSI_CONV_NO_HALO3")) {
            // 24- and 12-wide planes (the 48 x 24^2 / 48 x 12^2 levels of 192^2 patches: config 1, config 3, stitching windows): 16- and
            // 32-wide tiles cover them at 75 %.  The k32 kernel's 384-voxel tiles 4x4x24 / 8x4x12 (A tiles that straddle W-lines,
            // conv3_halo_k32.hip) cover them whole.  Same score as above (useful rows x fill of the 256 CUs x relative efficiency);
            // taken only when clearly ahead: 25 stitching windows at once 113 -> 101 ms per U-Net evaluation (L1 + L2 convs 57.9 ->
            // 46.2 ms), while at B = 4 (config 3) the fewer, larger tiles fill the CUs worse (L2: 288 blocks = 1.1 rounds) and
            // the 16-wide tiles stay (measured equal / slower: profiles/r04_notes.md).  CTSI_CONV_K32_NARROW = 0 / 1: never /
            // wherever the plane divides.
            auto sc = [&](int td, int th, int tw, double eff, int kmul) {
                const long long t = (long long)d.n * ceil_div(p->Dr, td) * ceil_div(p->Hr, th) * ceil_div(p->Wr, tw);
                const long long b = t * ceil_div(d.cout, 128) * kmul;
                const double useful = (double)rows * d.n / ((double)t * td * th * tw);
                return useful * (double)b / (double)(((b + 255) / 256) * 256) * eff;
            };
            double cur;
            if (p->halo3 == 7) {
                const int km = p->ksplit == 2 ? 2 : 1;
                cur = p->m512_w16 == 5 ? sc(3, 8, 16, 1.1, km) : p->m512_w16 == 3 ? sc(3, 4, 32, 1.1, 1)
                      : p->m512_w16 == 2 ? sc(4, 8, 16, 1.15, 1) : sc(4, 4, 32, 1.15, km);
            } else if (p->halo3 == 2) {
                cur = p->h32_w16 == 2 ? sc(3, 4, 16, 0.95, 1) : p->h32_w16 == 1 ? sc(4, 4, 16, 1.0, 1) : sc(4, 2, 32, 1.0, 1);
            } else {
                cur = 0.6;      // the gather kernel (halo tiles wasted > 30 % of their rows)
            }
            const char* nw = getenv("CTSI_CONV_K32_NARROW");
            const bool off = nw && !strcmp(nw, "0"), force = nw && !strcmp(nw, "1");
            struct { int td, th, tw, code; } cand[2] = {{4, 4, 24, 6}, {8, 4, 12, 7}};
            // 2-way split-K on these tiles where it fills the CUs better (B = 4, 48 x 24^2 x 256 couts: 576 blocks = 2.25 rounds ->
            // 1152 = 4.5; 48 x 12^2 x 512 couts: 288 -> 576); CTSI_CONV_K32_NARROW_SK = 0 / 1: never / wherever K allows
            const char* nsk = getenv("CTSI_CONV_K32_NARROW_SK");
            const char* skc = getenv("CTSI_CONV_K32_SK384_MIN");
            const bool nsk_ok = p->Cin % 128 == 0 && d.c1 % 16 == 0 && d.c2 % 16 == 0 && !(nsk && !strcmp(nsk, "0"));
            const bool nsk_deep = p->Cin >= (skc ? atoi(skc) : 256);
            for (auto& c : cand) {
                if (off || p->halo3 == 6 || p->Wr % c.tw != 0 || p->Wr % 16 == 0) continue;
                const double s1 = sc(c.td, c.th, c.tw, 1.07, 1), s2 = nsk_ok ? sc(c.td, c.th, c.tw, 1.07, 2) : 0.0;
                const bool sk2 = nsk_ok && ((nsk && !strcmp(nsk, "1")) || (nsk_deep && s2 > 1.1 * s1));
                if (force || (sk2 ? s2 : s1) > 1.05 * cur) {
                    p->halo3 = 7;
                    p->BM = 384;
                    p->BN = 128;
                    p->h32_w16 = 0;
                    p->m512_w16 = c.code;
                    p->ksplit = sk2 ? 2 : 0;
                    break;
                }
            }
        }
        // ConvTranspose3d (3,4,4) / (1,2,2) on the k32 kernel: each parity class is a 12-tap convolution on the input grid with
        // the 3x3x3 conv's halo tile (conv3_halo_k32.hip, TR = true); CTSI_CONV_K32T=0 keeps the gather kernel (A/B timing)
        if (d.transposed && d.c2 == 0 && d.c1 % 16 == 0 && d.cout >= 64 && d.cout % 8 == 0 && p->CinW == p->Cin &&
            extent < 2.0e9) {
            // tile: useful fraction of the tile rows x fill of the 256 CUs (4 classes x n-tiles blocks per input tile) x relative
            // efficiency (384-voxel tiles 0.96)
            // (4x4x24 / 8x4x12: the straddling 384-voxel tiles for 24- / 12-wide input planes, only where the plane divides:
            //  25 stitching windows at once, Upsample + Downsample layers 12.4 -> 11.1 ms; CTSI_CONV_K32_NARROW=0: never)
            const bool narrow_off = getenv("CTSI_CONV_K32_NARROW") && !strcmp(getenv("CTSI_CONV_K32_NARROW"), "0");
            struct { int td, th, tw, code; double eff; } cand[6] = {{4, 4, 32, 0, 1.0}, {4, 8, 16, 2, 1.0}, {3, 4, 32, 3, 0.96},
                                                                    {3, 8, 16, 5, 0.96}, {4, 4, 24, 6, 0.96}, {8, 4, 12, 7, 0.96}};
            double best = -1.0, best_useful = 0.0;
            int best_code = 0;
            for (auto& c : cand) {
                if (c.code >= 6 && (p->Wr % c.tw != 0 || p->Wr % 16 == 0 || narrow_off)) continue;
                const long long t = (long long)d.n * ceil_div(p->Dr, c.td) * ceil_div(p->Hr, c.th) * ceil_div(p->Wr, c.tw);
                const long long b = t * 4 * ceil_div(d.cout, 128);
                const double useful = (double)rows * d.n / ((double)t * c.td * c.th * c.tw);
                const double sc = useful * (double)b / (double)(((b + 255) / 256) * 256) * c.eff;
                if (sc > best) {
                    best = sc;
                    best_code = c.code;
                    best_useful = useful;
                }
            }
            const char* kt = getenv("CTSI_CONV_K32T");
            const bool force = getenv("CTSI_CONV_FORCE_HALO3") != nullptr;
            if ((best_useful >= 0.7 || force) && !(kt && !strcmp(kt, "0")) && !getenv("CTSI_CONV_NO_HALO3")) {
                p->halo3 = 7;
                p->BN = 128;
                p->m512_w16 = best_code;
                const char* w16 = getenv("CTSI_CONV_M512W16");    // "0" | "1" (tuning / test aid: 4x4x32 / 4x8x16)
                if (w16 && !strcmp(w16, "1")) p->m512_w16 = 2;
                if (w16 && !strcmp(w16, "0")) p->m512_w16 = 0;
                const char* t384 = getenv("CTSI_CONV_K32_384");   // "1": a 384-voxel tile of that width (test aid)
                if (t384 && !strcmp(t384, "1")) p->m512_w16 = (p->m512_w16 == 2 || p->m512_w16 == 5) ? 5 : 3;
                if (t384 && !strcmp(t384, "0")) p->m512_w16 = (p->m512_w16 == 2 || p->m512_w16 == 5) ? 2 : 0;
                p->BM = (p->m512_w16 == 3 || p->m512_w16 >= 5) ? 384 : 512;
            }
        }
        // Strided Conv3d (3,4,4) / (1,2,2) / pad 1 (Downsample3D, the VAE encoder's DownsampleBlock) on the k32 kernel: the four
        // input-parity sub-grids are 3x2x2-tap stride-1 convolutions on the OUTPUT grid's halo tile (conv3_halo_k32.hip, DS =
        // true).  CTSI_CONV_K32D=0 keeps the gather kernel (A/B timing).  Tile by the same score as the ConvTranspose form;
        // levels whose grid stays below one round of the 256 CUs with deep K take the 2-way split-K form of the 3x8x16 tile.
        if (!d.transposed && d.kd == 3 && d.kh == 4 && d.kw == 4 && d.sh == 2 && d.sw == 2 && d.pd == 1 && d.ph == 1 && d.pw == 1 &&
            d.c2 == 0 && d.c1 % 16 == 0 && d.cout >= 64 && d.cout % 8 == 0 && p->CinW == p->Cin && d.hi % 2 == 0 && d.wi % 2 == 0 &&
            extent < 2.0e9) {
            struct { int td, th, tw, code; double eff; } cand[6] = {{4, 4, 32, 0, 1.0}, {4, 8, 16, 2, 1.0}, {3, 4, 32, 3, 0.96},
                                                                    {3, 8, 16, 5, 0.96}, {4, 4, 24, 6, 0.96}, {8, 4, 12, 7, 0.96}};
            const bool narrow_off = getenv("CTSI_CONV_K32_NARROW") && !strcmp(getenv("CTSI_CONV_K32_NARROW"), "0");
            double best = -1.0, best_useful = 0.0;
            int best_code = 0;
            long long best_blocks = 0;
            for (auto& c : cand) {
                if (c.code >= 6 && (p->Wr % c.tw != 0 || p->Wr % 16 == 0 || narrow_off)) continue;
                const long long t = (long long)d.n * ceil_div(p->Dr, c.td) * ceil_div(p->Hr, c.th) * ceil_div(p->Wr, c.tw);
                const long long b = t * ceil_div(d.cout, 128);
                const double useful = (double)rows * d.n / ((double)t * c.td * c.th * c.tw);
                const double sc = useful * (double)b / (double)(((b + 255) / 256) * 256) * c.eff;
                if (useful < 0.7 && !getenv("CTSI_CONV_FORCE_HALO3")) continue;   // (a 32-wide tile on a 16-wide plane)
                if (sc > best) {
                    best = sc;
                    best_code = c.code;
                    best_useful = useful;
                    best_blocks = b;
                }
            }
            const char* kd_ = getenv("CTSI_CONV_K32D");       // "0": gather kernel (tuning / test aid)
            const bool force = getenv("CTSI_CONV_FORCE_HALO3") != nullptr;
            if ((best_useful >= 0.7 || force) && !(kd_ && !strcmp(kd_, "0")) && !getenv("CTSI_CONV_NO_HALO3")) {
                p->halo3 = 7;
                p->ds = 1;
                p->BN = 128;
                p->m512_w16 = best_code;
                const char* w16 = getenv("CTSI_CONV_M512W16");    // "0" | "1" (tuning / test aid: 4x4x32 / 4x8x16)
                if (w16 && !strcmp(w16, "1")) p->m512_w16 = 2;
                if (w16 && !strcmp(w16, "0")) p->m512_w16 = 0;
                const char* t384 = getenv("CTSI_CONV_K32_384");   // "1": a 384-voxel tile of that width (test aid)
                if (t384 && !strcmp(t384, "1")) p->m512_w16 = (p->m512_w16 == 2 || p->m512_w16 == 5) ? 5 : 3;
                if (t384 && !strcmp(t384, "0")) p->m512_w16 = (p->m512_w16 == 2 || p->m512_w16 == 5) ? 2 : 0;
                // split-K: 3x8x16 tiles, two blocks per (tile, n-tile) -- when even the best tile leaves the grid at <= half a
                // round of the CUs (48x16x16 x 512 couts: 128 blocks) and K is deep (48 taps x Cin)
                const char* sk = getenv("CTSI_CONV_K32_SPLITK");  // "0" | "1" (tuning / test aid)
                const long long t5 = (long long)d.n * ceil_div(p->Dr, 3) * ceil_div(p->Hr, 8) * ceil_div(p->Wr, 16) * ceil_div(d.cout, 128);
                bool use_sk = p->Cin % 32 == 0 && p->Cin >= 256 && best_blocks <= 160 && 2 * t5 <= 512;
                if (sk && !strcmp(sk, "0")) use_sk = false;
                if (sk && !strcmp(sk, "1") && p->Cin % 32 == 0) use_sk = true;
                if (use_sk) {
                    p->m512_w16 = 5;
                    p->ksplit = 2;
                }
                p->BM = (p->m512_w16 == 3 || p->m512_w16 >= 5) ? 384 : 512;
            }
        }
        // few output channels (network heads: 128 -> 8, 128 -> 1): halo tile 4x2x16 x 16 couts, see conv3_head.hip
        const long long padded_h = (long long)ceil_div(p->Dr, 4) * ceil_div(p->Hr, 2) * ceil_div(p->Wr, 16) * 128;
        if (!p->halo3 && k3 && !p->small && d.c2 == 0 && d.c1 % 32 == 0 && d.cout <= 16 && (rows * 10 >= padded_h * 7 || getenv("CTSI_CONV_FORCE_HALO3")) &&
            extent < 2.0e9 && !getenv("CTSI_CONV_NO_HEAD3")) {
            p->halo3 = 6;
            p->BM = 128;
            p->BN = 16;
            p->head2 = ctsi_conv3_head2_supported(p->Cin, d.cout) && p->CinW == p->Cin && !getenv("CTSI_CONV_NO_HEAD2");
        }
    }
    p->CoutPad = ceil_div(d.cout, p->BN) * p->BN;
    p->ntiles_n = p->CoutPad / p->BN;
    if (p->halo3 == 7 && p->m512_w16 == 6) {
        p->TD = 4; p->TH = 4; p->TW = 24;
    } else if (p->halo3 == 7 && p->m512_w16 == 7) {
        p->TD = 8; p->TH = 4; p->TW = 12;
    } else if (p->halo3 == 7 && p->m512_w16 == 5) {
        p->TD = 3; p->TH = 8; p->TW = 16;
    } else if (p->halo3 == 7 && p->m512_w16 == 3) {
        p->TD = 3; p->TH = 4; p->TW = 32;
    } else if (p->halo3 == 7 && p->m512_w16 == 2) {
        p->TD = 4; p->TH = 8; p->TW = 16;
    } else if (p->halo3 == 7) {
        p->TD = 4; p->TH = 4; p->TW = 32;
    } else if (p->halo3 == 2) {
        p->TD = p->h32_w16 == 2 ? 3 : 4; p->TH = p->h32_w16 ? 4 : 2; p->TW = p->h32_w16 ? 16 : 32;
    } else if (p->halo3 == 6) {
        p->TD = 4; p->TH = 2; p->TW = 16;
    } else {
        choose_tile(p->BM, p->Dr, p->Hr, p->Wr, &p->TD, &p->TH, &p->TW);
    }
    p->lTH = ilog2(p->TH);
    p->lTW = ilog2(p->TW);
    p->tilesD = ceil_div(p->Dr, p->TD);
    p->tilesH = ceil_div(p->Hr, p->TH);
    p->tilesW = ceil_div(p->Wr, p->TW);
    p->tps = p->tilesD * p->tilesH * p->tilesW;
    if (!p->halo3) {
        // gather kernel on small planes: a power-of-two box tile over e.g. a 6 x 6 plane is 44 % padding rows; runs of
        // BM consecutive voxels have none (only the last tile of a sample is ragged)
        const long long rows = (long long)p->Dr * p->Hr * p->Wr;
        const long long boxed = (long long)p->tps * p->BM;
        const char* lin = getenv("CTSI_CONV_LINEAR");   // "0" | "1" (tuning aid)
        if ((boxed * 100 > rows * 115 && !(lin && !strcmp(lin, "0"))) || (lin && !strcmp(lin, "1"))) {
            p->linear = 1;
            p->tps = (int)((rows + p->BM - 1) / p->BM);
            p->TD = (int)(p->BM / ((long long)p->Hr * p->Wr)) + 2;   // depth slices one tile can touch (fast-path extent)
        }
    }
    p->mtiles = d.n * p->tps;
    {   // buffer-addressed fast path: whole 64-channel chunks per source and a tile halo that fits 2^31 bytes
        const int cmax = d.c1 > d.c2 ? d.c1 : d.c2;
        const double extent = ((double)(p->TD + 4) * d.hi * d.wi + 2.0 * d.wi + 8) * cmax * 2.0 * (p->sH > 1 ? 1 : 1);
        p->fast = !p->small && d.c1 % 64 == 0 && d.c2 % 64 == 0 && extent < 2.0e9 && !getenv("CTSI_CONV_NO_FAST");
    }
    p->gsplit = 0;
    if (!p->halo3 && !p->small && p->BM == 128 && p->BN == 128) {
        // S-way split-K on the gather kernel: a layer that is a few dozen blocks (the 6 x 6 level of ONE 192^2 patch: 14 m-tiles
        // x 4 n-tiles = 56 blocks on 256 CUs) with hundreds of sequential K-steps is bound by its K-step latency; S blocks
        // per tile walk 1 / S of the steps each (csrc/conv_mfma.hip, hand-off by ticket).  CTSI_CONV_GSPLIT = 0 | 2..8 overrides.
        const long long blocks = (long long)p->nclass * p->mtiles * p->ntiles_n;
        int S = 0;
        if (blocks * 2 <= 256 && p->ksteps >= 32) {
            S = (int)(256 / blocks);
            if (S > 4) S = 4;
            while (S > 1 && p->ksteps / S < 16) --S;
        } else if (blocks <= 256 && p->ksteps >= 64) {
            S = 2;     // 129-256 blocks: the grid doubles past the ring mode's one-block-per-CU limit, so the two half-K blocks of
        }              // a tile share a CU in the 2-stage mode: 560 -> 700 TFLOP/s on the 6 x 6 level of config 3 (B = 4)
        const char* gs = getenv("CTSI_CONV_GSPLIT");
        if (gs) S = atoi(gs) >= 2 && atoi(gs) <= 8 && p->ksteps >= atoi(gs) ? atoi(gs) : 0;
        p->gsplit = S >= 2 ? S : 0;
    }
    if (!d.transposed)
        p->flops = 2.0 * d.n * (double)p->Do * p->Ho * p->Wo * p->Cin * d.cout * KK;
    else
        p->flops = 2.0 * d.n * (double)di_own * d.hi * d.wi * p->Cin * d.cout * KK;
    *out = p;
    return CTSI_OK;
}

extern "C" void ctsi_conv_plan_destroy(ctsi_conv_plan* plan) { free(plan); }

// few-cout heads (halo3 == 6): the packed buffer holds conv3_head_kernel's image, padded to 256 B, then conv3_head2_kernel's
static size_t head1_bytes(const ctsi_conv_plan* p) {
    return ((size_t)p->Cin * 27 * (p->d.cout <= 8 ? 8 : 16) * 2 + 1024 + 255) / 256 * 256;
}

extern "C" int ctsi_conv_plan_out_dims(const ctsi_conv_plan* p, int* d, int* h, int* w) {
    CTSI_CHECK_ARG(p, "ctsi_conv_plan_out_dims: null plan");
    if (d) *d = p->Do;
    if (h) *h = p->Ho;
    if (w) *w = p->Wo;
    return CTSI_OK;
}
extern "C" size_t ctsi_conv_plan_weight_bytes(const ctsi_conv_plan* p) {
    if (!p) return 0;
    if (p->stem) return ctsi_conv3_stem_weight_bytes(p->CoutPad);
    if (p->stream1) return (size_t)p->d.cout * p->Cin * 2;
    if (p->halo3 == 6)   // head kernels: conv3_head's image (8 weight rows when cout <= 8; + 1 KB: its last DMA piece is read whole),
        return head1_bytes(p) + (ctsi_conv3_head2_supported(p->Cin, p->d.cout) ? ctsi_conv3_head2_weight_bytes(p->d.cout) : 0);   // then conv3_head2's
    if (p->halo3 == 7) return ctsi_conv3_halo_k32_weight_bytes(p->Cin, p->CoutPad, p->BN, p->ds ? 2 : p->d.transposed);   // entries padded to whole steps
    if (p->halo3) return (size_t)p->Cin * 27 * p->CoutPad * 2;   // [chunk][27][cout_pad][32 | 16 ch] bf16
    return (size_t)p->nclass * p->CoutPad * p->Ktot * 2;
}
extern "C" int ctsi_conv_plan_tiles(const ctsi_conv_plan* p) { return p ? p->nclass * p->mtiles : 0; }
extern "C" int ctsi_conv_plan_tiles_per_sample(const ctsi_conv_plan* p) { return p ? p->tps : 0; }
extern "C" int ctsi_conv_plan_cout_pad(const ctsi_conv_plan* p) { return p ? p->CoutPad : 0; }
extern "C" double ctsi_conv_plan_flops(const ctsi_conv_plan* p) { return p ? p->flops : 0.0; }
extern "C" size_t ctsi_conv_plan_workspace_bytes(const ctsi_conv_plan* p) {
    // split-K plans: tickets / flags + fp32 partial accumulators (ctsi_conv_out.workspace; zero the first 8 * tiles bytes once)
    if (p && !p->halo3 && p->gsplit >= 2) {   // gather kernel: [tile] tickets (padded to 256 B) + [tile][split][128 x 128] fp32
        const size_t tiles = (size_t)p->nclass * p->mtiles * p->ntiles_n;
        return (tiles * 4 + 255) / 256 * 256 + tiles * p->gsplit * (size_t)(128 * 128) * sizeof(float);
    }
    if (!p || p->ksplit != 2) return 0;
    return ctsi_conv3_halo_k32_splitk_bytes(p->mtiles * p->ntiles_n);
}
extern "C" int ctsi_conv_plan_config(const ctsi_conv_plan* p, int* bm, int* bn, int* mode) {
    CTSI_CHECK_ARG(p, "ctsi_conv_plan_config: null plan");
    if (bm) *bm = p->BM;
    if (bn) *bn = p->BN;
    if (mode) *mode = p->stem ? 11 : p->stream1 ? 10 : (p->halo3 ? 2 + p->halo3 : (p->small ? 1 : (p->fast ? 2 : 0)));
    if (p->stream1) {
        if (bm) *bm = 16;
        if (bn) *bn = p->stream1 * 16;
    }
    return CTSI_OK;
}

// A 1x1x1 stride-1 conv that will run with the fused GroupNorm tail (ctsi_conv_out.gn_x) or as a plain bf16 conv + bias may
// take the streaming kernel of conv1_stream.hip (another packed-weight layout: call this BEFORE ctsi_conv_plan_weight_bytes /
// _pack_weights).  on = 1 selects it where the layer qualifies (whole 128-channel chunks per source, cout in whole n-tiles)
// and is a no-op otherwise -- ctsi_conv_plan_config reports mode 10 when it is active; on = 0 returns to the gather kernel.
// CTSI_CONV1_STREAM=0 (tuning / test aid) keeps every plan on the gather kernel.
extern "C" int ctsi_conv_plan_set_stream_tail(ctsi_conv_plan* p, int on) {
    CTSI_CHECK_ARG(p, "ctsi_conv_plan_set_stream_tail: null plan");
    p->stream1 = 0;
    const char* e = getenv("CTSI_CONV1_STREAM");
    if (!on || (e && atoi(e) == 0)) return CTSI_OK;
    const ctsi_conv_desc& d = p->d;
    if (d.transposed || d.kd != 1 || d.kh != 1 || d.kw != 1 || d.sh != 1 || d.sw != 1 || d.pd || d.ph || d.pw || p->dshift)
        return CTSI_OK;
    p->stream1 = ctsi_conv1_stream_nt(d.c1, d.c2, d.cout);
    return CTSI_OK;
}

// The weight tensor may carry fewer input channels than the (padded) activation tensor: the
// VAE encoder's first conv sees a 1-channel volume stored as 8 channels (7 zero).
extern "C" int ctsi_conv_plan_set_weight_cin(ctsi_conv_plan* p, int cin_w) {
    CTSI_CHECK_ARG(p && cin_w > 0 && cin_w <= p->Cin, "ctsi_conv_plan_set_weight_cin: bad cin %d", cin_w);
    p->CinW = cin_w;
    // a 3x3x3 stride-1 conv of a ONE-channel volume (the VAE encoder's first layer: the CT volume is stored with 8 channels,
    // 7 of them padding) with >= 64 couts: the 27 taps become the K of one MFMA (conv3_stem.hip); CTSI_CONV_NO_STEM keeps the
    // gather kernel's small-Cin form (A/B timing, tests)
    const ctsi_conv_desc& d = p->d;
    if (cin_w == 1 && !d.transposed && d.kd == 3 && d.kh == 3 && d.kw == 3 && d.sh == 1 && d.sw == 1 && d.pd == 1 && d.ph == 1 &&
        d.pw == 1 && d.c2 == 0 && d.c1 == 8 && d.cout >= 64 && d.cout % 8 == 0 && !p->dshift && !getenv("CTSI_CONV_NO_STEM")) {
        p->stem = 1;
        p->halo3 = 0;
        p->gsplit = 0;
        p->linear = 0;
        p->BM = 512;
        p->BN = 128;
        ctsi_conv3_stem_tile(&p->TD, &p->TH, &p->TW);
        p->tilesD = ceil_div(p->Dr, p->TD);
        p->tilesH = ceil_div(p->Hr, p->TH);
        p->tilesW = ceil_div(p->Wr, p->TW);
        p->tps = p->tilesD * p->tilesH * p->tilesW;
        p->mtiles = d.n * p->tps;
        p->CoutPad = ceil_div(d.cout, 128) * 128;
        p->ntiles_n = p->CoutPad / 128;
    }
    return CTSI_OK;
}

extern "C" int ctsi_conv_plan_pack_weights(const ctsi_conv_plan* p, const float* w, void* packed,
                                           void* stream) {
    CTSI_CHECK_ARG(p && w && packed, "ctsi_conv_plan_pack_weights: null argument");
    if (p->stem) return ctsi_conv3_stem_pack(w, packed, p->d.cout, p->CoutPad, p->CinW, stream);
    if (p->stream1) return ctsi_conv1_stream_pack(w, packed, p->d.cout, p->Cin, p->CinW, p->stream1, stream);
    if (p->halo3 == 7)
        return ctsi_conv3_halo_k32_pack(w, packed, p->d.cout, p->CoutPad, p->Cin, p->CinW, p->BN, p->ds ? 2 : p->d.transposed,
                                        ctsi_conv3_halo_k32_direct(p->m512_w16, p->ksplit, p->ds), stream);
    if (p->halo3 == 6) {
        hipMemsetAsync((char*)packed + head1_bytes(p) - 1280, 0, 1280, (hipStream_t)stream);
        if (ctsi_conv3_head2_supported(p->Cin, p->d.cout)) {
            const int rc = ctsi_conv3_head2_pack(w, (char*)packed + head1_bytes(p), p->d.cout, p->Cin, p->CinW, stream);
            if (rc != CTSI_OK) return rc;
        }
        return ctsi_conv3_halo_pack(w, packed, p->d.cout, p->d.cout <= 8 ? 8 : 16, p->Cin, p->CinW, stream);
    }
    if (p->halo3) return ctsi_conv3_halo_pack(w, packed, p->d.cout, p->CoutPad, p->Cin, p->CinW, stream);
    PackParams q;
    memset(&q, 0, sizeof(q));
    q.w = w;
    q.out = (bf16_t*)packed;
    q.nclass = p->nclass;
    q.T = p->T;
    q.Cout = p->d.cout;
    q.CoutPad = p->CoutPad;
    q.Cin = p->Cin;
    q.CinW = p->CinW;
    q.Ktot = p->Ktot;
    q.small = p->small;
    q.kc_per_tap = p->kc_per_tap;
    q.lcpt = p->lcpt;
    const long long KK = (long long)p->d.kd * p->d.kh * p->d.kw;
    if (!p->d.transposed) {
        q.s_co = (long long)p->CinW * KK;
        q.s_ci = KK;
    } else {
        q.s_ci = (long long)p->d.cout * KK;
        q.s_co = KK;
    }
    memcpy(q.tapk, p->tapk, sizeof(q.tapk));
    const long long total = (long long)q.nclass * q.CoutPad * q.Ktot;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(conv_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, q);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

template <int WM, int WN, int TM, int TN>
static int launch_conv(const ConvKParams& k, int mode, int grid, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr size_t lds = 2 * (BM * 128 + BN * 128) + BM * 8 + WM * BN * 8;
    constexpr int STAGE = BM * 128 + BN * 128;
    constexpr int NSTG = STAGE <= 32768 ? 4 : (STAGE <= 49152 ? 3 : 2);
    constexpr size_t lds_ring = (size_t)NSTG * STAGE + BM * 8 + WM * BN * 8;
    static CtsiPerDeviceOnce attr_once;
    if (attr_once.first()) {
        hipFuncSetAttribute((const void*)conv_gather_mfma_kernel<WM, WN, TM, TN, 3>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ring);
        hipFuncSetAttribute((const void*)conv_gather_mfma_kernel<WM, WN, TM, TN, 0>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)conv_gather_mfma_kernel<WM, WN, TM, TN, 1>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)conv_gather_mfma_kernel<WM, WN, TM, TN, 2>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const dim3 g(grid), b(WM * WN * 64);
    if (mode == 1)
        hipLaunchKernelGGL((conv_gather_mfma_kernel<WM, WN, TM, TN, 1>), g, b, lds, st, k);
    else if (mode == 3)
        hipLaunchKernelGGL((conv_gather_mfma_kernel<WM, WN, TM, TN, 3>), g, b, lds_ring, st, k);
    else if (mode == 2)
        hipLaunchKernelGGL((conv_gather_mfma_kernel<WM, WN, TM, TN, 2>), g, b, lds, st, k);
    else
        hipLaunchKernelGGL((conv_gather_mfma_kernel<WM, WN, TM, TN, 0>), g, b, lds, st, k);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

extern "C" int ctsi_conv_fwd(const ctsi_conv_plan* p, const void* x1, const void* x2, const void* packed_w,
                             const float* bias, const ctsi_conv_out* o, void* stream) {
    CTSI_CHECK_ARG(p && x1 && packed_w && o && o->y, "ctsi_conv_fwd: null argument");
    CTSI_CHECK_ARG(p->d.c2 == 0 || x2 != nullptr, "ctsi_conv_fwd: plan has two sources but x2 is null");
    CTSI_CHECK_ARG(o->mode == 0 || o->mode == 1, "ctsi_conv_fwd: bad output mode %d", o->mode);
    if (o->mode == 0) {
        CTSI_CHECK_ARG(p->d.cout % 8 == 0 && o->cout_stride % 8 == 0 && o->c_off % 8 == 0 &&
                           o->cout_stride >= o->c_off + p->d.cout,
                       "ctsi_conv_fwd: bf16 output needs cout, cout_stride, c_off multiples of 8 "
                       "(cout=%d stride=%d off=%d)", p->d.cout, o->cout_stride, o->c_off);
    }
    if (p->stem && o->mode == 0 && o->act == 0 && o->gn_x == nullptr) {
        StemParams q;
        memset(&q, 0, sizeof(q));
        q.x = (const bf16_t*)x1;
        q.w = (const bf16_t*)packed_w;
        q.bias = bias;
        q.y = (bf16_t*)o->y;
        q.colsum = (float*)o->colsum;
        q.cx = p->d.c1;
        q.D = p->d.di; q.H = p->d.hi; q.W = p->d.wi;
        q.Cout = p->d.cout; q.CoutPad = p->CoutPad;
        q.cout_stride = o->cout_stride; q.c_off = o->c_off;
        q.tilesD = p->tilesD; q.tilesH = p->tilesH; q.tilesW = p->tilesW; q.tps = p->tps; q.mtiles = p->mtiles;
        return ctsi_conv3_stem_launch(&q, stream);
    }
    CTSI_CHECK_ARG(!p->stem, "ctsi_conv_fwd: a one-channel stem plan writes bf16 without activation or fused tail");
    if (p->stream1) {
        CTSI_CHECK_ARG(o->mode == 0 && o->act == 0 && o->colsum == nullptr,
                       "ctsi_conv_fwd: a streaming 1x1x1 plan writes bf16 without activation or column sums");
        Conv1StreamParams q;
        memset(&q, 0, sizeof(q));
        q.x1 = (const bf16_t*)x1;
        q.x2 = (const bf16_t*)(x2 ? x2 : x1);
        q.w = (const bf16_t*)packed_w;
        q.bias = bias;
        q.h = (const bf16_t*)o->gn_x;
        q.y = (bf16_t*)o->y;
        if (o->gn_x != nullptr) {
            CTSI_CHECK_ARG(o->gn_sums && o->gn_gamma && o->gn_beta && o->gn_groups > 0 && p->d.cout % o->gn_groups == 0 &&
                               o->gn_count > 0,
                           "ctsi_conv_fwd: bad fused GroupNorm arguments (groups=%d, cout=%d)", o->gn_groups, p->d.cout);
            q.gn_sums = (const double*)o->gn_sums;
            q.gn_gamma = (const float*)o->gn_gamma;
            q.gn_beta = (const float*)o->gn_beta;
            q.gn_groups = o->gn_groups;
            q.gn_eps = o->gn_eps;
            q.gn_count = (double)o->gn_count;
            q.gn_silu = o->gn_silu;
        }
        q.C1 = p->d.c1; q.C2 = p->d.c2; q.Cout = p->d.cout;
        q.cout_stride = o->cout_stride; q.c_off = o->c_off;
        q.V = (long long)p->Do * p->Ho * p->Wo;
        return ctsi_conv1_stream_launch(&q, p->d.n, p->stream1, stream);
    }
    if (o->gn_x != nullptr) {
        CTSI_CHECK_ARG(!p->halo3 && o->mode == 0 && o->act == 0 && o->colsum == nullptr && p->nclass == 1,
                       "ctsi_conv_fwd: the fused GroupNorm tail needs a gather-kernel plan, bf16 output, no act / colsum");
        CTSI_CHECK_ARG(o->gn_sums && o->gn_gamma && o->gn_beta && o->gn_groups > 0 && p->d.cout % o->gn_groups == 0 &&
                           o->gn_count > 0,
                       "ctsi_conv_fwd: bad fused GroupNorm arguments (groups=%d, cout=%d)", o->gn_groups, p->d.cout);
    }
    if (p->halo3 && (p->halo3 == 6 || (o->mode == 0 && o->act == 0))) {
        Conv3HaloParams h;
        memset(&h, 0, sizeof(h));
        h.x1 = (const bf16_t*)x1;
        h.x2 = (const bf16_t*)(x2 ? x2 : x1);
        h.w = (const bf16_t*)packed_w;
        h.bias = bias;
        h.y = o->y;
        h.colsum = o->colsum;
        h.C1 = p->d.c1; h.C2 = p->d.c2;
        h.Di = p->d.di; h.Hi = p->d.hi; h.Wi = p->d.wi;
        h.Do = p->Do; h.Ho = p->Ho; h.Wo = p->Wo;
        h.dshift = p->dshift;
        h.tilesD = p->tilesD; h.tilesH = p->tilesH; h.tilesW = p->tilesW; h.tps = p->tps; h.mtiles = p->mtiles;
        h.ntiles_n = p->ntiles_n;
        h.nchunks = p->Cin / (p->halo3 == 7 ? 16 : 32) * (p->ds ? 4 : 1);   // (Downsample form: 4 virtual chunks per 16 channels)
        h.ds = p->ds;
        h.Cout = p->d.cout; h.CoutPad = p->CoutPad;
        h.cout_stride = o->cout_stride; h.c_off = o->c_off;
        {   // k32 kernel: streaming (non-temporal) output stores for tensors the next pass cannot find in a cache anyway (>= 128 MB:
            // the U-Net's 48 x 128^2 level, the VAE's full-resolution levels): the tile's 128 KB no longer displace halos and
            // weight slabs from the XCD's 4 MB L2 (128->128 @48x128^2 1318 -> 1343 TFLOP/s; step 25.90 -> 25.78 ms with the
            // following gn_apply 0.04-0.1 ms slower).  CTSI_CONV_NT_STORE=0 / 1 overrides.
            static const char* nts = getenv("CTSI_CONV_NT_STORE");
            const long long out_bytes = (long long)p->d.n * p->Do * p->Ho * p->Wo * o->cout_stride * 2;
            h.nt_store = nts ? atoi(nts) != 0 : out_bytes >= (128ll << 20);
        }
        {   // n-major block order (one n-tile's 3.5 MB weight slab at a time per XCD instead of all of them: the 48x32x32 /
            // 512-cout layers move 508 MB of HBM traffic per launch against 114 MB algorithmic because 14 MB of weights
            // thrash the 4 MB L2).  Interleaved timing shows no speed difference (1214 vs 1211 TFLOP/s), so it stays opt-in.
            const char* nm = getenv("CTSI_CONV_NMAJOR");   // "1" (tuning aid; read per launch)
            h.n_major = nm ? atoi(nm) : 0;
            // (ConvTranspose on the k32 kernel: number of sibling (class, n-tile) blocks of an input tile kept adjacent;
            //  4 = one group's weight slabs fit an XCD's L2: +0.5-1 % over all 8 / 16 siblings adjacent)
            if (p->d.transposed && !nm) h.n_major = 4;
        }
        {
            h.dbg = ctsi_debug_flags();                 // release library: the timeline bit only (ctsi_internal.h)
            h.nchunks = ctsi_debug_ksteps(h.nchunks);   // ablation builds only: truncate the chunk loop
        }
        {   // k32 kernel, block -> tile order: 8 x 4 super-tiles inside a depth band (the kernel falls back to the plain order where
            // the tile grid does not divide): on 512-wide planes the 32 blocks an XCD runs at a time then share H and W halos in its
            // L2 -- FETCH_SIZE of the VAE decoder's k32 launches -34 %, decode -0.7 ms (profiles/r04_notes.md); identical to the
            // plain order on the U-Net's 128-wide level (4 tiles per row).  CTSI_CONV_TILE_ORDER = 0 / 1 / 2 overrides (A/B timing)
            const char* to = getenv("CTSI_CONV_TILE_ORDER");
            h.tile_order = to ? atoi(to) : 2;
            const char* eb = getenv("CTSI_CONV_EPI_BARRIER");   // "1": A/B timing of the barrier the direct epilogue drops (read per launch)
            h.dbg_epi_barrier = eb ? atoi(eb) : 0;
        }
        if (p->halo3 == 6) {
            // second form (taps as the GEMM's N dimension, input read once straight into the MFMA layout: conv3_head2.hip) for the
            // layers it covers; conv3_head_kernel keeps the others and the GroupNorm column sums.  CTSI_CONV_NO_HEAD2: A/B timing
            if (p->head2 && o->colsum == nullptr)
                return ctsi_conv3_head2_launch(&h, p->d.n, (const char*)packed_w + head1_bytes(p), o->mode, o->act, o->sn, o->sc, o->sd,
                                               o->sh, o->sw, stream);
            return ctsi_conv3_head_launch(&h, p->d.cout <= 8 ? 8 : 16, o->mode, o->act, o->sn, o->sc, o->sd, o->sh, o->sw, stream);
        }
        h.tr = p->d.transposed;
        if (p->halo3 == 7 && p->ksplit == 2) {
            CTSI_CHECK_ARG(o->workspace, "ctsi_conv_fwd: this plan needs ctsi_conv_out.workspace (ctsi_conv_plan_workspace_bytes)");
            const int tiles = p->mtiles * p->ntiles_n;
            h.ksplit = 2;
            h.sk_sync = (int*)o->workspace;
            h.sk_ws = (float*)((char*)o->workspace + ((size_t)tiles * 8 + 255) / 256 * 256);
        }
        if (p->halo3 == 7) return ctsi_conv3_halo_k32_launch(&h, p->m512_w16, p->BN, stream);
        return ctsi_conv3_halo_launch(&h, p->h32_w16 == 2 ? 4 : (p->h32_w16 ? 3 : 1), stream);
    }
    CTSI_CHECK_ARG(!p->halo3, "ctsi_conv_fwd: the 3x3x3 halo-tile plan supports bf16 NDHWC output without activation");
    ConvKParams k;
    memset(&k, 0, sizeof(k));
    k.x1 = (const bf16_t*)x1;
    k.x2 = (const bf16_t*)(x2 ? x2 : x1);
    k.w = (const bf16_t*)packed_w;
    k.bias = bias;
    k.y = o->y;
    k.colsum = o->colsum;
    k.C1 = p->d.c1; k.C2 = p->d.c2; k.Cin = p->Cin;
    k.Di = p->d.di; k.Hi = p->d.hi; k.Wi = p->d.wi;
    k.Dr = p->Dr; k.Hr = p->Hr; k.Wr = p->Wr;
    k.sH = p->sH; k.sW = p->sW;
    k.lTH = p->lTH; k.lTW = p->lTW;
    k.linear = p->linear;
    k.tilesD = p->tilesD; k.tilesH = p->tilesH; k.tilesW = p->tilesW; k.tps = p->tps; k.mtiles = p->mtiles;
    k.ntiles_n = p->ntiles_n;
    k.T = p->T; k.ksteps = p->ksteps; k.kc_per_tap = p->kc_per_tap; k.lcpt = p->lcpt;
    k.Cout = p->d.cout; k.CoutPad = p->CoutPad; k.Ktot = p->Ktot;
    k.Do = p->Do; k.Ho = p->Ho; k.Wo = p->Wo; k.uH = p->uH; k.uW = p->uW;
    k.nclass = p->nclass;
    k.out_mode = o->mode; k.cout_stride = o->cout_stride; k.c_off = o->c_off; k.act = o->act;
    k.osn = o->sn; k.osc = o->sc; k.osd = o->sd; k.osh = o->sh; k.osw = o->sw;
    memcpy(k.tapdelta, p->tapdelta, sizeof(k.tapdelta));
    k.NA = p->NA; k.NB = p->NB; k.NC = p->NC;
    memcpy(k.ad, p->ad, sizeof(k.ad));
    memcpy(k.bh, p->bh, sizeof(k.bh));
    memcpy(k.cw, p->cw, sizeof(k.cw));
    for (int c = 0; c < 4; ++c) { k.pH[c] = p->pH[c]; k.pW[c] = p->pW[c]; }
    int grid = p->nclass * p->mtiles * p->ntiles_n;
    if (p->gsplit >= 2 && o->workspace != nullptr && !(p->BM != 128 || p->BN != 128)) {
        const size_t tiles = (size_t)grid;
        k.ksplit = p->gsplit;
        k.sk_sync = (int*)o->workspace;
        k.sk_ws = (float*)((char*)o->workspace + (tiles * 4 + 255) / 256 * 256);
        grid *= p->gsplit;
    }
    hipStream_t st = (hipStream_t)stream;
    k.ksteps = ctsi_debug_ksteps(k.ksteps);   // (ablation builds only: truncate the K loop to price prologue + epilogue)
    k.dbg = ctsi_debug_flags();
    k.gn_x = (const bf16_t*)o->gn_x;
    k.gn_sums = o->gn_sums;
    k.gn_gamma = o->gn_gamma;
    k.gn_beta = o->gn_beta;
    k.gn_groups = o->gn_groups;
    k.gn_silu = o->gn_silu;
    k.gn_eps = o->gn_eps;
    k.gn_count = (double)o->gn_count;
    k.dshift = p->dshift;
    memcpy(k.tap_margin, p->tap_margin, sizeof(k.tap_margin));
    memcpy(k.ad_min, p->ad_min, sizeof(k.ad_min));
    int mode = p->small ? 1 : (p->fast ? 2 : 0);
    {   // deep-ring variant for launches that cannot hide DMA latency behind other blocks of the same CU
        const char* ring = getenv("CTSI_CONV_RING");   // "0" | "1" (tuning aid; read per launch)
        const bool small_grid = grid <= 256 && !(p->BM == 256 && p->BN == 256);   // at most one block per CU: the ring's
                                                                                   // 128-144 KB of LDS evict no second block
        if (mode == 2 && ((small_grid && !(ring && !strcmp(ring, "0"))) || (ring && !strcmp(ring, "1")))) mode = 3;
    }
    if (p->BM == 256 && p->BN == 256) return launch_conv<2, 4, 4, 2>(k, mode, grid, st);
    if (p->BM == 256 && p->BN == 128) return launch_conv<4, 2, 2, 2>(k, mode, grid, st);
    if (p->BN == 128) return launch_conv<2, 2, 2, 2>(k, mode, grid, st);
    return launch_conv<4, 1, 1, 1>(k, mode, grid, st);
}
