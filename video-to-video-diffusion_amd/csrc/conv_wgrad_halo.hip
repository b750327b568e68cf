// Weight gradient of 3x3x3 stride-1 'same' convolutions as a HALO-TILE kernel (round 3): the forward kernel's data movement
// (conv3_halo_k32.hip) with the roles of M and K exchanged.
//
//   dW[t][co][ci] = sum over voxels v of  dY[v][co] * X[v + t][ci]                     (t = one of the 27 taps)
//
// conv_wgrad_kernel / conv_wgrad_s1_kernel (conv_wgrad.hip) run ONE tap (or the 3 kw taps) of a 128 x 128 channel tile per
// block: every K-step re-stages an R slab and a shifted G slab (64 / 190 flop per byte of LDS fill) and every wave re-reads
// both operands from LDS for 16 (32) accumulator tiles: 36-38 % MFMA busy, 0.26-0.29 of peak (profiles/r03_pmc_mfma_busy_train.json).
// Here a block owns ONE 16-input-channel chunk x 128 output channels x ALL 27 taps and walks 192-voxel tiles (3x8x8, 6x4x8,
// 3x4x16, 4x4x12 for 12- / 24-wide planes, 8x6x4 for 6-wide ones: the host picks the one with the largest useful fraction).
// Per tile it stages
//   * the X halo tile of its chunk, (TD+2)(TH+2)(TW+2) voxels x 32 B -- the 27 taps read it at shifted LDS addresses, exactly
//     like the forward kernel's A operand (out-of-volume rows are zero-filled by the DMA: no validity masks), and
//   * the dY tile, 192 voxels x 128 couts, as eight [16-cout chunk][voxel][32 B] slabs,
// 67-78 KB for 27 x 16 x 128 x 192 x 2 = 21.2 MFLOP: 317 flop per byte of LDS fill (the forward kernel: 385).  Both operands are
// k-major ([voxel][channel]) and are read with gfx950's transposing LDS read ds_read_b64_tr_b16 straight into
// v_mfma_f32_16x16x32_bf16 operands: M = 16 input channels of one tap, N = 16 output channels, K = 32 voxels.  The 8 waves
// are 4 tap groups x 2 cout halves: a wave holds 7 taps x 4 cout tiles = 28 accumulator tiles (112 registers) and reads
// 7 + 4 fragments per 28 MFMAs (0.39 per MFMA; the forward kernel: 0.375).
// K <-> voxel mapping: the tile's voxels (w fastest) form 48 blocks of 4 voxels along w; K-step s owns blocks 8 s .. 8 s + 7 and
// k-group kg / half e reads block 8 s + 4 e + kg -- table-driven per lane (a_step[s][e]), so TW only has to be a multiple of 4
// and blocks may straddle W-lines and depth slices.
// LDS layout for conflict-free transposed reads (a 16-lane group reads 4 voxel rows x 32 B = 128 contiguous bytes; the two
// groups served together -- k-groups kg, kg + 1 = CONSECUTIVE voxel blocks -- must fall into different 128-byte halves of the
// 256-byte bank period): the halo tile's w-stride is padded to HWP rows, a multiple of 4 with an odd quarter (same line: + 4
// rows; next line: + HWP - 4 (TW - 4) rows; next slice: both = 128 B modulo 256), and the dY voxels of a K-step are stored in
// the order (e, kg, q) (the DMA's per-lane source address does the permutation).
// Main loop: the A fragment of unit u + 2 and the next step's B fragments load under unit u's MFMAs; the next tile's DMA pieces
// are issued one per 4 units, SIMD partners two units apart.  1055-1150 TFLOP/s on the large layers (the one-tap kernel: 700-850),
// MFMA busy 57-58 % (36 %): profiles/r03_notes.md.
// Split-K over voxel-tile ranges: partial [slice][tap][cout][cin] fp32 tiles in the workspace, summed in a fixed order by
// conv_wgrad_reduce_t_kernel (deterministic), as for the other weight-gradient kernels.
#include "conv3_halo_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) short wh_s16x4;
typedef wh_s16x4 __attribute__((address_space(3))) * wh_lds_ptr;
__device__ __forceinline__ wh_s16x4 wh_tr_read(unsigned lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((wh_lds_ptr)(unsigned long long)lds_addr);
}

struct WgradHaloParams {
    const bf16_t* X;     // layer input, [n][D][H][W][CXs] bf16 (the gathered tensor)
    const bf16_t* dY;    // output gradient, [n][D][H][W][CYs] bf16
    float* part;         // [S][27][CRp][CGp] fp32
    int CX, CXs, CY, CYs;
    int N, D, H, W;
    int tilesD, tilesH, tilesW, ntiles;   // voxel tiles per sample axis, total over the batch
    int nchunks, ncot;                    // 16-cin chunks, 128-cout tiles
    int S, tps;                           // slices, tiles per slice
    int CRp, CGp;
    int dbg;
};

template <int TD_, int TH_, int TW_>
struct WhCfg {
    static constexpr int TD = TD_, TH = TH_, TW = TW_;
    static constexpr int NV = TD * TH * TW;
    static constexpr int KS = NV / 32;                    // K-steps (32 voxels) per tile
    static constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
    // padded w-stride of the halo tile in rows: a multiple of 4 whose quarter is ODD, so that the 128-byte blocks (4 voxels x
    // 32 B) of two consecutive voxel blocks -- same line or across a line / slice boundary -- fall into different halves of the
    // 256-byte bank period (see the k-group mapping below)
    static constexpr int HWP = ((HW + 3) / 4) % 2 ? ((HW + 3) / 4) * 4 : ((HW + 3) / 4 + 1) * 4;
    static constexpr int HROWS = HD * HH * HWP;
    static constexpr int XPIECES = (HROWS + 31) / 32;
    static constexpr int XBYTES = XPIECES * 1024;
    static constexpr int DYPIECES = 8 * KS;               // [16-cout chunk j][K-step s]
    static constexpr int STAGE = XBYTES + DYPIECES * 1024;
    static constexpr int LDS_BYTES = 2 * STAGE;
    static constexpr int NTH = 512, NWAVE = 8;
    static constexpr int XPW = (XPIECES + NWAVE - 1) / NWAVE;   // X pieces per wave
    static constexpr int DPW = DYPIECES / NWAVE;                // dY pieces per wave
    static constexpr int NBW = TW / 4;                          // 4-voxel blocks per w-line
    static_assert(NV % 32 == 0 && TW % 4 == 0 && LDS_BYTES <= 160 * 1024 && DYPIECES % NWAVE == 0 && (HWP / 4) % 2 == 1,
                  "unsupported tile");
};

// K position (step s, k-group kg, half e, 0..3) -> tile voxel.  The tile's voxels, w fastest, form NV / 4 blocks of 4 voxels
// along w; step s owns blocks 8 s .. 8 s + 7, and (kg, e) takes block 8 s + 4 e + kg: the two k-groups a transposing read serves
// together (kg, kg + 1) hold CONSECUTIVE blocks, whose halo rows are 128 B apart modulo 256 by the choice of HWP.
template <int TH, int TW>
__device__ __forceinline__ void wh_block(int s, int kg, int e, int* d, int* h, int* w) {
    const int b = 8 * s + 4 * e + kg;
    const int line = b / (TW / 4);
    *w = 4 * (b - line * (TW / 4));
    *d = line / TH;
    *h = line - *d * TH;
}

template <int TD, int TH, int TW>
__global__ void __attribute__((amdgpu_flat_work_group_size(1, 512)))
conv_wgrad_halo_kernel(const WgradHaloParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = WhCfg<TD, TH, TW>;
    constexpr int KS = Cfg::KS, HH = Cfg::HH, HW = Cfg::HW, HWP = Cfg::HWP, HROWS = Cfg::HROWS, XPIECES = Cfg::XPIECES;
    static_assert(Cfg::TD == TD && Cfg::TH == TH && Cfg::TW == TW, "");
    constexpr int XBYTES = Cfg::XBYTES, STAGE = Cfg::STAGE, XPW = Cfg::XPW, DPW = Cfg::DPW, NWAVE = Cfg::NWAVE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr3_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mg = wave >> 1, ng = wave & 1;                 // tap group (taps mg, mg + 4, ...), cout half

    // ---- block decode: (cin chunk, cout tile) fastest, then slice: the blocks of one slice read the same voxels and sit in
    //      one XCD's L2 (xcd_remap_h gives every XCD a contiguous range of logical ids) -----------------------------------
    int bid = xcd_remap_h((int)blockIdx.x, (int)gridDim.x);
    const int combos = p.nchunks * p.ncot;
    const int sl = bid / combos;
    bid -= sl * combos;
    const int cot = bid / p.nchunks, cc = bid - cot * p.nchunks;
    const int t_begin = sl * p.tps, t_end = min(t_begin + p.tps, p.ntiles);

    const v4i_t rsX = h3_make_rsrc(p.X, 0x7fffffffu);
    const v4i_t rsY = h3_make_rsrc(p.dY, 0x7fffffffu);

    // ---- per-lane DMA constants ----------------------------------------------------------------------------------------
    // X piece j = wave + 8 i: halo rows 32 j .. 32 j + 31, lane -> row 32 j + lane / 2, 16-byte half lane & 1
    int xc[XPW];            // packed halo coordinates hd | hh << 8 | hw << 16 (-1: padding row / beyond the tile)
#pragma unroll
    for (int i = 0; i < XPW; ++i) {
        const int j = wave + NWAVE * i;
        const int r = j * 32 + (lane >> 1);
        const int hd = r / (HH * HWP), rem = r - hd * (HH * HWP);
        const int hh = rem / HWP, hw = rem - hh * HWP;
        xc[i] = (j < XPIECES && r < HROWS && hw < HW) ? (hd | (hh << 8) | (hw << 16)) : -1;
    }
    // dY piece id = wave + 8 i -> (cout chunk j = id / KS, K-step s = id % KS); lane -> LDS position lane / 2 of the step's
    // 32, which holds tile voxel (kg, e, q): position = (4 e + kg) * 4 + q
    int yc[DPW];            // packed tile coordinates d | h << 8 | w << 16 of this lane's voxel in that piece
    {
        const int pos = lane >> 1;
        const int kg = (pos >> 2) & 3, e = pos >> 4, q = pos & 3;
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int s = (wave + NWAVE * i) % KS;
            int d, h, w;
            wh_block<TH, TW>(s, kg, e, &d, &h, &w);
            yc[i] = d | (h << 8) | ((w + q) << 16);
        }
    }
    const unsigned half16 = (unsigned)((lane & 1) * 16);
    const unsigned xcol = (unsigned)(cc * 32) + half16;                       // byte offset of the chunk inside a voxel row
    const unsigned xrow_bytes = (unsigned)(p.CXs * 2), yrow_bytes = (unsigned)(p.CYs * 2);

    // tile -> origin (wave-uniform); one X / dY piece of that tile -> LDS stage
    struct Origin { int nb, d0, h0, w0; };
    auto tile_origin = [&](int tile) -> Origin {
        const int per_n = p.tilesD * p.tilesH * p.tilesW;
        const int nb = tile / per_n;
        int r0 = tile - nb * per_n;
        const int tD = r0 / (p.tilesH * p.tilesW);
        r0 -= tD * p.tilesH * p.tilesW;
        const int tH = r0 / p.tilesW, tW = r0 - tH * p.tilesW;
        return Origin{nb, tD * TD, tH * TH, tW * TW};
    };
    auto issue_x = [&](const Origin& o, int i, unsigned base) {          // i: compile-time piece slot of this wave
        const int j = wave + NWAVE * i;
        if (j >= XPIECES) return;
        const int c = xc[i];
        const int gd = o.d0 - 1 + (c & 0xff), gh = o.h0 - 1 + ((c >> 8) & 0xff), gw = o.w0 - 1 + ((c >> 16) & 0xff);
        const bool ok = c >= 0 && (unsigned)gd < (unsigned)p.D && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
        const unsigned vox = (unsigned)(((o.nb * p.D + gd) * p.H + gh) * p.W + gw);
        const unsigned voff = ok ? vox * xrow_bytes + xcol : 0x80000000u;
        h3_dma16(rsX, (unsigned)__builtin_amdgcn_readfirstlane((int)(base + j * 1024)), voff, 0);
    };
    auto issue_y = [&](const Origin& o, int i, unsigned base) {
        const int id = wave + NWAVE * i;
        const int j = id / KS;                                             // 16-cout chunk of the block's 128
        const int c = yc[i];
        const int gd = o.d0 + (c & 0xff), gh = o.h0 + ((c >> 8) & 0xff), gw = o.w0 + ((c >> 16) & 0xff);
        const int co = cot * 128 + j * 16 + (lane & 1) * 8;
        const bool ok = gd < p.D && gh < p.H && gw < p.W && co < p.CY;
        const unsigned vox = (unsigned)(((o.nb * p.D + gd) * p.H + gh) * p.W + gw);
        const unsigned voff = ok ? vox * yrow_bytes + (unsigned)co * 2u : 0x80000000u;
        h3_dma16(rsY, (unsigned)__builtin_amdgcn_readfirstlane((int)(base + XBYTES + id * 1024)), voff, 0);
    };
    constexpr int NPW = XPW + DPW;                                         // DMA pieces per wave and tile (3 + 6)
    auto issue_piece = [&](const Origin& o, int k, unsigned base) {        // k: compile-time 0 .. NPW - 1
        if (k < XPW) issue_x(o, k, base); else issue_y(o, k - XPW, base);
    };

    // ---- fragment addressing ---------------------------------------------------------------------------------------------
    // operand lane (r16 = lane & 15, kg = lane >> 4): the 16-lane group kg issues the transposing read of 4 voxel rows x 32 B;
    // lane addresses row q = r16 >> 2, 8-byte piece pp = r16 & 3 and receives channel r16's 4 consecutive voxels
    const int r16 = lane & 15, kg = lane >> 4;
    const int q = r16 >> 2, pp = r16 & 3;
    unsigned a_step[KS][2];                                  // halo byte offset of this lane's row for (step, e), tap (0, 0, 0)
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
        for (int e_ = 0; e_ < 2; ++e_) {
            int d, h, w;
            wh_block<TH, TW>(s_, kg, e_, &d, &h, &w);
            a_step[s_][e_] = (unsigned)(((d * HH + h) * HWP + w + q) * 32 + pp * 8);
        }
    const unsigned b_lane = (unsigned)(XBYTES + (kg * 4 + q) * 32 + pp * 8);          // + (j KS + s) 1024, 512 e

    f32x4 acc[7][4];
#pragma unroll
    for (int i = 0; i < 7; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.0f;

    // A fragment of unit u = 7 s + i (tap slot i of K-step s), B fragment j of K-step s
    auto load_a = [&](unsigned sb, int u) -> bf16x8 {
        const int s_ = u / 7, i_ = u % 7;
        int tap = mg + 4 * i_;
        tap = tap < 27 ? tap : 26;                       // (tap group 3's last slot repeats tap 26; its tile is not stored)
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        const unsigned toff = (unsigned)(((kd * HH + kh) * HWP + kw) * 32);
        const wh_s16x4 a0 = wh_tr_read(sb + a_step[s_][0] + toff), a1 = wh_tr_read(sb + a_step[s_][1] + toff);
        return __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto load_b = [&](unsigned sb, int s_, int j) -> bf16x8 {
        const unsigned a = sb + b_lane + (unsigned)(((ng * 4 + j) * KS + s_) * 1024);
        const wh_s16x4 b0 = wh_tr_read(a), b1 = wh_tr_read(a + 512);
        return __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    // Main loop.  Per tile: KS x 7 units of 4 MFMAs (one tap slot x the wave's 4 cout tiles).  The A fragment of unit u + 2 and
    // (during the first four units of a step) the B fragments of the NEXT step are loaded while unit u is on the matrix cores:
    // 3 + 8 fragment registers sets instead of 7 + 4 loaded in a burst.  The DMA pieces of the NEXT tile (NPW per wave) are
    // issued one per 4 units -- waves 0-3 at units 4 k, their SIMD partners 4-7 at units 4 k + 2 -- so that a piece's ~60-100
    // issue cycles (conv3_halo_k32.hip) fall under the partner's MFMAs instead of stalling all 8 waves behind the barrier.
    constexpr int NU = 7 * KS;
    static_assert(NU / 4 >= NPW, "not enough issue slots for the tile's DMA pieces");
    if (t_begin < t_end) {
        {
            const Origin o0 = tile_origin(t_begin);
#pragma unroll
            for (int k = 0; k < NPW; ++k) issue_piece(o0, k, lds0);
        }
        for (int tile = t_begin; tile < t_end; ++tile) {
            const int stage = (tile - t_begin) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // tile landed; every wave is done with the other stage
            const bool more = tile + 1 < t_end && !(CTSI_DBG(p.dbg, 1));
            const Origin on = tile_origin(more ? tile + 1 : tile);
            const unsigned sb = lds0 + stage * STAGE, nb_ = lds0 + (stage ^ 1) * STAGE;
            bf16x8 bfr[4], bnx[4], af[3];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = load_b(sb, 0, j);
            af[0] = load_a(sb, 0);
            af[1] = load_a(sb, 1);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int s_ = u / 7, i_ = u % 7;
                if (u + 2 < NU) af[(u + 2) % 3] = load_a(sb, u + 2);
                if (i_ < 4 && s_ + 1 < KS) bnx[i_] = load_b(sb, s_ + 1, i_);
                if (more) {
                    if ((u & 3) == 0 && (u >> 2) < NPW) { if (wave < 4) issue_piece(on, u >> 2, nb_); }
                    if ((u & 3) == 2 && (u >> 2) < NPW) { if (wave >= 4) issue_piece(on, u >> 2, nb_); }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i_][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[u % 3], bfr[j], acc[i_][j], 0, 0, 0);
                if (i_ == 6) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bfr[j] = bnx[j];
                }
            }
        }
    }

    // ---- partial tiles -> workspace [slice][tap][CRp couts][CGp cins]: accumulator (i, j)[r] = cin 4 kg + r, cout r16 -----------
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int tap = mg + 4 * i;
        if (tap < 27) {
            float* dst = p.part + ((size_t)sl * 27 + tap) * (size_t)p.CRp * p.CGp;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = cot * 128 + (ng * 4 + j) * 16 + r16;
                *reinterpret_cast<float4*>(dst + (size_t)co * p.CGp + cc * 16 + 4 * kg) =
                    make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
        }
    }
#endif
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
// tile choice among 3x8x8, 6x4x8, 3x4x16, 4x4x12 (12- / 24-wide planes) and 8x6x4 (6-wide planes): the one with the largest
// useful fraction of its 192 voxels, which must be >= 0.7.  Returns 0 when the layer does not qualify (the caller falls back to conv_wgrad.hip's kernels).
struct WhPlan {
    int code;      // 1: 3x8x8, 2: 6x4x8, 3: 3x4x16, 4: 4x4x12, 5: 8x6x4
    int td, th, tw, tilesD, tilesH, tilesW, ntiles, nchunks, ncot, S, tps, CRp, CGp;
};

static int wh_plan(int n, int d, int h, int w, int cx, int cy, WhPlan* o) {
    if (cx % 16 != 0 || cy % 8 != 0) return 0;
    const int cand[5][4] = {{1, 3, 8, 8}, {2, 6, 4, 8}, {3, 3, 4, 16}, {4, 4, 4, 12}, {5, 8, 6, 4}};
    double best = 0.0;
    int bi = -1;
    for (int i = 0; i < 5; ++i) {
        const int td = cand[i][1], th = cand[i][2], tw = cand[i][3];
        const double tiles = (double)((d + td - 1) / td) * ((h + th - 1) / th) * ((w + tw - 1) / tw);
        const double useful = (double)d * h * w / (tiles * td * th * tw);
        if (useful > best + 1e-9) {
            best = useful;
            bi = i;
        }
    }
    if (bi < 0 || best < 0.7) return 0;
    o->code = cand[bi][0];
    o->td = cand[bi][1]; o->th = cand[bi][2]; o->tw = cand[bi][3];
    o->tilesD = (d + o->td - 1) / o->td; o->tilesH = (h + o->th - 1) / o->th; o->tilesW = (w + o->tw - 1) / o->tw;
    const long long nt = (long long)n * o->tilesD * o->tilesH * o->tilesW;
    if (nt >= (1ll << 30)) return 0;
    o->ntiles = (int)nt;
    o->nchunks = cx / 16;
    o->ncot = (cy + 127) / 128;
    o->CRp = o->ncot * 128;
    o->CGp = o->nchunks * 16;
    // one 8-wave block per CU (134 KB of LDS): aim at whole rounds of the 256 CUs, at least 8 tiles per block
    const int combos = o->nchunks * o->ncot;
    const char* tgt = getenv("CTSI_WGRAD_HALO_TARGET");   // blocks to aim for (tuning aid)
    const int target = tgt && atoi(tgt) > 0 ? atoi(tgt) : 256;   // (measured: 256 = one round beats 512 / 768: half the partial-sum traffic)
    int S = target / combos;
    if (S < 1) S = 1;
    const int smax = (o->ntiles + 7) / 8;
    if (S > smax) S = smax;
    if (S < 1) S = 1;
    o->tps = (o->ntiles + S - 1) / S;
    o->S = (o->ntiles + o->tps - 1) / o->tps;
    return 1;
}

extern "C" int ctsi_wgrad_halo_plan(int n, int d, int h, int w, int cx, int cy, int* S, size_t* ws_bytes, int* CRp, int* CGp) {
    WhPlan pl;
    if (!wh_plan(n, d, h, w, cx, cy, &pl)) return 0;
    if (S) *S = pl.S;
    if (CRp) *CRp = pl.CRp;
    if (CGp) *CGp = pl.CGp;
    if (ws_bytes) *ws_bytes = (size_t)pl.S * 27 * (size_t)pl.CRp * pl.CGp * sizeof(float);
    return pl.code;
}

template <int TD, int TH, int TW>
static void wh_launch(const WgradHaloParams& p, int blocks, hipStream_t st) {
    using Cfg = WhCfg<TD, TH, TW>;
    auto k = conv_wgrad_halo_kernel<TD, TH, TW>;
    static CtsiPerDeviceOnce attr_once;
    if (attr_once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(Cfg::NTH), Cfg::LDS_BYTES, st, p);
}

// X: layer input (n, d, h, w, cx_stride) bf16; dY: output gradient (n, d, h, w, cy_stride) bf16; part: workspace of
// ctsi_wgrad_halo_plan's size.  Leaves the partial sums [S][27][CRp][CGp] in `part` (the caller runs the reduce pass).
extern "C" int ctsi_wgrad_halo_launch(const void* X, const void* dY, void* part, int n, int d, int h, int w, int cx, int cx_stride,
                                      int cy, int cy_stride, void* stream) {
    WhPlan pl;
    CTSI_CHECK_ARG(X && dY && part && wh_plan(n, d, h, w, cx, cy, &pl), "ctsi_wgrad_halo_launch: layer does not qualify");
    const long long xb = (long long)n * d * h * w * cx_stride * 2, yb = (long long)n * d * h * w * cy_stride * 2;
    CTSI_CHECK_ARG(xb < 0x7fffffffll && yb < 0x7fffffffll, "ctsi_wgrad_halo_launch: tensor larger than one 2 GiB buffer window");
    WgradHaloParams p;
    p.X = (const bf16_t*)X; p.dY = (const bf16_t*)dY; p.part = (float*)part;
    p.CX = cx; p.CXs = cx_stride; p.CY = cy; p.CYs = cy_stride;
    p.N = n; p.D = d; p.H = h; p.W = w;
    p.tilesD = pl.tilesD; p.tilesH = pl.tilesH; p.tilesW = pl.tilesW; p.ntiles = pl.ntiles;
    p.nchunks = pl.nchunks; p.ncot = pl.ncot; p.S = pl.S; p.tps = pl.tps; p.CRp = pl.CRp; p.CGp = pl.CGp;
    p.dbg = ctsi_debug_flags();
    const int blocks = pl.nchunks * pl.ncot * pl.S;
    if (pl.code == 1) wh_launch<3, 8, 8>(p, blocks, (hipStream_t)stream);
    else if (pl.code == 2) wh_launch<6, 4, 8>(p, blocks, (hipStream_t)stream);
    else if (pl.code == 3) wh_launch<3, 4, 16>(p, blocks, (hipStream_t)stream);
    else if (pl.code == 4) wh_launch<4, 4, 12>(p, blocks, (hipStream_t)stream);
    else wh_launch<8, 6, 4>(p, blocks, (hipStream_t)stream);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
