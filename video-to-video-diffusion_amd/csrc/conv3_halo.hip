// 3x3x3 stride-1 convolution with an LDS-staged D x H x W input halo tile (gfx950 / MI355X), 32x32x16-MFMA kernels for the
// levels the 512- / 384-voxel tiles of conv3_halo_k32.hip fill badly (48 x 16 x 16 with 512 couts).
//
// The gather-GEMM kernel (conv_mfma.hip) re-stages the activation slab for each of the 27 taps and runs
// at a constant L2->LDS fill rate (~8-10 TB/s): bytes per flop bound it.  These kernels stage, per
// 32-channel chunk, the (TD+2) x (TH+2) x (TW+2) halo tile of the input ONCE (buffer_load ... lds,
// out-of-volume voxels are out-of-range buffer offsets -> hardware zero fill = the conv's zero padding) and
// let all 27 taps read their A operand from it at shifted LDS addresses; only the weights stream per tap.
// (The round-1 16x16x32 form of the 4 x 4 x 16 tile lives under experiments/conv3_halo_w16.hip.)
#include "conv3_halo_common.h"
#include <type_traits>
#include <string.h>

// =====================================================================================================
// 32-wide variant: output tile 4 x 2 x 32 voxels, v_mfma_f32_32x32x16_bf16.  An A operand tile is a W-line
// of 32 consecutive halo rows, read with ds_read_b128 (each 16-lane group of a b128 read then covers 16
// distinct (row & 3, chunk ^ swizzle) bank slots for every tap shift).  Half the MFMA and LDS-read
// instruction count of the 16-wide kernel for the same FLOPs (the 16x16x32 form holds the SIMD's issue
// port for 8 of its 16 cycles, the 32x32x16 form for 8 of 32), used whenever W >= 32-ish.
// =====================================================================================================

// Tile shapes <TD, TH, TW, TM, TN> (TM x TN = 32x32 MFMA tiles per wave; 8 waves = WM x WN with WM * TM * 32 = BM rows and
// WN * TN * 32 = 128 couts):
//   <4,2,32,2,2> = 4 x 2 x 32 voxels, an A tile of 32 rows = one W-line; waves 4 (M) x 2 (N), 2 x 2 tiles each;
//   <4,4,16,2,2> = 4 x 4 x 16 voxels for 16-wide levels: an A tile = two W-lines of 16 (lanes 16-31 read the next H
//                  line), so those levels run on 32x32x16 MFMAs with half the LDS reads per flop of the 16x16x32 form;
//   <3,4,16,3,1> = 3 x 4 x 16 = 192 voxels, waves 2 (M) x 4 (N), 3 x 1 tiles each: a 48 x 16 x 16 level with 512 couts is
//                  16 x 4 x 4 = 256 blocks -- one per CU -- where 256-voxel tiles give 192 blocks on 256 CUs.
template <int TD_, int TH_, int TW_, int TM_, int TN_>
struct H32Cfg {
    static constexpr int TD = TD_, TH = TH_, TW = TW_, TM = TM_, TN = TN_;
    static constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
    static constexpr int HV = HD * HH * HW;               // <2,32>: 816 halo voxels, <4,16>: 648
    static constexpr int HALO_INSTR = (HV + 15) / 16;     // 51 / 41
    static constexpr int HALO_BYTES = HALO_INSTR * 1024;
    static constexpr int BM = TD * TH * TW;               // 256 / 192
    static constexpr int BN = 128;
    static constexpr int WM = BM / (TM * 32), WN = BN / (TN * 32);
    static constexpr int LPT = 32 / TW;                   // W-lines per 32-row A tile
    static constexpr int WSLOT_BYTES = 3 * BN * 64;
    static constexpr int NTH = 512;
    static constexpr int OFF_W = 2 * HALO_BYTES;
    static constexpr int OFF_ROW = OFF_W + 2 * WSLOT_BYTES;
    static constexpr int OFF_CS = OFF_ROW + BM * 8;
    static constexpr int LDS_BYTES = OFF_CS + WM * BN * 8;  // <4,2,32>: 159744 <= 163840
    static constexpr int NPIECE = (HALO_INSTR + 7) / 8;    // halo DMA instructions per wave and chunk (issued at g < NPIECE)
    static_assert(WM * WN == 8 && BM % (TM * 32) == 0 && 32 % TW == 0 && (TW == 32 || TH % 2 == 0) && NPIECE <= 9 && LDS_BYTES <= 160 * 1024 &&
                      BM * BN * 2 <= OFF_W, "unsupported tile");
};

template <int TD_, int TH_, int TW_, int TM_, int TN_>
__global__ void __launch_bounds__(512)
conv3_halo32_kernel(const Conv3HaloParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = H32Cfg<TD_, TH_, TW_, TM_, TN_>;
    constexpr int TD = Cfg::TD, TH = Cfg::TH, TW = Cfg::TW, HH = Cfg::HH, HW = Cfg::HW, HV = Cfg::HV, LPT = Cfg::LPT;
    constexpr int TM = Cfg::TM, TN = Cfg::TN, WM = Cfg::WM, WN = Cfg::WN;
    (void)TD;
    constexpr int HALO_INSTR = Cfg::HALO_INSTR, HALO_BYTES = Cfg::HALO_BYTES, BM = Cfg::BM, BN = Cfg::BN, NTH = Cfg::NTH;
    constexpr int WSLOT_BYTES = Cfg::WSLOT_BYTES, OFF_W = Cfg::OFF_W, OFF_ROW = Cfg::OFF_ROW, OFF_CS = Cfg::OFF_CS;
    constexpr int NPIECE = Cfg::NPIECE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long* s_rowoff = reinterpret_cast<long long*>(smem + OFF_ROW);
    float* s_cs = reinterpret_cast<float*>(smem + OFF_CS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int bid = xcd_remap_h(blockIdx.x, gridDim.x);
    int mt, nt;
    h3_decode_tile(bid, p.mtiles, p.ntiles_n, p.n_major, &mt, &nt);
    const int n0 = nt * BN;
    const int nb = mt / p.tps;
    int r0 = mt - nb * p.tps;
    const int tD = r0 / (p.tilesH * p.tilesW);
    r0 -= tD * p.tilesH * p.tilesW;
    const int tH = r0 / p.tilesW;
    const int tW = r0 - tH * p.tilesW;
    const int d0 = tD * TD, h0 = tH * TH, w0 = tW * TW;

    // 16-wide tiles: an A tile of 32 rows is two W-lines, HW = 18 halo voxels apart -- the second line's rows would sit 2 voxels
    // off the 16-voxel bank pattern of the first (42 % LDS bank conflicts in conv3_halo32_kernel<3,4,16,3,1>).  Its rows are
    // therefore ROTATED by 2: row 16 + r of a tile holds voxel (r + 14) % 16 of the odd line, which reads halo voxel
    // v0 + 18 + (r + 14) % 16 = v0 + 16 + r (+ 16 for r < 2): the bank pattern of 32 consecutive voxels.  The same rotation
    // in the row-offset table is all the epilogue needs (column sums do not depend on the row order).
    auto rot16 = [](int m, int line) -> int { return (TW == 16 && (line & 1)) ? ((m + 14) & 15) : m; };
    if (tid < BM) {   // row = line*TW + m, line = ld*TH + lh
        const int line = tid / TW, mm = rot16(tid % TW, line);
        const int d = d0 + line / TH, h = h0 + line % TH, w = w0 + mm;
        long long off = -1;
        if (d < p.Do && h < p.Ho && w < p.Wo)
            off = ((((long long)nb * p.Do + d) * p.Ho + h) * p.Wo + w) * p.cout_stride + p.c_off;
        s_rowoff[tid] = off;
    }

    int dlo = d0 + p.dshift - 1;
    dlo = dlo < 0 ? 0 : dlo;
    const long long basevox = ((long long)(nb * p.Di + dlo) * p.Hi) * p.Wi;
    const v4i_t rs1 = h3_make_rsrc(reinterpret_cast<const char*>(p.x1) + basevox * p.C1 * 2, 0x7fffffffu);
    const v4i_t rs2 = h3_make_rsrc(reinterpret_cast<const char*>(p.x2) + basevox * p.C2 * 2, 0x7fffffffu);
    const v4i_t rsw = h3_make_rsrc(reinterpret_cast<const char*>(p.w) + (long long)n0 * 64, 0x7fffffffu);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr3_t)smem;

    const int hq = (lane & 3) ^ (lane >> 4);
    int hrel[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int j = wave + 8 * i;
        const int v = j * 16 + (lane >> 2);
        const int hd = v / (HH * HW), rem = v - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        const int gd = d0 + p.dshift - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
        const bool ok = (j < HALO_INSTR) && (v < HV) && gd >= 0 && gd < p.Di && gh >= 0 && gh < p.Hi && gw >= 0 &&
                        gw < p.Wi;
        hrel[i] = ok ? ((gd - dlo) * p.Hi + gh) * p.Wi + gw : -1;
    }
    const unsigned w_voff = (unsigned)wave * 1024u + (unsigned)lane * 16u;
    const int C1 = p.C1, C2 = p.C2, CoutPad = p.CoutPad, nchunks = p.nchunks;

    auto issue_halo = [&](int cc, int i, int hoff) {
        const int j = wave + 8 * i;
        if (j >= HALO_INSTR) return;
        const int ch0 = cc * 32;
        const bool second = ch0 >= C1;
        const unsigned cbytes = (unsigned)((second ? C2 : C1) * 2);
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((second ? ch0 - C1 : ch0) * 2);
        int hsel = hrel[0];
#pragma unroll
        for (int q = 1; q < NPIECE; ++q) hsel = (i == q) ? hrel[q] : hsel;
        const unsigned voff = hsel >= 0 ? (unsigned)hsel * cbytes + (unsigned)hq * 16u : 0x80000000u;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + hoff + j * 1024));
        if (second)
            h3_dma16(rs2, dst, voff, soff);
        else
            h3_dma16(rs1, dst, voff, soff);
    };
    auto issue_weights = [&](int s, int woff) {
        const int cc = s / 9, g = s - cc * 9;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(((cc * 27 + g * 3 + i) * CoutPad) * 64);
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + woff + i * (BN * 64) + wave * 1024));
            h3_dma16(rsw, dst, w_voff, soff);
        }
    };

    // fragment addressing: lane -> row r = lane & 31 of the 32-row operand tile, k-group hk = lane >> 5
    const int hk = lane >> 5, r = lane & 31;
    int vline[TM];                                           // halo voxel of this lane's row in each of the wave's A tiles
#pragma unroll
    for (int i = 0; i < TM; ++i) {                           // A tile = rows [(wm*TM + i)*32, +32) = LPT whole W-lines
        const int line = (wm * TM + i) * LPT + r / TW;
        vline[i] = ((line / TH) * HH + (line % TH)) * HW + rot16(r % TW, line);
    }
    const int rowb = wn * (TN * 32) + r;
    const int b_off = rowb * 64 + ((hk ^ ((rowb >> 2) & 3)) << 4);   // k-step 0; k-step 1 = ^32; n-tile 1 = +2048

    f32x16 acc[TM][TN];                                      // start value = the bias of the lane's cout (no bias pass later)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n0 + (wn * TN + j) * 32 + (lane & 31);
        const float bv = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = bv;
    }

    bf16x8 fa0[TM][2], fb0[TN][2], fa1[TM][2], fb1[TN][2], fa2[TM][2], fb2[TN][2];   // [tile][k-step]
    const int S = nchunks * 9;
    int boffj[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        boffj[j] = j * 2048;
        asm volatile("" : "+v"(boffj[j]));
    }

#define H32_LOAD(FA, FB, HBUF, WBUF, VS, KW)                                                                   \
    {                                                                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) {                                                    \
            const int v_ = (VS) + vline[i_] + (KW);                                                            \
            const int o_ = v_ * 64 + ((hk ^ ((v_ >> 2) & 3)) << 4);                                            \
            FA[i_][0] = *reinterpret_cast<const bf16x8*>((HBUF) + o_);                                         \
            FA[i_][1] = *reinterpret_cast<const bf16x8*>((HBUF) + (o_ ^ 32));                                  \
        }                                                                                                      \
        _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) {                                                    \
            const int o_ = b_off + (KW) * (BN * 64) + boffj[j_];                                               \
            FB[j_][0] = *reinterpret_cast<const bf16x8*>((WBUF) + o_);                                         \
            FB[j_][1] = *reinterpret_cast<const bf16x8*>((WBUF) + (o_ ^ 32));                                  \
        }                                                                                                      \
    }
#define H32_MFMA(FA, FB)                                                                                       \
    {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_) _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)      \
            _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] =                                    \
                __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i_][k_], FB[j_][k_], acc[i_][j_], 0, 0, 0);         \
        __builtin_amdgcn_s_setprio(0);                                                                         \
    }

    // Software pipeline (blocks b0,b1,b2 = the three kw taps of a step; block b always uses fragment set F[b]):
    //   phase 0: MFMA b0(s)  interleaved with LOAD b2(s)      -> F2
    //   lgkmcnt(0), vmcnt, barrier; DMA W(s+2) into the slot just drained, one halo piece of the next chunk
    //   phase 1: MFMA b1(s)  interleaved with LOAD b0(s+1)    -> F0
    //   phase 2: MFMA b2(s)  interleaved with LOAD b1(s+1)    -> F1
    // Every fragment set is loaded two phases before it is used; inside a phase the 8 ds_read_b128 are
    // placed one per MFMA gap (sched_group_barrier), so neither wave of a SIMD has a load-only phase.
#define H32_PHASE(FAc, FBc, FAl, FBl, HBUF, WBUF, VS, KW, DOLOAD)                                              \
    {                                                                                                          \
        H32_LOAD(FAl, FBl, HBUF, WBUF, VS, KW);                                                                \
        _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_) _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)      \
            _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) acc[i_][j_] =                                    \
                __builtin_amdgcn_mfma_f32_32x32x16_bf16(FAc[i_][k_], FBc[j_][k_], acc[i_][j_], 0, 0, 0);       \
        /* NLD = 2 (TM + TN) ds_read_b128 spread over the NMF = 2 TM TN MFMA gaps (the first NLD - NMF gaps take two) */ \
        _Pragma("unroll") for (int q_ = 0; q_ < 2 * (TM + TN) - 2 * TM * TN; ++q_) {                           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                 \
        }                                                                                                      \
        _Pragma("unroll") for (int q_ = 2 * (TM + TN) - 2 * TM * TN; q_ < 2 * TM * TN; ++q_) {                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }

#pragma unroll
    for (int i = 0; i < NPIECE; ++i) issue_halo(0, i, 0);
    issue_weights(0, OFF_W);
    if (S > 1) issue_weights(1, OFF_W + WSLOT_BYTES);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    H32_LOAD(fa0, fb0, smem, smem + OFF_W, 0, 0);
    H32_LOAD(fa1, fb1, smem, smem + OFF_W, 0, 1);
    __builtin_amdgcn_sched_barrier(0);

    int cc = 0, g = 0;
    bool halo_in_flight = false;   // wave-uniform
    for (int s = 0; s < S; ++s) {
        const char* hbuf = smem + (cc & 1) * HALO_BYTES;
        const char* wbuf = smem + OFF_W + (s & 1) * WSLOT_BYTES;
        const int kd = g / 3, kh = g - kd * 3;
        const int vs = (kd * HH + kh) * HW;
        H32_PHASE(fa0, fb0, fa2, fb2, hbuf, wbuf, vs, 2, true);
        // every wave has drained this step's weight slot; the weight DMA of the previous step has landed.
        // A halo piece issued in the previous step (the youngest VMEM op of this wave) may stay in flight
        // one more step: it streams from HBM and is only consumed after the chunk switch.
        if (halo_in_flight)
            asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // DMA issue placement: behind the first MFMA phase after the barrier (its 8 MFMAs drain through the matrix pipe while
        // the pieces are issued) unless p.dbg & 16 (old placement, right behind the barrier: A/B timing)
#define H32_ISSUE()                                                                                            \
        {                                                                                                      \
            if (s + 2 < S && !(CTSI_DBG(p.dbg, 2))) issue_weights(s + 2, OFF_W + (s & 1) * WSLOT_BYTES);                \
            halo_in_flight = (g < NPIECE - 1) && (cc + 1 < nchunks) && !(CTSI_DBG(p.dbg, 1)) && (wave + 8 * g < HALO_INSTR); \
            if (g < NPIECE && cc + 1 < nchunks && !(CTSI_DBG(p.dbg, 1))) issue_halo(cc + 1, g, ((cc + 1) & 1) * HALO_BYTES); \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
        }
        if (CTSI_DBG(p.dbg, 16)) H32_ISSUE();
        int g2 = g + 1, cc2 = cc;
        if (g2 == 9) {
            g2 = 0;
            ++cc2;
        }
        // (the last step's look-ahead loads read stale but in-bounds LDS and are never consumed)
        const char* hbuf2 = smem + (cc2 & 1) * HALO_BYTES;
        const char* wbuf2 = smem + OFF_W + ((s + 1) & 1) * WSLOT_BYTES;
        const int kd2 = g2 / 3, kh2 = g2 - kd2 * 3;
        const int vs2 = (kd2 * HH + kh2) * HW;
        __builtin_amdgcn_sched_barrier(0);
        H32_PHASE(fa1, fb1, fa0, fb0, hbuf2, wbuf2, vs2, 0, true);
        // SIMD partners (waves w, w + 4) issue their pieces at different phase boundaries (+1.7-2.4 %, profiles/r02_notes.md):
        // while one issues -- and feeds no MFMAs -- the other runs a phase on the matrix pipe (p.dbg & 32: all waves here)
        if (!(CTSI_DBG(p.dbg, 16)) && (wave < 4 || (CTSI_DBG(p.dbg, 32)))) H32_ISSUE();
        H32_PHASE(fa2, fb2, fa1, fb1, hbuf2, wbuf2, vs2, 1, true);
        if (!(CTSI_DBG(p.dbg, 16)) && wave >= 4 && !(CTSI_DBG(p.dbg, 32))) H32_ISSUE();
        g = g2;
        cc = cc2;
    }
#undef H32_ISSUE
#undef H32_PHASE
#undef H32_LOAD
#undef H32_MFMA
    __syncthreads();

    // ---- epilogue ------------------------------------------------------------------------------------------------
    if (CTSI_DBG(p.dbg, 8)) return;
    bf16_t* s_tile = reinterpret_cast<bf16_t*>(smem);  // [BM][BN] bf16 = 64 KB
    const bool want_sums = p.colsum != nullptr;
    const int lhi = lane >> 5, lcol = lane & 31;
    unsigned vbits[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        unsigned vb = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (wm * TM + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * lhi;
            vb |= (unsigned)(s_rowoff[row] >= 0) << q;
        }
        vbits[i] = vb;
    }
    // per accumulator: half a v_cvt_pk_bf16_f32 (rows q, q + 1 share it), one 2-byte LDS write, one add + one fma for the column
    // sums; the validity select only in waves that own rows outside the volume
    auto tile_out = [&](auto masked_tag, auto sums_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value, SUMS = decltype(sums_tag)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = (wn * TN + j) * 32 + lcol;
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                bf16_t* trow = s_tile + ((wm * TM + i) * 32 + 4 * lhi) * BN + col;
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    const float v0 = acc[i][j][q], v1 = acc[i][j][q + 1];
                    const uint32_t pk = pack_bf16x2_v(f32x2_t{v0, v1});
                    const int rr = (q & 3) + 8 * (q >> 2);
                    trow[rr * BN] = (bf16_t)(pk & 0xffffu);
                    trow[(rr + 1) * BN] = (bf16_t)(pk >> 16);
                    if (SUMS) {
                        const float m0 = (!MASKED || ((vbits[i] >> q) & 1u)) ? v0 : 0.0f;
                        const float m1 = (!MASKED || ((vbits[i] >> (q + 1)) & 1u)) ? v1 : 0.0f;
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
    {
        using T = std::true_type;
        using F = std::false_type;
        bool rg = false;
#pragma unroll
        for (int i = 0; i < TM; ++i) rg = rg || vbits[i] != 0xffffu;
        const bool ragged = __builtin_amdgcn_ballot_w64(rg) != 0ull;   // wave-uniform
        if (!want_sums) tile_out(F{}, F{});
        else if (ragged) tile_out(T{}, T{});
        else tile_out(F{}, T{});
    }
    __syncthreads();
    if (want_sums && tid < BN) {
        float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
        for (int q = 0; q < WM; ++q) {
            t1 += s_cs[(q * BN + tid) * 2 + 0];
            t2 += s_cs[(q * BN + tid) * 2 + 1];
        }
        const long long slab = (long long)p.mtiles * CoutPad;
        p.colsum[(long long)mt * CoutPad + n0 + tid] = t1;
        p.colsum[slab + (long long)mt * CoutPad + n0 + tid] = t2;
    }
    {
        constexpr int CPR = BN / 8;
        bf16_t* y = reinterpret_cast<bf16_t*>(p.y);
        for (int c = tid; c < BM * CPR; c += NTH) {
            const int row = c / CPR, ch = c - row * CPR;
            const long long off = s_rowoff[row];
            const int co = n0 + ch * 8;
            if (off >= 0 && co < p.Cout && !(CTSI_DBG(p.dbg, 4))) {
                const uint4 v = *reinterpret_cast<const uint4*>(s_tile + row * BN + ch * 8);
                *reinterpret_cast<uint4*>(y + off + co) = v;
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- weight packing: fp32 (cout, cin, 3,3,3) -> bf16 [chunk][tap][cout_pad][32], 16-B chunks swizzled per row --------
__global__ void conv3_halo_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int CoutPad,
                                       int Cin, int CinW, int nchunks) {
    const long long total = (long long)nchunks * 27 * CoutPad * 32;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 31);                 // physical element within the 64-B row
        const long long row = idx >> 5;                // (chunk*27 + tap)*CoutPad + cout
        const int co = (int)(row % CoutPad);
        const long long ct = row / CoutPad;
        const int tap = (int)(ct % 27), cc = (int)(ct / 27);
        const int slot = e >> 3;
        const int q = slot ^ ((co >> 2) & 3);          // logical chunk stored in this physical slot
        const int ci = cc * 32 + q * 8 + (e & 7);
        float v = 0.0f;
        if (co < Cout && ci < CinW) v = w[((long long)co * CinW + ci) * 27 + tap];
        out[idx] = f32_to_bf16(v);
    }
}

extern "C" int ctsi_conv3_halo_pack(const float* w, void* packed, int cout, int cout_pad, int cin, int cin_w,
                                    void* stream) {
    CTSI_CHECK_ARG(w && packed && cin % 32 == 0, "ctsi_conv3_halo_pack: bad arguments");
    const int nchunks = cin / 32;
    const long long total = (long long)nchunks * 27 * cout_pad * 32;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(conv3_halo_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)packed,
                       cout, cout_pad, cin, cin_w, nchunks);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

extern "C" int ctsi_conv3_halo_launch(const Conv3HaloParams* hp, int wide, void* stream) {
    static CtsiPerDeviceOnce attr_once;
    if (attr_once.first()) {
        hipFuncSetAttribute((const void*)conv3_halo32_kernel<4, 2, 32, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)H32Cfg<4, 2, 32, 2, 2>::LDS_BYTES);
        hipFuncSetAttribute((const void*)conv3_halo32_kernel<4, 4, 16, 2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)H32Cfg<4, 4, 16, 2, 2>::LDS_BYTES);
        hipFuncSetAttribute((const void*)conv3_halo32_kernel<3, 4, 16, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)H32Cfg<3, 4, 16, 3, 1>::LDS_BYTES);
    }
    const int grid = hp->mtiles * hp->ntiles_n;
    constexpr int lds_232 = H32Cfg<4, 2, 32, 2, 2>::LDS_BYTES, lds_416 = H32Cfg<4, 4, 16, 2, 2>::LDS_BYTES;
    constexpr int lds_316 = H32Cfg<3, 4, 16, 3, 1>::LDS_BYTES;
    if (wide == 1)
        hipLaunchKernelGGL((conv3_halo32_kernel<4, 2, 32, 2, 2>), dim3(grid), dim3(512), lds_232, (hipStream_t)stream, *hp);
    else if (wide == 3)
        hipLaunchKernelGGL((conv3_halo32_kernel<4, 4, 16, 2, 2>), dim3(grid), dim3(512), lds_416, (hipStream_t)stream, *hp);
    else if (wide == 4)
        hipLaunchKernelGGL((conv3_halo32_kernel<3, 4, 16, 3, 1>), dim3(grid), dim3(512), lds_316, (hipStream_t)stream, *hp);
    else {
        ctsi_set_error("ctsi_conv3_halo_launch: unknown tile %d", wide);
        return CTSI_ERR_INVALID;
    }
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
