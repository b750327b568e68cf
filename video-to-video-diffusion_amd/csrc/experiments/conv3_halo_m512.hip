// EXPERIMENT / A-B REFERENCE, not part of libctsi.so (compile-checked by `make experiments`): the 32x32x16-MFMA form of the
// 512-voxel tile, superseded by conv3_halo_k32.hip (+10-12 % on real data), and its opt-in normalise-on-load variant
// (NIN = true: GroupNorm + SiLU + time bias applied to the staged halo tile in LDS; bit-identical to the separate pass but
// +0.9 ms per step).  Measurements: profiles/r02_notes.md.
//
// 3x3x3 stride-1 convolution, LDS halo tile 4 x 4 x 32 voxels (512 output voxels) x 128 couts, 16-channel chunks.
//
// Measured on conv3_halo32_kernel (4x2x32 x 128 couts, 32-channel chunks): the LDS-DMA *instructions* (1 KB per wave
// instruction, ~100+ issue cycles each inside an MFMA phase) cost 11-15 % of the kernel, most of them weight pieces
// (27 x 8 KB per chunk and block).  Doubling the voxel tile halves the weight pieces per MFMA; 16-channel chunks
// (32-byte halo rows) keep the double-buffered halo at 2 x 39 KB.  Per wave: 64 voxels (2 W-lines) x 128 couts =
// 2 x 4 MFMA tiles of 32x32 (128 accumulators), 6 ds_read_b128 per 8 MFMAs; DMA instructions per MFMA: 0.085 instead
// of 0.155; LDS fill 385 flop/B instead of 211; prologue / epilogue amortised over twice the work.
// Weight image and swizzles as in conv3_halo_n64.hip ([chunk16][27][cout_pad][16], slot = khalf ^ ((row >> 3) & 1)),
// 4-deep weight ring with counted vmcnt waits.  The 128 KB output tile goes through LDS in one pass (it reuses the
// halo + weight buffers; the row-offset table and the column-sum scratch sit behind it).
#include "../conv3_halo_common.h"
#include <stdlib.h>

// Tile shapes: <4,4> = 4 x 4 x 32 = 512 voxels (8 waves) and <6,2> = 6 x 2 x 32 = 384 voxels (6 waves; a 48 x 32 x 32 level
// with 512 couts gives 128 x 4 = 512 blocks = two full rounds on 256 CUs, where 512-voxel tiles give 384 blocks).
// <4,8,16> = 4 x 8 x 16 = 512 voxels for levels whose width is a multiple of 16 but not of 32 (48-wide latents of 192^2
// patches): an A tile of 32 rows is then two W-lines of 16 (lanes 16-31 read the next H line); with 32-byte rows that costs
// 2-way bank conflicts on half of the A-fragment reads (no 1-bit slot swizzle separates rows 16 AND 24 apart), which the
// 0.75 ds_read_b128 per MFMA of this kernel absorbs.
template <int TD_, int TH_, int TW_ = 32>
struct HmCfg {
    static constexpr int TD = TD_, TH = TH_, TW = TW_;
    static constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
    static constexpr int HV = HD * HH * HW;                 // <4,4>: 1224 halo voxels
    static constexpr int HALO_INSTR = (HV + 31) / 32;       // <4,4>: 39 wave-DMAs of 32 voxels x 32 B
    static constexpr int HALO_BYTES = HALO_INSTR * 1024;
    static constexpr int BM = TD * TH * TW;                 // 512 / 384
    static constexpr int BN = 128;
    static constexpr int TAP_BYTES = BN * 32;               // 4096
    static constexpr int WSLOT_BYTES = 3 * TAP_BYTES;       // 12288
    static constexpr int NWS = 4;                           // weight ring: the DMA of step s+4 is issued when step s's slot is
                                                            // drained (5 slots / 3 steps in flight measured 2-4 % slower)
    static constexpr int NWAVE = BM / 64;
    static constexpr int NTH = 64 * NWAVE;                  // == BM: one output row per thread in the row table
    static constexpr int NPIECE = (HALO_INSTR + NWAVE - 1) / NWAVE;   // halo DMAs per wave and chunk (issued at g < NPIECE)
    static constexpr int OFF_W = 2 * HALO_BYTES;
    static constexpr int LOOP_END = OFF_W + NWS * WSLOT_BYTES;
    static constexpr int OFF_ROW = LOOP_END > BM * BN * 2 ? LOOP_END : BM * BN * 2;   // behind the loop buffers and the epilogue tile
    static constexpr int OFF_CS = OFF_ROW + BM * 8;         // column-sum scratch [NWAVE][BN][2] floats
    static constexpr int LDS_BYTES = OFF_CS + NWAVE * BN * 8;
    static_assert(BM % 64 == 0 && NPIECE <= 6 && LDS_BYTES <= 160 * 1024, "unsupported tile");
};

__device__ __forceinline__ void hm_wait_vm(int allowed) {   // wave-uniform `allowed`
    switch (allowed) {
        case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory"); break;
    }
}

template <int TD_, int TH_, int TW_ = 32, bool NIN = false>
__global__ void __attribute__((amdgpu_flat_work_group_size(1, HmCfg<TD_, TH_, TW_>::NTH)))
conv3_halo32m_kernel(const Conv3HaloParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = HmCfg<TD_, TH_, TW_>;
    constexpr int TD = Cfg::TD, TH = Cfg::TH, TW = Cfg::TW, HH = Cfg::HH, HW = Cfg::HW, HV = Cfg::HV;
    constexpr int HALO_INSTR = Cfg::HALO_INSTR, HALO_BYTES = Cfg::HALO_BYTES, BM = Cfg::BM, BN = Cfg::BN;
    constexpr int TAP_BYTES = Cfg::TAP_BYTES, WSLOT_BYTES = Cfg::WSLOT_BYTES, NWS = Cfg::NWS, NWAVE = Cfg::NWAVE;
    constexpr int NTH = Cfg::NTH, NPIECE = Cfg::NPIECE, OFF_W = Cfg::OFF_W, OFF_ROW = Cfg::OFF_ROW, OFF_CS = Cfg::OFF_CS;
    (void)TD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long* s_rowoff = reinterpret_cast<long long*>(smem + OFF_ROW);
    float* s_cs = reinterpret_cast<float*>(smem + OFF_CS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // lines 2*wave, 2*wave+1 of the tile (line = ld*TH + lh)
    const int wm = wave;

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

    {   // row = line*32 + m, line = ld*TH + lh
        const int mm = tid % TW, line = tid / TW;
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
    const v4i_t rsw = h3_make_rsrc(reinterpret_cast<const char*>(p.w) + (long long)n0 * 32, 0x7fffffffu);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr3_t)smem;

    // halo DMA: piece j = wave + NWAVE*i covers halo voxels 32*j .. 32*j+31; lane -> voxel 32*j + lane/2, physical slot lane&1
    int hrel[NPIECE];
    unsigned hq16[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int j = wave + NWAVE * i;
        const int v = j * 32 + (lane >> 1);
        const int hd = v / (HH * HW), rem = v - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        const int gd = d0 + p.dshift - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
        bool ok = (j < HALO_INSTR) && (v < HV) && gd >= 0 && gd < p.Di && gh >= 0 && gh < p.Hi && gw >= 0 && gw < p.Wi;
        // normalise-on-load of a depth-sharded input: a halo slice at a volume end holds zeros = the conv's zero padding, which
        // must stay zero: treat it as out of volume (hardware zero fill, no normalisation)
        if (NIN && ((p.nin_pad_lo && gd == 0) || (p.nin_pad_hi && gd == p.Di - 1))) ok = false;
        hrel[i] = ok ? ((gd - dlo) * p.Hi + gh) * p.Wi + gw : -1;
        hq16[i] = (unsigned)(((lane & 1) ^ ((v >> 3) & 1)) * 16);   // logical 8-channel half stored in this slot
    }
    const int C1 = p.C1, C2 = p.C2, CoutPad = p.CoutPad, nchunks = p.nchunks;

    auto issue_halo = [&](int cc, int i, int hoff) -> int {
        const int j = wave + NWAVE * i;
        if (j >= HALO_INSTR) return 0;
        const int ch0 = cc * 16;
        const bool second = ch0 >= C1;
        const unsigned cbytes = (unsigned)((second ? C2 : C1) * 2);
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((second ? ch0 - C1 : ch0) * 2);
        int hsel = hrel[0];
        unsigned qsel = hq16[0];
#pragma unroll
        for (int q = 1; q < NPIECE; ++q) {
            hsel = (i == q) ? hrel[q] : hsel;
            qsel = (i == q) ? hq16[q] : qsel;
        }
        const unsigned voff = hsel >= 0 ? (unsigned)hsel * cbytes + qsel : 0x80000000u;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + hoff + j * 1024));
        if (second)
            h3_dma16(rs2, dst, voff, soff);
        else
            h3_dma16(rs1, dst, voff, soff);
        return 1;
    };
    // weights of step s = 3 taps x 4 KB = 12 pieces of 1 KB: wave w copies piece w, waves 0..3 also piece 8 + w
    const unsigned w_voff = (unsigned)lane * 16u;
    auto issue_weights = [&](int s) -> int {
        const int cc = s / 9, g = s - cc * 9;
        const unsigned slot = lds0 + OFF_W + (s % NWS) * WSLOT_BYTES;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = wave + NWAVE * k;
            if (piece < 12) {
                const int tap = piece >> 2, quarter = piece & 3;
                const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(
                    ((cc * 27 + g * 3 + tap) * CoutPad) * 32 + quarter * 1024);
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(slot + piece * 1024));
                h3_dma16(rsw, dst, w_voff, soff);
                ++cnt;
            }
        }
        return cnt;
    };

    // ---- normalise-on-load --------------------------------------------------------------------------------------------------
    // per-channel (scale, shift, time bias) tables behind the loop / epilogue buffers; every lane later rewrites, in place, the
    // 16 bytes (8 channels) of each halo piece it DMA'd itself, once that piece has landed (covered by the wave's own vmcnt
    // wait); the barrier that already separates "chunk landed" from "chunk read" publishes the result to the other waves.
    constexpr bool nin = NIN;     // normalise-on-load is its own instantiation: the plain kernel carries none of it
    float* s_nsc = reinterpret_cast<float*>(smem + Cfg::LDS_BYTES);
    float* s_nsh = s_nsc + C1;
    float* s_ntb = s_nsh + C1;
    if (nin) {
        const int cpg = C1 / p.nin_groups;
        long long trow = nb;
        if (p.nin_step_ptr) trow += (long long)(*p.nin_step_ptr) * p.nin_n_total;
        for (int ch = tid; ch < C1; ch += NTH) {
            const int gi = ch / cpg;
            const double mean = p.nin_sums[((long long)nb * p.nin_groups + gi) * 2 + 0] / p.nin_count;
            double var = p.nin_sums[((long long)nb * p.nin_groups + gi) * 2 + 1] / p.nin_count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)p.nin_eps));
            const float sc = p.nin_gamma[ch] * rstd;
            s_nsc[ch] = sc;
            s_nsh[ch] = p.nin_beta[ch] - (float)mean * sc;
            s_ntb[ch] = p.nin_tbias ? p.nin_tbias[trow * p.nin_tb_stride + ch] : 0.0f;
        }
    }
    const int nq8 = ((lane & 1) ^ ((lane >> 4) & 1)) * 8;   // first channel (within a 16-channel chunk) of this lane's 16 bytes:
                                                            // the same for every piece (hq16 does not depend on the piece)
    auto xform_piece = [&](int i, int hoff, int ccn) {
        const int j = wave + NWAVE * i;
        if (j >= HALO_INSTR) return;
        int hsel = hrel[0];
#pragma unroll
        for (int q = 1; q < NPIECE; ++q) hsel = (i == q) ? hrel[q] : hsel;
        if (hsel < 0) return;                               // zero padding stays zero
        uint4* ptr = reinterpret_cast<uint4*>(smem + hoff + j * 1024 + lane * 16);
        const uint4 raw = *ptr;
        const int ch = ccn * 16 + nq8;
        const float4 sc0 = *reinterpret_cast<const float4*>(s_nsc + ch), sc1 = *reinterpret_cast<const float4*>(s_nsc + ch + 4);
        const float4 sh0 = *reinterpret_cast<const float4*>(s_nsh + ch), sh1 = *reinterpret_cast<const float4*>(s_nsh + ch + 4);
        const float4 tb0 = *reinterpret_cast<const float4*>(s_ntb + ch), tb1 = *reinterpret_cast<const float4*>(s_ntb + ch + 4);
        const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
        const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
        const float tb[8] = {tb0.x, tb0.y, tb0.z, tb0.w, tb1.x, tb1.y, tb1.z, tb1.w};
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = __uint_as_float(w[k] << 16) * sc[2 * k] + sh[2 * k];
            float b = __uint_as_float(w[k] & 0xffff0000u) * sc[2 * k + 1] + sh[2 * k + 1];
            if (p.nin_silu) {
                a = silu_f(a);
                b = silu_f(b);
            }
            o[k] = pack_bf16x2(a + tb[2 * k], b + tb[2 * k + 1]);
        }
        *ptr = make_uint4(o[0], o[1], o[2], o[3]);
    };

    // fragment addressing: lane -> row r = lane & 31 of the 32-row operand tile, k-half hk = lane >> 5
    const int hk = lane >> 5, r = lane & 31;
    int vline[2];                                            // halo voxel of this lane's row in the wave's two A tiles, before taps
#pragma unroll
    for (int i = 0; i < 2; ++i) {                            // A tile = rows [(2*wm + i)*32, +32) = 32 / TW whole W-lines
        const int line = (2 * wm + i) * (32 / TW) + r / TW;
        vline[i] = ((line / TH) * HH + (line % TH)) * HW + r % TW;
    }
    const int b_off = r * 32 + ((hk ^ ((r >> 3) & 1)) << 4);    // n-tile j: + j*1024 (rows +32 keep (row>>3)&1)

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    bf16x8 fa0[2], fb0[4], fa1[2], fb1[4], fa2[2], fb2[4];
    const int S = nchunks * 9;

#define HN_LOAD(FA, FB, HBUF, WBUF, VS, KW)                                                                    \
    {                                                                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                     \
            const int v_ = (VS) + vline[i_] + (KW);                                                            \
            FA[i_] = *reinterpret_cast<const bf16x8*>((HBUF) + v_ * 32 + ((hk ^ ((v_ >> 3) & 1)) << 4));       \
        }                                                                                                      \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                       \
            FB[j_] = *reinterpret_cast<const bf16x8*>((WBUF) + b_off + (KW) * TAP_BYTES + j_ * 1024);          \
    }
    // one phase = the MFMAs of one tap (set c) interleaved with the fragment loads of the tap two phases ahead (set l)
#define HN_PHASE(FAc, FBc, FAl, FBl, HBUF, WBUF, VS, KW)                                                       \
    {                                                                                                          \
        HN_LOAD(FAl, FBl, HBUF, WBUF, VS, KW);                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)       \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FAc[i_], FBc[j_], acc[i_][j_], 0, 0, 0);     \
        _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                                                     \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
        }                                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }

    // prologue: halo of chunk 0, weights of steps 0..3
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) issue_halo(0, i, 0);
    int n_prev1 = 0, n_prev2 = 0;   // DMAs this wave issued in the previous step / the one before (wave-uniform)
#pragma unroll
    for (int s = 0; s < NWS; ++s)
        if (s < S) issue_weights(s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (nin) {
        __syncthreads();             // the coefficient tables are complete
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) xform_piece(i, 0, 0);
    }
    __syncthreads();
    HN_LOAD(fa0, fb0, smem, smem + OFF_W, 0, 0);
    HN_LOAD(fa1, fb1, smem, smem + OFF_W, 0, 1);
    __builtin_amdgcn_sched_barrier(0);

    int cc = 0, g = 0;
    // DMA issue placement (p.dbg & 16 selects the old one, right behind the barrier, for A/B timing): the pieces of a step are
    // issued AFTER the first MFMA phase that follows the barrier, i.e. while that phase's 8 MFMAs drain through the matrix
    // pipe -- behind the barrier nothing of this wave is queued there, and both SIMD partners would spend their ~60-100
    // issue cycles per piece (MI355X_MICROARCH.md, LDS-DMA piece issue cost) with the pipe idle.
#define HM_SYNC()                                                                                              \
    {                                                                                                          \
        /* slot s is drained by this wave once the loads above have returned.  Needed next: weights of step    \
           s+1 (issued 3 steps ago); the next chunk's halo pieces are issued at g < NPIECE <= 6, so they are   \
           older than the two most recent steps' DMAs by the time they are read (g = 8). */                    \
        hm_wait_vm(n_prev1 + n_prev2);                                                                         \
        if (!(CTSI_DBG(p.dbg, 64))) __builtin_amdgcn_s_barrier();   /* (64: timing-only ablation) */                    \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
#define HM_ISSUE()                                                                                             \
    {                                                                                                          \
        int issued = 0;                                                                                        \
        if (s + NWS < S && !(CTSI_DBG(p.dbg, 2))) issued += issue_weights(s + NWS);                                     \
        if (g < NPIECE && cc + 1 < nchunks && !(CTSI_DBG(p.dbg, 1)))                                                    \
            issued += issue_halo(cc + 1, g, ((cc + 1) & 1) * HALO_BYTES);                                      \
        n_prev2 = n_prev1;                                                                                     \
        n_prev1 = issued;                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
    const bool issue_early = (CTSI_DBG(p.dbg, 16)) != 0, issue_last = (CTSI_DBG(p.dbg, 32)) != 0;
    for (int s = 0; s < S; ++s) {
        const char* hbuf = smem + (cc & 1) * HALO_BYTES;
        const char* wbuf = smem + OFF_W + (s % NWS) * WSLOT_BYTES;
        const int kd = g / 3, kh = g - kd * 3;
        const int vs = (kd * HH + kh) * HW;
        HN_PHASE(fa0, fb0, fa2, fb2, hbuf, wbuf, vs, 2);
        HM_SYNC();
        if (issue_early) HM_ISSUE();
        /* normalise-on-load: the piece issued at step g-3 is covered by the wait above (older than the two most recent
           steps); its rewrite is published by the following barriers, the last one (g = 7) by barrier(8), after which the
           next chunk is first read */
        if (nin && wave < 4 && g >= 3 && g - 3 < NPIECE && cc + 1 < nchunks)
            xform_piece(g - 3, ((cc + 1) & 1) * HALO_BYTES, cc + 1);
        int g2 = g + 1, cc2 = cc;
        if (g2 == 9) {
            g2 = 0;
            ++cc2;
        }
        // (the last step's look-ahead loads read stale but in-bounds LDS and are never consumed)
        const char* hbuf2 = smem + (cc2 & 1) * HALO_BYTES;
        const char* wbuf2 = smem + OFF_W + ((s + 1) % NWS) * WSLOT_BYTES;
        const int kd2 = g2 / 3, kh2 = g2 - kd2 * 3;
        const int vs2 = (kd2 * HH + kh2) * HW;
        __builtin_amdgcn_sched_barrier(0);
        HN_PHASE(fa1, fb1, fa0, fb0, hbuf2, wbuf2, vs2, 0);
        if (!issue_early && !issue_last) HM_ISSUE();
        HN_PHASE(fa2, fb2, fa1, fb1, hbuf2, wbuf2, vs2, 1);
        if (issue_last) HM_ISSUE();
        // normalise-on-load, second half of the waves: the SIMD partners of waves 0-3 rewrite their piece at the END of
        // the step, so that one partner's VALU burst runs beside the other's MFMA phases instead of both stalling the
        // matrix pipe at the same time (still before the next barrier, which publishes it)
        if (nin && wave >= 4 && g >= 3 && g - 3 < NPIECE && cc + 1 < nchunks) {
            xform_piece(g - 3, ((cc + 1) & 1) * HALO_BYTES, cc + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        g = g2;
        cc = cc2;
    }
#undef HM_SYNC
#undef HM_ISSUE
#undef HN_PHASE
#undef HN_LOAD
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: bias, GroupNorm column sums, the 512 x 128 bf16 tile through LDS (128 KB), 16-byte row stores -----
    if (CTSI_DBG(p.dbg, 8)) return;
    bf16_t* s_tile = reinterpret_cast<bf16_t*>(smem);  // [BM][BN] bf16
    const bool want_sums = p.colsum != nullptr;
    const int lhi = lane >> 5, lcol = lane & 31;
    unsigned vbits[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        unsigned vb = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (wm * 2 + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * lhi;
            vb |= (unsigned)(s_rowoff[row] >= 0) << q;
        }
        vbits[i] = vb;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = j * 32 + lcol;
        const int co = n0 + col;
        const float bv = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.0f;
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16_t* trow = s_tile + ((wm * 2 + i) * 32 + 4 * lhi) * BN + col;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float v = acc[i][j][q] + bv;
                trow[((q & 3) + 8 * (q >> 2)) * BN] = f32_to_bf16(v);
                if (want_sums) {
                    const float vm = ((vbits[i] >> q) & 1u) ? v : 0.0f;
                    s1 += vm;
                    s2 += vm * vm;
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
    __syncthreads();
    if (want_sums && tid < BN) {
        float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
        for (int q = 0; q < NWAVE; ++q) {
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

// ---- weight packing: fp32 (cout, cin, 3,3,3) -> bf16 [chunk16][tap][cout_pad][16], 16-B halves swizzled per row ----
__global__ void conv3_halo_c16_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int CoutPad,
                                           int CinW, int nchunks) {
    const long long total = (long long)nchunks * 27 * CoutPad * 16;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 15);                 // physical element within the 32-B row
        const long long row = idx >> 4;                // (chunk*27 + tap)*CoutPad + cout
        const int co = (int)(row % CoutPad);
        const long long ct = row / CoutPad;
        const int tap = (int)(ct % 27), cc = (int)(ct / 27);
        const int q = (e >> 3) ^ (((co & 63) >> 3) & 1);   // logical half stored in this physical slot (row = co within the 64-cout tile)
        const int ci = cc * 16 + q * 8 + (e & 7);
        float v = 0.0f;
        if (co < Cout && ci < CinW) v = w[((long long)co * CinW + ci) * 27 + tap];
        out[idx] = f32_to_bf16(v);
    }
}

extern "C" int ctsi_conv3_halo_c16_pack(const float* w, void* packed, int cout, int cout_pad, int cin, int cin_w,
                                        void* stream) {
    CTSI_CHECK_ARG(w && packed && cin % 16 == 0 && cout_pad % 64 == 0, "ctsi_conv3_halo_c16_pack: bad arguments");
    const int nchunks = cin / 16;
    const long long total = (long long)nchunks * 27 * cout_pad * 16;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(conv3_halo_c16_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)packed,
                       cout, cout_pad, cin_w, nchunks);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

extern "C" int ctsi_conv3_halo_m512_launch(const Conv3HaloParams* hp, int tile /* 0: 4x4x32, 2: 4x8x16 */, void* stream) {
    using C44 = HmCfg<4, 4>;
    using C48 = HmCfg<4, 8, 16>;
    auto k44 = conv3_halo32m_kernel<4, 4, 32, false>;
    auto k48 = conv3_halo32m_kernel<4, 8, 16, false>;
    auto k44n = conv3_halo32m_kernel<4, 4, 32, true>;
    auto k48n = conv3_halo32m_kernel<4, 8, 16, true>;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)k44, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k48, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k44n, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k48n, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const int grid = hp->mtiles * hp->ntiles_n;
    const int extra = hp->nin_sums ? 3 * hp->C1 * (int)sizeof(float) : 0;   // normalise-on-load coefficient tables
    if (C44::LDS_BYTES + extra > 160 * 1024 || C48::LDS_BYTES + extra > 160 * 1024) {
        ctsi_set_error("conv3_halo32m: normalise-on-load tables for %d channels do not fit the LDS", hp->C1);
        return CTSI_ERR_UNSUPPORTED;
    }
    if (hp->nin_sums) {
        if (tile == 2)
            hipLaunchKernelGGL(k48n, dim3(grid), dim3(C48::NTH), C48::LDS_BYTES + extra, (hipStream_t)stream, *hp);
        else
            hipLaunchKernelGGL(k44n, dim3(grid), dim3(C44::NTH), C44::LDS_BYTES + extra, (hipStream_t)stream, *hp);
    } else if (tile == 2)
        hipLaunchKernelGGL(k48, dim3(grid), dim3(C48::NTH), C48::LDS_BYTES, (hipStream_t)stream, *hp);
    else
        hipLaunchKernelGGL(k44, dim3(grid), dim3(C44::NTH), C44::LDS_BYTES, (hipStream_t)stream, *hp);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
