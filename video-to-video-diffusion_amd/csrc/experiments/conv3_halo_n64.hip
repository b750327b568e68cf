// 3x3x3 stride-1 convolution, LDS halo tile 4 x 2 x 32 voxels, HALF-SIZE blocks: 4 waves, 64 couts, 16-channel chunks.
//
// conv3_halo32_kernel (conv3_halo.hip) fills the whole 160 KB of a CU's LDS with one 8-wave block, so a block's
// prologue (row table, first halo / weight DMA latency) and epilogue (64 KB output store) are never overlapped with
// MFMA work: 14 % of a 128-channel layer.  This variant halves every LDS region -- 32-byte halo rows (16 channels per
// chunk: 2 x 26 KB), 6 KB weight steps in a 4-deep ring (24 KB) -- so that TWO independent 4-wave blocks share a CU
// (78 KB each, one wave of each block per SIMD): while one block stores its tile or waits for its first DMAs the other
// one keeps the matrix cores busy, and the two blocks' barriers de-synchronise.
//
// Per wave: 64 voxels (one depth slice of the tile: 2 W-lines of 32) x 64 couts = 2 x 2 MFMA tiles of 32x32,
// v_mfma_f32_32x32x16_bf16 with k = the chunk's 16 channels: one MFMA per (tile, tap).  Fragment reads are
// ds_read_b128 (8 channels of a row); 16-byte slot swizzle  slot = khalf ^ ((row >> 3) & 1)  makes the 16-lane
// groups of a b128 read conflict-free for every tap shift (rows 8 or 24 apart land in different slots).
// Weight image: [chunk16][27 taps][cout_pad][16], pre-swizzled the same way; a step = the 3 kw taps of one (kd, kh)
// = 6 KB; the DMA of step s+4 is issued as soon as step s's slot is drained and waits are counted (the DMAs of the
// two most recent steps may stay in flight).
#include "../conv3_halo_common.h"
#include <stdlib.h>

namespace hn {
constexpr int TD = 4, TH = 2, TW = 32;
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int HV = HD * HH * HW;                 // 816 halo voxels
constexpr int HALO_INSTR = (HV + 31) / 32;       // 26 wave-DMAs of 32 voxels x 32 B
constexpr int HALO_BYTES = HALO_INSTR * 1024;    // 26624
constexpr int BM = TD * TH * TW;                 // 256
constexpr int BN = 64;
constexpr int TAP_BYTES = BN * 32;               // 2048
constexpr int WSLOT_BYTES = 3 * TAP_BYTES;       // 6144
constexpr int NWS = 4;
constexpr int NTH = 256;
constexpr int NPIECE = (HALO_INSTR + 3) / 4;     // 7 halo DMAs per wave and chunk
constexpr int OFF_W = 2 * HALO_BYTES;            // 53248
constexpr int OFF_ROW = OFF_W + NWS * WSLOT_BYTES;   // 77824
constexpr int LDS_BYTES = OFF_ROW + BM * 8;      // 79872 <= 80 KB: two blocks per CU
constexpr int OFF_CS = BM * BN * 2;              // epilogue only: column-sum scratch behind the 32 KB output tile
}  // namespace hn

__device__ __forceinline__ void hn_wait_vm(int allowed) {   // wave-uniform `allowed`
    switch (allowed) {
        case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
    }
}

__global__ void __launch_bounds__(256)
conv3_halo32n_kernel(const Conv3HaloParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace hn;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long* s_rowoff = reinterpret_cast<long long*>(smem + OFF_ROW);
    float* s_cs = reinterpret_cast<float*>(smem + OFF_CS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = depth slice of the tile this wave computes
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
        const int mm = tid & 31, line = tid >> 5;
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

    // halo DMA: piece j = wave + 4*i covers halo voxels 32*j .. 32*j+31; lane -> voxel 32*j + lane/2, physical slot lane&1
    int hrel[NPIECE];
    unsigned hq16[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int j = wave + 4 * i;
        const int v = j * 32 + (lane >> 1);
        const int hd = v / (HH * HW), rem = v - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        const int gd = d0 + p.dshift - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
        const bool ok = (j < HALO_INSTR) && (v < HV) && gd >= 0 && gd < p.Di && gh >= 0 && gh < p.Hi && gw >= 0 &&
                        gw < p.Wi;
        hrel[i] = ok ? ((gd - dlo) * p.Hi + gh) * p.Wi + gw : -1;
        hq16[i] = (unsigned)(((lane & 1) ^ ((v >> 3) & 1)) * 16);   // logical 8-channel half stored in this slot
    }
    const int C1 = p.C1, C2 = p.C2, CoutPad = p.CoutPad, nchunks = p.nchunks;

    auto issue_halo = [&](int cc, int i, int hoff) -> int {
        const int j = wave + 4 * i;
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
    // weights of step s = 3 taps x 2 KB = 6 pieces of 1 KB: wave w copies piece w, waves 0 and 1 also piece 4 + w
    const unsigned w_voff = (unsigned)lane * 16u;
    auto issue_weights = [&](int s) -> int {
        const int cc = s / 9, g = s - cc * 9;
        const unsigned slot = lds0 + OFF_W + (s & (NWS - 1)) * WSLOT_BYTES;
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = wave + 4 * k;
            if (piece < 6) {
                const int tap = piece >> 1, half = piece & 1;
                const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(
                    ((cc * 27 + g * 3 + tap) * CoutPad) * 32 + half * 1024);
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(slot + piece * 1024));
                h3_dma16(rsw, dst, w_voff, soff);
                ++cnt;
            }
        }
        return cnt;
    };

    // fragment addressing: lane -> row r = lane & 31 of the 32-row operand tile, k-half hk = lane >> 5
    const int hk = lane >> 5, r = lane & 31;
    const int va = (wm * HH) * HW + r;                       // halo voxel of (ld = wm, lh = 0) before taps
    const int b_off = r * 32 + ((hk ^ ((r >> 3) & 1)) << 4);    // n-tile 1: +1024 (rows +32 keep (row>>3)&1)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    bf16x8 fa0[2], fb0[2], fa1[2], fb1[2], fa2[2], fb2[2];
    const int S = nchunks * 9;

#define HN_LOAD(FA, FB, HBUF, WBUF, VS, KW)                                                                    \
    {                                                                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                     \
            const int v_ = (VS) + i_ * HW + (KW);                                                              \
            FA[i_] = *reinterpret_cast<const bf16x8*>((HBUF) + v_ * 32 + ((hk ^ ((v_ >> 3) & 1)) << 4));       \
        }                                                                                                      \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                       \
            FB[j_] = *reinterpret_cast<const bf16x8*>((WBUF) + b_off + (KW) * TAP_BYTES + j_ * 1024);          \
    }
    // one phase = the MFMAs of one tap (set c) interleaved with the fragment loads of the tap two phases ahead (set l)
#define HN_PHASE(FAc, FBc, FAl, FBl, HBUF, WBUF, VS, KW)                                                       \
    {                                                                                                          \
        HN_LOAD(FAl, FBl, HBUF, WBUF, VS, KW);                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)       \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FAc[i_], FBc[j_], acc[i_][j_], 0, 0, 0);     \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                     \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
        }                                                                                                      \
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
    __syncthreads();
    HN_LOAD(fa0, fb0, smem, smem + OFF_W, va, 0);
    HN_LOAD(fa1, fb1, smem, smem + OFF_W, va, 1);
    __builtin_amdgcn_sched_barrier(0);

    int cc = 0, g = 0;
    for (int s = 0; s < S; ++s) {
        const char* hbuf = smem + (cc & 1) * HALO_BYTES;
        const char* wbuf = smem + OFF_W + (s & (NWS - 1)) * WSLOT_BYTES;
        const int kd = g / 3, kh = g - kd * 3;
        const int vs = va + (kd * HH + kh) * HW;
        HN_PHASE(fa0, fb0, fa2, fb2, hbuf, wbuf, vs, 2);
        // slot s is drained by this wave once the loads above have returned.  Needed next: weights of step s+1
        // (issued 3 steps ago) and, before the chunk switch (g = 8), the whole next halo: its last piece was issued at
        // g = 6, and the only DMAs issued after it are the weights of step g = 7 -> those alone may stay in flight.
        hn_wait_vm(g == 8 ? n_prev1 : n_prev1 + n_prev2);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        int issued = 0;
        if (s + NWS < S && !(CTSI_DBG(p.dbg, 2))) issued += issue_weights(s + NWS);
        if (g < NPIECE && cc + 1 < nchunks && !(CTSI_DBG(p.dbg, 1))) issued += issue_halo(cc + 1, g, ((cc + 1) & 1) * HALO_BYTES);
        n_prev2 = n_prev1;
        n_prev1 = issued;
        int g2 = g + 1, cc2 = cc;
        if (g2 == 9) {
            g2 = 0;
            ++cc2;
        }
        // (the last step's look-ahead loads read stale but in-bounds LDS and are never consumed)
        const char* hbuf2 = smem + (cc2 & 1) * HALO_BYTES;
        const char* wbuf2 = smem + OFF_W + ((s + 1) & (NWS - 1)) * WSLOT_BYTES;
        const int kd2 = g2 / 3, kh2 = g2 - kd2 * 3;
        const int vs2 = va + (kd2 * HH + kh2) * HW;
        __builtin_amdgcn_sched_barrier(0);
        HN_PHASE(fa1, fb1, fa0, fb0, hbuf2, wbuf2, vs2, 0);
        HN_PHASE(fa2, fb2, fa1, fb1, hbuf2, wbuf2, vs2, 1);
        g = g2;
        cc = cc2;
    }
#undef HN_PHASE
#undef HN_LOAD
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: bias, bf16 tile through LDS, 16-byte row stores, GroupNorm column sums ---------------------------
    if (CTSI_DBG(p.dbg, 8)) return;
    bf16_t* s_tile = reinterpret_cast<bf16_t*>(smem);  // [BM][BN] bf16 = 32 KB
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
    for (int j = 0; j < 2; ++j) {
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
        for (int q = 0; q < 4; ++q) {
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

// (the [chunk16][tap][cout_pad][16] weight packing this kernel reads lives in ../conv3_halo_m512.hip: ctsi_conv3_halo_c16_pack)

extern "C" int ctsi_conv3_halo_n64_launch(const Conv3HaloParams* hp, void* stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)conv3_halo32n_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)hn::LDS_BYTES);
        attr_done = true;
    }
    const int grid = hp->mtiles * hp->ntiles_n;
    hipLaunchKernelGGL(conv3_halo32n_kernel, dim3(grid), dim3(hn::NTH), hn::LDS_BYTES, (hipStream_t)stream, *hp);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
