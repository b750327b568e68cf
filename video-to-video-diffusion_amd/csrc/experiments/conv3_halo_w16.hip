// EXPERIMENT / A-B REFERENCE, not part of libctsi.so (compile-checked by `make experiments`): the round-1 16-wide halo-tile
// kernel on v_mfma_f32_16x16x32_bf16 (tile 4 x 4 x 16, 32-channel chunks, two weight slots, no look-ahead).  Superseded by
// conv3_halo32_kernel<4,4,16,2,2> / <3,4,16,3,1> (+15-20 % on the 48 x 16 x 16 level, profiles/r02_notes.md).
//
// 3x3x3 stride-1 convolution with an LDS-staged D x H x W input halo tile (gfx950 / MI355X).
//
// The gather-GEMM kernel (conv_mfma.hip) re-stages the activation slab for each of the 27 taps and runs
// at a constant L2->LDS fill rate (~8-10 TB/s): bytes per flop bound it.  This kernel stages, per
// 32-channel chunk, the (TD+2) x (TH+2) x (TW+2) = 6 x 6 x 18 halo tile of the input ONCE (buffer_load ... lds,
// out-of-volume voxels are out-of-range buffer offsets -> hardware zero fill = the conv's zero padding) and
// lets all 27 taps read their A operand from it at shifted LDS addresses; only the weights stream per tap.
// Fill traffic per 256x128 output tile and 32-channel chunk: 41 KB halo + 27 x 8 KB weights for 56.6 MFLOP
// (220 flop/B, vs 85 flop/B for the 256x128 gather tile).
//
// MFMA: v_mfma_f32_16x16x32_bf16.  An A operand tile is one W-line of 16 output voxels x 32 channels, so a
// 32-lane half of a ds_read_b64 always touches 16 CONSECUTIVE 64-byte halo rows whatever the tap shift is;
// with the 16-byte chunk swizzle  slot = chunk ^ ((row >> 2) & 3)  and the two 8-byte halves read in
// opposite order by odd/even k-groups, every fragment read is bank-conflict free.  The weight tile uses the
// same 64-byte-row layout (pre-swizzled at pack time, so its DMA is a linear copy) and the same read code;
// the half swap permutes k identically in both operands, which leaves the dot product unchanged.
//
// Block = 8 waves (4 along M x 2 along N), output tile 4 x 4 x 16 voxels x 128 couts, 64 fp32 accumulators
// per lane.  Per step = one (kd, kh) pair = 3 taps: 3 weight DMAs + <=1 halo DMA per wave, 48 ds_read_b64,
// 48 MFMA, one barrier.  Halo tiles are double buffered across chunks, weight slots across steps.
#include "../conv3_halo_common.h"
#include <string.h>

// Tile shapes: <4,4> = 4 x 4 x 16 voxels (8 waves) and <6,2> = 6 x 2 x 16 voxels (6 waves; 48 x 16 x 16 volumes split
// into exactly 64 tiles -> 256 blocks with 4 cout tiles: one per CU instead of 192 blocks on 256 CUs).
template <int TD_, int TH_>
struct H3Cfg {
    static constexpr int TD = TD_, TH = TH_, TW = 16;
    static constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
    static constexpr int HV = HD * HH * HW;               // <4,4>: 648 halo voxels
    static constexpr int HALO_INSTR = (HV + 15) / 16;     // <4,4>: 41 DMA wave-instructions of 16 voxels x 64 B
    static constexpr int HALO_BYTES = HALO_INSTR * 1024;
    static constexpr int BM = TD * TH * TW;               // 256 / 192
    static constexpr int BN = 128;
    static constexpr int WM = BM / 64;                    // M-waves (64 voxels = 4 W-lines each)
    static constexpr int NW = 2 * WM, NTH = 64 * NW;
    static constexpr int NPIECE = (HALO_INSTR + NW - 1) / NW;   // halo DMAs per wave and chunk (<= 6: issued at g < 6)
    static constexpr int WPIECE = 24 / NW;                // 1 KB weight pieces per wave and step (3 taps x 8 KB)
    static constexpr int WSLOT_BYTES = 3 * BN * 64;       // 3 taps x 128 couts x 32 ch bf16
    static constexpr int OFF_W = 2 * HALO_BYTES;
    static constexpr int OFF_ROW = OFF_W + 2 * WSLOT_BYTES;
    static constexpr int OFF_CS = OFF_ROW + BM * 8;
    static constexpr int LDS_BYTES = OFF_CS + WM * BN * 8;  // column-sum scratch [WM][BN][2] floats
    static_assert(BM % 64 == 0 && 24 % NW == 0 && NPIECE <= 6, "unsupported tile");
};

template <int TD_, int TH_>
__global__ void __attribute__((amdgpu_flat_work_group_size(1, H3Cfg<TD_, TH_>::NTH)))
conv3_halo_kernel(const Conv3HaloParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = H3Cfg<TD_, TH_>;
    constexpr int TD = Cfg::TD, TH = Cfg::TH, TW = Cfg::TW, HH = Cfg::HH, HW = Cfg::HW, HV = Cfg::HV;
    constexpr int HALO_INSTR = Cfg::HALO_INSTR, HALO_BYTES = Cfg::HALO_BYTES, BM = Cfg::BM, BN = Cfg::BN;
    constexpr int WM = Cfg::WM, NW = Cfg::NW, NTH = Cfg::NTH, NPIECE = Cfg::NPIECE, WPIECE = Cfg::WPIECE;
    constexpr int WSLOT_BYTES = Cfg::WSLOT_BYTES, OFF_W = Cfg::OFF_W, OFF_ROW = Cfg::OFF_ROW, OFF_CS = Cfg::OFF_CS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long* s_rowoff = reinterpret_cast<long long*>(smem + OFF_ROW);
    float* s_cs = reinterpret_cast<float*>(smem + OFF_CS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- block decode: n-tiles of one m-tile adjacent, XCD-contiguous m ranges --------------------------
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

    // output row offsets (row = line*16 + m, line = ld*TH + lh)
    if (tid < BM) {
        const int m = tid & 15, line = tid >> 4;
        const int d = d0 + line / TH, h = h0 + line % TH, w = w0 + m;
        long long off = -1;
        if (d < p.Do && h < p.Ho && w < p.Wo)
            off = ((((long long)nb * p.Do + d) * p.Ho + h) * p.Wo + w) * p.cout_stride + p.c_off;
        s_rowoff[tid] = off;
    }

    // ---- halo DMA descriptors ------------------------------------------------------------------------------
    // base = first input plane this tile can touch; per-lane voxel offsets are relative to it
    int dlo = d0 + p.dshift - 1;
    dlo = dlo < 0 ? 0 : dlo;
    const long long basevox = ((long long)(nb * p.Di + dlo) * p.Hi) * p.Wi;
    const char* b1 = reinterpret_cast<const char*>(p.x1) + basevox * p.C1 * 2;
    const char* b2 = reinterpret_cast<const char*>(p.x2) + basevox * p.C2 * 2;
    const v4i_t rs1 = h3_make_rsrc(b1, 0x7fffffffu);
    const v4i_t rs2 = h3_make_rsrc(b2, 0x7fffffffu);
    const char* wb = reinterpret_cast<const char*>(p.w) + (long long)n0 * 64;
    const v4i_t rsw = h3_make_rsrc(wb, 0x7fffffffu);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr3_t)smem;   // LDS byte address of the dynamic region

    // this wave issues halo instructions j = wave + NW*i (i < NPIECE, j < HALO_INSTR); lane -> voxel 16j + (lane>>2)
    const int hq = (lane & 3) ^ (lane >> 4);  // logical 8-channel chunk landing in this lane's 16-B slot
    int hrel[NPIECE];                          // voxel index relative to basevox, or -1
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int j = wave + NW * i;
        const int v = j * 16 + (lane >> 2);
        const int hd = v / (HH * HW), rem = v - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        const int gd = d0 + p.dshift - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
        const bool ok = (j < HALO_INSTR) && (v < HV) && gd >= 0 && gd < p.Di && gh >= 0 && gh < p.Hi && gw >= 0 &&
                        gw < p.Wi;
        hrel[i] = ok ? ((gd - dlo) * p.Hi + gh) * p.Wi + gw : -1;
    }
    const unsigned w_voff = (unsigned)lane * 16u;
    const int C1 = p.C1, C2 = p.C2, CoutPad = p.CoutPad, nchunks = p.nchunks;

    // `hoff`/`woff`: byte offset of the destination buffer inside the dynamic LDS region
    auto issue_halo = [&](int cc, int i, int hoff) {  // one DMA instruction (16 voxels) of chunk cc
        const int j = wave + NW * i;
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
    auto issue_weights = [&](int s, int woff) {  // the 3 taps of step s = 24 pieces of 1 KB (16 cout rows): WPIECE per wave
        const int cc = s / 9, g = s - cc * 9;
#pragma unroll
        for (int i = 0; i < WPIECE; ++i) {
            const int piece = wave * WPIECE + i;
            const int tap = piece >> 3, sub = piece & 7;
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(((cc * 27 + g * 3 + tap) * CoutPad) * 64 + sub * 1024);
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + woff + tap * (BN * 64) + sub * 1024));
            h3_dma16(rsw, dst, w_voff, soff);
        }
    };

    // ---- fragment addressing -----------------------------------------------------------------------------------
    const int kg = lane >> 4, m = lane & 15;
    const int half0 = (kg & 1) * 8;
    int vline[4];                                         // halo voxel of this wave's 4 W-lines (line = 4*wm + i) before taps
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int line = wm * 4 + i;
        vline[i] = ((line / TH) * HH + (line % TH)) * HW + m;
    }
    const int rowb = wn * 64 + m;                          // weight row of n-tile 0 of this wave
    const int b_off = rowb * 64 + ((kg ^ ((rowb >> 2) & 3)) << 4) + half0;   // + j*1024 + kw*BN*64

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- software-pipelined main loop ---------------------------------------------------------------------------------
    // Block (s, kw) always uses fragment set F[kw]; per step:
    //   LOAD F1(s,1) | MFMA F0 | LOAD F2(s,2) | MFMA F1 | lgkmcnt(0), vmcnt(0), barrier | DMA W(s+2), halo piece |
    //   LOAD F0(s+1,0) | MFMA F2
    // so every LDS read has 16 MFMAs (256 cycles) of cover and the barrier sits between two MFMA clusters.
    // DMA W(s+2) reuses weight slot s&1: every wave has finished reading it (its F2 load was waited for
    // before the barrier).  Halo pieces of chunk cc+1 go to the other halo buffer during steps g < 6 of chunk cc.
    bf16x8 fa0[4], fb0[4], fa1[4], fb1[4], fa2[4], fb2[4];
    const int S = nchunks * 9;
    // per-n-tile weight row offsets, made opaque so that hipcc cannot fuse two fragments' reads into
    // ds_read2st64_b64 (half the LDS rate and 32-bank addressing: it reintroduces bank conflicts)
    int boffj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        boffj[j] = j * 1024;
        asm volatile("" : "+v"(boffj[j]));
    }

#define H3_LOAD(FA, FB, HBUF, WBUF, VS, KW)                                                                    \
    {                                                                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                     \
            const int v_ = (VS) + vline[i_] + (KW);                                                            \
            const char* a_ = (HBUF) + v_ * 64 + ((kg ^ ((v_ >> 2) & 3)) << 4);                                 \
            const uint2 lo_ = *reinterpret_cast<const uint2*>(a_ + half0);                                     \
            const uint2 hi_ = *reinterpret_cast<const uint2*>(a_ + (half0 ^ 8));                               \
            const uint4 u_ = make_uint4(lo_.x, lo_.y, hi_.x, hi_.y);                                           \
            FA[i_] = *reinterpret_cast<const bf16x8*>(&u_);                                                    \
        }                                                                                                      \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                     \
            const char* b_ = (WBUF) + (KW) * (BN * 64) + boffj[j_];                                            \
            const uint2 lo_ = *reinterpret_cast<const uint2*>(b_);                                             \
            const uint2 hi_ = *reinterpret_cast<const uint2*>(b_ + ((half0 ^ 8) - half0));                     \
            const uint4 u_ = make_uint4(lo_.x, lo_.y, hi_.x, hi_.y);                                           \
            FB[j_] = *reinterpret_cast<const bf16x8*>(&u_);                                                    \
        }                                                                                                      \
    }
#define H3_MFMA(FA, FB)                                                                                        \
    {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)       \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[i_], FB[j_], acc[i_][j_], 0, 0, 0);       \
        __builtin_amdgcn_s_setprio(0);                                                                         \
    }

    // prologue: halo of chunk 0, weights of steps 0 and 1
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) issue_halo(0, i, 0);
    issue_weights(0, OFF_W);
    if (S > 1) issue_weights(1, OFF_W + WSLOT_BYTES);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    H3_LOAD(fa0, fb0, smem, smem + OFF_W + b_off, 0, 0);

    int cc = 0, g = 0;
    for (int s = 0; s < S; ++s) {
        const char* hbuf = smem + (cc & 1) * HALO_BYTES;
        const char* wbuf = smem + OFF_W + (s & 1) * WSLOT_BYTES + b_off;
        const int kd = g / 3, kh = g - kd * 3;
        const int vs = (kd * HH + kh) * HW;
        H3_LOAD(fa1, fb1, hbuf, wbuf, vs, 1);
        __builtin_amdgcn_sched_barrier(0);
        H3_MFMA(fa0, fb0);
        __builtin_amdgcn_sched_barrier(0);
        H3_LOAD(fa2, fb2, hbuf, wbuf, vs, 2);
        __builtin_amdgcn_sched_barrier(0);
        H3_MFMA(fa1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        // next step's operands: every wave's reads of this step's weight slot are complete, all DMA landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < S) issue_weights(s + 2, OFF_W + (s & 1) * WSLOT_BYTES);
        if (g < NPIECE && cc + 1 < nchunks) issue_halo(cc + 1, g, ((cc + 1) & 1) * HALO_BYTES);
        int g2 = g + 1, cc2 = cc;
        if (g2 == 9) {
            g2 = 0;
            ++cc2;
        }
        if (s + 1 < S) {
            const int kd2 = g2 / 3, kh2 = g2 - kd2 * 3;
            H3_LOAD(fa0, fb0, smem + (cc2 & 1) * HALO_BYTES, smem + OFF_W + ((s + 1) & 1) * WSLOT_BYTES + b_off,
                    (kd2 * HH + kh2) * HW, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        H3_MFMA(fa2, fb2);
        __builtin_amdgcn_sched_barrier(0);
        g = g2;
        cc = cc2;
    }
#undef H3_LOAD
#undef H3_MFMA
    __syncthreads();

    // ---- epilogue: + bias, column sums, bf16 tile through LDS, full-row stores ------------------------------------------
    bf16_t* s_tile = reinterpret_cast<bf16_t*>(smem);  // [BM][BN] bf16 = 64 KB (the two halo buffers)
    const bool want_sums = p.colsum != nullptr;
    unsigned vbits = 0;  // validity of this lane's 16 rows: bit (i*4 + r)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (wm * 4 + i) * 16 + kg * 4 + r;
            vbits |= (unsigned)(s_rowoff[row] >= 0) << (i * 4 + r);
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = wn * 64 + j * 16 + m;
        const int co = n0 + col;
        const float bv = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.0f;
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (wm * 4 + i) * 16 + kg * 4 + r;
                const float v = acc[i][j][r] + bv;
                s_tile[row * BN + col] = f32_to_bf16(v);
                if (want_sums) {
                    const float vm = ((vbits >> (i * 4 + r)) & 1u) ? v : 0.0f;
                    s1 += vm;
                    s2 += vm * vm;
                }
            }
        if (want_sums) {
            s1 += __shfl_xor(s1, 16);
            s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (kg == 0) {
                s_cs[(wm * BN + col) * 2 + 0] = s1;
                s_cs[(wm * BN + col) * 2 + 1] = s2;
            }
        }
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
            if (off >= 0 && co < p.Cout) {
                const uint4 v = *reinterpret_cast<const uint4*>(s_tile + row * BN + ch * 8);
                *reinterpret_cast<uint4*>(y + off + co) = v;
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

extern "C" int ctsi_conv3_halo_w16_launch(const Conv3HaloParams* hp, void* stream) {
    using C44 = H3Cfg<4, 4>;
    auto k44 = conv3_halo_kernel<4, 4>;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)k44, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C44::LDS_BYTES);
        attr_done = true;
    }
    hipLaunchKernelGGL(k44, dim3(hp->mtiles * hp->ntiles_n), dim3(C44::NTH), C44::LDS_BYTES, (hipStream_t)stream, *hp);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
