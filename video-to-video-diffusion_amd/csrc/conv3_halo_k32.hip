// 3x3x3 stride-1 convolution, 512-voxel LDS halo tile x 128 couts on v_mfma_f32_16x16x32_bf16: the K = 32 of one MFMA is
// TWO TAPS x 16 channels.
//
// Why a second 512-voxel kernel: conv3_halo32m_kernel (32x32x16 MFMAs) is not cycle- but CLOCK-bound on real data -- the same
// launch on all-zero operands runs 26-33 % faster (1236 -> 1579 TFLOP/s for 384->128 @48x128^2; profiles/r02_notes.md), i.e. the
// chip lowers its clock under the MFMA load (MI355X_MICROARCH.md, DVFS give-back) and cycles saved in the issue stream come back
// only in part.  What raises the clock for the same FLOPs: the 16x16x32 MFMA shape (give-back item 7: ~1.12-1.15 x the FLOP/s
// of the 32x32x16 loop at equal cycles) and fewer VALU instructions per MFMA (rule 28).  This kernel keeps conv3_halo32m's
// data movement (16-channel chunks, 32-byte halo rows, double-buffered halo, weight ring fed by LDS-DMA, one barrier per
// step, 128 KB output tile through LDS) and changes the arithmetic:
//   * the (chunk, tap) pairs of a block form ONE linear stream q = chunk * 27 + tap; a *unit* is two consecutive entries
//     (2 x 16 channels = K 32), a *step* two units (4 taps, 16 KB of weights, one barrier).  Lanes 0-31 of an operand read
//     the unit's first tap, lanes 32-63 the second: the tap shift is a per-lane LDS address, so pairs may straddle (kd, kh)
//     rows and chunks (27 taps per chunk is odd); the stream is zero-padded to whole steps in the packed image;
//   * per wave 64 voxels x 128 couts = 4 x 8 tiles of 16x16: 32 MFMAs and 12 ds_read_b128 per unit -- the same LDS bytes per
//     FLOP as the 32x32x16 form -- with every fragment address = one per-lane base + a per-unit tap offset + an immediate
//     (4 VALU per 32 MFMAs instead of ~5 per fragment);
//   * no LDS swizzle: a 16-lane group of a ds_read_b128 covers 16 consecutive 32-byte rows, rows r and r + 8 (same banks)
//     always with DIFFERENT 16-byte halves (group lanes {0-3, 12-15} read half h, {4-11} read half 1 - h), conflict-free as laid
//     out by the DMA.
#include "conv3_halo_common.h"
#include <stdlib.h>

template <int TD_, int TH_, int TW_ = 32>
struct HkCfg {
    static constexpr int TD = TD_, TH = TH_, TW = TW_;
    static constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
    static constexpr int HV = HD * HH * HW;                 // <4,4,32>: 1224 halo voxels
    static constexpr int HALO_INSTR = (HV + 31) / 32;       // wave-DMAs of 32 voxels x 32 B
    static constexpr int HALO_BYTES = HALO_INSTR * 1024;
    static constexpr int BM = TD * TH * TW;                 // 512
    static constexpr int BN = 128;
    static constexpr int TAP_BYTES = BN * 32;               // 4096: [128 couts][16 channels] bf16
    static constexpr int STEP_TAPS = 4;
    static constexpr int WSLOT_BYTES = STEP_TAPS * TAP_BYTES;   // 16384
    static constexpr int NWS = 4;                           // three steps in flight + the one being read
    static constexpr int NWAVE = BM / 64;
    static constexpr int NTH = 64 * NWAVE;
    static constexpr int NPIECE = (HALO_INSTR + NWAVE - 1) / NWAVE;   // halo DMAs per wave and chunk
    static constexpr int OFF_W = 2 * HALO_BYTES;
    static constexpr int LOOP_END = OFF_W + NWS * WSLOT_BYTES;
    static constexpr int OFF_ROW = LOOP_END > BM * BN * 2 ? LOOP_END : BM * BN * 2;
    static constexpr int OFF_CS = OFF_ROW + BM * 8;
    static constexpr int LDS_BYTES = OFF_CS + NWAVE * BN * 8;
    static constexpr int LPW = 64 / TW;                     // W-lines per wave
    static_assert(NWAVE == 8 && NPIECE <= 5 && TH % LPW == 0 && LDS_BYTES <= 160 * 1024, "unsupported tile");
    // A tile i (rows [16 i, 16 i + 16) of the wave's 64): halo-voxel offset from the wave's first voxel
    static constexpr int a_imm(int i) { return (((16 * i) / TW) * HW + (16 * i) % TW) * 32; }
};

__device__ __forceinline__ void hk_wait_vm(int allowed) {   // wave-uniform `allowed`
    switch (allowed) {
        case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
    }
}

template <int TD_, int TH_, int TW_ = 32>
__global__ void __attribute__((amdgpu_flat_work_group_size(1, 512)))
conv3_halo_k32_kernel(const Conv3HaloParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = HkCfg<TD_, TH_, TW_>;
    constexpr int TH = Cfg::TH, TW = Cfg::TW, HH = Cfg::HH, HW = Cfg::HW, HV = Cfg::HV;
    constexpr int HALO_INSTR = Cfg::HALO_INSTR, HALO_BYTES = Cfg::HALO_BYTES, BM = Cfg::BM, BN = Cfg::BN;
    constexpr int TAP_BYTES = Cfg::TAP_BYTES, WSLOT_BYTES = Cfg::WSLOT_BYTES, NWS = Cfg::NWS, NWAVE = Cfg::NWAVE;
    constexpr int NTH = Cfg::NTH, NPIECE = Cfg::NPIECE, OFF_W = Cfg::OFF_W, OFF_ROW = Cfg::OFF_ROW, OFF_CS = Cfg::OFF_CS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long* s_rowoff = reinterpret_cast<long long*>(smem + OFF_ROW);
    float* s_cs = reinterpret_cast<float*>(smem + OFF_CS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

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
    const int d0 = tD * Cfg::TD, h0 = tH * TH, w0 = tW * TW;

    {   // row = line * TW + m, line = ld * TH + lh
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

    // halo DMA: piece j = wave + NWAVE * i covers halo voxels 32 j .. 32 j + 31; lane -> voxel 32 j + lane / 2, 16-byte half lane & 1
    int hrel[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int j = wave + NWAVE * i;
        const int v = j * 32 + (lane >> 1);
        const int hd = v / (HH * HW), rem = v - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        const int gd = d0 + p.dshift - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
        const bool ok = (j < HALO_INSTR) && (v < HV) && gd >= 0 && gd < p.Di && gh >= 0 && gh < p.Hi && gw >= 0 && gw < p.Wi;
        hrel[i] = ok ? ((gd - dlo) * p.Hi + gh) * p.Wi + gw : -1;
    }
    const unsigned hq16 = (unsigned)((lane & 1) * 16);
    const int C1 = p.C1, C2 = p.C2, CoutPad = p.CoutPad, nchunks = p.nchunks;
    const int Q = nchunks * 27;                              // (chunk, tap) entries
    const int S = (Q + 3) >> 2;                              // steps of 4 entries (the packed image is zero-padded to 4 S)

    auto issue_halo = [&](int cc, int i) -> int {
        const int j = wave + NWAVE * i;
        if (j >= HALO_INSTR) return 0;
        const int ch0 = cc * 16;
        const bool second = ch0 >= C1;
        const unsigned cbytes = (unsigned)((second ? C2 : C1) * 2);
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((second ? ch0 - C1 : ch0) * 2);
        int hsel = hrel[0];
#pragma unroll
        for (int q = 1; q < NPIECE; ++q) {   // (the empty asm keeps hrel[] in registers: hipcc otherwise turns the select chain
            int cand = hrel[q];              //  into a dynamically indexed scratch array, and a scratch load counts in vmcnt)
            asm("" : "+v"(cand));
            hsel = (i == q) ? cand : hsel;
        }
        const unsigned voff = hsel >= 0 ? (unsigned)hsel * cbytes + hq16 : 0x80000000u;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (cc & 1) * HALO_BYTES + j * 1024));
        if (second)
            h3_dma16(rs2, dst, voff, soff);
        else
            h3_dma16(rs1, dst, voff, soff);
        return 1;
    };
    // weights of step s = 4 entries x 4 KB = 16 pieces of 1 KB: wave w copies pieces w and 8 + w
    const unsigned w_voff = (unsigned)lane * 16u;
    auto issue_weights = [&](int s) {
        const unsigned slot = lds0 + OFF_W + (s % NWS) * WSLOT_BYTES;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = wave + NWAVE * k;
            const int e = piece >> 2, quarter = piece & 3;
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(((s * 4 + e) * CoutPad) * 32 + quarter * 1024);
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(slot + piece * 1024));
            h3_dma16(rsw, dst, w_voff, soff);
        }
    };

    // fragment addressing: lane -> row r16 = lane & 15 of the 16-row operand tile, k group kg = lane >> 4:
    // kg >> 1 = which of the unit's two taps, kg & 1 = which 8 of the tap's 16 channels
    const int r16 = lane & 15, kg = lane >> 4;
    const bool tap1 = kg >= 2;
    int a_lane, b_lane;
    {
        const int line0 = wave * Cfg::LPW;
        const int vbase = ((line0 / TH) * HH + (line0 % TH)) * HW;
        a_lane = (vbase + r16) * 32 + (kg & 1) * 16;
        b_lane = OFF_W + (kg >> 1) * TAP_BYTES + r16 * 32 + (kg & 1) * 16;
    }
    // LDS byte offset of entry q's tap in its halo buffer (wave-uniform); entries past the end repeat the last one (their
    // weights are zero in the packed image; a repeated REAL tap keeps 0 x value finite wherever the real product is)
    auto tap_off = [&](int q) -> int {
        q = q < Q ? q : Q - 1;
        const int cc = q / 27, t = q - cc * 27;
        const int kd = t / 9, t2 = t - kd * 9;
        const int kh = t2 / 3, kw = t2 - kh * 3;
        return (cc & 1) * HALO_BYTES + ((kd * HH + kh) * HW + kw) * 32;
    };

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.0f;

    bf16x8 fa0[4], fa1[4], fbl[4], fbh[4];

    // A fragments i0, i0 + 1 of unit u into FA; B fragments j0 .. j0 + 3 of the unit `uu` (0 / 1) of weight slot `wslot` into FB
#define HK_LOAD_A(FA, I0, AADDR)                                                                               \
    {                                                                                                          \
        FA[I0] = *reinterpret_cast<const bf16x8*>(smem + (AADDR) + Cfg::a_imm(I0));                            \
        FA[I0 + 1] = *reinterpret_cast<const bf16x8*>(smem + (AADDR) + Cfg::a_imm(I0 + 1));                    \
    }
#define HK_LOAD_B(FB, J0, BADDR, UU)                                                                           \
    {                                                                                                          \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                       \
            FB[j_] = *reinterpret_cast<const bf16x8*>(smem + (BADDR) + (UU) * 2 * TAP_BYTES + ((J0) + j_) * 512); \
    }
#define HK_MFMA(FA, FB, J0)                                                                                    \
    {                                                                                                          \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)       \
            acc[i_][(J0) + j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[i_], FB[j_], acc[i_][(J0) + j_], 0, 0, 0); \
    }
    // 16 MFMAs with the phase's 6 ds_read_b128 in its first gaps (they feed the NEXT phase, 10 MFMAs = 160+ cycles later)
#define HK_SCHED()                                                                                             \
    {                                                                                                          \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                                     \
        _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                                                     \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
        }                                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }

    // prologue: halo of chunk 0, weights of steps 0..2
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) issue_halo(0, i);
#pragma unroll
    for (int s = 0; s < NWS - 1; ++s)
        if (s < S) issue_weights(s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const int oa = tap_off(0), ob = tap_off(1);
        const int aaddr = a_lane + (tap1 ? ob : oa);
        HK_LOAD_A(fa0, 0, aaddr);
        HK_LOAD_A(fa0, 2, aaddr);
        HK_LOAD_B(fbl, 0, b_lane, 0);
    }
    __builtin_amdgcn_sched_barrier(0);

    int n_prev = 0;                  // DMAs of this wave's most recent issue group (wave-uniform)
    int hc = 1, hp = 0, hfree = 0;   // next halo chunk to fetch, its next piece, the first step whose group may issue it
    auto issue_group = [&](int s) {
        int issued = 0;
        if (s + NWS - 1 < S && !(p.dbg & 2)) {
            issue_weights(s + NWS - 1);
            issued += 2;
        }
        if (hc < nchunks && s >= hfree && !(p.dbg & 1)) {
            issued += issue_halo(hc, hp);
            ++hp;
            if (hp == 1 && NPIECE > 1) {                 // two pieces in a chunk's first group: 5 pieces fit 4 groups
                issued += issue_halo(hc, hp);
                ++hp;
            }
            if (hp >= NPIECE) {
                hp = 0;
                hfree = (27 * hc - 1) >> 2;              // step of the unit that holds chunk hc - 1's last entry
                ++hc;
            }
        }
        n_prev = issued;
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < S; ++s) {
        const int baddr = b_lane + (s % NWS) * WSLOT_BYTES;
        const int baddr_n = b_lane + ((s + 1) % NWS) * WSLOT_BYTES;
        int aaddr1, aaddr2;
        {
            const int oa = tap_off(4 * s + 2), ob = tap_off(4 * s + 3);
            aaddr1 = a_lane + (tap1 ? ob : oa);              // unit 2 s + 1
            const int oc = tap_off(4 * s + 4), od = tap_off(4 * s + 5);
            aaddr2 = a_lane + (tap1 ? od : oc);              // unit 2 s + 2 (next step; the last step re-reads in-bounds data)
        }
        // ---- unit 2 s ----
        HK_LOAD_B(fbh, 4, baddr, 0);
        HK_LOAD_A(fa1, 0, aaddr1);
        HK_MFMA(fa0, fbl, 0);
        HK_SCHED();
        HK_LOAD_A(fa1, 2, aaddr1);
        HK_LOAD_B(fbl, 0, baddr, 1);
        HK_MFMA(fa0, fbh, 4);
        HK_SCHED();
        // ---- barrier B_s: everything step s + 1 reads has landed (issued two groups ago or earlier) and is visible; every wave
        //      has drained its reads of weight slot (s - 1) % NWS and of a halo chunk whose last entry lies in a unit <= 2 s + 1
        hk_wait_vm(n_prev);
        if (!(p.dbg & 64)) __builtin_amdgcn_s_barrier();      // (64: timing-only ablation)
        __builtin_amdgcn_sched_barrier(0);
        // ---- unit 2 s + 1 ----
        HK_LOAD_B(fbh, 4, baddr, 1);
        HK_LOAD_A(fa0, 0, aaddr2);
        HK_MFMA(fa1, fbl, 0);
        HK_SCHED();
        // issue group of step s (LDS-DMA pieces: weights of step s + 3, halo pieces of the next chunk).  SIMD partners (waves w
        // and w + 4) issue at DIFFERENT phase boundaries: a piece costs its wave ~60-100 issue cycles during which it feeds no
        // MFMAs, so waves 0-3 issue behind the first MFMA phase after the barrier -- while waves 4-7 run their second phase on
        // the matrix pipe -- and waves 4-7 behind that second phase (p.dbg & 16: all waves at the first boundary, for A/B timing)
        if (wave < 4 || (p.dbg & 16)) issue_group(s);
        HK_LOAD_A(fa0, 2, aaddr2);
        HK_LOAD_B(fbl, 0, baddr_n, 0);
        HK_MFMA(fa1, fbh, 4);
        HK_SCHED();
        if (wave >= 4 && !(p.dbg & 16)) issue_group(s);
    }
#undef HK_LOAD_A
#undef HK_LOAD_B
#undef HK_MFMA
#undef HK_SCHED
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: bias, GroupNorm column sums, the 512 x 128 bf16 tile through LDS (128 KB), 16-byte row stores -----
    // accumulator (i, j)[q]: row 16 i + 4 kg + q of the wave's 64, cout 16 j + r16
    if (p.dbg & 8) return;
    bf16_t* s_tile = reinterpret_cast<bf16_t*>(smem);  // [BM][BN] bf16
    const bool want_sums = p.colsum != nullptr;
    unsigned vbits = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) vbits |= (unsigned)(s_rowoff[wave * 64 + 16 * i + 4 * kg + q] >= 0) << (4 * i + q);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int col = j * 16 + r16;
        const int co = n0 + col;
        const float bv = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.0f;
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16_t* trow = s_tile + (wave * 64 + 16 * i + 4 * kg) * BN + col;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = acc[i][j][q] + bv;
                trow[q * BN] = f32_to_bf16(v);
                if (want_sums) {
                    const float vm = ((vbits >> (4 * i + q)) & 1u) ? v : 0.0f;
                    s1 += vm;
                    s2 += vm * vm;
                }
            }
        }
        if (want_sums) {
            s1 += __shfl_xor(s1, 16);
            s2 += __shfl_xor(s2, 16);
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (kg == 0) {
                s_cs[(wave * BN + col) * 2 + 0] = s1;
                s_cs[(wave * BN + col) * 2 + 1] = s2;
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
            if (off >= 0 && co < p.Cout && !(p.dbg & 4)) {
                const uint4 v = *reinterpret_cast<const uint4*>(s_tile + row * BN + ch * 8);
                *reinterpret_cast<uint4*>(y + off + co) = v;
            }
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- weight packing: fp32 (cout, cin, 3,3,3) -> bf16 [entry q = chunk16 * 27 + tap][cout_pad][16], zero entries up to 4 S ----
__global__ void conv3_halo_k32_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int CoutPad,
                                           int CinW, int nchunks, long long total) {
    const int Q = nchunks * 27;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 15);
        const long long row = idx >> 4;                // q * CoutPad + cout
        const int co = (int)(row % CoutPad);
        const long long q = row / CoutPad;
        float v = 0.0f;
        if (q < Q) {
            const int tap = (int)(q % 27), cc = (int)(q / 27);
            const int ci = cc * 16 + e;
            if (co < Cout && ci < CinW) v = w[((long long)co * CinW + ci) * 27 + tap];
        }
        out[idx] = f32_to_bf16(v);
    }
}

extern "C" size_t ctsi_conv3_halo_k32_weight_bytes(int cin, int cout_pad) {
    const long long q = (long long)(cin / 16) * 27;
    return (size_t)(((q + 3) / 4) * 4 * cout_pad * 32);
}

extern "C" int ctsi_conv3_halo_k32_pack(const float* w, void* packed, int cout, int cout_pad, int cin, int cin_w,
                                        void* stream) {
    CTSI_CHECK_ARG(w && packed && cin % 16 == 0 && cout_pad % 128 == 0, "ctsi_conv3_halo_k32_pack: bad arguments");
    const int nchunks = cin / 16;
    const long long total = (long long)ctsi_conv3_halo_k32_weight_bytes(cin, cout_pad) / 2;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(conv3_halo_k32_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)packed,
                       cout, cout_pad, cin_w, nchunks, total);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

extern "C" int ctsi_conv3_halo_k32_launch(const Conv3HaloParams* hp, int tile /* 0: 4x4x32, 2: 4x8x16 */, void* stream) {
    using C44 = HkCfg<4, 4>;
    using C48 = HkCfg<4, 8, 16>;
    auto k44 = conv3_halo_k32_kernel<4, 4, 32>;
    auto k48 = conv3_halo_k32_kernel<4, 8, 16>;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)k44, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k48, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const int grid = hp->mtiles * hp->ntiles_n;
    if (tile == 2)
        hipLaunchKernelGGL(k48, dim3(grid), dim3(C48::NTH), C48::LDS_BYTES, (hipStream_t)stream, *hp);
    else
        hipLaunchKernelGGL(k44, dim3(grid), dim3(C44::NTH), C44::LDS_BYTES, (hipStream_t)stream, *hp);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
