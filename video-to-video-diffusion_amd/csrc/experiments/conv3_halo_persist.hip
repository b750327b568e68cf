// Persistent version of the 32-wide 3x3x3 halo-tile convolution (see conv3_halo.hip for the tile, the LDS
// layout and the bank-conflict analysis).
//
// Why: with one 8-wave block per CU (the halo + weight buffers fill the 160 KiB LDS) nothing overlaps a
// block's prologue (cold halo / weight DMA) and epilogue (LDS transpose + 64 KB store burst at HBM rate,
// issued by every CU at the same moment); together they cost ~9 us per tile = 14 % of a Cin = 128 layer.
// Here each block walks a strided list of output tiles and the whole kernel is ONE continuous
// (tile, chunk, kd-kh step) pipeline:
//   * during the last chunk of tile t the halo pieces of chunk 0 of tile t+1 stream into the free halo
//     buffer, and the weight DMAs simply wrap around (step S, S+1 of tile t == step 0, 1 of tile t+1);
//   * the look-ahead fragment loads of the last step read tile t+1's operands, so the MFMA stream never
//     drains at a tile boundary;
//   * the epilogue writes the bf16 outputs straight from the accumulators (32 consecutive couts of a voxel
//     per half-wave store), no LDS, no barrier; the stores drain underneath the next tile's MFMAs (the first
//     wait after them is a counted vmcnt(63), every older DMA is complete by then);
//   * per-wave partial column sums go to 4 slab rows per tile (the GroupNorm finalize kernel just sees 4x
//     more tiles), so the epilogue needs no cross-wave reduction either.
#include "../conv3_halo_common.h"
#include <string.h>

namespace h32p {
namespace h32 {   // tile constants of the 4 x 2 x 32 kernel this experiment was derived from
constexpr int TD = 4, TH = 2, TW = 32;
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int HV = HD * HH * HW;
constexpr int HALO_INSTR = (HV + 15) / 16;
constexpr int HALO_BYTES = HALO_INSTR * 1024;
constexpr int BM = TD * TH * TW;
constexpr int BN = 128;
constexpr int WSLOT_BYTES = 3 * BN * 64;
constexpr int NTH = 512;
constexpr int OFF_W = 2 * HALO_BYTES;
constexpr int OFF_ROW = OFF_W + 2 * WSLOT_BYTES;
constexpr int OFF_CS = OFF_ROW + BM * 8;
constexpr int LDS_BYTES = OFF_CS + 4 * BN * 8;
constexpr int NPIECE = (HALO_INSTR + 7) / 8;
}  // namespace h32
using namespace h32;
// three rotating s_rowoff buffers (tile k uses buffer k % 3): the next tile's offsets are written while slow
// waves may still read the previous tile's buffer in their epilogue; the column-sum scratch is not needed
constexpr int ROW_BYTES = BM * 8;
constexpr int LDS_BYTES_P = OFF_ROW + 3 * ROW_BYTES;   // 159744
}  // namespace h32p

__device__ __forceinline__ v4i_t h32p_rsrc(int lo, int hi) {
    v4i_t r;
    r.x = __builtin_amdgcn_readfirstlane(lo);
    r.y = __builtin_amdgcn_readfirstlane(hi);
    r.z = 0x7fffffff;
    r.w = 0x00020000;
    return r;
}

__global__ void __launch_bounds__(512)
conv3_halo32p_kernel(const Conv3HaloParams p, const int total_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace h32p;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int G = gridDim.x;
    const int lb = xcd_remap_h(blockIdx.x, G);   // blocks of one XCD own neighbouring tiles (shared halos in its L2)
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr3_t)smem;
    const int C1 = p.C1, C2 = p.C2, CoutPad = p.CoutPad, nchunks = p.nchunks;
    const int hq = (lane & 3) ^ (lane >> 4);
    const unsigned w_voff = (unsigned)wave * 1024u + (unsigned)lane * 16u;
    const int hk = lane >> 5, r = lane & 31;
    const int lhi = hk, lcol = r;

    // ---- per-tile state lives in plain scalars (two sets: c_ = current tile, n_ = next tile); macros instead of
    //      structs/lambdas so that nothing is ever indexed dynamically or spilled to scratch memory ------------------
#define HP_DECL(P)                                                                                             \
    int P##mt = 0, P##n0 = 0, P##u1lo = 0, P##u1hi = 0, P##u2lo = 0, P##u2hi = 0, P##uwlo = 0, P##uwhi = 0;    \
    int P##h0 = -1, P##h1 = -1, P##h2 = -1, P##h3 = -1, P##h4 = -1, P##h5 = -1, P##h6 = -1;                    \
    float P##b0 = 0.0f, P##b1 = 0.0f;
#define HP_COPY(D, S_)                                                                                         \
    D##mt = S_##mt; D##n0 = S_##n0; D##u1lo = S_##u1lo; D##u1hi = S_##u1hi; D##u2lo = S_##u2lo;                \
    D##u2hi = S_##u2hi; D##uwlo = S_##uwlo; D##uwhi = S_##uwhi; D##h0 = S_##h0; D##h1 = S_##h1;                \
    D##h2 = S_##h2; D##h3 = S_##h3; D##h4 = S_##h4; D##h5 = S_##h5; D##h6 = S_##h6; D##b0 = S_##b0;            \
    D##b1 = S_##b1;
#define HP_HREL(I_, DST)                                                                                       \
    {                                                                                                          \
        const int j_ = wave + 8 * (I_);                                                                        \
        const int v_ = j_ * 16 + (lane >> 2);                                                                  \
        const int hd_ = v_ / (HH * HW), rem_ = v_ - hd_ * (HH * HW);                                           \
        const int hh_ = rem_ / HW, hw_ = rem_ - hh_ * HW;                                                      \
        const int gd_ = d0_ + p.dshift - 1 + hd_, gh_ = h0_ - 1 + hh_, gw_ = w0_ - 1 + hw_;                    \
        const bool ok_ = (j_ < HALO_INSTR) && (v_ < HV) && gd_ >= 0 && gd_ < p.Di && gh_ >= 0 && gh_ < p.Hi && \
                         gw_ >= 0 && gw_ < p.Wi;                                                               \
        DST = ok_ ? ((gd_ - dlo_) * p.Hi + gh_) * p.Wi + gw_ : -1;                                             \
    }
#define HP_SETUP(P, ID, RBUF)                                                                                  \
    {                                                                                                          \
        const int mt_ = (ID) / p.ntiles_n;                                                                     \
        const int nt_ = (ID) - mt_ * p.ntiles_n;                                                               \
        P##mt = mt_;                                                                                           \
        P##n0 = nt_ * BN;                                                                                      \
        const int nb_ = mt_ / p.tps;                                                                           \
        int r0_ = mt_ - nb_ * p.tps;                                                                           \
        const int tD_ = r0_ / (p.tilesH * p.tilesW);                                                           \
        r0_ -= tD_ * p.tilesH * p.tilesW;                                                                      \
        const int tH_ = r0_ / p.tilesW;                                                                        \
        const int tW_ = r0_ - tH_ * p.tilesW;                                                                  \
        const int d0_ = tD_ * TD, h0_ = tH_ * TH, w0_ = tW_ * TW;                                              \
        long long* s_ro_ = reinterpret_cast<long long*>(smem + OFF_ROW + (RBUF) * ROW_BYTES);                  \
        if (tid < BM) {                                                                                        \
            const int mm_ = tid & 31, line_ = tid >> 5;                                                        \
            const int d_ = d0_ + line_ / TH, h_ = h0_ + line_ % TH, w_ = w0_ + mm_;                            \
            long long off_ = -1;                                                                               \
            if (d_ < p.Do && h_ < p.Ho && w_ < p.Wo)                                                           \
                off_ = ((((long long)nb_ * p.Do + d_) * p.Ho + h_) * p.Wo + w_) * p.cout_stride + p.c_off;     \
            s_ro_[tid] = off_;                                                                                 \
        }                                                                                                      \
        int dlo_ = d0_ + p.dshift - 1;                                                                         \
        dlo_ = dlo_ < 0 ? 0 : dlo_;                                                                            \
        const long long bv_ = ((long long)(nb_ * p.Di + dlo_) * p.Hi) * p.Wi;                                  \
        const unsigned long long a1_ = reinterpret_cast<unsigned long long>(p.x1) + bv_ * C1 * 2;              \
        const unsigned long long a2_ = reinterpret_cast<unsigned long long>(p.x2) + bv_ * C2 * 2;              \
        const unsigned long long aw_ = reinterpret_cast<unsigned long long>(p.w) + (long long)P##n0 * 64;      \
        P##u1lo = (int)(a1_ & 0xffffffffull); P##u1hi = (int)((a1_ >> 32) & 0xffffull);                        \
        P##u2lo = (int)(a2_ & 0xffffffffull); P##u2hi = (int)((a2_ >> 32) & 0xffffull);                        \
        P##uwlo = (int)(aw_ & 0xffffffffull); P##uwhi = (int)((aw_ >> 32) & 0xffffull);                        \
        HP_HREL(0, P##h0) HP_HREL(1, P##h1) HP_HREL(2, P##h2) HP_HREL(3, P##h3) HP_HREL(4, P##h4)              \
        HP_HREL(5, P##h5) HP_HREL(6, P##h6)                                                                    \
        const int co0_ = P##n0 + wn * 64 + lcol, co1_ = co0_ + 32;                                             \
        P##b0 = (p.bias != nullptr && co0_ < p.Cout) ? p.bias[co0_] : 0.0f;                                    \
        P##b1 = (p.bias != nullptr && co1_ < p.Cout) ? p.bias[co1_] : 0.0f;                                    \
    }
#define HP_ISSUE_HALO(P, CC, I_, HOFF)  /* piece I_ (wave-uniform) of chunk CC of tile P */                     \
    {                                                                                                          \
        const int j_ = wave + 8 * (I_);                                                                        \
        if (j_ < HALO_INSTR) {                                                                                 \
            const int ch0_ = (CC) * 32;                                                                        \
            const bool sec_ = ch0_ >= C1;                                                                      \
            const unsigned cb_ = (unsigned)((sec_ ? C2 : C1) * 2);                                             \
            const unsigned so_ = (unsigned)__builtin_amdgcn_readfirstlane((sec_ ? ch0_ - C1 : ch0_) * 2);      \
            int hs_ = P##h0;                                                                                   \
            hs_ = ((I_) == 1) ? P##h1 : hs_;                                                                   \
            hs_ = ((I_) == 2) ? P##h2 : hs_;                                                                   \
            hs_ = ((I_) == 3) ? P##h3 : hs_;                                                                   \
            hs_ = ((I_) == 4) ? P##h4 : hs_;                                                                   \
            hs_ = ((I_) == 5) ? P##h5 : hs_;                                                                   \
            hs_ = ((I_) == 6) ? P##h6 : hs_;                                                                   \
            const unsigned vo_ = hs_ >= 0 ? (unsigned)hs_ * cb_ + (unsigned)hq * 16u : 0x80000000u;            \
            const unsigned ds_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (HOFF) + j_ * 1024));   \
            if (sec_)                                                                                          \
                h3_dma16(h32p_rsrc(P##u2lo, P##u2hi), ds_, vo_, so_);                                          \
            else                                                                                               \
                h3_dma16(h32p_rsrc(P##u1lo, P##u1hi), ds_, vo_, so_);                                          \
        }                                                                                                      \
    }
#define HP_ISSUE_W(P, S_, WOFF)  /* the 3 taps of step S_ of tile P */                                          \
    {                                                                                                          \
        const int cc_ = (S_) / 9, g_ = (S_) - cc_ * 9;                                                         \
        const v4i_t rw_ = h32p_rsrc(P##uwlo, P##uwhi);                                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                                     \
            const unsigned so_ =                                                                               \
                (unsigned)__builtin_amdgcn_readfirstlane(((cc_ * 27 + g_ * 3 + i_) * CoutPad) * 64);           \
            const unsigned ds_ = (unsigned)__builtin_amdgcn_readfirstlane(                                     \
                (int)(lds0 + (WOFF) + i_ * (BN * 64) + wave * 1024));                                          \
            h3_dma16(rw_, ds_, w_voff, so_);                                                                   \
        }                                                                                                      \
    }

    // ---- fragment addressing (tile independent) ---------------------------------------------------------------------
    const int va = (wm * HH) * HW + r;
    const int rowb = wn * 64 + r;
    const int b_off = rowb * 64 + ((hk ^ ((rowb >> 2) & 3)) << 4);
    int boffj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        boffj[j] = j * 2048;
        asm volatile("" : "+v"(boffj[j]));
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
    bf16x8 fa0[2][2], fb0[2][2], fa1[2][2], fb1[2][2], fa2[2][2], fb2[2][2];
    const int S = nchunks * 9;

#define HP_LOAD(FA, FB, HBUF, WBUF, VS, KW)                                                                    \
    {                                                                                                          \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                     \
            const int v_ = (VS) + i_ * HW + (KW);                                                              \
            const int o_ = v_ * 64 + ((hk ^ ((v_ >> 2) & 3)) << 4);                                            \
            FA[i_][0] = *reinterpret_cast<const bf16x8*>((HBUF) + o_);                                         \
            FA[i_][1] = *reinterpret_cast<const bf16x8*>((HBUF) + (o_ ^ 32));                                  \
        }                                                                                                      \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                                     \
            const int o_ = b_off + (KW) * (BN * 64) + boffj[j_];                                               \
            FB[j_][0] = *reinterpret_cast<const bf16x8*>((WBUF) + o_);                                         \
            FB[j_][1] = *reinterpret_cast<const bf16x8*>((WBUF) + (o_ ^ 32));                                  \
        }                                                                                                      \
    }
#define HP_PHASE(FAc, FBc, FAl, FBl, HBUF, WBUF, VS, KW)                                                       \
    {                                                                                                          \
        HP_LOAD(FAl, FBl, HBUF, WBUF, VS, KW);                                                                 \
        _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_) _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)       \
            _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) acc[i_][j_] =                                     \
                __builtin_amdgcn_mfma_f32_32x32x16_bf16(FAc[i_][k_], FBc[j_][k_], acc[i_][j_], 0, 0, 0);       \
        _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                                     \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }

    // ---- first tile: cold start ------------------------------------------------------------------------------------------
    HP_DECL(c_)
    HP_DECL(n_)
    int id = lb;
    int rbuf = 0;                       // s_rowoff buffer (0..2) of the current tile
    HP_SETUP(c_, id, 0)
    bool has_next = id + G < total_tiles;
    if (has_next) HP_SETUP(n_, id + G, 1)
    HP_ISSUE_HALO(c_, 0, 0, 0) HP_ISSUE_HALO(c_, 0, 1, 0) HP_ISSUE_HALO(c_, 0, 2, 0) HP_ISSUE_HALO(c_, 0, 3, 0)
    HP_ISSUE_HALO(c_, 0, 4, 0) HP_ISSUE_HALO(c_, 0, 5, 0) HP_ISSUE_HALO(c_, 0, 6, 0)
    HP_ISSUE_W(c_, 0, OFF_W)
    HP_ISSUE_W(c_, 1, OFF_W + WSLOT_BYTES)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    HP_LOAD(fa0, fb0, smem, smem + OFF_W, va, 0);
    HP_LOAD(fa1, fb1, smem, smem + OFF_W, va, 1);
    __builtin_amdgcn_sched_barrier(0);

    int hb = 0;        // halo buffer of the current chunk
    int wpar = 0;      // weight slot of the current step
    bool stores_in_flight = false;
    for (;;) {
        int cc = 0, g = 0;
        for (int s = 0; s < S; ++s) {
            const char* hbuf = smem + hb * HALO_BYTES;
            const char* wbuf = smem + OFF_W + wpar * WSLOT_BYTES;
            const int kd = g / 3, kh = g - kd * 3;
            const int vs = va + (kd * HH + kh) * HW;
            HP_PHASE(fa0, fb0, fa2, fb2, hbuf, wbuf, vs, 2);
            // all waves have drained this step's weight slot; every DMA issued before the previous barrier has
            // landed.  Right after an epilogue up to 64 stores are younger than those DMAs: leave them in flight.
            if (stores_in_flight)
                asm volatile("s_waitcnt vmcnt(63) lgkmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            stores_in_flight = false;
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // weights of step s+2 (wrapping into the next tile) go to the slot just drained
            if (s + 2 < S) {
                HP_ISSUE_W(c_, s + 2, OFF_W + wpar * WSLOT_BYTES)
            } else if (has_next) {
                HP_ISSUE_W(n_, s + 2 - S, OFF_W + wpar * WSLOT_BYTES)
            }
            // one halo piece of the next chunk (of this tile, or chunk 0 of the next tile) per step g < NPIECE
            if (g < NPIECE) {
                if (cc + 1 < nchunks) {
                    HP_ISSUE_HALO(c_, cc + 1, g, (hb ^ 1) * HALO_BYTES)
                } else if (has_next) {
                    HP_ISSUE_HALO(n_, 0, g, (hb ^ 1) * HALO_BYTES)
                }
            }
            int g2 = g + 1, cc2 = cc, hb2 = hb;
            if (g2 == 9) {
                g2 = 0;
                ++cc2;
                hb2 = hb ^ 1;
            }
            if (cc2 == nchunks) cc2 = 0;   // look-ahead crosses into the next tile (garbage if there is none)
            const char* hbuf2 = smem + hb2 * HALO_BYTES;
            const char* wbuf2 = smem + OFF_W + (wpar ^ 1) * WSLOT_BYTES;
            const int kd2 = g2 / 3, kh2 = g2 - kd2 * 3;
            const int vs2 = va + (kd2 * HH + kh2) * HW;
            __builtin_amdgcn_sched_barrier(0);
            HP_PHASE(fa1, fb1, fa0, fb0, hbuf2, wbuf2, vs2, 0);
            HP_PHASE(fa2, fb2, fa1, fb1, hbuf2, wbuf2, vs2, 1);
            g = g2;
            cc = cc2;
            hb = hb2;
            wpar ^= 1;
        }

        // ---- epilogue of the finished tile, straight from the accumulators ------------------------------------------
        if (!(CTSI_DBG(p.dbg, 8))) {
            const long long* s_rowoff = reinterpret_cast<const long long*>(smem + OFF_ROW + rbuf * ROW_BYTES);
            bf16_t* y = reinterpret_cast<bf16_t*>(p.y);
            const bool want_sums = p.colsum != nullptr;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = c_n0 + wn * 64 + j * 32 + lcol;
                const bool cok = co < p.Cout;
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = (wm * 2 + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * lhi;
                        const long long off = s_rowoff[row];
                        const float v = acc[i][j][q] + (j == 0 ? c_b0 : c_b1);
                        acc[i][j][q] = 0.0f;
                        if (off >= 0) {
                            s1 += v;
                            s2 += v * v;
                            if (cok && !(CTSI_DBG(p.dbg, 4))) y[off + co] = f32_to_bf16(v);
                        }
                    }
                }
                if (want_sums) {
                    s1 += __shfl_xor(s1, 32);
                    s2 += __shfl_xor(s2, 32);
                    if (lhi == 0) {
                        const long long row = (long long)c_mt * 4 + wm;
                        const long long slab = (long long)p.mtiles * 4 * CoutPad;
                        p.colsum[row * CoutPad + co] = s1;
                        p.colsum[slab + row * CoutPad + co] = s2;
                    }
                }
            }
            stores_in_flight = true;
        }
        if (!has_next) break;
        id += G;
        HP_COPY(c_, n_)
        rbuf = rbuf == 2 ? 0 : rbuf + 1;
        has_next = id + G < total_tiles;
        if (has_next) HP_SETUP(n_, id + G, (rbuf == 2 ? 0 : rbuf + 1))
    }
#undef HP_LOAD
#undef HP_PHASE
#endif  // __HIP_DEVICE_COMPILE__
}

extern "C" int ctsi_conv3_halo_persist_launch(const Conv3HaloParams* hp, int num_blocks, void* stream) {
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)conv3_halo32p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)h32p::LDS_BYTES_P);
        attr_done = true;
    }
    const int total = hp->mtiles * hp->ntiles_n;
    int grid = num_blocks < total ? num_blocks : total;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(conv3_halo32p_kernel, dim3(grid), dim3(512), h32p::LDS_BYTES_P, (hipStream_t)stream, *hp, total);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
