// 3x3x3 stride-1 convolution with FEW output channels (cout <= 16): the network heads -- U-Net conv_out 128 -> 8
// (models/unet3d.py:342-346), VAE decoder conv_out 128 -> 1 + tanh (models/vae.py:188, 202-203).
//
// These layers are not MFMA-bound: 2*27*Cin*Cout flop per voxel against 2*Cin bytes of input is 216 flop/B at
// cout = 8 -- the input stream and, inside the CU, the LDS reads of the A operand bound them.  The gather-GEMM kernel
// re-stages the activation slab for each of the 27 taps (5.3 GB of L2->LDS fill for the 48x128x128 U-Net head: 0.47 ms
// = 93 TFLOP/s); here the (TD+2)(TH+2)(TW+2) input halo tile is staged ONCE per 32-channel chunk and all 27 taps read
// it at shifted LDS addresses, as in conv3_halo.hip, with N = 16 (one v_mfma_f32_16x16x32_bf16 column tile, couts padded
// with zero rows).  What bounds it then: every A fragment (1 KB per wave read) feeds a single 16-cycle MFMA, so the LDS
// read rate is 27 x voxels x 2*Cin bytes (5.4 GB for the U-Net head) at 256 B/clk/CU -> ~45-70 us instead of 470.
//
// Block = 4 waves, output tile 4 x 2 x 16 voxels (8 W-lines, 2 per wave) x 16 couts; LDS = 27 KB halo + 14 / 27 KB weights
// (single-buffered: three / two blocks share a CU, one computes while the others wait for their DMA) + row table + sums.
// Output: fp32 with arbitrary strides (+ optional tanh) or bf16 NDHWC, + per-tile GroupNorm column sums on request.
#include "conv3_halo_common.h"

namespace hd3 {
// 4 x 2 x 16: H neighbours run at the same time on the same XCD and share their halo rows in L2; depth neighbours are
// hundreds of blocks apart, so the depth halo is what reaches HBM: (TD + 2) / TD = 1.5x the input with TD = 4 (a 2 x 4 x 16
// tile measured 422 MB per launch for the 201 MB input of the 48 x 128 x 128 head, profiles/r02_pmc_traffic.json).
constexpr int TD = 4, TH = 2, TW = 16;
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
constexpr int HV = HD * HH * HW;                 // 432 halo voxels
constexpr int HALO_INSTR = (HV + 15) / 16;       // 27 DMA wave-instructions of 16 voxels x 64 B
constexpr int HALO_BYTES = HALO_INSTR * 1024;
constexpr int BM = TD * TH * TW;                 // 128
constexpr int BN = 16;
constexpr int NW = 4, NTH = 256;
constexpr int HPIECE = (HALO_INSTR + NW - 1) / NW;   // 7
// weight image per chunk: 27 taps x WR cout rows x 32 ch bf16.  WR = 8 when cout <= 8 (both network heads): lanes 8-15 of the
// B operand re-read rows 0-7 (their output columns are never stored), which halves the weight slab -- 14 KB instead of 27 --
// so THREE blocks share a CU instead of two: this kernel waits on HBM latency, and resident blocks are what hides it.
template <int WR>
struct Cfg {
    static constexpr int WBYTES = (27 * WR * 64 + 1023) / 1024 * 1024;   // whole 1 KB DMA pieces: 14 KB / 27 KB
    static constexpr int WINSTR = WBYTES / 1024;
    static constexpr int WPIECE = (WINSTR + NW - 1) / NW;
    static constexpr int OFF_W = HALO_BYTES;
    static constexpr int OFF_ROW = OFF_W + WBYTES;
    static constexpr int OFF_CS = OFF_ROW + BM * 8;
    static constexpr int LDS_BYTES = OFF_CS + NW * BN * 8;   // 43.5 KB (WR 8) / 56.5 KB (WR 16)
};
}  // namespace hd3

template <int WR>
__global__ void __launch_bounds__(256)
conv3_head_kernel(const Conv3HaloParams p, const int out_mode, const int act, const long long osn, const long long osc,
                  const long long osd, const long long osh, const long long osw) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace hd3;
    constexpr int OFF_W = Cfg<WR>::OFF_W, OFF_ROW = Cfg<WR>::OFF_ROW, OFF_CS = Cfg<WR>::OFF_CS;
    constexpr int WINSTR = Cfg<WR>::WINSTR, WPIECE = Cfg<WR>::WPIECE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long* s_rowoff = reinterpret_cast<long long*>(smem + OFF_ROW);
    float* s_cs = reinterpret_cast<float*>(smem + OFF_CS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int mt = xcd_remap_h(blockIdx.x, gridDim.x);
    const int nb = mt / p.tps;
    int r0 = mt - nb * p.tps;
    const int tD = r0 / (p.tilesH * p.tilesW);
    r0 -= tD * p.tilesH * p.tilesW;
    const int tH = r0 / p.tilesW;
    const int tW = r0 - tH * p.tilesW;
    const int d0 = tD * TD, h0 = tH * TH, w0 = tW * TW;

    if (tid < BM) {   // row = line*16 + m, line = ld*TH + lh
        const int m = tid & 15, line = tid >> 4;
        const int d = d0 + line / TH, h = h0 + line % TH, w = w0 + m;
        long long off = -1;
        if (d < p.Do && h < p.Ho && w < p.Wo) {
            if (out_mode == 1)
                off = (long long)nb * osn + (long long)d * osd + (long long)h * osh + (long long)w * osw;
            else
                off = ((((long long)nb * p.Do + d) * p.Ho + h) * p.Wo + w) * p.cout_stride + p.c_off;
        }
        s_rowoff[tid] = off;
    }

    int dlo = d0 + p.dshift - 1;
    dlo = dlo < 0 ? 0 : dlo;
    const long long basevox = ((long long)(nb * p.Di + dlo) * p.Hi) * p.Wi;
    const v4i_t rs1 = h3_make_rsrc(reinterpret_cast<const char*>(p.x1) + basevox * p.C1 * 2, 0x7fffffffu);
    const v4i_t rsw = h3_make_rsrc(reinterpret_cast<const char*>(p.w), 0x7fffffffu);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr3_t)smem;

    // halo DMA: instruction j = wave + NW*i covers halo voxels 16j .. 16j+15; lane -> voxel 16j + lane/4, slot lane&3
    const unsigned hq16 = (unsigned)(((lane & 3) ^ (lane >> 4)) * 16);
    int hrel[HPIECE];
#pragma unroll
    for (int i = 0; i < HPIECE; ++i) {
        const int j = wave + NW * i;
        const int v = j * 16 + (lane >> 2);
        const int hd = v / (HH * HW), rem = v - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        const int gd = d0 + p.dshift - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
        const bool ok = (j < HALO_INSTR) && (v < HV) && gd >= 0 && gd < p.Di && gh >= 0 && gh < p.Hi && gw >= 0 &&
                        gw < p.Wi;
        hrel[i] = ok ? ((gd - dlo) * p.Hi + gh) * p.Wi + gw : -1;
    }
    const unsigned cbytes = (unsigned)(p.C1 * 2);
    const unsigned w_voff = (unsigned)lane * 16u;

    // fragment addressing (as conv3_halo_kernel): lane -> row m = lane & 15 of the 16-row operand tile, k-group kg
    const int kg = lane >> 4, m = lane & 15;
    const int half0 = (kg & 1) * 8;
    int vline[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int line = wave * 2 + i;
        vline[i] = ((line / TH) * HH + (line % TH)) * HW + m;
    }
    const int mb = m & (WR - 1);                              // weight row of this lane (rows repeat when WR = 8)
    const int b_off = mb * 64 + ((kg ^ ((mb >> 2) & 3)) << 4) + half0;

    f32x4 acc[2];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = p.nchunks;
    for (int cc = 0; cc < nchunks; ++cc) {
        if (cc > 0) __syncthreads();   // every wave has finished reading the previous chunk's buffers
        const unsigned soff_h = (unsigned)__builtin_amdgcn_readfirstlane(cc * 64);
#pragma unroll
        for (int i = 0; i < HPIECE; ++i) {
            const int j = wave + NW * i;
            if (j < HALO_INSTR) {
                const unsigned voff = hrel[i] >= 0 ? (unsigned)hrel[i] * cbytes + hq16 : 0x80000000u;
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + j * 1024));
                h3_dma16(rs1, dst, voff, soff_h);
            }
        }
#pragma unroll
        for (int i = 0; i < WPIECE; ++i) {
            const int piece = wave + NW * i;                  // 1 KB = 16 / WR taps (the image is padded to whole pieces)
            if (piece < WINSTR) {
                const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(cc * (27 * WR * 64) + piece * 1024);
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + OFF_W + piece * 1024));
                h3_dma16(rsw, dst, w_voff, soff);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* hbuf = smem;
        const char* wbuf = smem + OFF_W + b_off;
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            const int kd = g / 3, kh = g - kd * 3;
            const int vs = (kd * HH + kh) * HW;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const char* b_ = wbuf + (g * 3 + kw) * (WR * 64);
                const uint2 blo = *reinterpret_cast<const uint2*>(b_);
                const uint2 bhi = *reinterpret_cast<const uint2*>(b_ + ((half0 ^ 8) - half0));
                const uint4 bu = make_uint4(blo.x, blo.y, bhi.x, bhi.y);
                const bf16x8 fb = *reinterpret_cast<const bf16x8*>(&bu);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int v_ = vs + vline[i] + kw;
                    const char* a_ = hbuf + v_ * 64 + ((kg ^ ((v_ >> 2) & 3)) << 4);
                    const uint2 lo = *reinterpret_cast<const uint2*>(a_ + half0);
                    const uint2 hi = *reinterpret_cast<const uint2*>(a_ + (half0 ^ 8));
                    const uint4 u = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&u), fb, acc[i], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: + bias, activation, stores straight from the accumulators; optional per-tile column sums ----------
    const int co = m;
    const bool co_ok = co < p.Cout;
    const float bv = (p.bias != nullptr && co_ok) ? p.bias[co] : 0.0f;
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (wave * 2 + i) * 16 + kg * 4 + r;
            const long long off = s_rowoff[row];
            float v = acc[i][r] + bv;
            if (off >= 0 && co_ok) {
                s1 += v;
                s2 += v * v;
                if (act == 1) v = tanhf(v);
                if (out_mode == 1)
                    reinterpret_cast<float*>(p.y)[off + (long long)co * osc] = v;
                else
                    reinterpret_cast<bf16_t*>(p.y)[off + co] = f32_to_bf16(v);
            }
        }
    if (p.colsum != nullptr) {
        s1 += __shfl_xor(s1, 16);
        s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (kg == 0) {
            s_cs[(wave * BN + m) * 2 + 0] = s1;
            s_cs[(wave * BN + m) * 2 + 1] = s2;
        }
        __syncthreads();
        if (tid < BN) {
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                t1 += s_cs[(q * BN + tid) * 2 + 0];
                t2 += s_cs[(q * BN + tid) * 2 + 1];
            }
            const long long slab = (long long)p.mtiles * p.CoutPad;
            p.colsum[(long long)mt * p.CoutPad + tid] = t1;
            p.colsum[slab + (long long)mt * p.CoutPad + tid] = t2;
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

extern "C" int ctsi_conv3_head_launch(const Conv3HaloParams* hp, int rows, int out_mode, int act, long long sn, long long sc,
                                      long long sd, long long sh, long long sw, void* stream) {
    constexpr int l8 = hd3::Cfg<8>::LDS_BYTES, l16 = hd3::Cfg<16>::LDS_BYTES;
    if (rows == 8)
        hipLaunchKernelGGL(conv3_head_kernel<8>, dim3(hp->mtiles), dim3(hd3::NTH), l8, (hipStream_t)stream, *hp, out_mode,
                           act, sn, sc, sd, sh, sw);
    else
        hipLaunchKernelGGL(conv3_head_kernel<16>, dim3(hp->mtiles), dim3(hd3::NTH), l16, (hipStream_t)stream, *hp, out_mode,
                           act, sn, sc, sd, sh, sw);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
