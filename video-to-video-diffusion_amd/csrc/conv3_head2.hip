// 3x3x3 stride-1 convolution with FEW output channels, second form: the in-plane taps are the N dimension of the GEMM.
// Network heads: VAE decoder conv_out 128 -> 1 + tanh (models/vae.py:188, 202-203), U-Net conv_out 128 -> 8
// (models/unet3d.py:342-346, 412).
//
// conv3_head.hip stages the halo tile once per 32-channel chunk but reads every A fragment from LDS 27 times, each read feeding
// ONE 16-column MFMA of which 1 (or 8) columns are real: 35 % LDS bank conflicts, MFMA busy 21 %, 1.3 TB/s on a layer that
// only has to read its input once (profiles/r03_pmc_mfma_busy.json).  Here a voxel's 128 channels are read from HBM straight
// into the MFMA A layout (no LDS staging: 16 voxel rows x 64 B per wave instruction runs at the copy rate, 6.0-6.4 TB/s,
// tools/load_pattern_bench.hip) and multiplied ONCE per depth tap by B = W[kd][channel][(kh, kw, cout)] -- N = 9 cout columns,
// padded to whole 16-column tiles: 16 for cout = 1, 80 for cout = 8:
//     Q_kd[v][(kh, kw, co)] = sum_c X[v][c] W[co][c][kd][kh][kw]                for every voxel v of an input plane's halo tile
//     out[d][h][w][co]      = bias + sum_kd sum_(kh, kw) Q_kd[(d + kd - 1, h + kh - 1, w + kw - 1)][(kh, kw, co)]
// The depth sum happens in the accumulators: a block owns a (TH x TW) plane tile and MARCHES along depth; input plane ip adds
// its kd = 0 / 1 / 2 products to the accumulators of output planes ip + 1 / ip / ip - 1 (three live sets, rotated per plane),
// so every input plane is read exactly once per block; the finished set goes to LDS as Q[(tap, co)][halo voxel] (fp32) and the
// 9-term in-plane shifted sum + bias (+ tanh) is stored.  HBM reads: the input x (TH + 2)(TW + 2) / (TH TW) (in-plane halo,
// mostly L2 hits between neighbouring blocks marching in step) x (planes + 2) / planes per depth segment.
//
// Per wave and halo M-tile (16 voxels) and plane: 4 x 16-byte loads per lane, 12 NT MFMAs (v_mfma_f32_16x16x32_bf16), B
// fragments from LDS (lane-linear image: conflict-free ds_read_b128).  Loads run PF = R - 1 tiles ahead of the MFMAs in a
// register ring (plain loads: the compiler counts vmcnt).
#include "conv3_halo_common.h"
#include <type_traits>
#include <stdlib.h>

struct Head2Params {
    const bf16_t* x;      // bf16 NDHWC, 128 channels
    const bf16_t* w;      // packed B image (ctsi_conv3_head2_pack)
    const float* bias;
    void* y;
    int n, Di, Hi, Wi;    // input dims (Di includes the depth halo slices when dshift = 1)
    int Do, Ho, Wo;
    int dshift;
    int tilesH, tilesW, segs, pps;   // plane tiles, depth segments per sample, output planes per segment
    int Cout;
    int out_mode, act, cout_stride, c_off;
    long long osn, osc, osd, osh, osw;
};

namespace hd2 {
// <TH, TW, COUT, NW waves, MPW M tiles per wave and plane, PU planes per unrolled pass, R ring slots>: the wave's tiles form one
// stream q = plane * MPW + k; tile q lives in register ring slot q % R and is loaded R - 1 tiles ahead of its MFMAs.  R divides
// MPW * PU, so with the plane loop unrolled PU times every slot index is a compile-time constant.
template <int TH_, int TW_, int COUT_, int NW_, int MPW_, int PU_, int R_>
struct Cfg {
    static constexpr int TH = TH_, TW = TW_, COUT = COUT_, NW = NW_, MPW = MPW_, PU = PU_, R = R_;
    static constexpr int HH = TH + 2, HW = TW + 2, HV = HH * HW;
    static constexpr int MT = (HV + 15) / 16;                 // 16-voxel M tiles of the halo plane tile
    static constexpr int NQ = 9 * COUT;                       // real GEMM columns: (tap, cout)
    static constexpr int NT = (NQ + 15) / 16;
    static constexpr int NTH = NW * 64;
    static constexpr int QS = ((MT * 16 - 4 + 31) / 32) * 32 + 4;   // Q row stride (floats): = 4 (mod 32) -> the 16 columns of a
                                                                    // ds_write_b128 group land 16 B apart (mod 128 B)
    static constexpr int B_BYTES = 3 * NT * 4 * 1024;         // [kd][n-tile][k-step][lane][8 bf16]
    static constexpr int OFF_Q = B_BYTES;
    static constexpr int Q_BYTES = NQ * QS * 4;
    static constexpr int NQBUF = (B_BYTES + 2 * Q_BYTES <= 150 * 1024) ? 2 : 1;
    static constexpr int LDS_BYTES = B_BYTES + NQBUF * Q_BYTES;
    static constexpr int OPT = (TH * TW * COUT + NTH - 1) / NTH;   // outputs per thread and plane
    static_assert(MT <= NW * MPW && (MPW * PU) % R == 0 && R >= 2 && QS >= MT * 16 && QS % 4 == 0 && LDS_BYTES <= 160 * 1024, "bad head2 tile");
};
}  // namespace hd2

template <class C>
__global__ void __launch_bounds__(C::NTH)
conv3_head2_kernel(const Head2Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TH = C::TH, TW = C::TW, COUT = C::COUT, NW = C::NW, MPW = C::MPW, PU = C::PU, R = C::R, HW = C::HW, HV = C::HV;
    constexpr int MT = C::MT, NQ = C::NQ, NT = C::NT, NTH = C::NTH, QS = C::QS, OPT = C::OPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_q = reinterpret_cast<float*>(smem + C::OFF_Q);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (sample, depth segment, plane tile): an XCD owns a contiguous run of plane tiles of ONE segment, which march in
    // step and share their in-plane halo rows / columns in that XCD's L2
    int bid = xcd_remap_h(blockIdx.x, gridDim.x);
    const int tW = bid % p.tilesW;
    bid /= p.tilesW;
    const int tH = bid % p.tilesH;
    bid /= p.tilesH;
    const int seg = bid % p.segs;
    const int nb = bid / p.segs;
    const int h0 = tH * TH, w0 = tW * TW;
    const int o_lo = seg * p.pps;
    const int o_hi = o_lo + p.pps < p.Do ? o_lo + p.pps : p.Do;

    // B image -> LDS (12 KB per n-tile; from L2 after the first blocks)
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.w);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < C::B_BYTES / 16; i += NTH) dst[i] = src[i];
    }

    // A operand addressing: lane -> row r = lane & 15 of its M tile, 16-byte piece kg = lane >> 4 of every 64-byte k-step
    const int r16 = lane & 15, kg = lane >> 4;
    unsigned voff[MPW];
#pragma unroll
    for (int k = 0; k < MPW; ++k) {
        const int m = wave + NW * k;
        const int idx = m * 16 + r16;
        const int hh = idx / HW, hw = idx - hh * HW;
        const int gh = h0 - 1 + hh, gw = w0 - 1 + hw;
        const bool ok = m < MT && idx < HV && gh >= 0 && gh < p.Hi && gw >= 0 && gw < p.Wi;
        voff[k] = ok ? (unsigned)(gh * p.Wi + gw) * 256u + (unsigned)kg * 16u : 0x80000000u;   // out of range: the load returns 0
    }
    const long long plane_bytes = (long long)p.Hi * p.Wi * 256;
    const char* xs = reinterpret_cast<const char*>(p.x) + (long long)nb * p.Di * plane_bytes;

    bf16x8 fa[R][4];
    auto issue = [&](int ip, int k, bf16x8* f) {     // the 4 k-steps of tile k of input plane ip
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xs + (long long)ip * plane_bytes), 0,
                                                                            (int)plane_bytes, 0x00020000);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff[k], s * 64, 0);
            f[s] = *reinterpret_cast<const bf16x8*>(&v);
        }
    };

    f32x4 acc[MPW][3][NT];      // [tile][slot: 0 = output plane ahead, 1 = this plane, 2 = the plane that completes now][n-tile]
#pragma unroll
    for (int k = 0; k < MPW; ++k)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[k][t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // input planes of this segment: ip = od + dshift - 1 + kd; the valid ones (inside the tensor) form one contiguous range
    const int ip_lo = o_lo + p.dshift - 1, ip_hi = o_hi + p.dshift;      // inclusive
    const int va = ip_lo < 0 ? 0 : ip_lo, vb = ip_hi > p.Di - 1 ? p.Di - 1 : ip_hi;
    const int tiles_mine = (MT - wave + NW - 1) / NW;                     // my tiles: k < tiles_mine (wave-uniform)

    // prologue: the first R - 1 tiles of the stream, which starts at plane va (its pass parity is 0)
#pragma unroll
    for (int q = 0; q < R - 1; ++q) {
        const int po = q / MPW, kt = q % MPW;
        if (kt < tiles_mine && va + po <= vb) issue(va + po, kt, fa[q % R]);
    }
    __syncthreads();            // B image visible

    const char* bimg = smem + lane * 16;
    int qbuf = 0;
    static_assert(NTH % COUT == 0, "a thread's cout must not depend on the output it handles");
    const int co = tid % COUT;                                            // outputs o = tid + i NTH = (pixel, cout), cout fastest
    const float bv = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.0f;

    // loads + MFMAs of valid input plane ip, the PAR-th plane of an unrolled pass (tile k of it: stream index PAR * MPW + k)
    auto plane_mfma = [&](int ip, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        bool use_kd[3];      // which depth taps of this plane feed output planes of this segment (wave-uniform)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const int od = ip - p.dshift + 1 - kd;
            use_kd[kd] = od >= o_lo && od < o_hi;
        }
#pragma unroll
        for (int k = 0; k < MPW; ++k) {
            {   // prefetch the tile R - 1 positions ahead in the stream: plane ip + po, tile kt
                constexpr int dummy = 0;
                (void)dummy;
                const int kn = k + R - 1;
                const int po = kn / MPW, kt = kn % MPW;
                if (kt < tiles_mine && ip + po <= vb) issue(ip + po, kt, fa[(PAR * MPW + kn) % R]);
            }
            if (k < tiles_mine) {
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    if (!use_kd[kd]) continue;
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const bf16x8 fb = *reinterpret_cast<const bf16x8*>(bimg + ((kd * NT + j) * 4 + s) * 1024);
                            acc[k][kd][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[(PAR * MPW + k) % R][s], fb, acc[k][kd][j], 0, 0, 0);
                        }
                }
            }
        }
    };
    // after input plane ip: output plane od = ip - dshift - 1 is complete (planes beyond the tensor are zero padding): its
    // accumulator set goes to LDS as Q[(tap, cout)][halo voxel], the 9-term in-plane shifted sum is stored; then the sets rotate
    auto plane_emit = [&](int ip) {
        const int od = ip - p.dshift - 1;
        if (od >= o_lo && od < o_hi) {
            float* q = s_q + qbuf * (C::Q_BYTES / 4);
#pragma unroll
            for (int k = 0; k < MPW; ++k) {
                if (k >= tiles_mine) continue;
                const int m = wave + NW * k;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int nn = j * 16 + r16;
                    if (nn < NQ) *reinterpret_cast<f32x4*>(q + nn * QS + m * 16 + kg * 4) = acc[k][2][j];
                }
            }
            __syncthreads();
            const long long ybase = p.out_mode == 1 ? (long long)nb * p.osn + (long long)od * p.osd
                                                    : ((long long)nb * p.Do + od) * p.Ho * (long long)p.Wo * p.cout_stride + p.c_off;
#pragma unroll
            for (int i = 0; i < OPT; ++i) {
                const int pix = (tid + i * NTH) / COUT;
                const int hl = pix / TW, wl = pix - hl * TW;
                const int h = h0 + hl, w = w0 + wl;
                if (pix < TH * TW && h < p.Ho && w < p.Wo && co < p.Cout) {
                    float v = bv;
                    const float* qq = q + co * QS + hl * HW + wl;
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) v += qq[((kh * 3 + kw) * COUT) * QS + kh * HW + kw];
                    if (p.act == 1) v = tanhf(v);
                    if (p.out_mode == 1)
                        reinterpret_cast<float*>(p.y)[ybase + (long long)h * p.osh + (long long)w * p.osw + (long long)co * p.osc] = v;
                    else
                        reinterpret_cast<bf16_t*>(p.y)[ybase + ((long long)h * p.Wo + w) * p.cout_stride + co] = f32_to_bf16(v);
                }
            }
            if (C::NQBUF == 2)
                qbuf ^= 1;          // the next plane's Q goes to the other buffer: its barrier orders the reuse two planes later
            else
                __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < MPW; ++k)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[k][2][j] = acc[k][1][j];
                acc[k][1][j] = acc[k][0][j];
                acc[k][0][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    };
    // planes before the first valid one only exist when ip_lo = -1 (a volume start: nothing accumulated yet, nothing to emit)
    int ip = va;
    for (; ip + PU - 1 <= vb; ip += PU) {
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            if (u == 0) plane_mfma(ip, std::integral_constant<int, 0>{});
            if (u == 1) plane_mfma(ip + 1, std::integral_constant<int, 1 % PU>{});
            if (u == 2) plane_mfma(ip + 2, std::integral_constant<int, 2 % PU>{});
            if (u == 3) plane_mfma(ip + 3, std::integral_constant<int, 3 % PU>{});
            plane_emit(ip + u);
        }
    }
    {   // the remaining valid planes of the last, partial pass
        static_assert(PU <= 4, "plane unroll");
        int u = 0;
        if (ip <= vb) { plane_mfma(ip, std::integral_constant<int, 0>{}); plane_emit(ip); ++ip; ++u; }
        if (PU > 2 && ip <= vb) { plane_mfma(ip, std::integral_constant<int, 1 % PU>{}); plane_emit(ip); ++ip; ++u; }
        if (PU > 3 && ip <= vb) { plane_mfma(ip, std::integral_constant<int, 2 % PU>{}); plane_emit(ip); ++ip; ++u; }
    }
    for (; ip <= ip_hi; ++ip) plane_emit(ip);     // planes beyond the tensor's end: zero padding, emit only
#endif  // __HIP_DEVICE_COMPILE__
}

// fp32 (cout, 128, 3, 3, 3) -> bf16 [kd][n-tile j][k-step s][lane][8]: lane (n = lane & 15, kg = lane >> 4) holds
// B[k = channel 32 s + 8 kg + e][column 16 j + n], column = (kh * 3 + kw) * cout + co (zero beyond 9 cout)
__global__ void conv3_head2_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int cout, int cin_w, int nt) {
    const int total = 3 * nt * 4 * 512;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, s = (idx >> 9) & 3;
        const int j = (idx >> 11) % nt, kd = (idx >> 11) / nt;
        const int col = j * 16 + (lane & 15), ci = s * 32 + (lane >> 4) * 8 + e;
        float v = 0.0f;
        if (col < 9 * cout && ci < cin_w) {
            const int tap = col / cout, co = col - tap * cout;
            v = w[((long long)co * cin_w + ci) * 27 + kd * 9 + tap];
        }
        out[idx] = f32_to_bf16(v);
    }
}

extern "C" int ctsi_conv3_head2_supported(int cin, int cout) { return cin == 128 && (cout == 1 || cout == 8); }
extern "C" size_t ctsi_conv3_head2_weight_bytes(int cout) { return (size_t)3 * ((9 * cout + 15) / 16) * 4 * 1024; }

extern "C" int ctsi_conv3_head2_pack(const float* w, void* packed, int cout, int cin, int cin_w, void* stream) {
    CTSI_CHECK_ARG(w && packed && ctsi_conv3_head2_supported(cin, cout) && cin_w <= cin, "ctsi_conv3_head2_pack: bad arguments");
    const int nt = (9 * cout + 15) / 16;
    hipLaunchKernelGGL(conv3_head2_pack_kernel, dim3((3 * nt * 4 * 512 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       (bf16_t*)packed, cout, cin_w, nt);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

template <class C>
static int head2_launch(Head2Params& q, int target_blocks, hipStream_t stream) {
    q.tilesH = ceil_div(q.Ho, C::TH);
    q.tilesW = ceil_div(q.Wo, C::TW);
    // depth segments: about `target_blocks` blocks, at least 8 output planes each (every segment re-reads 2 input planes)
    static const char* tb = getenv("CTSI_HEAD2_BLOCKS");      // tuning aid
    const long long target = tb ? atoi(tb) : target_blocks;
    const long long tiles = (long long)q.n * q.tilesH * q.tilesW;
    int segs = (int)((target + tiles - 1) / tiles);
    if (segs > q.Do / 8) segs = q.Do / 8;
    if (segs < 1) segs = 1;
    q.pps = ceil_div(q.Do, segs);
    q.segs = ceil_div(q.Do, q.pps);
    auto k = conv3_head2_kernel<C>;
    static CtsiPerDeviceOnce attr_once;
    if (attr_once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    const long long grid = tiles * q.segs;
    CTSI_CHECK_ARG(grid < (1ll << 31) && (long long)q.Hi * q.Wi * 256 < (1ll << 31), "ctsi_conv3_head2_launch: plane too large");
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(C::NTH), C::LDS_BYTES, stream, q);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

extern "C" int ctsi_conv3_head2_launch(const Conv3HaloParams* hp, int n, const void* packed, int out_mode, int act, long long sn,
                                       long long sc, long long sd, long long sh, long long sw, void* stream) {
    CTSI_CHECK_ARG(hp && packed && ctsi_conv3_head2_supported(hp->C1, hp->Cout) && hp->C2 == 0, "ctsi_conv3_head2_launch: unsupported layer");
    Head2Params q;
    q.x = hp->x1;
    q.w = (const bf16_t*)packed;
    q.bias = hp->bias;
    q.y = hp->y;
    q.n = n;
    q.Di = hp->Di; q.Hi = hp->Hi; q.Wi = hp->Wi;
    q.Do = hp->Do; q.Ho = hp->Ho; q.Wo = hp->Wo;
    q.dshift = hp->dshift;
    q.Cout = hp->Cout;
    q.out_mode = out_mode; q.act = act; q.cout_stride = hp->cout_stride; q.c_off = hp->c_off;
    q.osn = sn; q.osc = sc; q.osd = sd; q.osh = sh; q.osw = sw;
    static const char* vs = getenv("CTSI_HEAD2_VARIANT");     // tuning aid (tools/head_bench.py): 1 = the runner-up tile
    const int v = vs ? atoi(vs) : 0;
    hipStream_t st = (hipStream_t)stream;
    using namespace hd2;
    // Measured on the real shapes (profiles/r04_head_bench.log; conv3_head_kernel: 2492 / 177 us):
    //   128 -> 1 @48x512x512  <8,32 | 8 waves x 3 tiles, ring 3> 650 us = 5.0 TB/s (one depth segment: 1024 blocks); <16,32 | 8 x 6,
    //                         ring 3> 656-688; 4-wave blocks 722-739; deeper rings (5, 6 tiles ahead) no gain: 64 KB in flight per CU
    //                         suffice; more depth segments only add their two re-read planes (2048 blocks 683-709, 4096 717-772)
    //   128 -> 8 @48x128x128  <8,16 | 12 waves x 1 tile, ring 2> 94-100 us; <.. ring 3> 93-105; <8,16 | 6 waves x 2 tiles> 133-135 (half
    //                         the bytes in flight).  One 118 KB block per CU (B image 60 KB + Q 56 KB): the MFMA phase and the shifted-sum
    //                         phase of a plane do not overlap -- 2.4 TB/s, not the 5 TB/s of the one-cout head
    if (hp->Cout == 1) {
        if (v == 1) return head2_launch<Cfg<16, 32, 1, 8, 6, 1, 3>>(q, 1024, st);
        return head2_launch<Cfg<8, 32, 1, 8, 3, 1, 3>>(q, 1024, st);
    }
    if (v == 1) return head2_launch<Cfg<8, 16, 8, 12, 1, 3, 3>>(q, 256, st);
    return head2_launch<Cfg<8, 16, 8, 12, 1, 2, 2>>(q, 256, st);
}
