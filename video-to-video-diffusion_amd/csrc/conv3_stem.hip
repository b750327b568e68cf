// 3x3x3 stride-1 convolution of a ONE-channel volume: the VAE encoder's first layer (models/vae.py:27, 109: Conv3DBlock(1, 128)
// on the CT volume, which the engine stores with 8 channels, 7 of them layout padding).
//
// The gather kernel's small-Cin form ran it at 1.4 TB/s of output (0.39 ms at 8 x 512^2, 1.29 ms at 4 x 48 x 192^2: one 96 KB
// block per CU whose tiny K loop, LDS transposition and row stores do not overlap) -- 1.9 % of an encode for 0.2 % of its FLOPs.
// Here the 27 taps are the K dimension of ONE v_mfma_f32_16x16x32_bf16 per 16 voxels x 16 couts (27 real + 5 zero columns):
//   * a block owns a 2 x 8 x 32 voxel tile; its one-channel halo tile (4 x 10 x 34 bf16 = 2.7 KB) is gathered once into LDS;
//   * A fragment of 16 voxels: lane (r16, kg) reads its 8 taps as eight 2-byte LDS loads at per-lane tap offsets;
//   * B = the layer's weights [tap][cout], held in registers for the block's lifetime (8 fragments for 128 couts);
//   * packed column 16 j + r holds cout 8 r + j (as in conv3_halo_k32_kernel's direct epilogue): a lane owns 8 consecutive
//     couts of its 4 voxel rows and stores 16 bytes each straight from the accumulators; GroupNorm column sums in the same sweep.
// Bound: the output write (128 couts x 2 B per voxel); the input is 1/64 of it.
#include "ctsi_internal.h"

namespace stem {
constexpr int TD = 2, TH = 8, TW = 32, BM = TD * TH * TW;          // 512 voxels
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, HV = HD * HH * HW;
constexpr int NW = 4, NTH = NW * 64, NJ = 8, BN = 128;
constexpr int APW = BM / 16 / NW;                                   // A tiles per wave: 8
}  // namespace stem

__global__ void __launch_bounds__(stem::NTH)
conv3_stem_kernel(const StemParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace stem;
    __shared__ bf16_t s_x[HV + 8];                 // halo tile, channel 0 only; s_x[HV] = 0 (the 5 padding columns of K read it)
    __shared__ float s_cs[NW][BN][2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    int mt = blockIdx.x;
    const int nt = blockIdx.y, n0 = nt * BN;
    const int nb = mt / p.tps;
    int r0 = mt - nb * p.tps;
    const int tD = r0 / (p.tilesH * p.tilesW);
    r0 -= tD * p.tilesH * p.tilesW;
    const int tH = r0 / p.tilesW, tW = r0 - tH * p.tilesW;
    const int d0 = tD * TD, h0 = tH * TH, w0 = tW * TW;

    // halo tile: voxel (d0 - 1 + hd, h0 - 1 + hh, w0 - 1 + hw), zero outside the volume (the conv's padding)
    const bf16_t* xs = p.x + (long long)nb * p.D * p.H * p.W * p.cx;
    for (int v = tid; v < HV + 8; v += NTH) {
        bf16_t val = 0;
        if (v < HV) {
            const int hd = v / (HH * HW), rem = v - hd * (HH * HW);
            const int hh = rem / HW, hw = rem - hh * HW;
            const int gd = d0 - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
            if (gd >= 0 && gd < p.D && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W)
                val = xs[(((long long)gd * p.H + gh) * p.W + gw) * p.cx];
        }
        s_x[v] = val;
    }
    // the layer's weights: 8 B fragments (cout tiles) for this n-tile, in registers for the block's lifetime
    bf16x8 fb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(p.w + ((size_t)(nt * NJ + j) * 64 + lane) * 8);
    // the lane's 8 taps (K columns 8 kg .. 8 kg + 7): halo offset of tap t = (kd, kh, kw); columns >= 27 read the zero slot
    int toff[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int t = kg * 8 + e;
        const int kd = t / 9, kh = (t - kd * 9) / 3, kw = t - kd * 9 - kh * 3;
        toff[e] = t < 27 ? (kd * HH + kh) * HW + kw : -1;
    }
    float bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int co = n0 + r16 * NJ + j;
        bv[j] = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.0f;
    }
    __syncthreads();

    const bool want_sums = p.colsum != nullptr;
    const int co = n0 + r16 * NJ;
    const bool co_ok = co < p.Cout;
    float s1[NJ], s2[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) s1[j] = s2[j] = 0.0f;

#pragma unroll 2
    for (int i = 0; i < APW; ++i) {
        const int row0 = (wave * APW + i) * 16;
        // A fragment: voxel row0 + r16 (line = row / TW, w = row % TW; TW = 32: an A tile lies in one W-line)
        const int row = row0 + r16;
        const int line = row / TW, wl = row - line * TW;
        const int hv = ((line / TH) * HH + (line % TH)) * HW + wl;
        unsigned short av[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = s_x[toff[e] >= 0 ? hv + toff[e] : HV];
        bf16x8 fa;
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[e] = (short)av[e];
        f32x4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            acc[j] = f32x4{bv[j], bv[j], bv[j], bv[j]};
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[j], acc[j], 0, 0, 0);
        }
        // accumulator (j)[q]: voxel row row0 + 4 kg + q, cout n0 + 8 r16 + j
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int orow = row0 + 4 * kg + q;
            const int ol = orow / TW, ow = orow - ol * TW;
            const int d = d0 + ol / TH, h = h0 + ol % TH, w = w0 + ow;
            const bool ok = d < p.D && h < p.H && w < p.W;
            if (want_sums && ok) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    s1[j] += acc[j][q];
                    s2[j] = __builtin_fmaf(acc[j][q], acc[j][q], s2[j]);
                }
            }
            if (ok && co_ok) {
                typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
                u4_t w4;
                w4.x = pack_bf16x2_v(f32x2_t{acc[0][q], acc[1][q]});
                w4.y = pack_bf16x2_v(f32x2_t{acc[2][q], acc[3][q]});
                w4.z = pack_bf16x2_v(f32x2_t{acc[4][q], acc[5][q]});
                w4.w = pack_bf16x2_v(f32x2_t{acc[6][q], acc[7][q]});
                const long long off = ((((long long)nb * p.D + d) * p.H + h) * p.W + w) * p.cout_stride + p.c_off + co;
                *reinterpret_cast<u4_t*>(p.y + off) = w4;
            }
        }
    }
    if (want_sums) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float a1 = s1[j], a2 = s2[j];
            a1 += __shfl_xor(a1, 16);
            a2 += __shfl_xor(a2, 16);
            a1 += __shfl_xor(a1, 32);
            a2 += __shfl_xor(a2, 32);
            if (kg == 0) {
                s_cs[wave][r16 * NJ + j][0] = a1;
                s_cs[wave][r16 * NJ + j][1] = a2;
            }
        }
        __syncthreads();
        if (tid < BN) {
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int q = 0; q < NW; ++q) {       // fixed order: deterministic
                t1 += s_cs[q][tid][0];
                t2 += s_cs[q][tid][1];
            }
            const long long slab = (long long)p.mtiles * p.CoutPad;
            p.colsum[(long long)mt * p.CoutPad + n0 + tid] = t1;
            p.colsum[slab + (long long)mt * p.CoutPad + n0 + tid] = t2;
        }
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// fp32 (cout, 1, 3, 3, 3) -> bf16 [n-tile][cout tile j][lane][8]: lane (n = lane & 15, kg = lane >> 4) holds
// B[k = tap 8 kg + e][cout = 128 n-tile + 8 n + j] (zero for taps >= 27 and couts >= cout)
__global__ void conv3_stem_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int cout, int cin_w, int total) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, j = (idx >> 9) & 7, ntile = idx >> 12;
        const int t = (lane >> 4) * 8 + e, co = ntile * 128 + (lane & 15) * 8 + j;
        out[idx] = f32_to_bf16((t < 27 && co < cout) ? w[((long long)co * cin_w) * 27 + t] : 0.0f);
    }
}

extern "C" void ctsi_conv3_stem_tile(int* td, int* th, int* tw) {
    *td = stem::TD; *th = stem::TH; *tw = stem::TW;
}
extern "C" size_t ctsi_conv3_stem_weight_bytes(int cout_pad) { return (size_t)cout_pad / 128 * 8 * 64 * 8 * 2; }

extern "C" int ctsi_conv3_stem_pack(const float* w, void* packed, int cout, int cout_pad, int cin_w, void* stream) {
    CTSI_CHECK_ARG(w && packed && cout_pad % 128 == 0 && cin_w >= 1, "ctsi_conv3_stem_pack: bad arguments");
    const int total = cout_pad / 128 * 8 * 64 * 8;
    hipLaunchKernelGGL(conv3_stem_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)packed, cout,
                       cin_w, total);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

extern "C" int ctsi_conv3_stem_launch(const StemParams* q, void* stream) {
    CTSI_CHECK_ARG(q && q->x && q->w && q->y && q->CoutPad % 128 == 0 && q->mtiles > 0, "ctsi_conv3_stem_launch: bad arguments");
    hipLaunchKernelGGL(conv3_stem_kernel, dim3(q->mtiles, q->CoutPad / 128), dim3(stem::NTH), 0, (hipStream_t)stream, *q);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
