// 1x1x1 convolution with the fused GroupNorm tail of a ResBlock, as a STREAMING pass:
//     y[v][co] = silu?( gn(h)[v][co] + sum_ci W[co][ci] [x1 | x2][v][ci] + b[co] )            (models/unet3d.py:102, 112-133)
// The six residual tails of a U-Net evaluation are HBM-bound (they read x, the skip and c2 and write the block output: 77 flop
// per byte at 384 -> 128), yet the gather kernel ran them at 2.6-3.4 TB/s: one 96 KB block per CU whose load, MFMA and store
// phases do not overlap, and a 2-stage ring that keeps 48 KB per CU in flight (profiles/r03_notes.md).  Here nothing is staged:
//   * the weight image of the block's n-tile (K x 128 couts: 32-128 KB) sits in LDS for the block's lifetime, laid out as the
//     MFMA A operand fragment by fragment (lane-linear ds_read_b128: conflict-free);
//   * every WAVE walks its own 16-voxel tiles (grid-stride over the sample): a voxel's channels go from HBM straight into the
//     B operand layout of v_mfma_f32_16x16x32_bf16 (16 voxel rows x 64 B per wave instruction: the copy rate,
//     tools/load_pattern_bench.hip), the loads of tile i + 1 are issued chunk by chunk into the registers tile i's MFMAs have
//     just consumed, h of tile i + 1 behind tile i's stores: a wave always has one whole tile (8-32 KB) in flight, 8 waves per CU;
//   * A = weights, B = voxels: a lane ends up with 4 NT CONSECUTIVE couts of ONE voxel (the cout <-> A row mapping is a
//     permutation fixed at pack time), so h is read and y written in 16-byte pieces straight from the accumulators: no LDS
//     transposition, no barrier after the prologue, waves drift apart and overlap each other's load / MFMA / store phases.
// Bound: HBM (algorithmic bytes 2 (Cin + 2 Cout) per voxel); LDS reads (1 KB per MFMA) cap the matrix pipe at ~50 %.
// Measured (tools/tail_bench.py, profiles/r04_tail_bench.log; gather kernel -> this one): 384 -> 128 @48x128^2 287 -> 186-194 us
// (5.2-5.4 TB/s), 128 -> 256 @48x64^2 83 -> 56 us, 256 -> 512 @48x32^2 54 -> 38 us.  Tried and dropped: 12 waves per block
// (55.3 vs 56.1 us on the small layers, no better on the large one), 512 blocks (-3 .. -9 %), and NON-TEMPORAL loads / stores,
// which are disastrous for this access shape (16 rows x 64 B per wave instruction: 0.9 TB/s on the cache-sized tensors,
// 3.1 instead of 5.3 TB/s on the 1 GB pass -- they seem to defeat the merging of a row's 16-byte pieces in the TCP).
#include "ctsi_internal.h"
#include <stdlib.h>
#include <type_traits>

// NCH 128-channel chunks of K, NT 16-cout tiles per n-tile (BN = 16 NT couts), NW waves
template <int NCH, int NT, int NW>
__global__ void __launch_bounds__(NW * 64)
conv1_stream_kernel(const Conv1StreamParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KS = NCH * 4, BN = NT * 16, W_BYTES = KS * NT * 1024;
    static_assert(NT % 2 == 0, "a lane stores its couts in 16-byte pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // scale / shift table: row g (the 4 NT couts of lane group g) padded by 8 floats -- with 256-cout n-tiles the four groups'
    // 16-byte reads were exactly 256 B apart, i.e. on the same banks (16-24 % LDS bank conflicts in the first PMC pass)
    constexpr int GS = 4 * NT + 8;
    float* s_sc = reinterpret_cast<float*>(smem + W_BYTES);
    float* s_sh = s_sc + 4 * GS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int nt = blockIdx.y, nb = blockIdx.z;
    const int n0 = nt * BN;

    {   // this n-tile's weight image -> LDS (from L2 after the first blocks)
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w) + (size_t)nt * W_BYTES);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = tid; i < W_BYTES / 16; i += NW * 64) dst[i] = src[i];
    }
    if (tid < BN) {
        // per-channel scale / shift of the sample's GroupNorm from the fp64 group statistics (as gn_apply_kernel computes
        // them); the conv bias rides in the shift
        const int co = n0 + tid;
        float sc = 0.0f, sh = p.bias != nullptr ? p.bias[co] : 0.0f;
        if (p.h != nullptr) {
            const int cpg = p.Cout / p.gn_groups, gi = co / cpg;
            const double* sm = p.gn_sums + ((long long)nb * p.gn_groups + gi) * 2;
            const double mean = sm[0] / p.gn_count;
            double var = sm[1] / p.gn_count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)p.gn_eps));
            sc = p.gn_gamma[co] * rstd;
            sh += p.gn_beta[co] - (float)mean * sc;
        }
        s_sc[(tid / (4 * NT)) * GS + tid % (4 * NT)] = sc;
        s_sh[(tid / (4 * NT)) * GS + tid % (4 * NT)] = sh;
    }

    const long long vb = (long long)nb * p.V;
    const int nwg = p.P * NW;
    const int co_l = n0 + g * (4 * NT);                       // the lane's first cout
    const bool has_h = p.h != nullptr;

    bf16x8 xr[NCH][4];
    uint4 hr[NT / 2];
    // chunk c of the row's channels: source 1 holds channels [0, C1), source 2 the rest (both multiples of 128)
    auto issue_x = [&](long long row, int c) {
        const bf16_t* pc = (c * 128 < p.C1) ? p.x1 + (vb + row) * p.C1 + g * 8 : p.x2 + (vb + row) * p.C2 + g * 8 - p.C1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            xr[c][s] = *reinterpret_cast<const bf16x8*>(pc + c * 128 + s * 32);
        }
    };
    auto issue_h = [&](long long row) {
        const bf16_t* ph = p.h + (vb + row) * p.cout_stride + p.c_off + co_l;
#pragma unroll
        for (int q = 0; q < NT / 2; ++q) hr[q] = *reinterpret_cast<const uint4*>(ph + q * 8);
    };
    auto clamp_row = [&](int tile) {
        long long row = (long long)tile * 16 + j;
        return row < p.V ? row : p.V - 1;                     // rows past the sample: any valid address, never stored
    };

    int tile = blockIdx.x * NW + wave;
    if (tile < p.tiles) {
        const long long row = clamp_row(tile);
#pragma unroll
        for (int c = 0; c < NCH; ++c) issue_x(row, c);
        if (has_h) issue_h(row);
    }
    __syncthreads();                                          // weight image + scale / shift visible; no barrier after this
    unsigned woff = lane * 16;

    for (; tile < p.tiles; tile += nwg) {
        // the image is the same for every tile: without an opaque base hipcc hoists all its KS x NT fragment reads out of the
        // tile loop and parks them in scratch
        asm volatile("" : "+v"(woff));
        const char* wimg = smem + woff;
        const int nxt = tile + nwg;
        const bool more = nxt < p.tiles;                      // (wave-uniform)
        const long long nrow = clamp_row(more ? nxt : tile);
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // weight fragments one k-step ahead of their MFMAs (two register sets), one ds_read_b128 per MFMA gap
        bf16x8 fa[2][NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) fa[0][t] = *reinterpret_cast<const bf16x8*>(wimg + t * 1024);
        __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);   // (the first set as its own group: the pairs below start one k-step ahead)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                constexpr int dummy = 0;
                (void)dummy;
                const int ks = c * 4 + s;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (ks + 1 < KS) fa[(ks + 1) & 1][t] = *reinterpret_cast<const bf16x8*>(wimg + ((ks + 1) * NT + t) * 1024);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks & 1][t], xr[c][s], acc[t], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (ks + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            if (more) issue_x(nrow, c);                       // the next tile's chunk into the registers just consumed
        }
        // tail: lane (g, j) holds couts co_l + 4 t + r of voxel j
        const long long row = (long long)tile * 16 + j;
        const bool ok = row < p.V;
        bf16_t* py = p.y + (vb + (ok ? row : 0)) * p.cout_stride + p.c_off + co_l;
        auto tail = [&](auto silu_tag) {
            constexpr bool SILU = decltype(silu_tag)::value;
#pragma unroll
            for (int q = 0; q < NT / 2; ++q) {
                float sc[8], sh[8];
                *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(s_sc + g * GS + q * 8);
                *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(s_sc + g * GS + q * 8 + 4);
                *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(s_sh + g * GS + q * 8);
                *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(s_sh + g * GS + q * 8 + 4);
                const uint4 hv = has_h ? hr[q] : make_uint4(0, 0, 0, 0);
                const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w};
                const float cv[8] = {acc[2 * q][0], acc[2 * q][1], acc[2 * q][2], acc[2 * q][3],
                                     acc[2 * q + 1][0], acc[2 * q + 1][1], acc[2 * q + 1][2], acc[2 * q + 1][3]};
                uint32_t ow[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    f32x2_t a = {__uint_as_float(hw[k] << 16), __uint_as_float(hw[k] & 0xffff0000u)};
                    a = a * f32x2_t{sc[2 * k], sc[2 * k + 1]} + f32x2_t{sh[2 * k], sh[2 * k + 1]} + f32x2_t{cv[2 * k], cv[2 * k + 1]};
                    if (SILU) a = silu2_f(a);
                    ow[k] = pack_bf16x2_v(a);
                }
                if (ok) *reinterpret_cast<uint4*>(py + q * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            }
        };
        if (p.gn_silu) tail(std::true_type{});
        else tail(std::false_type{});
        if (more && has_h) issue_h(nrow);
    }
#endif  // __HIP_DEVICE_COMPILE__
}

// fp32 [cout][cin_w] -> bf16 [n-tile][k-step ks][tile t][lane][8]: lane (i = lane & 15, kg = lane >> 4) holds
// W[cout = n-tile * BN + (i >> 2) * 4 NT + 4 t + (i & 3)][channel 32 ks + 8 kg + e] (zero beyond cin_w)
__global__ void conv1_stream_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int cout, int cin, int cin_w, int nt_) {
    const long long total = (long long)cout * cin;
    const int ks_n = cin / 32;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        long long r = idx >> 9;
        const int t = (int)(r % nt_);
        r /= nt_;
        const int ks = (int)(r % ks_n);
        const int ntile = (int)(r / ks_n);
        const int i = lane & 15, kg = lane >> 4;
        const int co = ntile * nt_ * 16 + (i >> 2) * 4 * nt_ + 4 * t + (i & 3);
        const int ci = ks * 32 + kg * 8 + e;
        out[idx] = f32_to_bf16(ci < cin_w ? w[(long long)co * cin_w + ci] : 0.0f);
    }
}

// n-tile width (in 16-cout tiles) for a layer, 0 = not served: K in whole 128-channel chunks from each source, the n-tile's
// weight image within 128 KB of LDS
extern "C" int ctsi_conv1_stream_nt(int c1, int c2, int cout) {
    if (c1 <= 0 || c1 % 128 != 0 || c2 % 128 != 0) return 0;
    const int nch = (c1 + c2) / 128;
    int nt;
    if (nch <= 2) nt = 16;
    else if (nch <= 4) nt = 8;
    else if (nch == 6 || nch == 8) {
        // deep K: the weight image only fits LDS in 64-cout n-tiles, every voxel row is then fetched cout / 64 times through the
        // L2 -> CU path and the pass is no faster than the gather kernel's (768 -> 256 @48x64^2: 160 vs 153 us, 1024 -> 512
        // @48x32^2: 103 vs 91 us; profiles/r04_tail_bench.log).  Parity-tested, selected only on request (CTSI_CONV1_STREAM=2)
        const char* e = getenv("CTSI_CONV1_STREAM");
        if (!(e && atoi(e) == 2)) return 0;
        nt = 4;
    } else return 0;
    const char* fnt = getenv("CTSI_CONV1_STREAM_NT");              // tuning / test aid (read per plan): force a narrower n-tile
    if (fnt && atoi(fnt) >= 4 && atoi(fnt) < nt) nt = atoi(fnt);
    while (nt > 4 && cout % (nt * 16) != 0) nt >>= 1;
    if (cout % (nt * 16) != 0) return 0;
    if (nch == 1 && nt != 16 && nt != 8) return 0;
    return nt;
}

extern "C" int ctsi_conv1_stream_pack(const float* w, void* packed, int cout, int cin, int cin_w, int nt, void* stream) {
    CTSI_CHECK_ARG(w && packed && nt > 0 && cin % 128 == 0 && cout % (nt * 16) == 0 && cin_w <= cin, "ctsi_conv1_stream_pack: bad arguments");
    const long long total = (long long)cout * cin;
    hipLaunchKernelGGL(conv1_stream_pack_kernel, dim3((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w, (bf16_t*)packed, cout, cin, cin_w, nt);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

template <int NCH, int NT, int NW>
static int c1s_launch(const Conv1StreamParams& q, int n, hipStream_t stream) {
    constexpr int LDS = NCH * 4 * NT * 1024 + 2 * 4 * (4 * NT + 8) * 4;
    static_assert(LDS <= 160 * 1024, "weight image does not fit");
    auto k = conv1_stream_kernel<NCH, NT, NW>;
    static CtsiPerDeviceOnce attr_once;
    if (attr_once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    hipLaunchKernelGGL(k, dim3(q.P, q.Cout / (NT * 16), n), dim3(NW * 64), LDS, stream, q);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

extern "C" int ctsi_conv1_stream_launch(Conv1StreamParams* q, int n, int nt, void* stream) {
    CTSI_CHECK_ARG(q && nt == ctsi_conv1_stream_nt(q->C1, q->C2, q->Cout), "ctsi_conv1_stream_launch: unsupported layer");
    CTSI_CHECK_ARG(q->V > 0 && (q->V + 15) / 16 < (1ll << 31), "ctsi_conv1_stream_launch: bad voxel count");
    q->tiles = (int)((q->V + 15) / 16);
    static int cus = 0;
    if (!cus) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
        else
            cus = 256;
    }
    const int nch = (q->C1 + q->C2) / 128;
    constexpr int NW = 8;
    // one block per CU over all (sample, n-tile) pairs; every wave gets at least one tile
    const int ntn = q->Cout / (nt * 16);
    const char* fb = getenv("CTSI_CONV1_STREAM_BLOCKS");            // tuning aid (read per launch): blocks per launch
    int P = (fb ? atoi(fb) : cus) / (ntn * n);
    if (P > (q->tiles + NW - 1) / NW) P = (q->tiles + NW - 1) / NW;
    if (P < 1) P = 1;
    q->P = P;
    hipStream_t st = (hipStream_t)stream;
#define C1S_CASE(NCH_, NT_) if (nch == NCH_ && nt == NT_) return c1s_launch<NCH_, NT_, NW>(*q, n, st)
    C1S_CASE(1, 16); C1S_CASE(1, 8);
    C1S_CASE(2, 16); C1S_CASE(2, 8); C1S_CASE(2, 4);
    C1S_CASE(3, 8); C1S_CASE(3, 4);
    C1S_CASE(4, 8); C1S_CASE(4, 4);
    C1S_CASE(6, 4); C1S_CASE(8, 4);
#undef C1S_CASE
    ctsi_set_error("ctsi_conv1_stream_launch: no instantiation for %d chunks x %d tiles", nch, nt);
    return CTSI_ERR_UNSUPPORTED;
}
