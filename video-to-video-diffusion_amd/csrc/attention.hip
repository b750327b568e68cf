// TemporalAttention of the reference (models/unet3d.py:136-194) as HBM-bound passes.
//
// As written, the module computes out = einsum('bhqk,bhvc->bhqc', softmax(q k^T / sqrt(hd)), v):
// k and v are independent summation indices, so out[q] = rowsum(softmax)[q] * sum_t V_t and
// rowsum(softmax) == 1.  With V = W_v gn(x) + b_v (1x1x1 conv, linear):
//     module(x) = x + W_p ( W_v * sum_d gn(x)_d + D*b_v ) + b_p          (broadcast over depth)
// Pass 1 (attn_depthsum_kernel) reads x once: GroupNorm column sums + S = sum_d x.
// Pass 2 (attn_normsum_kernel) forms sum_d gn(x)_d = gamma*rstd*(S - D*mean) + D*beta.
// The two pointwise matrices are folded into one 1x1x1 conv (conv_mfma.hip) by the host.
// Pass 3 (attn_broadcast_add_kernel) adds the (h,w,c) result back over depth.
// Exact mode evaluates rowsum(softmax(q k^T * hd^-0.5)) in fp32 (attn_softmax_rowsum_kernel) and
// scales the broadcast term with it, reproducing the einsum term by term.
#include "ctsi_internal.h"

__device__ __forceinline__ void unpack8a(const uint4 v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

#define ATTN_POS_PER_BLOCK_ROUNDS 1   // more, smaller blocks: 4 rounds left half of the CUs idle at 64x64 positions

// grid: (tiles, n); block 256.  Each thread owns one 8-channel chunk of `rounds` positions.
__global__ void __launch_bounds__(256)
attn_depthsum_kernel(const bf16_t* __restrict__ x, float* __restrict__ depthsum, float* __restrict__ colsum,
                     int n_total, int c, int d, int hw, int tps) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* s_red = reinterpret_cast<float*>(smem_raw);  // [rows_par][c][2]
    const int cpr = c >> 3;
    const int rows_par = 256 / cpr;
    const int tid = threadIdx.x;
    const int q = tid % cpr, rl = tid / cpr;
    const int tile = blockIdx.x, nb = blockIdx.y;
    const int ppb = rows_par * ATTN_POS_PER_BLOCK_ROUNDS;
    float c1[8], c2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c1[k] = c2[k] = 0.0f;
    if (rl < rows_par) {
        for (int rnd = 0; rnd < ATTN_POS_PER_BLOCK_ROUNDS; ++rnd) {
            const int pos = tile * ppb + rnd * rows_par + rl;
            if (pos >= hw) break;
            float s[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] = 0.0f;
            const bf16_t* base = x + (((long long)nb * d) * hw + pos) * c + q * 8;
            const long long dstride = (long long)hw * c;
            int dd = 0;
            for (; dd + 4 <= d; dd += 4) {     // four depth slices in flight per thread
                uint4 raw[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint4*>(base + (dd + u) * dstride);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float f[8];
                    unpack8a(raw[u], f);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        s[k] += f[k];
                        c2[k] += f[k] * f[k];
                    }
                }
            }
            for (; dd < d; ++dd) {
                float f[8];
                unpack8a(*reinterpret_cast<const uint4*>(base + dd * dstride), f);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    s[k] += f[k];
                    c2[k] += f[k] * f[k];
                }
            }
            float* o = depthsum + ((long long)nb * hw + pos) * c + q * 8;
            *reinterpret_cast<float4*>(o) = make_float4(s[0], s[1], s[2], s[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(s[4], s[5], s[6], s[7]);
#pragma unroll
            for (int k = 0; k < 8; ++k) c1[k] += s[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s_red[(rl * c + q * 8 + k) * 2 + 0] = c1[k];
            s_red[(rl * c + q * 8 + k) * 2 + 1] = c2[k];
        }
    }
    __syncthreads();
    for (int col = tid; col < c; col += 256) {
        float t1 = 0.0f, t2 = 0.0f;
        for (int r = 0; r < rows_par; ++r) {
            t1 += s_red[(r * c + col) * 2 + 0];
            t2 += s_red[(r * c + col) * 2 + 1];
        }
        const long long tg = (long long)nb * tps + tile;
        const long long slab = (long long)n_total * tps * c;
        colsum[tg * c + col] = t1;
        colsum[slab + tg * c + col] = t2;
    }
}

extern "C" int ctsi_attn_depthsum_tiles(int c, int h, int w) {
    if (c < 8 || c % 8 != 0 || c > 2048) return 0;
    const int rows_par = 256 / (c / 8);
    const int ppb = rows_par * ATTN_POS_PER_BLOCK_ROUNDS;
    return (h * w + ppb - 1) / ppb;
}

extern "C" int ctsi_attn_depthsum(const void* x, float* depthsum, float* colsum, int n, int c, int d, int h,
                                  int w, void* stream) {
    CTSI_CHECK_ARG(x && depthsum && colsum, "ctsi_attn_depthsum: null argument");
    CTSI_CHECK_ARG(c % 8 == 0 && c >= 8 && c <= 2048, "ctsi_attn_depthsum: bad c=%d", c);
    const int tps = ctsi_attn_depthsum_tiles(c, h, w);
    const int rows_par = 256 / (c / 8);
    const size_t lds = (size_t)rows_par * c * 2 * sizeof(float);
    hipLaunchKernelGGL(attn_depthsum_kernel, dim3(tps, n), dim3(256), lds, (hipStream_t)stream,
                       (const bf16_t*)x, depthsum, colsum, n, c, d, h * w, tps);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// xhat_sum[n][pos][ch] = gamma*rstd*(S - D*mean) + D*beta   (bf16)
__global__ void __launch_bounds__(256)
attn_normsum_kernel(const float* __restrict__ depthsum, const double* __restrict__ sums,
                    const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ out,
                    int c, int d, long long hw, int groups, float eps, long long total) {
    const int cpg = c / groups;
    const double cnt = (double)cpg * (double)d * (double)hw;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int ch = (int)(e % c);
        const long long nb = e / (hw * c);
        const int g = ch / cpg;
        const double m = sums[(nb * groups + g) * 2 + 0] / cnt;
        double var = sums[(nb * groups + g) * 2 + 1] / cnt - m * m;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float v = gamma[ch] * rstd * (depthsum[e] - (float)d * (float)m) + (float)d * beta[ch];
        out[e] = f32_to_bf16(v);
    }
}

extern "C" int ctsi_attn_normsum(const float* depthsum, const double* sums, const float* gamma,
                                 const float* beta, void* out, int n, int c, int d, int h, int w, int groups,
                                 float eps, void* stream) {
    CTSI_CHECK_ARG(depthsum && sums && gamma && beta && out, "ctsi_attn_normsum: null argument");
    CTSI_CHECK_ARG(groups > 0 && c % groups == 0, "ctsi_attn_normsum: bad groups");
    const long long total = (long long)n * h * w * c;
    long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(attn_normsum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, depthsum,
                       sums, gamma, beta, (bf16_t*)out, c, d, (long long)h * w, groups, eps, total);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// ---- fused "normalise the depth sum + (W_p W_v) product" -----------------------------------------------------------------
// P[r][co] = bias[co] + sum_ci W[co][ci] * xs[r][ci],   xs[r][ci] = gamma rstd (S[r][ci] - D mean) + D beta   (r = n * hw + pos)
// i.e. attn_normsum_kernel and the 1x1x1 "attn.pv" conv in ONE launch.  The product is tiny (0.1-0.5 GFLOP on <= 4096 rows)
// and ran at launch / pipeline-fill latency on the gather conv kernel (17-22 us, 11 launches per U-Net evaluation, plus 11
// normsum launches).  Here nothing is staged through LDS: the MFMA operand layouts are read straight from global memory --
// an A fragment is 8 consecutive channels of one fp32 row of S (normalised in registers), a B fragment 8 consecutive input
// channels of one row of W (bf16 [cout][cin], 0.1-0.5 MB: L2-resident) -- and ALL loads of a wave are issued before its
// first MFMA: one memory latency per launch.  A block = 16 rows x 64 couts; its 4 waves split K (c / 4 input channels
// each, KS = c / 128 steps of 32) and reduce their partial tiles through 16 KB of LDS.
template <int KS>
__global__ void __launch_bounds__(256)
attn_pv_kernel(const float* __restrict__ S, const double* __restrict__ sums, const float* __restrict__ gamma,
               const float* __restrict__ beta, const bf16_t* __restrict__ W, const float* __restrict__ bias,
               bf16_t* __restrict__ P, int c, int d, int hw, int groups, float eps, int rows) {
    __shared__ float s_part[4][16][64 + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int row0 = blockIdx.x * 16, col0 = blockIdx.y * 64;
    int row = row0 + r16;
    row = row < rows ? row : rows - 1;                      // (ragged last tile: clamped loads, masked stores)
    const int nb = row / hw;
    const int cpg = c / groups;
    const double cnt = (double)cpg * (double)d * (double)hw;
    const int kbase = wave * (c >> 2) + kg * 8;             // this lane's first channel of K-step 0
    float4 a_raw[KS][2];
    uint4 b_raw[KS][4];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int ch = kbase + ks * 32;
        const float* sp = S + (long long)row * c + ch;
        a_raw[ks][0] = *reinterpret_cast<const float4*>(sp);
        a_raw[ks][1] = *reinterpret_cast<const float4*>(sp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            b_raw[ks][j] = *reinterpret_cast<const uint4*>(W + (long long)(col0 + j * 16 + r16) * c + ch);
    }
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][q] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int ch = kbase + ks * 32;
        const int g = ch / cpg;                              // cpg % 8 == 0: the 8 channels share a group
        const double m = sums[((long long)nb * groups + g) * 2 + 0] / cnt;
        double var = sums[((long long)nb * groups + g) * 2 + 1] / cnt - m * m;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float dm = (float)d * (float)m, df = (float)d;
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + ch), g1 = *reinterpret_cast<const float4*>(gamma + ch + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + ch), b1 = *reinterpret_cast<const float4*>(beta + ch + 4);
        const float sv[8] = {a_raw[ks][0].x, a_raw[ks][0].y, a_raw[ks][0].z, a_raw[ks][0].w,
                             a_raw[ks][1].x, a_raw[ks][1].y, a_raw[ks][1].z, a_raw[ks][1].w};
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        union { bf16x8 v; unsigned u[4]; } fa;
#pragma unroll
        for (int k = 0; k < 4; ++k)   // same expression as attn_normsum_kernel, rounded to bf16 like its output
            fa.u[k] = pack_bf16x2(gv[2 * k] * rstd * (sv[2 * k] - dm) + df * bv[2 * k],
                                  gv[2 * k + 1] * rstd * (sv[2 * k + 1] - dm) + df * bv[2 * k + 1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            union { bf16x8 v; uint4 u; } fb;
            fb.u = b_raw[ks][j];
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, acc[j], 0, 0, 0);
        }
    }
    // accumulator [j][q]: row 4 kg + q, cout 16 j + r16
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) s_part[wave][4 * kg + q][16 * j + r16] = acc[j][q];
    __syncthreads();
    const int orow = tid >> 4, oc = (tid & 15) * 4;
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        o[k] = bias[col0 + oc + k] + ((s_part[0][orow][oc + k] + s_part[1][orow][oc + k]) +
                                      (s_part[2][orow][oc + k] + s_part[3][orow][oc + k]));
    if (row0 + orow < rows) {
        uint2 v;
        v.x = pack_bf16x2(o[0], o[1]);
        v.y = pack_bf16x2(o[2], o[3]);
        *reinterpret_cast<uint2*>(P + (long long)(row0 + orow) * c + col0 + oc) = v;
    }
}

extern "C" int ctsi_attn_pv_supported(int c, int groups) {
    return (c == 128 || c == 256 || c == 512 || c == 1024) && groups > 0 && c % groups == 0 && (c / groups) % 8 == 0;
}

extern "C" int ctsi_attn_pv(const float* depthsum, const double* sums, const float* gamma, const float* beta, const void* w_bf16,
                            const float* bias, void* out, int n, int c, int d, int h, int w, int groups, float eps, void* stream) {
    CTSI_CHECK_ARG(depthsum && sums && gamma && beta && w_bf16 && bias && out, "ctsi_attn_pv: null argument");
    CTSI_CHECK_ARG(ctsi_attn_pv_supported(c, groups), "ctsi_attn_pv: unsupported c=%d / groups=%d (c in {128,256,512,1024}, "
                   "(c / groups) %% 8 == 0)", c, groups);
    const int rows = n * h * w;
    const dim3 grid((rows + 15) / 16, c / 64), block(256);
    hipStream_t st = (hipStream_t)stream;
#define CTSI_PV(KS_)                                                                                                    \
    hipLaunchKernelGGL(attn_pv_kernel<KS_>, grid, block, 0, st, depthsum, sums, gamma, beta, (const bf16_t*)w_bf16, bias, \
                       (bf16_t*)out, c, d, h * w, groups, eps, rows)
    if (c == 128) CTSI_PV(1);
    else if (c == 256) CTSI_PV(2);
    else if (c == 512) CTSI_PV(4);
    else CTSI_PV(8);
#undef CTSI_PV
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// y[n][d][pos][ch] = x + p[n][pos][ch] * (rowsum ? rowsum[n][d][pos][head(ch)] : 1)
// grid (blocks, n * d): blockIdx.y = one (sample, depth) slice, whose hw * c/8 16-byte chunks line up one to one with the
// chunks of the sample's p term -- no index arithmetic per chunk in the fast mode (the first version decoded a flat 64-bit
// chunk index with five 64-bit divisions per 16 bytes).
template <bool EXACT>
__global__ void __launch_bounds__(256)
attn_broadcast_add_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ pterm,
                          const float* __restrict__ rowsum, int heads, bf16_t* __restrict__ y, int c, int d,
                          int slice_chunks /* hw * c / 8 */) {
    const int cpr = c >> 3;
    const int slice = blockIdx.y, nb = slice / d;
    const bf16_t* xs = x + (long long)slice * slice_chunks * 8;
    bf16_t* ys = y + (long long)slice * slice_chunks * 8;
    const bf16_t* ps = pterm + (long long)nb * slice_chunks * 8;
    const float* rs = EXACT ? rowsum + (long long)slice * (slice_chunks / cpr) * heads : nullptr;
    const int hd8 = EXACT ? (c / heads) >> 3 : 1;            // 8-channel chunks per head
    const int stride = 256;                                  // a block walks its own contiguous 512 chunks (see ctsi_gn_apply)
    auto one = [&](const uint4 xr, const uint4 pr, int e) {
        float scale = 1.0f;
        if (EXACT) {
            const int pos = e / cpr, q = e - pos * cpr;
            scale = rs[pos * heads + q / hd8];
        }
        const uint32_t xw[4] = {xr.x, xr.y, xr.z, xr.w}, pw[4] = {pr.x, pr.y, pr.z, pr.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x2_t xv = {__uint_as_float(xw[k] << 16), __uint_as_float(xw[k] & 0xffff0000u)};
            const f32x2_t pv = {__uint_as_float(pw[k] << 16), __uint_as_float(pw[k] & 0xffff0000u)};
            o[k] = pack_bf16x2_v(xv + pv * scale);
        }
        *reinterpret_cast<uint4*>(ys + (long long)e * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    };
    int e = blockIdx.x * 512 + threadIdx.x;
    if (e + stride < slice_chunks) {                         // two independent chunks in flight
        const uint4 x0 = *reinterpret_cast<const uint4*>(xs + (long long)e * 8);
        const uint4 p0 = *reinterpret_cast<const uint4*>(ps + (long long)e * 8);
        const uint4 x1 = *reinterpret_cast<const uint4*>(xs + (long long)(e + stride) * 8);
        const uint4 p1 = *reinterpret_cast<const uint4*>(ps + (long long)(e + stride) * 8);
        one(x0, p0, e);
        one(x1, p1, e + stride);
    } else if (e < slice_chunks)
        one(*reinterpret_cast<const uint4*>(xs + (long long)e * 8), *reinterpret_cast<const uint4*>(ps + (long long)e * 8), e);
}

extern "C" int ctsi_attn_broadcast_add(const void* x, const void* p, const float* rowsum, int heads, void* y,
                                       int n, int c, int d, int h, int w, void* stream) {
    CTSI_CHECK_ARG(x && p && y, "ctsi_attn_broadcast_add: null argument");
    CTSI_CHECK_ARG(c % 8 == 0, "ctsi_attn_broadcast_add: c=%d must be a multiple of 8", c);
    if (rowsum) CTSI_CHECK_ARG(heads > 0 && c % heads == 0 && (c / heads) % 8 == 0,
                               "ctsi_attn_broadcast_add: exact mode needs head_dim %% 8 == 0");
    const long long slice = (long long)h * w * (c / 8);
    CTSI_CHECK_ARG(slice < (1ll << 30) && (long long)n * d < 65536, "ctsi_attn_broadcast_add: slice too large");
    const long long blocks = (slice + 511) / 512;
    if (rowsum)
        hipLaunchKernelGGL(attn_broadcast_add_kernel<true>, dim3((unsigned)blocks, n * d), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, (const bf16_t*)p, rowsum, heads, (bf16_t*)y, c, d, (int)slice);
    else
        hipLaunchKernelGGL(attn_broadcast_add_kernel<false>, dim3((unsigned)blocks, n * d), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)x, (const bf16_t*)p, rowsum, heads, (bf16_t*)y, c, d, (int)slice);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// Exact-mode helper.  qk: bf16 NDHWC with 2c channels (q = first c, k = next c, heads split inside
// each, models/unet3d.py:171-177).  One wave per (sample, position, head); lane = query index
// (depth <= 64 per pass), keys looped; fp32 throughout.
__global__ void __launch_bounds__(256)
attn_softmax_rowsum_kernel(const bf16_t* __restrict__ qk, float* __restrict__ rowsum, int c, int d,
                           long long hw, int heads, long long total_items) {
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= total_items) return;
    const int head = (int)(item % heads);
    const long long np = item / heads;   // n*hw + pos
    const long long pos = np % hw, nb = np / hw;
    const int hd = c / heads;
    const float scale = rsqrtf((float)hd);
    const int c2 = 2 * c;
    for (int q0 = 0; q0 < d; q0 += 64) {
        const int qi = q0 + lane;
        const bool act = qi < d;
        const bf16_t* qp = qk + (((nb * d + (act ? qi : 0)) * hw + pos) * c2 + head * hd);
        // pass 1: max, pass 2: sum exp, pass 3 is implicit (rowsum = sum exp / sum exp evaluated
        // the way softmax does: e_k / denom summed over k)
        float mx = -3.0e38f;
        for (int kk = 0; kk < d; ++kk) {
            const bf16_t* kp = qk + (((nb * d + kk) * hw + pos) * c2 + c + head * hd);
            float dot = 0.0f;
            for (int ch = 0; ch < hd; ++ch) dot += bf16_to_f32(qp[ch]) * bf16_to_f32(kp[ch]);
            mx = fmaxf(mx, dot * scale);
        }
        float den = 0.0f;
        for (int kk = 0; kk < d; ++kk) {
            const bf16_t* kp = qk + (((nb * d + kk) * hw + pos) * c2 + c + head * hd);
            float dot = 0.0f;
            for (int ch = 0; ch < hd; ++ch) dot += bf16_to_f32(qp[ch]) * bf16_to_f32(kp[ch]);
            den += expf(dot * scale - mx);
        }
        float rs = 0.0f;
        for (int kk = 0; kk < d; ++kk) {
            const bf16_t* kp = qk + (((nb * d + kk) * hw + pos) * c2 + c + head * hd);
            float dot = 0.0f;
            for (int ch = 0; ch < hd; ++ch) dot += bf16_to_f32(qp[ch]) * bf16_to_f32(kp[ch]);
            rs += expf(dot * scale - mx) / den;
        }
        if (act) rowsum[((nb * d + qi) * hw + pos) * heads + head] = rs;
    }
}

extern "C" int ctsi_attn_softmax_rowsum(const void* qk, float* rowsum, int n, int c, int d, int h, int w,
                                        int heads, void* stream) {
    CTSI_CHECK_ARG(qk && rowsum, "ctsi_attn_softmax_rowsum: null argument");
    CTSI_CHECK_ARG(heads > 0 && c % heads == 0, "ctsi_attn_softmax_rowsum: c=%d not divisible by heads=%d", c, heads);
    const long long items = (long long)n * h * w * heads;
    const long long blocks = (items + 3) / 4;
    hipLaunchKernelGGL(attn_softmax_rowsum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)qk, rowsum, c, d, (long long)h * w, heads, items);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
