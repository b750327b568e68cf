// Training-path kernels other than the convolutions (all HBM- or latency-bound):
//   q_sample + Min-SNR-5 weighted MSE loss        models/diffusion.py:81-106, 108-203
//   GroupNorm (+SiLU, +time bias, +residual) backward   (what autograd derives for models/unet3d.py:51-133)
//   data-gradient weight re-layout, channel sums (bias gradients), small fp32 Linear backward (time embedding)
// Activations and their gradients are bf16 NDHWC with 16-byte (8-channel) accesses; statistics and parameter
// gradients are fp32 / fp64.  Reductions are two-stage with a fixed order (deterministic, no float atomics).
#include "ctsi_internal.h"
#include <math.h>

#define GNB_TILE_ROWS 512

__device__ __forceinline__ void t_unpack8(const uint4 v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 t_pack8(const float* f) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
    return v;
}
// d/dz [z * sigmoid(z)]
__device__ __forceinline__ float silu_grad_f(float z) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
    return s * (1.0f + z * (1.0f - s));
}

// ==== GroupNorm backward ============================================================================================
// forward (ctsi_gn_apply):  h = gn(x);  a = silu_pre ? silu(h) : h;  b = a + tbias;  c = b + residual;
//                           y = silu_post ? silu(c) : c
// backward: gc = dy * (silu_post ? silu'(c) : 1)   (= grad of residual; its per-sample channel sum = grad of tbias)
//           g  = gc * (silu_pre ? silu'(h) : 1)    (= grad of h)
//           dx = rstd * (gamma*g - mean_grp(gamma*g) - xhat * mean_grp(gamma*g*xhat))
//           dgamma = sum g*xhat, dbeta = sum g.
// Pass 1 (this kernel): writes g (bf16) and per-tile column sums [n][tiles][4][c] = (sum g, sum g*xhat, sum gc, sum xhat).
// grid (tiles, n), block 256.  dy may be a depth-broadcast tensor (n, 1, h, w, c): dy_mod = h*w, else 0.
// SILU_PRE / RES / SILU_POST are template parameters: as run-time flags the compiler evaluated both SiLU sites for every
// element and selected afterwards (the kernel was VALU-bound, like gn_apply before its options became template parameters).
template <bool SILU_PRE, bool RES, bool SILU_POST>
__global__ void __launch_bounds__(256)
gn_bwd_reduce_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, long long dy_mod,
                     const double* __restrict__ sums, const float* __restrict__ gamma,
                     const float* __restrict__ beta, int c, long long vox, int groups, float eps,
                     const bf16_t* __restrict__ residual, bf16_t* __restrict__ g_out,
                     float* __restrict__ colsum3, int tiles, int tile_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* s_rs = reinterpret_cast<float*>(smem_raw);   // rstd
    float* s_mr = s_rs + c;                              // -mean*rstd
    float* s_ga = s_mr + c;
    float* s_be = s_ga + c;
    float* s_red = s_be + c;                             // [rows_par][4][c]
    const int nb = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
    const int cpg = c / groups;
    const double cnt = (double)cpg * (double)vox;
    for (int ch = tid; ch < c; ch += 256) {
        const int g = ch / cpg;
        const double m = sums[((long long)nb * groups + g) * 2 + 0] / cnt;
        double var = sums[((long long)nb * groups + g) * 2 + 1] / cnt - m * m;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        s_rs[ch] = rstd;
        s_mr[ch] = -(float)m * rstd;
        s_ga[ch] = gamma[ch];
        s_be[ch] = beta[ch];
    }
    __syncthreads();
    const int cpr = c >> 3;
    const int rows_par = 256 / cpr;
    const int q = tid % cpr, rl = tid / cpr;
    const long long v0 = (long long)tile * tile_rows;
    long long v1 = v0 + tile_rows;
    if (v1 > vox) v1 = vox;
    float a0[8], a1[8], a2[8], a3[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a0[k] = a1[k] = a2[k] = a3[k] = 0.0f;
    if (rl < rows_par) {
        const bf16_t* xb = x + (long long)nb * vox * c + q * 8;
        const bf16_t* rb = (RES && SILU_POST) ? residual + (long long)nb * vox * c + q * 8 : nullptr;
        bf16_t* gb = g_out != nullptr ? g_out + (long long)nb * vox * c + q * 8 : nullptr;   // (NULL: pass 3 re-derives g)
        const bf16_t* db = dy + (long long)nb * (dy_mod ? dy_mod : vox) * c + q * 8;
        float rs[8], mr[8], ga[8], be[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            rs[k] = s_rs[q * 8 + k]; mr[k] = s_mr[q * 8 + k]; ga[k] = s_ga[q * 8 + k]; be[k] = s_be[q * 8 + k];
        }
        // 4 voxels per iteration: all 8-12 loads are issued before the first use (memory-level parallelism)
        constexpr int U = 4;
        for (long long vb = v0 + rl; vb < v1; vb += (long long)rows_par * U) {
            uint4 xr[U], dr[U], rr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long v = vb + (long long)u * rows_par;
                if (v < v1) {
                    xr[u] = *reinterpret_cast<const uint4*>(xb + v * c);
                    const long long dv = dy_mod ? (v % dy_mod) : v;
                    dr[u] = *reinterpret_cast<const uint4*>(db + dv * c);
                    if (RES && SILU_POST) rr[u] = *reinterpret_cast<const uint4*>(rb + v * c);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long v = vb + (long long)u * rows_par;
                if (v >= v1) break;
                float xf[8], df[8], rf[8], gf[8];
                t_unpack8(xr[u], xf);
                t_unpack8(dr[u], df);
                if (RES && SILU_POST) t_unpack8(rr[u], rf);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float xh = xf[k] * rs[k] + mr[k];
                    const float h = xh * ga[k] + be[k];
                    float gc = df[k];
                    if (SILU_POST) {
                        float cc = SILU_PRE ? silu_f(h) : h;
                        if (RES) cc += rf[k];
                        gc *= silu_grad_f(cc);
                    }
                    float g = gc;
                    if (SILU_PRE) g *= silu_grad_f(h);
                    gf[k] = g;
                    a0[k] += g;
                    a1[k] += g * xh;
                    a2[k] += gc;
                    a3[k] += xh;
                }
                if (gb != nullptr) *reinterpret_cast<uint4*>(gb + v * c) = t_pack8(gf);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s_red[(rl * 4 + 0) * c + q * 8 + k] = a0[k];
            s_red[(rl * 4 + 1) * c + q * 8 + k] = a1[k];
            s_red[(rl * 4 + 2) * c + q * 8 + k] = a2[k];
            s_red[(rl * 4 + 3) * c + q * 8 + k] = a3[k];
        }
    }
    __syncthreads();
    float* out = colsum3 + ((long long)nb * tiles + tile) * 4 * c;
    for (int e = tid; e < 4 * c; e += 256) {
        float t = 0.0f;
        for (int r = 0; r < rows_par; ++r) t += s_red[r * 4 * c + e];
        out[e] = t;
    }
}

// Pass 2a: grid (group chunks, n), block 256.  A block owns `gpb` whole groups (CB = gpb*cpg <= max(64, cpg)
// channels) and 256/CB tile lanes; it sums the tiles, forms the group terms and writes per-sample partial grads:
//   s12[n][groups][2] = (sum_grp gamma*g, sum_grp gamma*g*xhat) / m
//   pgrad[n][4][c] = (sum g*xhat, sum g, sum gc, sum_v dx)   with  sum_v dx = rstd*(gamma*sum g - V*S1/m - S2/m*sum xhat)
//   (the last one is the bias gradient of the convolution that produced x)
__global__ void __launch_bounds__(256)
gn_bwd_finalize_kernel(const float* __restrict__ colsum3, const float* __restrict__ gamma,
                       const double* __restrict__ sums, float eps, int c, long long vox, int groups, int tiles,
                       int gpb, float* __restrict__ s12, float* __restrict__ pgrad) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int cpg = c / groups;
    const int CB = gpb * cpg;
    const int TL = CB >= 256 ? 1 : 256 / CB;
    float* s_part = reinterpret_cast<float*>(smem_raw);   // [TL][4][CB]
    float* s_tot = s_part + TL * 4 * CB;                   // [4][CB]: sum g, sum g*xhat, sum gc, sum xhat
    float* s_g1 = s_tot + 4 * CB;                          // [gpb] S1/m
    float* s_g2 = s_g1 + gpb;
    const int nb = blockIdx.y, tid = threadIdx.x;
    const int g0 = blockIdx.x * gpb;
    const int ngr = min(gpb, groups - g0);
    const int c0 = g0 * cpg, cn = ngr * cpg;
    const float* base = colsum3 + (long long)nb * tiles * 4 * c + c0;
    for (int chl = tid % CB, tl = tid / CB; chl < cn && tl < TL; chl += 256 * ((CB + 255) / 256)) {
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
        int t = tl;
        for (; t + 3 * TL < tiles; t += 4 * TL) {       // 16 independent loads in flight (fixed order: deterministic)
            float r[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* row = base + (long long)(t + u * TL) * 4 * c + chl;
                r[u][0] = row[0]; r[u][1] = row[c]; r[u][2] = row[2 * c]; r[u][3] = row[3 * c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                t0 += r[u][0]; t1 += r[u][1]; t2 += r[u][2]; t3 += r[u][3];
            }
        }
        for (; t < tiles; t += TL) {
            const float* row = base + (long long)t * 4 * c + chl;
            t0 += row[0];
            t1 += row[c];
            t2 += row[2 * c];
            t3 += row[3 * c];
        }
        s_part[(tl * 4 + 0) * CB + chl] = t0;
        s_part[(tl * 4 + 1) * CB + chl] = t1;
        s_part[(tl * 4 + 2) * CB + chl] = t2;
        s_part[(tl * 4 + 3) * CB + chl] = t3;
        if (CB <= 256) break;
    }
    __syncthreads();
    for (int e = tid; e < 4 * CB; e += 256) {
        const int k = e / CB, chl = e - k * CB;
        float t = 0.0f;
        if (chl < cn)
            for (int tl = 0; tl < TL; ++tl) t += s_part[(tl * 4 + k) * CB + chl];
        s_tot[e] = t;
    }
    __syncthreads();
    const double cnt = (double)cpg * (double)vox;
    const float inv_m = (float)(1.0 / cnt);
    for (int g = tid; g < ngr; g += 256) {
        float sa = 0.0f, sb = 0.0f;
        for (int k = 0; k < cpg; ++k) {
            const float ga = gamma[c0 + g * cpg + k];
            sa += ga * s_tot[0 * CB + g * cpg + k];
            sb += ga * s_tot[1 * CB + g * cpg + k];
        }
        s_g1[g] = sa * inv_m;
        s_g2[g] = sb * inv_m;
        s12[((long long)nb * groups + g0 + g) * 2 + 0] = sa * inv_m;
        s12[((long long)nb * groups + g0 + g) * 2 + 1] = sb * inv_m;
    }
    __syncthreads();
    for (int chl = tid; chl < cn; chl += 256) {
        const int g = chl / cpg, ch = c0 + chl;
        const double m = sums[((long long)nb * groups + g0 + g) * 2 + 0] / cnt;
        double var = sums[((long long)nb * groups + g0 + g) * 2 + 1] / cnt - m * m;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        pgrad[((long long)nb * 4 + 0) * c + ch] = s_tot[1 * CB + chl];
        pgrad[((long long)nb * 4 + 1) * c + ch] = s_tot[0 * CB + chl];
        pgrad[((long long)nb * 4 + 2) * c + ch] = s_tot[2 * CB + chl];
        pgrad[((long long)nb * 4 + 3) * c + ch] =
            rstd * (gamma[ch] * s_tot[0 * CB + chl] - (float)vox * s_g1[g] - s_g2[g] * s_tot[3 * CB + chl]);
    }
}

// Pass 2b: dgamma[c] = sum_n pgrad[n][0][c], dbeta[c] = sum_n pgrad[n][1][c]; dtb[n][c] = pgrad[n][2][c] (optional,
// row stride dtb_stride); dxsum[c] = sum_n pgrad[n][3][c] (optional)
__global__ void __launch_bounds__(256)
gn_bwd_param_kernel(const float* __restrict__ pgrad, int n, int c, float* __restrict__ dgamma,
                    float* __restrict__ dbeta, float* __restrict__ dtb, long long dtb_stride,
                    float* __restrict__ dxsum) {
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= c) return;
    float a = 0.0f, b = 0.0f, d = 0.0f;
    for (int i = 0; i < n; ++i) {
        a += pgrad[((long long)i * 4 + 0) * c + ch];
        b += pgrad[((long long)i * 4 + 1) * c + ch];
        d += pgrad[((long long)i * 4 + 3) * c + ch];
        if (dtb) dtb[(long long)i * dtb_stride + ch] = pgrad[((long long)i * 4 + 2) * c + ch];
    }
    dgamma[ch] = a;
    dbeta[ch] = b;
    if (dxsum) dxsum[ch] = d;
}

// Pass 3: dx = rstd*(gamma*g - S1/m - xhat*S2/m) (+ add).  grid (blocks, n), block 256.  The grid stride is a multiple of the
// row's chunk count (the host picks the block count so), so a thread keeps its 8 channels' five coefficients in registers.
// GMODE 0: g is read from pass 1's buffer; 1: g = dy * silu'(gn(x)) re-derived from dy (pass 1 wrote no g: nothing else
// needs it -- one tensor write less, and pass 3 re-reads what pass 1 just read through the Infinity Cache); 2: g = dy
// (dy_mod > 0: a depth-broadcast dy).  In modes 1 / 2 `g` is dy.
template <bool ADD, int GMODE>
__global__ void __launch_bounds__(256)
gn_bwd_apply_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ x, const double* __restrict__ sums,
                    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ s12,
                    const bf16_t* __restrict__ add, bf16_t* __restrict__ dx, int c, long long vox, int groups,
                    float eps, int contig, long long dy_mod) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* s_rs = reinterpret_cast<float*>(smem_raw);
    float* s_mr = s_rs + c;
    float* s_ag = s_mr + c;   // rstd*gamma
    float* s_b1 = s_ag + c;   // rstd*S1/m
    float* s_b2 = s_b1 + c;   // rstd*S2/m
    const int nb = blockIdx.y, tid = threadIdx.x;
    const int cpg = c / groups;
    const double cnt = (double)cpg * (double)vox;
    for (int ch = tid; ch < c; ch += 256) {
        const int gi = ch / cpg;
        const double m = sums[((long long)nb * groups + gi) * 2 + 0] / cnt;
        double var = sums[((long long)nb * groups + gi) * 2 + 1] / cnt - m * m;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        s_rs[ch] = rstd;
        s_mr[ch] = -(float)m * rstd;
        s_ag[ch] = rstd * gamma[ch];
        s_b1[ch] = rstd * s12[((long long)nb * groups + gi) * 2 + 0];
        s_b2[ch] = rstd * s12[((long long)nb * groups + gi) * 2 + 1];
    }
    __syncthreads();
    const int cpr = c >> 3;
    const long long total = vox * cpr;
    const bf16_t* gb = g + (long long)nb * ((GMODE != 0 && dy_mod) ? dy_mod : vox) * c;
    const bf16_t* xb = x + (long long)nb * vox * c;
    const bf16_t* ab = ADD ? add + (long long)nb * vox * c : nullptr;
    bf16_t* ob = dx + (long long)nb * vox * c;
    // contig: one contiguous batch of 2 x 256 chunks per block, blocks in address order (see ctsi_gn_apply); else grid-stride
    const long long stride = contig ? 256 : (long long)gridDim.x * 256;
    const long long e0 = (contig ? (long long)blockIdx.x * 512 : (long long)blockIdx.x * 256) + tid;
    const long long total_end = contig && (long long)(blockIdx.x + 1) * 512 < total ? (long long)(blockIdx.x + 1) * 512 : total;
    const int q = (int)(e0 % cpr);                       // stride % cpr == 0: the same chunk every iteration
    float rs[8], mr[8], ag[8], b1[8], b2[8], ga[8], be[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        ga[k] = GMODE == 1 ? gamma[q * 8 + k] : 0.0f;
        be[k] = GMODE == 1 ? beta[q * 8 + k] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 8; k += 4) {
        *reinterpret_cast<float4*>(rs + k) = *reinterpret_cast<const float4*>(s_rs + q * 8 + k);
        *reinterpret_cast<float4*>(mr + k) = *reinterpret_cast<const float4*>(s_mr + q * 8 + k);
        *reinterpret_cast<float4*>(ag + k) = *reinterpret_cast<const float4*>(s_ag + q * 8 + k);
        *reinterpret_cast<float4*>(b1 + k) = *reinterpret_cast<const float4*>(s_b1 + q * 8 + k);
        *reinterpret_cast<float4*>(b2 + k) = *reinterpret_cast<const float4*>(s_b2 + q * 8 + k);
    }
    auto one = [&](const uint4 graw, const uint4 xraw, const uint4 araw, long long e) {
        float gf[8], xf[8], af[8], of[8];
        t_unpack8(graw, gf);
        t_unpack8(xraw, xf);
        if (ADD) t_unpack8(araw, af);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xh = xf[k] * rs[k] + mr[k];
            float gg = gf[k];
            if (GMODE == 1) gg = bf16_to_f32(f32_to_bf16(gg * silu_grad_f(xh * ga[k] + be[k])));   // (rounded as pass 1's buffer was)
            float v = ag[k] * gg - b1[k] - xh * b2[k];
            if (ADD) v += af[k];
            of[k] = v;
        }
        *reinterpret_cast<uint4*>(ob + e * 8) = t_pack8(of);
    };
    constexpr int U = 2;                                 // independent iterations in flight (4-6 16-byte loads)
    long long e = e0;
    for (; e + (U - 1) * stride < total_end; e += U * stride) {
        uint4 gr[U], xr[U], ar[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long eg = (GMODE != 0 && dy_mod) ? ((e + u * stride) / cpr % dy_mod) * cpr + q : e + u * stride;
            gr[u] = *reinterpret_cast<const uint4*>(gb + eg * 8);
            xr[u] = *reinterpret_cast<const uint4*>(xb + (e + u * stride) * 8);
            ar[u] = ADD ? *reinterpret_cast<const uint4*>(ab + (e + u * stride) * 8) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) one(gr[u], xr[u], ar[u], e + u * stride);
    }
    for (; e < total_end; e += stride) {
        const long long eg = (GMODE != 0 && dy_mod) ? (e / cpr % dy_mod) * cpr + q : e;
        one(*reinterpret_cast<const uint4*>(gb + eg * 8), *reinterpret_cast<const uint4*>(xb + e * 8),
            ADD ? *reinterpret_cast<const uint4*>(ab + e * 8) : make_uint4(0, 0, 0, 0), e);
    }
}

extern "C" int ctsi_gn_bwd_tiles(int d, int h, int w) {      // (tiles at the full 512-row tile; see gnb_tile_rows)
    const long long vox = (long long)d * h * w;
    return (int)((vox + GNB_TILE_ROWS - 1) / GNB_TILE_ROWS);
}

// Rows per tile of pass 1: 512 for large tensors; halved until the launch has ~2048 blocks, down to one batch of four rows per
// thread lane.  With fixed 512-row tiles the coarse levels of a 192^2 patch were 16-216 blocks whose threads walked 32-128
// voxels one load batch after the other: 38-47 us per launch whatever the tensor size (profiles/r03_train_kernel_stats.csv).
static int gnb_tile_rows(int n, int c, long long vox) {
    const int rows_par = 256 / (c >> 3);
    const int min_rows = rows_par * 4 > 16 ? rows_par * 4 : 16;
    int rows = GNB_TILE_ROWS;
    while (rows > min_rows && (long long)n * ((vox + rows - 1) / rows) < 2048) rows >>= 1;
    return rows;
}

extern "C" int ctsi_gn_bwd(const void* x, const void* dy, int dy_bcast_d, const double* sums, const float* gamma,
                           const float* beta, int n, int c, int d, int h, int w, int groups, float eps, int silu_pre,
                           const void* residual, int silu_post, const void* add, void* g_buf, void* dx,
                           float* workspace, float* dgamma, float* dbeta, float* dtbias, long long dtbias_stride,
                           float* dxsum, void* stream) {
    CTSI_CHECK_ARG(x && dy && sums && gamma && beta && dx && workspace && dgamma && dbeta,
                   "ctsi_gn_bwd: null argument");
    // g_buf may be NULL when nothing but pass 3 needs the GroupNorm output's gradient: pass 3 re-derives it from dy
    CTSI_CHECK_ARG(g_buf || (!residual && !silu_post), "ctsi_gn_bwd: g_buf may only be omitted without residual / post-SiLU");
    CTSI_CHECK_ARG(c % 8 == 0 && c <= 2048 && groups > 0 && c % groups == 0, "ctsi_gn_bwd: bad c=%d groups=%d", c,
                   groups);
    const long long vox = (long long)d * h * w;
    const int tile_rows = gnb_tile_rows(n, c, vox);
    const int tiles = (int)((vox + tile_rows - 1) / tile_rows);
    hipStream_t st = (hipStream_t)stream;
    // workspace: colsum3 [n][tiles][4][c] | s12 [n][groups][2] | pgrad [n][4][c]
    float* colsum3 = workspace;
    float* s12 = colsum3 + (long long)n * tiles * 4 * c;
    float* pgrad = s12 + (long long)n * groups * 2;
    const int cpr = c >> 3;
    const int rows_par = 256 / cpr;
    CTSI_CHECK_ARG(rows_par >= 1, "ctsi_gn_bwd: c=%d too wide", c);
    const size_t lds1 = (size_t)(4 * c + rows_par * 4 * c) * sizeof(float);
    CTSI_CHECK_ARG(lds1 <= 160 * 1024, "ctsi_gn_bwd: LDS budget exceeded for c=%d", c);
    typedef void (*reduce_fn)(const bf16_t*, const bf16_t*, long long, const double*, const float*, const float*, int, long long,
                              int, float, const bf16_t*, bf16_t*, float*, int, int);
    static const reduce_fn reduce_tab[8] = {
        gn_bwd_reduce_kernel<false, false, false>, gn_bwd_reduce_kernel<true, false, false>,
        gn_bwd_reduce_kernel<false, true, false>,  gn_bwd_reduce_kernel<true, true, false>,
        gn_bwd_reduce_kernel<false, false, true>,  gn_bwd_reduce_kernel<true, false, true>,
        gn_bwd_reduce_kernel<false, true, true>,   gn_bwd_reduce_kernel<true, true, true>};
    const reduce_fn reduce = reduce_tab[(silu_pre ? 1 : 0) | (residual ? 2 : 0) | (silu_post ? 4 : 0)];
    if (lds1 > 64 * 1024)
        CTSI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(reduce),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    hipLaunchKernelGGL(reduce, dim3(tiles, n), dim3(256), lds1, st, (const bf16_t*)x, (const bf16_t*)dy,
                       dy_bcast_d ? (long long)h * w : 0ll, sums, gamma, beta, c, vox, groups, eps,
                       (const bf16_t*)residual, (bf16_t*)g_buf, colsum3, tiles, tile_rows);
    CTSI_LAUNCH_CHECK();
    {
        const int cpg = c / groups;
        // channels per block: 64 (4 tile lanes) when there are few tiles; 16 (16 lanes, 4 x the blocks) when a (sample, group)
        // has many -- with the smaller statistics tiles of pass 1 a 4-lane block walked up to 216 tile rows one batch after
        // the other (21 us per launch on average against 9 before the tile change)
        int gpb = (tiles > 32 ? 16 : 64) / cpg;
        if (gpb < 1) gpb = 1;
        if (gpb > groups) gpb = groups;
        const int CB = gpb * cpg;
        const int TL = CB >= 256 ? 1 : 256 / CB;
        const size_t lds2 = (size_t)(TL * 4 * CB + 4 * CB + 2 * gpb) * sizeof(float);
        hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((groups + gpb - 1) / gpb, n), dim3(256), lds2, st, colsum3, gamma,
                           sums, eps, c, vox, groups, tiles, gpb, s12, pgrad);
    }
    CTSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((c + 255) / 256), dim3(256), 0, st, pgrad, n, c, dgamma, dbeta, dtbias,
                       dtbias_stride, dxsum);
    CTSI_LAUNCH_CHECK();
    const long long total = vox * cpr;
    long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    int contig = 0;
    if (256 % cpr == 0) {   // contiguous batches of 512 chunks per block
        contig = 1;
        blocks = (total + 511) / 512;
    } else {   // grid stride = a multiple of the chunks per voxel row (cpr <= 256: one block's 256 threads times need)
        int gq = cpr, g256 = 256;
        while (g256) { const int t = gq % g256; gq = g256; g256 = t; }     // gcd(cpr, 256)
        const int need = cpr / gq;
        blocks = (blocks + need - 1) / need * need;
    }
    {
        typedef void (*apply_fn)(const bf16_t*, const bf16_t*, const double*, const float*, const float*, const float*,
                                 const bf16_t*, bf16_t*, int, long long, int, float, int, long long);
        static const apply_fn apply_tab[6] = {gn_bwd_apply_kernel<false, 0>, gn_bwd_apply_kernel<true, 0>,
                                              gn_bwd_apply_kernel<false, 1>, gn_bwd_apply_kernel<true, 1>,
                                              gn_bwd_apply_kernel<false, 2>, gn_bwd_apply_kernel<true, 2>};
        const int gmode = g_buf ? 0 : (silu_pre ? 1 : 2);
        const long long dy_mod = (gmode != 0 && dy_bcast_d) ? (long long)h * w : 0ll;
        hipLaunchKernelGGL(apply_tab[gmode * 2 + (add ? 1 : 0)], dim3((unsigned)blocks, n), dim3(256), 5 * c * sizeof(float), st,
                           (const bf16_t*)(g_buf ? g_buf : dy), (const bf16_t*)x, sums, gamma, beta, s12, (const bf16_t*)add,
                           (bf16_t*)dx, c, vox, groups, eps, contig, dy_mod);
    }
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

extern "C" size_t ctsi_gn_bwd_workspace_floats(int n, int c, int d, int h, int w, int groups) {
    const long long vox = (long long)d * h * w;
    const int tile_rows = gnb_tile_rows(n, c, vox);
    const long long tiles = (vox + tile_rows - 1) / tile_rows;
    return (size_t)((long long)n * tiles * 4 * c + (long long)n * groups * 2 + (long long)n * 4 * c);
}

// ==== channel sums (bias gradients) ==================================================================================
// partial[b][c] over row blocks, then out[c] = scale * sum_b partial[b][c]
__global__ void __launch_bounds__(256)
channel_sum_partial_kernel(const bf16_t* __restrict__ x, long long rows, int c, int c_stride,
                           float* __restrict__ partial, long long rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* s_red = reinterpret_cast<float*>(smem_raw);  // [rows_par][c]
    const int cpr = c >> 3;
    const int rows_par = 256 / cpr;
    const int tid = threadIdx.x, q = tid % cpr, rl = tid / cpr;
    const long long v0 = (long long)blockIdx.x * rows_per_block;
    long long v1 = v0 + rows_per_block;
    if (v1 > rows) v1 = rows;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.0f;
    if (rl < rows_par) {
        long long v = v0 + rl;
        for (; v + 3ll * rows_par < v1; v += 4ll * rows_par) {   // four rows in flight per thread, added in row order
            uint4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint4*>(x + (v + (long long)u * rows_par) * c_stride + q * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8];
                t_unpack8(raw[u], f);
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += f[k];
            }
        }
        for (; v < v1; v += rows_par) {
            float f[8];
            t_unpack8(*reinterpret_cast<const uint4*>(x + v * c_stride + q * 8), f);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += f[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s_red[rl * c + q * 8 + k] = a[k];
    }
    __syncthreads();
    for (int ch = tid; ch < c; ch += 256) {
        float t = 0.0f;
        for (int r = 0; r < rows_par; ++r) t += s_red[r * c + ch];
        partial[(long long)blockIdx.x * c + ch] = t;
    }
}
// one block = 16 channels x 16 lanes over the row blocks (one serial thread per channel took 50-100 us for the 432 partial rows
// of a 4 x 48^3 tensor); every lane sums its strided subset in order, lane 0 adds the 16 lane sums in order: a fixed order
__global__ void __launch_bounds__(256)
channel_sum_final_kernel(const float* __restrict__ partial, int blocks, int c, float* __restrict__ out, float scale) {
    __shared__ float s_part[16][17];
    const int cl = threadIdx.x & 15, r = threadIdx.x >> 4;
    const int ch = blockIdx.x * 16 + cl;
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
    if (ch < c) {
        int b = r;
        for (; b + 48 < blocks; b += 64) {
            t0 += partial[(long long)b * c + ch];
            t1 += partial[(long long)(b + 16) * c + ch];
            t2 += partial[(long long)(b + 32) * c + ch];
            t3 += partial[(long long)(b + 48) * c + ch];
        }
        for (; b < blocks; b += 16) t0 += partial[(long long)b * c + ch];
    }
    s_part[r][cl] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (r == 0 && ch < c) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s_part[k][cl];
        out[ch] = t * scale;
    }
}
#define CHSUM_ROWS 1024
extern "C" size_t ctsi_channel_sum_workspace_floats(long long rows, int c) {
    return (size_t)(((rows + CHSUM_ROWS - 1) / CHSUM_ROWS) * c);
}
extern "C" int ctsi_channel_sum(const void* x, long long rows, int c, int c_stride, float* workspace, float* out,
                                float scale, void* stream) {
    CTSI_CHECK_ARG(x && workspace && out && rows > 0 && c > 0 && c % 8 == 0 && c <= 2048 && c_stride % 8 == 0 &&
                       c_stride >= c, "ctsi_channel_sum: bad arguments (c=%d stride=%d)", c, c_stride);
    const long long blocks = (rows + CHSUM_ROWS - 1) / CHSUM_ROWS;
    const int rows_par = 256 / (c >> 3);
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3((unsigned)blocks), dim3(256), (size_t)rows_par * c * sizeof(float),
                       (hipStream_t)stream, (const bf16_t*)x, rows, c, c_stride, workspace, (long long)CHSUM_ROWS);
    CTSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3((c + 15) / 16), dim3(256), 0, (hipStream_t)stream, workspace,
                       (int)blocks, c, out, scale);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// ==== small elementwise helpers ======================================================================================
// (one contiguous batch of 4 x 256 16-byte chunks per block, blocks in address order: see ctsi_gn_apply)
__global__ void __launch_bounds__(256) add_bf16_kernel(bf16_t* __restrict__ a, const bf16_t* __restrict__ b, long long chunks) {
    const long long e0 = (long long)blockIdx.x * 1024 + threadIdx.x;
    uint4 ra[4], rb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (e0 + u * 256 < chunks) {
            ra[u] = *reinterpret_cast<const uint4*>(a + (e0 + u * 256) * 8);
            rb[u] = *reinterpret_cast<const uint4*>(b + (e0 + u * 256) * 8);
        }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (e0 + u * 256 < chunks) {
            float fa[8], fb[8];
            t_unpack8(ra[u], fa);
            t_unpack8(rb[u], fb);
#pragma unroll
            for (int k = 0; k < 8; ++k) fa[k] += fb[k];
            *reinterpret_cast<uint4*>(a + (e0 + u * 256) * 8) = t_pack8(fa);
        }
}
extern "C" int ctsi_add_bf16(void* a, const void* b, long long count, void* stream) {
    CTSI_CHECK_ARG(a && b && count >= 0 && count % 8 == 0, "ctsi_add_bf16: count must be a multiple of 8");
    if (count == 0) return CTSI_OK;
    const long long chunks = count / 8;
    const long long blocks = (chunks + 1023) / 1024;
    hipLaunchKernelGGL(add_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (bf16_t*)a,
                       (const bf16_t*)b, chunks);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// fp32 [rows][c] -> bf16 [rows][c]
__global__ void __launch_bounds__(256) f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long count) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256)
        dst[e] = f32_to_bf16(src[e]);
}
extern "C" int ctsi_f32_to_bf16(const float* src, void* dst, long long count, void* stream) {
    CTSI_CHECK_ARG(src && dst && count >= 0, "ctsi_f32_to_bf16: bad arguments");
    if (count == 0) return CTSI_OK;
    long long blocks = (count + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, count);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// out (cout' = ci_cnt, cin' = cout, T):  out[(ci' * cout + co) * T + (T-1-t)] = w[(co * cin + ci_off + ci') * T + t]
__global__ void __launch_bounds__(256)
weight_dgrad_layout_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int taps, int ci_off,
                           int ci_cnt, long long total) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int t = (int)(e % taps);
        const long long r = e / taps;
        const int co = (int)(r % cout);
        const int ci = (int)(r / cout);
        out[e] = w[((long long)co * cin + ci_off + ci) * taps + (taps - 1 - t)];
    }
}
extern "C" int ctsi_weight_dgrad_layout(const float* w, float* out, int cout, int cin, int taps, int ci_off, int ci_cnt,
                                        void* stream) {
    CTSI_CHECK_ARG(w && out && cout > 0 && cin > 0 && taps > 0 && ci_off >= 0 && ci_cnt > 0 && ci_off + ci_cnt <= cin,
                   "ctsi_weight_dgrad_layout: bad arguments");
    const long long total = (long long)ci_cnt * cout * taps;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(weight_dgrad_layout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, out, cout,
                       cin, taps, ci_off, ci_cnt, total);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// ==== q_sample and the loss ============================================================================================
// z_t = sqrt_ac[t_b] * z0 + sqrt_1mac[t_b] * noise     (fp32 NCDHW in) -> bf16 NDHWC channels [c_off, c_off + c)
__global__ void __launch_bounds__(256)
q_sample_kernel(const float* __restrict__ z0, const float* __restrict__ noise, const float* __restrict__ sqrt_ac,
                const float* __restrict__ sqrt_1mac, const int* __restrict__ t, bf16_t* __restrict__ dst, int c,
                long long vox, int c_total, int c_off, long long total /* n*vox */) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long nb = e / vox, v = e - nb * vox;
        const int tt = t[nb];
        const float a = sqrt_ac[tt], s = sqrt_1mac[tt];
        for (int ch = 0; ch < c; ++ch) {
            const long long i = (nb * c + ch) * vox + v;
            dst[e * c_total + c_off + ch] = f32_to_bf16(a * z0[i] + s * noise[i]);
        }
    }
}
extern "C" int ctsi_q_sample(const float* z0, const float* noise, const float* sqrt_ac, const float* sqrt_1mac,
                             const int* t, void* dst, int n, int c, int d, int h, int w, int c_total, int c_off,
                             void* stream) {
    CTSI_CHECK_ARG(z0 && noise && sqrt_ac && sqrt_1mac && t && dst && n > 0 && c > 0 && c_off >= 0 && c_off + c <= c_total,
                   "ctsi_q_sample: bad arguments");
    const long long vox = (long long)d * h * w, total = vox * n;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(q_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z0, noise, sqrt_ac,
                       sqrt_1mac, t, (bf16_t*)dst, c, vox, c_total, c_off, total);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// loss = sum_b norm[b] * sum_e mask * (pred - noise)^2.   pred: fp32 NDHWC (n, vox, c); noise: fp32 NCDHW;
// mask: fp32 (n, c, d) or NULL.  Also d_pred (bf16 NDHWC, c_stride channels per voxel) = 2*norm[b]*mask*(pred-noise)*gscale.
// Stage 1: per-block partial sums in double; stage 2 (one block): loss_out[0] = total, loss_out[1+b] = per-sample
// unnormalised sums.
#define LOSS_BLOCKS_PER_SAMPLE 64
__global__ void __launch_bounds__(256)
mse_loss_partial_kernel(const float* __restrict__ pred, const float* __restrict__ noise, const float* __restrict__ mask,
                        int c, long long vox, long long hw, int d, double* __restrict__ partial) {
    __shared__ double s_red[256];
    const int nb = blockIdx.y;
    const long long total = vox * c;
    double acc = 0.0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int ch = (int)(e % c);
        const long long v = e / c;
        float df = pred[(long long)nb * total + e] - noise[((long long)nb * c + ch) * vox + v];
        float m = 1.0f;
        if (mask) m = mask[((long long)nb * c + ch) * d + (v / hw)];
        acc += (double)(m * df * df);
    }
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) s_red[threadIdx.x] += s_red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(long long)nb * gridDim.x + blockIdx.x] = s_red[0];
}
__global__ void mse_loss_final_kernel(const double* __restrict__ partial, const float* __restrict__ norm, int n, int bps,
                                      float* __restrict__ loss_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double total = 0.0;
    for (int b = 0; b < n; ++b) {
        double s = 0.0;
        for (int k = 0; k < bps; ++k) s += partial[(long long)b * bps + k];
        loss_out[1 + b] = (float)s;
        total += (double)norm[b] * s;
    }
    loss_out[0] = (float)total;
}
extern "C" int ctsi_mse_loss_fwd(const float* pred, const float* noise, const float* mask, const float* norm, int n,
                                 int c, int d, int h, int w, double* workspace, float* loss_out, void* stream) {
    CTSI_CHECK_ARG(pred && noise && norm && workspace && loss_out && n > 0 && c > 0, "ctsi_mse_loss_fwd: bad arguments");
    const long long hw = (long long)h * w, vox = hw * d;
    hipLaunchKernelGGL(mse_loss_partial_kernel, dim3(LOSS_BLOCKS_PER_SAMPLE, n), dim3(256), 0, (hipStream_t)stream, pred,
                       noise, mask, c, vox, hw, d, workspace);
    CTSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(mse_loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, norm, n,
                       LOSS_BLOCKS_PER_SAMPLE, loss_out);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
extern "C" size_t ctsi_mse_loss_workspace_doubles(int n) { return (size_t)n * LOSS_BLOCKS_PER_SAMPLE; }

__global__ void __launch_bounds__(256)
mse_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ noise, const float* __restrict__ mask,
                    const float* __restrict__ norm, const float* __restrict__ gscale, int c, long long vox, long long hw,
                    int d, bf16_t* __restrict__ dpred, int c_stride, long long total /* n*vox*c_stride */) {
    const float gs = gscale ? gscale[0] : 1.0f;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int ch = (int)(e % c_stride);
        const long long nv = e / c_stride;
        const long long nb = nv / vox, v = nv - nb * vox;
        float out = 0.0f;
        if (ch < c) {
            const float df = pred[nv * c + ch] - noise[(nb * c + ch) * vox + v];
            float m = 1.0f;
            if (mask) m = mask[(nb * c + ch) * d + (v / hw)];
            out = 2.0f * norm[nb] * m * df * gs;
        }
        dpred[e] = f32_to_bf16(out);
    }
}
extern "C" int ctsi_mse_loss_bwd(const float* pred, const float* noise, const float* mask, const float* norm,
                                 const float* gscale, int n, int c, int d, int h, int w, void* dpred, int c_stride,
                                 void* stream) {
    CTSI_CHECK_ARG(pred && noise && norm && dpred && n > 0 && c > 0 && c_stride >= c, "ctsi_mse_loss_bwd: bad arguments");
    const long long hw = (long long)h * w, vox = hw * d, total = vox * n * c_stride;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(mse_loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, noise, mask,
                       norm, gscale, c, vox, hw, d, (bf16_t*)dpred, c_stride, total);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// ==== fp32 Linear backward for the time embedding (rows <= 64) ========================================================
// y = act_in(x) W^T + b with act_in = SiLU when silu_in (x is then the pre-activation):
//   dw[o][i] = sum_r dy[r][o] * a[r][i];  db[o] = sum_r dy[r][o];  dx[r][i] = (sum_o dy[r][o] w[o][i]) * act_in'(x[r][i])
__global__ void __launch_bounds__(256)
linear_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dy, int rows, int in_dim, int out_dim,
                    int silu_in, float* __restrict__ dw, float* __restrict__ db) {
    const long long total = (long long)out_dim * in_dim;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int o = (int)(e / in_dim), i = (int)(e - (long long)o * in_dim);
        float s = 0.0f;
        for (int r = 0; r < rows; ++r) {
            float a = x[(long long)r * in_dim + i];
            if (silu_in) a = a / (1.0f + expf(-a));
            s += dy[(long long)r * out_dim + o] * a;
        }
        dw[e] = s;
        if (i == 0 && db) {
            float b = 0.0f;
            for (int r = 0; r < rows; ++r) b += dy[(long long)r * out_dim + o];
            db[o] = b;
        }
    }
}
// grid (in_dim/32, rows): 32 consecutive inputs x 8 output lanes per block, LDS reduction over the lanes
__global__ void __launch_bounds__(256)
linear_bwd_x_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy, int rows,
                    int in_dim, int out_dim, int silu_in, float* __restrict__ dx) {
    __shared__ float s_red[8][32];
    const int il = threadIdx.x & 31, ol = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + il, r = blockIdx.y;
    float s = 0.0f;
    if (i < in_dim)
        for (int o = ol; o < out_dim; o += 8) s += dy[(long long)r * out_dim + o] * w[(long long)o * in_dim + i];
    s_red[ol][il] = s;
    __syncthreads();
    if (ol == 0 && i < in_dim) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += s_red[k][il];
        if (silu_in) {
            const float z = x[(long long)r * in_dim + i];
            const float sg = 1.0f / (1.0f + expf(-z));
            t *= sg * (1.0f + z * (1.0f - sg));
        }
        dx[(long long)r * in_dim + i] = t;
    }
}
extern "C" int ctsi_linear_bwd(const float* x, const float* w, const float* dy, int rows, int in_dim, int out_dim,
                               int silu_in, float* dw, float* db, float* dx, void* stream) {
    CTSI_CHECK_ARG(x && w && dy && rows > 0 && rows <= 64 && in_dim > 0 && out_dim > 0, "ctsi_linear_bwd: bad arguments");
    if (dw) {
        const long long total = (long long)out_dim * in_dim;
        long long blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(linear_bwd_w_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, dy, rows,
                           in_dim, out_dim, silu_in, dw, db);
        CTSI_LAUNCH_CHECK();
    }
    if (dx) {
        hipLaunchKernelGGL(linear_bwd_x_kernel, dim3((in_dim + 31) / 32, rows), dim3(256), 0, (hipStream_t)stream, x, w, dy,
                           rows, in_dim, out_dim, silu_in, dx);
        CTSI_LAUNCH_CHECK();
    }
    return CTSI_OK;
}

// ==== weight + bias gradients of many SMALL pointwise layers in one launch ============================================
// The 22 pointwise layers inside the 11 TemporalAttention blocks (proj_out and the V third of qkv: models/unet3d.py:152-153,
// 185-190) act on depth-summed tensors of 144-2304 rows: their weight gradients were 22 x (split-K kernel + reduce pass) and
// their bias gradients 22 x (partial + final channel sum) = 88 launches of 5-10 us, 14 TFLOP/s (0.9 ms per config-3 micro-step).
// Their operands are tiny (<= 1.2 MB), so the training engine keeps them until the end of the backward pass and issues ONE
// launch: a block owns a 64 x 64 tile of one layer's dW (and, in the first cin tile, 64 entries of db) and walks ALL rows in
// order -- fp32 FMAs on bf16 operands staged through LDS, no split, no atomics: deterministic.
struct LinGradEntry {
    const bf16_t* x;     // [rows][cin]   layer input
    const bf16_t* dy;    // [rows][cout]  output gradient
    float* dw;           // [cout][dw_stride] (+ element offset applied by the host)
    float* db;           // [cout] or NULL
    int rows, cin, cout, dw_stride;
    float b_scale;
    int pad_;
};
struct LinGradBlock { int entry, ct, it, pad_; };

__global__ void __launch_bounds__(256)
linear_wgrad_multi_kernel(const LinGradEntry* __restrict__ entries, const LinGradBlock* __restrict__ blocks) {
    __shared__ __attribute__((aligned(16))) bf16_t s_dy[32][64 + 8];   // (+8: rows 16 B apart mod 128 B)
    __shared__ __attribute__((aligned(16))) bf16_t s_x[32][64 + 8];
    const LinGradBlock b = blocks[blockIdx.x];
    const LinGradEntry e = entries[b.entry];
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int co0 = b.ct * 64, ci0 = b.it * 64;
    const int lr = tid >> 3, lc = (tid & 7) * 8;                        // staging: row lr, 8 channels from lc
    float acc[4][4], bs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bs[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
    }
    for (int r0 = 0; r0 < e.rows; r0 += 32) {
        const int r = r0 + lr;
        uint4 vd = make_uint4(0, 0, 0, 0), vx = make_uint4(0, 0, 0, 0);
        if (r < e.rows) {
            if (co0 + lc < e.cout) vd = *reinterpret_cast<const uint4*>(e.dy + (long long)r * e.cout + co0 + lc);
            if (ci0 + lc < e.cin) vx = *reinterpret_cast<const uint4*>(e.x + (long long)r * e.cin + ci0 + lc);
        }
        __syncthreads();
        *reinterpret_cast<uint4*>(&s_dy[lr][lc]) = vd;
        *reinterpret_cast<uint4*>(&s_x[lr][lc]) = vx;
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            const uint2 d2 = *reinterpret_cast<const uint2*>(&s_dy[k][ty * 4]);
            const uint2 x2 = *reinterpret_cast<const uint2*>(&s_x[k][tx * 4]);
            const float dv[4] = {__uint_as_float(d2.x << 16), __uint_as_float(d2.x & 0xffff0000u), __uint_as_float(d2.y << 16),
                                 __uint_as_float(d2.y & 0xffff0000u)};
            const float xv[4] = {__uint_as_float(x2.x << 16), __uint_as_float(x2.x & 0xffff0000u), __uint_as_float(x2.y << 16),
                                 __uint_as_float(x2.y & 0xffff0000u)};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bs[i] += dv[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(dv[i], xv[j], acc[i][j]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + ty * 4 + i;
        if (co >= e.cout) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = ci0 + tx * 4 + j;
            if (ci < e.cin) e.dw[(long long)co * e.dw_stride + ci] = acc[i][j];
        }
        if (e.db != nullptr && b.it == 0 && tx == 0) e.db[co] = bs[i] * e.b_scale;
    }
}

// entries / blocks: device tables built by the caller (n_blocks rows of {entry, cout tile, cin tile}); channel counts multiples of 8
extern "C" int ctsi_linear_wgrad_multi(const void* entries, const void* blocks, int n_blocks, void* stream) {
    CTSI_CHECK_ARG(entries && blocks && n_blocks > 0, "ctsi_linear_wgrad_multi: bad arguments");
    hipLaunchKernelGGL(linear_wgrad_multi_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const LinGradEntry*)entries, (const LinGradBlock*)blocks);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
