// Boundary layout conversions, trilinear depth upsample, timestep embedding and the DDIM / DDPM
// elementwise updates (all HBM- or latency-bound; vectorised 16-B accesses where the layout allows).
// Reference semantics: models/model.py:252-343 (fp32 NCDHW API tensors, nan_to_num guards,
// F.interpolate trilinear), models/unet3d.py:18-48,88-91,123-125 (time embedding),
// inference/sampler.py:294-334 (DDIM update), models/diffusion.py:270-338 (DDPM update).
#include "ctsi_internal.h"
#include <math.h>

// ---- fp32 NCDHW -> bf16 NDHWC channel slice -------------------------------------------------------
__global__ void __launch_bounds__(256)
ncdhw_f32_to_ndhwc_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int c, long long vox,
                               int c_total, int c_off, long long total /* n*vox */) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long nb = e / vox, v = e - nb * vox;
        const float* s = src + nb * c * vox + v;
        bf16_t* o = dst + e * c_total + c_off;
        int ch = 0;
        if (((c_total | c_off) & 7) == 0) {
            for (; ch + 8 <= c; ch += 8) {
                uint4 pk;
                pk.x = pack_bf16x2(s[(ch + 0) * vox], s[(ch + 1) * vox]);
                pk.y = pack_bf16x2(s[(ch + 2) * vox], s[(ch + 3) * vox]);
                pk.z = pack_bf16x2(s[(ch + 4) * vox], s[(ch + 5) * vox]);
                pk.w = pack_bf16x2(s[(ch + 6) * vox], s[(ch + 7) * vox]);
                *reinterpret_cast<uint4*>(o + ch) = pk;
            }
        }
        for (; ch < c; ++ch) o[ch] = f32_to_bf16(s[ch * vox]);
    }
}

extern "C" int ctsi_ncdhw_f32_to_ndhwc_bf16(const float* src, void* dst, int n, int c, int d, int h, int w,
                                            int c_total, int c_off, void* stream) {
    CTSI_CHECK_ARG(src && dst && n > 0 && c > 0 && c_off >= 0 && c_off + c <= c_total,
                   "ctsi_ncdhw_f32_to_ndhwc_bf16: bad arguments (c=%d c_total=%d c_off=%d)", c, c_total, c_off);
    const long long vox = (long long)d * h * w, total = vox * n;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(ncdhw_f32_to_ndhwc_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       src, (bf16_t*)dst, c, vox, c_total, c_off, total);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// ---- bf16 NDHWC -> fp32 NCDHW ------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
ndhwc_bf16_to_ncdhw_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int c, long long vox,
                               long long total) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long nb = e / vox, v = e - nb * vox;
        const bf16_t* s = src + e * c;
        float* o = dst + nb * c * vox + v;
        for (int ch = 0; ch < c; ++ch) o[ch * vox] = bf16_to_f32(s[ch]);
    }
}

extern "C" int ctsi_ndhwc_bf16_to_ncdhw_f32(const void* src, float* dst, int n, int c, int d, int h, int w,
                                            void* stream) {
    CTSI_CHECK_ARG(src && dst && n > 0 && c > 0, "ctsi_ndhwc_bf16_to_ncdhw_f32: bad arguments");
    const long long vox = (long long)d * h * w, total = vox * n;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(ndhwc_bf16_to_ncdhw_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, dst, c, vox, total);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// ---- fp32 NCDHW <-> fp32 NDHWC ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
f32_layout_kernel(const float* __restrict__ src, float* __restrict__ dst, int c, long long vox, long long total,
                  int to_ndhwc) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long nb = e / vox, v = e - nb * vox;
        if (to_ndhwc) {
            for (int ch = 0; ch < c; ++ch) dst[e * c + ch] = src[(nb * c + ch) * vox + v];
        } else {
            for (int ch = 0; ch < c; ++ch) dst[(nb * c + ch) * vox + v] = src[e * c + ch];
        }
    }
}
static int f32_layout(const float* src, float* dst, int n, int c, int d, int h, int w, int to_ndhwc, void* stream) {
    CTSI_CHECK_ARG(src && dst && n > 0 && c > 0, "fp32 layout conversion: bad arguments");
    const long long vox = (long long)d * h * w, total = vox * n;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(f32_layout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, c,
                       vox, total, to_ndhwc);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
extern "C" int ctsi_ncdhw_f32_to_ndhwc_f32(const float* src, float* dst, int n, int c, int d, int h, int w,
                                           void* stream) {
    return f32_layout(src, dst, n, c, d, h, w, 1, stream);
}
extern "C" int ctsi_ndhwc_f32_to_ncdhw_f32(const float* src, float* dst, int n, int c, int d, int h, int w,
                                           void* stream) {
    return f32_layout(src, dst, n, c, d, h, w, 0, stream);
}

// ---- trilinear depth upsample (h, w unchanged) ---------------------------------------------------------------
// src index (align_corners=False): s = max((d + 0.5) * d_in/d_out - 0.5, 0); i0 = floor(s);
// i1 = min(i0 + 1, d_in - 1); out = (1 - (s - i0)) * x[i0] + (s - i0) * x[i1]    (fp32)
__global__ void __launch_bounds__(256)
trilinear_depth_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst_bf16, float* __restrict__ dst_f32,
                       int c, int d_in, int d_out, long long hw, int c_total, int c_off, long long total) {
    const float scale = (float)d_in / (float)d_out;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long p = e % hw;
        const long long nd = e / hw;
        const int dd = (int)(nd % d_out);
        const long long nb = nd / d_out;
        float s = scale * ((float)dd + 0.5f) - 0.5f;
        if (s < 0.0f) s = 0.0f;
        const int i0 = (int)s;
        const int i1 = i0 + (i0 < d_in - 1 ? 1 : 0);
        const float l1 = s - (float)i0, l0 = 1.0f - l1;
        for (int ch = 0; ch < c; ++ch) {
            const float* sp = src + ((nb * c + ch) * d_in) * hw + p;
            const float v = l0 * sp[(long long)i0 * hw] + l1 * sp[(long long)i1 * hw];
            if (dst_bf16) dst_bf16[e * c_total + c_off + ch] = f32_to_bf16(v);
            if (dst_f32) dst_f32[((nb * c + ch) * d_out + dd) * hw + p] = v;
        }
    }
}

extern "C" int ctsi_trilinear_depth_fwd(const float* src, void* dst_bf16, int n, int c, int d_in, int d_out,
                                        int h, int w, int c_total, int c_off, float* dst_f32, void* stream) {
    CTSI_CHECK_ARG(src && (dst_bf16 || dst_f32) && d_in > 0 && d_out > 0, "ctsi_trilinear_depth_fwd: bad arguments");
    CTSI_CHECK_ARG(!dst_bf16 || (c_off >= 0 && c_off + c <= c_total), "ctsi_trilinear_depth_fwd: bad channel slice");
    const long long hw = (long long)h * w, total = (long long)n * d_out * hw;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(trilinear_depth_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst_bf16, dst_f32, c, d_in, d_out, hw, c_total, c_off, total);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// ---- timestep embedding ---------------------------------------------------------------------------------------------
// rows = (steps x batch) timestep values.  Three launches:
//   (1) sincos + Linear1 + SiLU   (2) Linear2   (3) all per-ResBlock Linear(SiLU(temb)) stacked.
// One wave per output feature; the wave keeps its weight row in registers and walks the rows.
__global__ void __launch_bounds__(256)
time_sincos_kernel(const int* __restrict__ t_rows, int rows, int dim, float* __restrict__ out) {
    const int half = dim / 2;
    const float step = logf(10000.0f) / (float)(half - 1);
    for (int e = blockIdx.x * 256 + threadIdx.x; e < rows * half; e += gridDim.x * 256) {
        const int r = e / half, i = e - r * half;
        const float freq = expf((float)i * -step);
        const float arg = (float)t_rows[r] * freq;
        out[r * dim + i] = sinf(arg);
        out[r * dim + half + i] = cosf(arg);
    }
}

// out[r][o] = act_out( W[o] . act_in(in[r]) + b[o] );  in_dim <= 2048
__global__ void __launch_bounds__(256)
time_linear_kernel(const float* __restrict__ in, int rows, int in_dim, const float* __restrict__ w,
                   const float* __restrict__ b, int out_dim, float* __restrict__ out, int silu_in, int silu_out) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= out_dim) return;
    constexpr int MAXK = 32;  // in_dim <= 64*32
    float wr[MAXK];
    const int nk = (in_dim + 63) / 64;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        const int idx = k * 64 + lane;
        wr[k] = (k < nk && idx < in_dim) ? w[(long long)o * in_dim + idx] : 0.0f;
    }
    const float bias = b[o];
    for (int r = 0; r < rows; ++r) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            const int idx = k * 64 + lane;
            if (k < nk && idx < in_dim) {
                float v = in[(long long)r * in_dim + idx];
                if (silu_in) v = silu_f(v);
                acc += wr[k] * v;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) {
            float v = acc + bias;
            if (silu_out) v = silu_f(v);
            out[(long long)r * out_dim + o] = v;
        }
    }
}

extern "C" int ctsi_time_embed_fwd(const int* t_rows, int rows, int dim, int time_dim, const float* w1,
                                   const float* b1, const float* w2, const float* b2, const float* w_all,
                                   const float* b_all, int total_out, float* scratch, float* tbias_out,
                                   void* stream) {
    CTSI_CHECK_ARG(t_rows && w1 && b1 && w2 && b2 && scratch, "ctsi_time_embed_fwd: null argument");
    CTSI_CHECK_ARG(rows > 0 && dim >= 4 && dim % 2 == 0 && dim <= 2048 && time_dim > 0 && time_dim <= 2048,
                   "ctsi_time_embed_fwd: unsupported sizes dim=%d time_dim=%d", dim, time_dim);
    hipStream_t st = (hipStream_t)stream;
    float* sincos = scratch;                           // rows*dim
    float* hidden = scratch + (long long)rows * dim;   // rows*time_dim
    float* temb = hidden + (long long)rows * time_dim; // rows*time_dim
    int blocks = (rows * (dim / 2) + 255) / 256;
    hipLaunchKernelGGL(time_sincos_kernel, dim3(blocks), dim3(256), 0, st, t_rows, rows, dim, sincos);
    CTSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(time_linear_kernel, dim3((time_dim + 3) / 4), dim3(256), 0, st, sincos, rows, dim, w1, b1,
                       time_dim, hidden, 0, 1);
    CTSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(time_linear_kernel, dim3((time_dim + 3) / 4), dim3(256), 0, st, hidden, rows, time_dim, w2,
                       b2, time_dim, temb, 0, 0);
    CTSI_LAUNCH_CHECK();
    if (total_out > 0) {
        CTSI_CHECK_ARG(w_all && b_all && tbias_out, "ctsi_time_embed_fwd: null stacked projection");
        hipLaunchKernelGGL(time_linear_kernel, dim3((total_out + 3) / 4), dim3(256), 0, st, temb, rows, time_dim,
                           w_all, b_all, total_out, tbias_out, 1, 0);
        CTSI_LAUNCH_CHECK();
    }
    return CTSI_OK;
}

// Training variant: scratch keeps the PRE-activation of the first Linear (the backward needs SiLU' of it);
// the SiLU moves into the second Linear's input.  Same values as ctsi_time_embed_fwd.
extern "C" int ctsi_time_embed_train_fwd(const int* t_rows, int rows, int dim, int time_dim, const float* w1,
                                         const float* b1, const float* w2, const float* b2, const float* w_all,
                                         const float* b_all, int total_out, float* scratch, float* tbias_out,
                                         void* stream) {
    CTSI_CHECK_ARG(t_rows && w1 && b1 && w2 && b2 && scratch && w_all && b_all && tbias_out,
                   "ctsi_time_embed_train_fwd: null argument");
    CTSI_CHECK_ARG(rows > 0 && dim >= 4 && dim % 2 == 0 && dim <= 2048 && time_dim > 0 && time_dim <= 2048,
                   "ctsi_time_embed_train_fwd: unsupported sizes dim=%d time_dim=%d", dim, time_dim);
    hipStream_t st = (hipStream_t)stream;
    float* sincos = scratch;                           // rows*dim
    float* lin1 = scratch + (long long)rows * dim;     // rows*time_dim, pre-activation
    float* temb = lin1 + (long long)rows * time_dim;   // rows*time_dim
    int blocks = (rows * (dim / 2) + 255) / 256;
    hipLaunchKernelGGL(time_sincos_kernel, dim3(blocks), dim3(256), 0, st, t_rows, rows, dim, sincos);
    CTSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(time_linear_kernel, dim3((time_dim + 3) / 4), dim3(256), 0, st, sincos, rows, dim, w1, b1,
                       time_dim, lin1, 0, 0);
    CTSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(time_linear_kernel, dim3((time_dim + 3) / 4), dim3(256), 0, st, lin1, rows, time_dim, w2, b2,
                       time_dim, temb, 1, 0);
    CTSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(time_linear_kernel, dim3((total_out + 3) / 4), dim3(256), 0, st, temb, rows, time_dim, w_all,
                       b_all, total_out, tbias_out, 1, 0);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// ---- sampler updates ---------------------------------------------------------------------------------------------------
// coef row layout (8 floats), built on the host with fp32 torch ops exactly as the reference does:
//  DDIM: [0]=sqrt(1-a_t+1e-8) [1]=sqrt(a_t+1e-8)+1e-8 [2]=sqrt(a_prev+1e-8) [3]=sqrt(1-a_prev+1e-8) [4]=sigma_t
//  DDPM: [0]=sqrt(1-abar_t)   [1]=sqrt(abar_t)        [2]=post_mean_coef1   [3]=post_mean_coef2     [4]=(t!=0)*exp(0.5*logvar)
// z, eps: fp32 NDHWC.  noise: fp32 NCDHW (the layout torch.randn_like(z) has in the reference) or NULL.
// nonfinite: optional [steps][6] int32 table, row = *step_ptr: how many elements the reference's guards would have
// reported at this step -- {noise_pred NaN, Inf, z_0_pred NaN, Inf, z after update NaN, Inf} (inference/sampler.py:
// 288-292, 307-311, 331-334).  The guards themselves (nan_to_num) are applied unconditionally; the host reads the table
// once after the loop and logs what the reference logs, so the captured step never synchronises.
__device__ __forceinline__ void count_nonfinite(float v, int& n_nan, int& n_inf) {
    n_nan += (v != v) ? 1 : 0;
    n_inf += (v == __builtin_inff() || v == -__builtin_inff()) ? 1 : 0;
}

template <bool DDPM>
__global__ void __launch_bounds__(256)
sampler_step_kernel(float* __restrict__ z, const float* __restrict__ eps, const float* __restrict__ noise,
                    bf16_t* __restrict__ zin, int c_total, int c_off, const float* __restrict__ coef,
                    const int* __restrict__ step_ptr, int c, long long vox, long long total, int* __restrict__ nonfinite) {
    const int step = step_ptr ? *step_ptr : 0;
    const float* cf = coef + (long long)step * 8;
    const float c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3], c4 = cf[4];
    int cnt[6] = {0, 0, 0, 0, 0, 0};
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int ch = (int)(e % c);
        const long long nv = e / c;  // n*vox + v
        float ep = eps[e];
        const float zt = z[e];
        float zn;
        if (DDPM) {
            float z0 = (zt - c0 * ep) / c1;
            z0 = fminf(fmaxf(z0, -1.0f), 1.0f);
            zn = c2 * z0 + c3 * zt;
            if (noise) {
                const long long nb = nv / vox, v = nv - nb * vox;
                zn += c4 * noise[(nb * c + ch) * vox + v];
            }
        } else {
            count_nonfinite(ep, cnt[0], cnt[1]);
            ep = nan_to_num_f(ep);
            float z0 = (zt - c0 * ep) / c1;
            count_nonfinite(z0, cnt[2], cnt[3]);
            z0 = nan_to_num_f(z0);
            z0 = fminf(fmaxf(z0, -10.0f), 10.0f);
            zn = c2 * z0 + c3 * ep;
            if (noise) {
                const long long nb = nv / vox, v = nv - nb * vox;
                zn += c4 * noise[(nb * c + ch) * vox + v];
            }
            count_nonfinite(zn, cnt[4], cnt[5]);
            zn = nan_to_num_f(zn);
        }
        z[e] = zn;
        if (zin) zin[nv * c_total + c_off + ch] = f32_to_bf16(zn);
    }
    if (!DDPM && nonfinite != nullptr) {
        const int any = cnt[0] | cnt[1] | cnt[2] | cnt[3] | cnt[4] | cnt[5];
        if (__any(any != 0)) {   // never taken on healthy runs: no atomics, no divergence cost
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (cnt[k]) atomicAdd(&nonfinite[step * 6 + k], cnt[k]);
        }
    }
}

template <bool DDPM>
static int sampler_step(float* z, const float* eps, const float* noise, void* zin, int c_total, int c_off,
                        const float* coef, const int* step_ptr, int n, int c, int d, int h, int w, int* nonfinite,
                        void* stream) {
    CTSI_CHECK_ARG(z && eps && coef, "sampler step: null argument");
    CTSI_CHECK_ARG(!zin || (c_off >= 0 && c_off + c <= c_total), "sampler step: bad channel slice");
    const long long vox = (long long)d * h * w, total = vox * n * c;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((sampler_step_kernel<DDPM>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z,
                       eps, noise, (bf16_t*)zin, c_total, c_off, coef, step_ptr, c, vox, total, nonfinite);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

extern "C" int ctsi_ddim_step(float* z, const float* eps, const float* noise, void* zin, int c_total, int c_off,
                              const float* coef, const int* step_ptr, int n, int c, int d, int h, int w,
                              int* nonfinite, void* stream) {
    return sampler_step<false>(z, eps, noise, zin, c_total, c_off, coef, step_ptr, n, c, d, h, w, nonfinite, stream);
}
extern "C" int ctsi_ddpm_step(float* z, const float* eps, const float* noise, void* zin, int c_total, int c_off,
                              const float* coef, const int* step_ptr, int n, int c, int d, int h, int w,
                              void* stream) {
    return sampler_step<true>(z, eps, noise, zin, c_total, c_off, coef, step_ptr, n, c, d, h, w, nullptr, stream);
}

// ---- DDPM posterior with PER-SAMPLE timesteps (models/diffusion.py:249-338) --------------------------------------
// fp32 NCDHW in the API layout (what p_mean_variance / p_sample take and return).  Sample b reads coefficient row
// coef[b*8 ..]: {sqrt(1-abar_t), sqrt(abar_t), posterior_mean_coef1, posterior_mean_coef2, [t != 0] * exp(0.5 logvar)}.
//   z0   = (z - c0 * eps) / c1           (_predict_z_0_from_noise; clamped to [-1, 1] when clip != 0)
//   mean = c2 * z0 + c3 * z              (p_mean_variance)
//   out  = mean + c4 * noise             (p_sample; noise == NULL: out = mean)
__global__ void __launch_bounds__(256)
ddpm_posterior_kernel(const float* __restrict__ z, const float* __restrict__ eps, const float* __restrict__ noise,
                      float* __restrict__ z0_out, float* __restrict__ out, const float* __restrict__ coef,
                      long long per_sample, long long total, int clip) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const float* cf = coef + (e / per_sample) * 8;
        const float zt = z[e];
        float z0 = (zt - cf[0] * eps[e]) / cf[1];
        if (clip) z0 = fminf(fmaxf(z0, -1.0f), 1.0f);
        if (z0_out) z0_out[e] = z0;
        if (out) {
            float m = cf[2] * z0 + cf[3] * zt;
            if (noise) m += cf[4] * noise[e];
            out[e] = m;
        }
    }
}
extern "C" int ctsi_ddpm_posterior(const float* z, const float* eps, const float* noise, float* z0_out, float* out,
                                   const float* coef, int n, long long per_sample, int clip, void* stream) {
    CTSI_CHECK_ARG(z && eps && coef && (z0_out || out) && n > 0 && per_sample > 0, "ctsi_ddpm_posterior: bad arguments");
    const long long total = per_sample * n;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ddpm_posterior_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z, eps, noise,
                       z0_out, out, coef, per_sample, total, clip);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

__global__ void step_advance_kernel(int* step_ptr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *step_ptr += 1;
}
extern "C" int ctsi_step_advance(int* step_ptr, void* stream) {
    CTSI_CHECK_ARG(step_ptr, "ctsi_step_advance: null argument");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_ptr);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

__global__ void __launch_bounds__(256) nan_to_num_kernel(float* x, long long count, int sanitize, int* counts) {
    int n_nan = 0, n_inf = 0;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256) {
        const float v = x[e];
        count_nonfinite(v, n_nan, n_inf);
        if (sanitize) x[e] = nan_to_num_f(v);
    }
    if (counts != nullptr && __any((n_nan | n_inf) != 0)) {
        if (n_nan) atomicAdd(&counts[0], n_nan);
        if (n_inf) atomicAdd(&counts[1], n_inf);
    }
}
extern "C" int ctsi_nan_to_num_f32(float* x, long long count, void* stream) {
    CTSI_CHECK_ARG(x && count >= 0, "ctsi_nan_to_num_f32: bad arguments");
    if (count == 0) return CTSI_OK;
    long long blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nan_to_num_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, count, 1,
                       (int*)nullptr);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
// counts[0] += #NaN, counts[1] += #Inf of x (the reference's `torch.isnan(z).sum()` / `torch.isinf(z).sum()` of its
// NaN/Inf checkpoints, inference/sampler.py:268-275) without a host round trip; sanitize != 0 also applies nan_to_num.
extern "C" int ctsi_count_nonfinite_f32(float* x, long long count, int sanitize, int* counts, void* stream) {
    CTSI_CHECK_ARG(x && counts && count >= 0, "ctsi_count_nonfinite_f32: bad arguments");
    if (count == 0) return CTSI_OK;
    long long blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nan_to_num_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, count, sanitize,
                       counts);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// ---- sliding-window stitching (inference/sampler.py:157-170, 438-451) ------------------------------------------
// acc[.., d0+d, h0+h, w0+w] += patch[.., d, h, w] * wd[d]*wh[h]*ww[w];  wsum[...] += wd*wh*ww   (fp32 NCDHW)
// One launch per patch (patches are produced one after the other), 1-D separable Gaussian windows.
__global__ void __launch_bounds__(256)
blend_accumulate_kernel(float* __restrict__ acc, float* __restrict__ wsum, const float* __restrict__ patch,
                        const float* __restrict__ wd, const float* __restrict__ wh, const float* __restrict__ ww,
                        int nc, int pd, int ph, int pw, int D, int H, int W, int d0, int h0, int w0, long long total) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int w = (int)(e % pw);
        long long r = e / pw;
        const int h = (int)(r % ph);
        r /= ph;
        const int d = (int)(r % pd);
        const long long c = r / pd;
        const float g = wd[d] * wh[h] * ww[w];
        const long long o = ((c * D + d0 + d) * H + h0 + h) * W + w0 + w;
        acc[o] += patch[e] * g;
        wsum[o] += g;
    }
}
extern "C" int ctsi_blend_accumulate(float* acc, float* wsum, const float* patch, const float* wd, const float* wh,
                                     const float* ww, int nc, int pd, int ph, int pw, int d_full, int h_full,
                                     int w_full, int d0, int h0, int w0, void* stream) {
    CTSI_CHECK_ARG(acc && wsum && patch && wd && wh && ww, "ctsi_blend_accumulate: null argument");
    CTSI_CHECK_ARG(d0 >= 0 && h0 >= 0 && w0 >= 0 && d0 + pd <= d_full && h0 + ph <= h_full && w0 + pw <= w_full,
                   "ctsi_blend_accumulate: patch (%d,%d,%d)+(%d,%d,%d) outside the volume (%d,%d,%d)", d0, h0, w0, pd, ph,
                   pw, d_full, h_full, w_full);
    const long long total = (long long)nc * pd * ph * pw;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(blend_accumulate_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, acc, wsum,
                       patch, wd, wh, ww, nc, pd, ph, pw, d_full, h_full, w_full, d0, h0, w0, total);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

__global__ void __launch_bounds__(256) blend_normalize_kernel(float* acc, const float* wsum, long long count) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256)
        acc[e] = acc[e] / (wsum[e] + 1e-8f);
}
extern "C" int ctsi_blend_normalize(float* acc, const float* wsum, long long count, void* stream) {
    CTSI_CHECK_ARG(acc && wsum && count >= 0, "ctsi_blend_normalize: bad arguments");
    if (count == 0) return CTSI_OK;
    long long blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(blend_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, acc, wsum, count);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// ---- validation metrics (utils/metrics.py:14-193): per-slice MSE and 11x11 box-window SSIM on device ---------------
// a, b: fp32 (n, c, d, h, w).  SSIM as the reference writes it: means / second moments are avg_pool2d(window,
// stride 1, padding window/2, zeros counted) of a, b, a^2, b^2, ab; variances clamped at 0; map clamped to [0, 1].
// grid (tiles_w, tiles_h, n*c*d), block 16x16 outputs, (16+2R)^2 input tile of a and b in LDS (R <= 7).
// Per-block partials (sum of squared differences, sum of SSIM, NaN counts) go to a workspace; a second kernel adds
// them per slice in a fixed order.
#define MET_T 16
#define MET_RMAX 7
__global__ void __launch_bounds__(256)
slice_metrics_kernel(const float* __restrict__ a, const float* __restrict__ b, int d, int h, int w, int R, float c1,
                     float c2, double* __restrict__ partial) {
    __shared__ float s_a[MET_T + 2 * MET_RMAX][MET_T + 2 * MET_RMAX + 1];
    __shared__ float s_b[MET_T + 2 * MET_RMAX][MET_T + 2 * MET_RMAX + 1];
    __shared__ double s_red[4][256];
    const int plane = blockIdx.z;                    // (n*c + ch)*d + slice
    const int x0 = blockIdx.x * MET_T, y0 = blockIdx.y * MET_T;
    const float* pa = a + (long long)plane * h * w;
    const float* pb = b + (long long)plane * h * w;
    const int TS = MET_T + 2 * R;
    for (int e = threadIdx.x; e < TS * TS; e += 256) {
        const int ty = e / TS, tx = e - ty * TS;
        const int y = y0 + ty - R, x = x0 + tx - R;
        const bool in = y >= 0 && y < h && x >= 0 && x < w;
        s_a[ty][tx] = in ? pa[(long long)y * w + x] : 0.0f;
        s_b[ty][tx] = in ? pb[(long long)y * w + x] : 0.0f;
    }
    __syncthreads();
    const int lx = threadIdx.x % MET_T, ly = threadIdx.x / MET_T;
    const int x = x0 + lx, y = y0 + ly;
    double sq = 0.0, ss = 0.0, na = 0.0, nb = 0.0;
    if (x < w && y < h) {
        float sa = 0.0f, sb = 0.0f, saa = 0.0f, sbb = 0.0f, sab = 0.0f;
        for (int j = 0; j <= 2 * R; ++j)
            for (int i = 0; i <= 2 * R; ++i) {
                const float va = s_a[ly + j][lx + i], vb = s_b[ly + j][lx + i];
                sa += va; sb += vb; saa += va * va; sbb += vb * vb; sab += va * vb;
            }
        const float inv = 1.0f / (float)((2 * R + 1) * (2 * R + 1));
        const float mu1 = sa * inv, mu2 = sb * inv;
        float v1 = saa * inv - mu1 * mu1, v2 = sbb * inv - mu2 * mu2;
        const float v12 = sab * inv - mu1 * mu2;
        v1 = fmaxf(v1, 0.0f);
        v2 = fmaxf(v2, 0.0f);
        const float num = (2.0f * mu1 * mu2 + c1) * (2.0f * v12 + c2);
        const float den = (mu1 * mu1 + mu2 * mu2 + c1) * (v1 + v2 + c2) + 1e-8f;
        float s = num / den;
        s = fminf(fmaxf(s, 0.0f), 1.0f);     // NaN stays NaN through fminf/fmaxf? no: they drop NaN -> counted below
        const float ca = s_a[ly + R][lx + R], cb = s_b[ly + R][lx + R];
        const float df = ca - cb;
        sq = (double)df * (double)df;
        ss = (double)s;
        na = (ca != ca) ? 1.0 : 0.0;
        nb = (cb != cb) ? 1.0 : 0.0;
    }
    s_red[0][threadIdx.x] = sq; s_red[1][threadIdx.x] = ss; s_red[2][threadIdx.x] = na; s_red[3][threadIdx.x] = nb;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 4; ++k) s_red[k][threadIdx.x] += s_red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 4) {
        const long long blk = ((long long)plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[blk * 4 + threadIdx.x] = s_red[threadIdx.x][0];
    }
}
// out[slice][4] = (mean squared difference, mean SSIM, NaN count of a, NaN count of b) over (n, c, h, w)
__global__ void slice_metrics_final_kernel(const double* __restrict__ partial, int nc, int d, long long tiles, long long hw,
                                           double* __restrict__ out) {
    const int sl = blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= d) return;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int p = 0; p < nc; ++p) {
        const double* src = partial + ((long long)p * d + sl) * tiles * 4;
        for (long long t = 0; t < tiles; ++t)
            for (int k = 0; k < 4; ++k) acc[k] += src[t * 4 + k];
    }
    const double cnt = (double)nc * (double)hw;
    out[sl * 4 + 0] = acc[0] / cnt;
    out[sl * 4 + 1] = acc[1] / cnt;
    out[sl * 4 + 2] = acc[2];
    out[sl * 4 + 3] = acc[3];
}
extern "C" size_t ctsi_slice_metrics_workspace_doubles(int n, int c, int d, int h, int w) {
    const long long tiles = (long long)((h + MET_T - 1) / MET_T) * ((w + MET_T - 1) / MET_T);
    return (size_t)((long long)n * c * d * tiles * 4);
}
extern "C" int ctsi_slice_metrics(const float* a, const float* b, int n, int c, int d, int h, int w, int window,
                                  float max_val, double* workspace, double* out, void* stream) {
    CTSI_CHECK_ARG(a && b && workspace && out && n > 0 && c > 0 && d > 0 && h > 0 && w > 0, "ctsi_slice_metrics: bad arguments");
    CTSI_CHECK_ARG(window >= 1 && (window & 1) && window / 2 <= MET_RMAX, "ctsi_slice_metrics: window must be odd and <= %d",
                   2 * MET_RMAX + 1);
    const long long planes = (long long)n * c * d;
    CTSI_CHECK_ARG(planes <= 65535, "ctsi_slice_metrics: too many (n, c, d) planes for one launch (%lld)", planes);
    const dim3 grid((w + MET_T - 1) / MET_T, (h + MET_T - 1) / MET_T, (unsigned)planes);
    const float c1 = (0.01f * max_val) * (0.01f * max_val), c2 = (0.03f * max_val) * (0.03f * max_val);
    hipLaunchKernelGGL(slice_metrics_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, b, d, h, w, window / 2, c1, c2,
                       workspace);
    CTSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(slice_metrics_final_kernel, dim3((d + 63) / 64), dim3(64), 0, (hipStream_t)stream, workspace, n * c, d,
                       (long long)grid.x * grid.y, (long long)h * w, out);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
