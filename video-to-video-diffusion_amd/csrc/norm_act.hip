// GroupNorm statistics + fused normalise / SiLU / time-bias / residual kernels (HBM-bound).
// Replaces nn.GroupNorm + nn.SiLU + the ResBlock tails of the reference:
//   models/unet3d.py:59-60,70-74,95-98,116-133,151,167,328-330 and models/vae.py:28-35,44-56,69-76,90-97.
// All tensors are bf16 NDHWC; every global access is a 16-byte (8-channel) vector per lane.
// Statistics path: per-tile column sums (conv epilogue or gn_colsum_kernel) -> gn_finalize_kernel
// (fp64 (sum, sumsq) per (sample, group)) -> gn_apply_kernel.
#include "ctsi_internal.h"
#include <stdlib.h>

#define GN_TILE_ROWS 512

__device__ __forceinline__ void unpack8(const uint4 v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
    return v;
}

// ---- column sums of a tensor no conv produced ------------------------------------------------
// grid: (tiles_per_sample, n); block 256.  cpr = c/8 16-byte chunks per voxel.
__global__ void __launch_bounds__(256)
gn_colsum_kernel(const bf16_t* __restrict__ x, float* __restrict__ colsum, int n_total, int c,
                 long long vox_per_sample, int tps) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* s_red = reinterpret_cast<float*>(smem_raw);  // [rows_par][c][2]
    const int cpr = c >> 3;
    const int rows_par = 256 / cpr;
    const int tid = threadIdx.x;
    const int q = tid % cpr, rl = tid / cpr;
    const int tile = blockIdx.x, nb = blockIdx.y;
    const long long v0 = (long long)tile * GN_TILE_ROWS;
    long long v1 = v0 + GN_TILE_ROWS;
    if (v1 > vox_per_sample) v1 = vox_per_sample;
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.0f;
    if (rl < rows_par) {
        const bf16_t* base = x + ((long long)nb * vox_per_sample) * c + q * 8;
        for (long long v = v0 + rl; v < v1; v += rows_par) {
            const uint4 raw = *reinterpret_cast<const uint4*>(base + v * c);
            float f[8];
            unpack8(raw, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s1[k] += f[k];
                s2[k] += f[k] * f[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            s_red[((rl * c) + q * 8 + k) * 2 + 0] = s1[k];
            s_red[((rl * c) + q * 8 + k) * 2 + 1] = s2[k];
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

extern "C" int ctsi_gn_colsum_tiles(int d, int h, int w) {
    const long long vox = (long long)d * h * w;
    return (int)((vox + GN_TILE_ROWS - 1) / GN_TILE_ROWS);
}

extern "C" int ctsi_gn_colsum(const void* x, float* colsum, int n, int c, int d, int h, int w,
                              int* tiles_per_sample, void* stream) {
    CTSI_CHECK_ARG(x && colsum, "ctsi_gn_colsum: null argument");
    CTSI_CHECK_ARG(c % 8 == 0 && c >= 8 && c <= 2048, "ctsi_gn_colsum: c=%d must be a multiple of 8 in [8,2048]", c);
    const int tps = ctsi_gn_colsum_tiles(d, h, w);
    if (tiles_per_sample) *tiles_per_sample = tps;
    const int cpr = c / 8, rows_par = 256 / cpr;
    const size_t lds = (size_t)rows_par * c * 2 * sizeof(float);
    hipLaunchKernelGGL(gn_colsum_kernel, dim3(tps, n), dim3(256), lds, (hipStream_t)stream,
                       (const bf16_t*)x, colsum, n, c, (long long)d * h * w, tps);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// ---- finalize: column-sum slab -> (sum, sumsq) per (sample, group), fp64 -----------------------
// grid: (groups, n); block 1024.  ONE block owns one (sample, group): every thread sums a fixed, strided subset of the
// tile partials in a fixed order, the block combines them with a fixed-shape tree, and the result is WRITTEN (not
// accumulated) -- the statistics are bit-identical from run to run by construction (no atomics, no dependence on block
// scheduling), and the sums buffer needs no zeroing.
// Split form (gridDim.z = S > 1: tensors with tens of thousands of tiles -- the VAE at full resolution: 24576 tiles, 3 MB of
// partials per group, 71 us per launch when ONE block read them): slice s of a (sample, group) sums items
// [s * per, (s + 1) * per) the same way, parks its pair in a scratch row (agent-scope stores) and takes a ticket; the block that
// draws the last ticket adds the S pairs IN SLICE ORDER and writes the result: still no dependence on block scheduling.
// The scratch rows and tickets are process-global and the slot rotates per call: launches that could run CONCURRENTLY on one
// device (two streams, two host threads, two graphs replayed side by side) must not both take the split form -- the engine
// issues everything of a device on ONE stream (Ctx), which is what makes this safe; a caller that cannot promise that sets
// CTSI_GN_FIN_SPLIT=0.
#define GN_FIN_ROWS 2048
#define GN_FIN_SMAX 64
#define GN_FIN_SLOTS 4
__device__ double g_fin_part[GN_FIN_SLOTS][GN_FIN_ROWS][GN_FIN_SMAX][2];
__device__ unsigned int g_fin_ticket[GN_FIN_SLOTS][GN_FIN_ROWS];

__global__ void __launch_bounds__(1024)
gn_finalize_kernel(const float* __restrict__ colsum, double* __restrict__ sums, int n_total, int c_pad,
                   int groups, int cpg, int tps, int nclass, int accumulate, int slot) {
    const int g = blockIdx.x, nb = blockIdx.y;
    const int tid = threadIdx.x;
    const int S = gridDim.z, sl = blockIdx.z;
    const long long slab = (long long)nclass * n_total * tps * c_pad;
    const int items = tps * cpg;
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
    if ((cpg & 3) == 0 && (c_pad & 3) == 0) {
        // 16-byte loads, 8 of them in flight per thread and statistic: a group's cpg channels of one tile row are contiguous.
        // Large tensors (the VAE decoder at full resolution: 24576 tiles per sample) are latency-, not bandwidth-bound here:
        // the scalar form below took 30-140 us per launch on them (15 launches per decode), this one a few us.
        const int q4 = cpg >> 2, items4_all = tps * q4;
        const int per = S > 1 ? ((items4_all + S - 1) / S + 1023) / 1024 * 1024 : items4_all;   // (whole thread rounds per slice)
        const int it0 = sl * per;
        const int items4 = it0 + per < items4_all ? it0 + per : items4_all;
        const float4* cs1 = reinterpret_cast<const float4*>(colsum);
        const float4* cs2 = reinterpret_cast<const float4*>(colsum + slab);
        for (int cls = 0; cls < nclass; ++cls) {
            const long long tbase = ((long long)cls * n_total + nb) * tps;
            int it = it0 + tid;
            for (; it + 7 * 1024 < items4; it += 8 * 1024) {
                float4 v1[8], v2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i2 = it + u * 1024;
                    const long long idx = (((tbase + i2 / q4) * c_pad + g * cpg) >> 2) + i2 % q4;
                    v1[u] = cs1[idx];
                    v2[u] = cs2[idx];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a1[u & 3] += ((double)v1[u].x + (double)v1[u].y) + ((double)v1[u].z + (double)v1[u].w);
                    a2[u & 3] += ((double)v2[u].x + (double)v2[u].y) + ((double)v2[u].z + (double)v2[u].w);
                }
            }
            for (int u = 0; it < items4; it += 1024, ++u) {
                const long long idx = (((tbase + it / q4) * c_pad + g * cpg) >> 2) + it % q4;
                const float4 w1 = cs1[idx], w2 = cs2[idx];
                a1[u & 3] += ((double)w1.x + (double)w1.y) + ((double)w1.z + (double)w1.w);
                a2[u & 3] += ((double)w2.x + (double)w2.y) + ((double)w2.z + (double)w2.w);
            }
        }
    } else
    for (int cls = 0; cls < nclass; ++cls) {
        const long long tbase = ((long long)cls * n_total + nb) * tps;
        int it = tid;
        for (; it + 3 * 1024 < items; it += 4 * 1024) {   // four independent loads in flight per statistic
            float v1[4], v2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i2 = it + u * 1024;
                const long long idx = (tbase + i2 / cpg) * c_pad + g * cpg + i2 % cpg;
                v1[u] = colsum[idx];
                v2[u] = colsum[slab + idx];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a1[u] += (double)v1[u];
                a2[u] += (double)v2[u];
            }
        }
        for (int u = 0; it < items; it += 1024, ++u) {
            const long long idx = (tbase + it / cpg) * c_pad + g * cpg + it % cpg;
            a1[u] += (double)colsum[idx];
            a2[u] += (double)colsum[slab + idx];
        }
    }
    // fixed-shape reduction: butterfly inside each wave (6 steps, no barrier), then the 16 wave partials in wave order
    double r1 = (a1[0] + a1[1]) + (a1[2] + a1[3]), r2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        r1 += __shfl_xor(r1, off);
        r2 += __shfl_xor(r2, off);
    }
    __shared__ double s1[16], s2[16];
    if ((tid & 63) == 0) {
        s1[tid >> 6] = r1;
        s2[tid >> 6] = r2;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < 16; ++w) {
            s1[0] += s1[w];
            s2[0] += s2[w];
        }
    }
    if (tid == 0) {
        double t1 = s1[0], t2 = s2[0];
        bool last = true;
        if (S > 1) {
            const int row = nb * groups + g;
            double* part = &g_fin_part[slot][row][0][0];
            __hip_atomic_store(part + 2 * sl + 0, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part + 2 * sl + 1, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned tk = __hip_atomic_fetch_add(&g_fin_ticket[slot][row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = tk == (unsigned)(S - 1);
            if (last) {
                __hip_atomic_store(&g_fin_ticket[slot][row], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                t1 = 0.0;
                t2 = 0.0;
                for (int k = 0; k < S; ++k) {
                    t1 += __hip_atomic_load(part + 2 * k + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    t2 += __hip_atomic_load(part + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (last) {
            double* o = sums + ((long long)nb * groups + g) * 2;
            o[0] = (accumulate ? o[0] : 0.0) + t1;
            o[1] = (accumulate ? o[1] : 0.0) + t2;
        }
    }
}

extern "C" int ctsi_gn_finalize(const float* colsum, double* sums, int n, int c, int c_pad, int groups,
                                int tiles_per_sample, int nclass, int accumulate, void* stream) {
    CTSI_CHECK_ARG(colsum && sums, "ctsi_gn_finalize: null argument");
    CTSI_CHECK_ARG(groups > 0 && c % groups == 0, "ctsi_gn_finalize: c=%d not divisible by groups=%d", c, groups);
    const int cpg = c / groups;
    // slices: one per 2048 16-byte items of a (sample, group) (the vector path only), at most 64; scratch rows rotate over 4
    // slots per call so that consecutive launches never share one
    int S = 1;
    static int call_no = 0;
    if ((cpg & 3) == 0 && (c_pad & 3) == 0 && nclass == 1 && (long long)n * groups <= GN_FIN_ROWS) {
        const long long items4 = (long long)tiles_per_sample * (cpg >> 2);
        S = (int)(items4 / 2048);       // (round 4: 2048 instead of 8192 items per slice, up to 64 slices: the VAE's 8-group norms gave
                                        //  96 blocks of 256 KB each at full resolution -- 27 us per launch, latency-bound)
        if (S > GN_FIN_SMAX) S = GN_FIN_SMAX;
        if (S < 1) S = 1;
        static const char* fs = getenv("CTSI_GN_FIN_SPLIT");
        if (fs && atoi(fs) == 0) S = 1;
    }
    const int slot = S > 1 ? (call_no++ & (GN_FIN_SLOTS - 1)) : 0;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, n, S), dim3(1024), 0, (hipStream_t)stream, colsum, sums, n, c_pad,
                       groups, cpg, tiles_per_sample, nclass, accumulate, slot);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// 16-byte accesses, optionally non-temporal (tensors far larger than the Infinity Cache: streamed once)
typedef unsigned int gn_u4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 gn_ld(const bf16_t* p) {
    if (NT) {
        const gn_u4_t v = __builtin_nontemporal_load(reinterpret_cast<const gn_u4_t*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const uint4*>(p);
}
template <bool NT>
__device__ __forceinline__ void gn_st(bf16_t* p, const uint4 v) {
    if (NT) {
        const gn_u4_t w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<gn_u4_t*>(p));
    } else {
        *reinterpret_cast<uint4*>(p) = v;
    }
}

// ---- apply --------------------------------------------------------------------------------------
// grid: (blocks_per_sample, n); block 256.  The four options are template parameters: with run-time flags the compiler
// if-converted both SiLU sites into unconditional code (2 exp + 2 rcp per element and a select chain: ~25 VALU instructions
// per element, 2.3 TB/s each way -- VALU-bound, not HBM-bound).  CONSTQ: the grid stride is a multiple of the row's chunk
// count, so a thread meets the same 8 channels every iteration and keeps their scale / shift / time bias in registers.
template <bool SILU_PRE, bool TB, bool RES, bool SILU_POST, bool CONSTQ, bool NT>
__global__ void __launch_bounds__(256)
gn_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, const double* __restrict__ sums,
                const float* __restrict__ gamma, const float* __restrict__ beta, int c, long long vox,
                long long vox_stat, int groups, float eps, const float* __restrict__ tbias, int tbias_stride,
                const int* __restrict__ step_ptr, int n_total, const bf16_t* __restrict__ residual, long long per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* s_scale = reinterpret_cast<float*>(smem_raw);
    float* s_shift = s_scale + c;
    float* s_tb = s_shift + c;
    const int nb = blockIdx.y, tid = threadIdx.x;
    const int cpg = c / groups;
    const double cnt = (double)cpg * (double)vox_stat;
    long long trow = nb;
    if (TB && step_ptr) trow += (long long)(*step_ptr) * n_total;
    for (int ch = tid; ch < c; ch += 256) {
        const int g = ch / cpg;
        const double m = sums[((long long)nb * groups + g) * 2 + 0] / cnt;
        double var = sums[((long long)nb * groups + g) * 2 + 1] / cnt - m * m;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[ch] * rstd;
        s_scale[ch] = sc;
        s_shift[ch] = beta[ch] - (float)m * sc;
        if (TB) s_tb[ch] = tbias[trow * tbias_stride + ch];
    }
    __syncthreads();
    const int cpr = c >> 3;
    // per_block > 0: every block walks its own contiguous range of per_block chunks (stride 256); 0: grid-stride
    long long total = vox * cpr;
    const bf16_t* xb = x + (long long)nb * vox * c;
    bf16_t* yb = y + (long long)nb * vox * c;
    const bf16_t* rb = RES ? residual + (long long)nb * vox * c : nullptr;
    const long long stride = per_block > 0 ? 256 : (long long)gridDim.x * 256;
    const int dq = CONSTQ ? 0 : (int)(stride % cpr);
    constexpr int U = 4;   // independent 16-byte loads in flight per thread
    long long e = (per_block > 0 ? (long long)blockIdx.x * per_block : (long long)blockIdx.x * 256) + tid;
    if (per_block > 0 && (long long)(blockIdx.x + 1) * per_block < total) total = (long long)(blockIdx.x + 1) * per_block;
    int q = (int)(e % cpr);   // 8-channel chunk index inside the voxel row
    float csc[8], csh[8], ctb[8];
    auto coef = [&](int qq, float* sc, float* sh, float* tb) {   // 16-byte LDS reads
        *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(s_scale + qq * 8);
        *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(s_scale + qq * 8 + 4);
        *reinterpret_cast<float4*>(sh) = *reinterpret_cast<const float4*>(s_shift + qq * 8);
        *reinterpret_cast<float4*>(sh + 4) = *reinterpret_cast<const float4*>(s_shift + qq * 8 + 4);
        if (TB) {
            *reinterpret_cast<float4*>(tb) = *reinterpret_cast<const float4*>(s_tb + qq * 8);
            *reinterpret_cast<float4*>(tb + 4) = *reinterpret_cast<const float4*>(s_tb + qq * 8 + 4);
        }
    };
    if (CONSTQ) coef(q, csc, csh, ctb);
    // one dword = channels (2 j, 2 j + 1): the pair stays a 2-vector from the unpack to the single v_cvt_pk_bf16_f32
    auto one = [&](const uint4 raw, const uint4 rraw, long long ee, int qq) {
        float lsc[8], lsh[8], ltb[8];
        if (!CONSTQ) coef(qq, lsc, lsh, ltb);
        const float* sc = CONSTQ ? csc : lsc;
        const float* sh = CONSTQ ? csh : lsh;
        const float* tb = CONSTQ ? ctb : ltb;
        const uint32_t xin[4] = {raw.x, raw.y, raw.z, raw.w}, rin[4] = {rraw.x, rraw.y, rraw.z, rraw.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2_t v = {__uint_as_float(xin[j] << 16), __uint_as_float(xin[j] & 0xffff0000u)};
            const f32x2_t s2 = {sc[2 * j], sc[2 * j + 1]}, h2 = {sh[2 * j], sh[2 * j + 1]};
            v = v * s2 + h2;
            if (SILU_PRE) v = silu2_f(v);
            if (TB) v += f32x2_t{tb[2 * j], tb[2 * j + 1]};
            if (RES) v += f32x2_t{__uint_as_float(rin[j] << 16), __uint_as_float(rin[j] & 0xffff0000u)};
            if (SILU_POST) v = silu2_f(v);
            o[j] = pack_bf16x2_v(v);
        }
        gn_st<NT>(yb + ee * 8, make_uint4(o[0], o[1], o[2], o[3]));
    };
    for (; e + (U - 1) * stride < total; e += U * stride) {
        uint4 raw[U], rraw[U];
        int qs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            raw[u] = gn_ld<NT>(xb + (e + u * stride) * 8);
            rraw[u] = RES ? gn_ld<NT>(rb + (e + u * stride) * 8) : make_uint4(0, 0, 0, 0);
            qs[u] = q;
            if (!CONSTQ) q = (q + dq >= cpr) ? q + dq - cpr : q + dq;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) one(raw[u], rraw[u], e + u * stride, qs[u]);
    }
    for (; e < total; e += stride) {
        const uint4 raw = gn_ld<NT>(xb + e * 8);
        const uint4 rraw = RES ? gn_ld<NT>(rb + e * 8) : make_uint4(0, 0, 0, 0);
        one(raw, rraw, e, q);
        if (!CONSTQ) q = (q + dq >= cpr) ? q + dq - cpr : q + dq;
    }
}

typedef void (*gn_apply_fn)(const bf16_t*, bf16_t*, const double*, const float*, const float*, int, long long, long long, int,
                            float, const float*, int, const int*, int, const bf16_t*, long long);
template <int I>
static gn_apply_fn gn_apply_pick(int idx) {
    if constexpr (I >= 64) {
        return nullptr;
    } else {
        if (idx == I) return gn_apply_kernel<(I & 1) != 0, (I & 2) != 0, (I & 4) != 0, (I & 8) != 0, (I & 16) != 0, (I & 32) != 0>;
        return gn_apply_pick<I + 1>(idx);
    }
}

extern "C" int ctsi_gn_apply(const void* x, void* y, const double* sums, const float* gamma, const float* beta,
                             int n, int c, int d, int h, int w, int d_stat, int groups, float eps, int silu_pre,
                             const float* tbias, int tbias_stride, const int* step_ptr,
                             const void* residual, int silu_post, void* stream) {
    CTSI_CHECK_ARG(x && y && sums && gamma && beta, "ctsi_gn_apply: null argument");
    CTSI_CHECK_ARG(c % 8 == 0 && groups > 0 && c % groups == 0, "ctsi_gn_apply: bad c=%d groups=%d", c, groups);
    CTSI_CHECK_ARG(d_stat >= d, "ctsi_gn_apply: d_stat=%d < d=%d", d_stat, d);
    const long long vox = (long long)d * h * w;
    const long long total = vox * (c / 8);
    long long blocks = (total + 256 * 8 - 1) / (256 * 8);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    // a grid stride that is a multiple of the chunks per voxel row (c / 8) keeps every thread on one 8-channel chunk
    const int cpr = c / 8;
    int gq = cpr, g256 = 256;
    while (g256) { const int t = gq % g256; gq = g256; g256 = t; }     // gq = gcd(cpr, 256)
    const int need = cpr / gq;                                         // blocks must be a multiple of this
    if (blocks >= need) blocks -= blocks % need;
    bool constq = (blocks * 256) % cpr == 0;
    // Contiguous ranges: every block takes ONE batch of 4 x 256 consecutive 16-byte chunks (16 KB) and the blocks are dispatched in
    // address order, so the chip's working front moves linearly through the tensor.  Against 2048 grid-striding blocks (each
    // thread's loads 8 MB apart) the decoder-size tensors go from 4.7-4.8 to 6.0-6.3 TB/s (3.2 GB at 128 channels x 48 x 512^2: 1345 ->
    // 1069 us; with a residual 2044 -> 1554 us), the U-Net's 201 MB tensors from 6.5 to 6.7 TB/s.  Needs 256 % (c / 8) == 0 (a
    // thread then stays on one 8-channel chunk); other channel counts keep the grid-stride form.  CTSI_GN_CONTIG=0 forces it.
    long long per_block = 0;
    {
        static const char* cg = getenv("CTSI_GN_CONTIG");
        if (!(cg && atoi(cg) == 0) && 256 % cpr == 0) {
            per_block = 1024;
            blocks = (total + per_block - 1) / per_block;
            constq = true;
        }
    }
    const size_t lds = (size_t)c * 3 * sizeof(float);
    // tensors well beyond the 256 MB Infinity Cache (> 512 MB: measured threshold) are streamed with non-temporal accesses (CTSI_GN_NT=0 / 1 overrides)
    static const char* nt_env = getenv("CTSI_GN_NT");
    const bool nt = nt_env ? atoi(nt_env) != 0 : (long long)n * vox * c * 2 > (512ll << 20);
    const int idx = (silu_pre ? 1 : 0) | (tbias ? 2 : 0) | (residual ? 4 : 0) | (silu_post ? 8 : 0) | (constq ? 16 : 0) | (nt ? 32 : 0);
    hipLaunchKernelGGL(gn_apply_pick<0>(idx), dim3((unsigned)blocks, n), dim3(256), lds, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)y, sums, gamma, beta, c, vox, (long long)d_stat * h * w, groups, eps,
                       tbias, tbias_stride, step_ptr, n, (const bf16_t*)residual, per_block);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
