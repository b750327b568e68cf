/*
 * ctsi.h — C ABI of the MI355X (gfx950) CT slice-interpolation engine.
 *
 * This is the drop-in boundary underneath the reference's Python module surface
 * (models.* / inference.*).  The reference has no FFI of its own: every entry point
 * below replaces a PyTorch op family that the reference calls on its hot path, cited
 * as `reference file:line`.  The Python host (video-to-video-diffusion_amd/) binds
 * these with ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *  - extern "C", plain pointers / sizes / ints, no torch or C++ types.
 *  - every function returns 0 on success, <0 on error; ctsi_last_error() returns a
 *    thread-local message.  Nothing throws.
 *  - no entry point allocates device memory, synchronises the stream or the device:
 *    the caller passes all buffers (query sizes with the *_bytes/_floats helpers).
 *    All launches go to the `stream` argument (a hipStream_t passed as void*), so a
 *    caller may capture any sequence of calls into a hipGraph (ctsi_graph_*).
 *  - activations inside the engine are bf16, channels-last ("NDHWC": n, d, h, w, c with
 *    c fastest).  The API boundary tensors of the reference are fp32 NCDHW; the two
 *    conversion entry points sit at that boundary.  Accumulation, GroupNorm statistics,
 *    the time embedding and the DDIM/DDPM state are fp32.
 */
#ifndef CTSI_H
#define CTSI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTSI_OK 0
#define CTSI_ERR_INVALID (-1)
#define CTSI_ERR_UNSUPPORTED (-2)
#define CTSI_ERR_HIP (-3)

/* library / error handling ------------------------------------------------------------ */
int ctsi_version(void);
/* 1 for a build with the timing-only ablation switches compiled in (`make ablate`: libctsi_ablate.so, for tools/ only:
 * CTSI_DEBUG_FLAGS / CTSI_DEBUG_KSTEPS can make its kernels skip work); 0 for the release library, which ignores them. */
int ctsi_ablation_build(void);
const char* ctsi_last_error(void);
/* 1 when a HIP device is visible to this process, 0 otherwise (never raises). */
int ctsi_device_available(void);

/* boundary layout conversion ---------------------------------------------------------- *
 * fp32 NCDHW (reference API tensors, models/model.py:252-259) <-> bf16 NDHWC (engine).
 * `c_total`/`c_off` let the caller write a channel slice of a wider NDHWC tensor
 * (used for torch.cat([x, c], dim=1), models/unet3d.py:372).                            */
int ctsi_ncdhw_f32_to_ndhwc_bf16(const float* src, void* dst, int n, int c, int d, int h, int w,
                                 int c_total, int c_off, void* stream);
int ctsi_ndhwc_bf16_to_ncdhw_f32(const void* src, float* dst, int n, int c, int d, int h, int w,
                                 void* stream);
/* fp32 NDHWC <-> fp32 NCDHW (sampler state z lives as fp32 NDHWC inside the engine). */
int ctsi_ncdhw_f32_to_ndhwc_f32(const float* src, float* dst, int n, int c, int d, int h, int w,
                                void* stream);
int ctsi_ndhwc_f32_to_ncdhw_f32(const float* src, float* dst, int n, int c, int d, int h, int w,
                                void* stream);

/* convolution family ------------------------------------------------------------------ *
 * One gather-GEMM MFMA kernel serves nn.Conv3d k=3 p=1 (models/unet3d.py:56,96,257,331;
 * models/vae.py:27,45,134,188), k=1 (unet3d.py:102,152-153; vae.py:137,161),
 * k=(3,4,4) s=(1,2,2) p=1 (unet3d.py:204-207; vae.py:65-68) and
 * nn.ConvTranspose3d k=(3,4,4) s=(1,2,2) p=1 (unet3d.py:218-221; vae.py:86-89).
 *
 * A plan is a host-only opaque object (no device memory) describing one layer at one
 * input shape.  The input may be the channel concatenation of two NDHWC tensors
 * (c1 + c2 channels; c2 = 0 for a single source) so torch.cat along channels
 * (unet3d.py:372, 401) is never materialised.                                           */
typedef struct ctsi_conv_plan ctsi_conv_plan;

typedef struct ctsi_conv_desc {
    int transposed;      /* 0: Conv3d, 1: ConvTranspose3d                                   */
    int kd, kh, kw;      /* kernel size                                                     */
    int sh, sw;          /* stride in H, W (depth stride is always 1 on this path)         */
    int pd, ph, pw;      /* padding                                                         */
    int n, c1, c2;       /* batch, channels of source 1 and source 2 (c2 may be 0)        */
    int cout;
    int di, hi, wi;      /* input spatial size                                             */
    int halo_d;          /* 1: the input tensor carries one halo slice below and above its own
                            di-2 slices (depth-sharded volume); outputs cover the own slices only */
} ctsi_conv_desc;

/* epilogue description for ctsi_conv_fwd */
typedef struct ctsi_conv_out {
    void* y;             /* output tensor                                                   */
    int mode;            /* 0: bf16 NDHWC, channel stride = cout_stride, offset c_off      */
                         /* 1: fp32 with explicit element strides (sn, sc, sd, sh, sw)     */
    int cout_stride;     /* mode 0: channels per voxel of y (>= c_off + cout)              */
    int c_off;           /* mode 0: first channel written                                   */
    long long sn, sc, sd, sh, sw; /* mode 1 strides, in elements                           */
    int act;             /* 0: none, 1: tanh (models/vae.py:203)                           */
    float* colsum;       /* optional: per-tile column sums for a following GroupNorm
                            ([2][ctsi_conv_plan_tiles()][cout_pad] floats), or NULL       */
    /* optional fused ResBlock tail (models/unet3d.py:130-133, `self.activation(h + self.residual_conv(x))` with h =
     * GroupNorm(conv2 output)): when gn_x != NULL the conv result r is not stored as is but
     *     y = silu?( gn(gn_x) * gamma + beta + r ),   gn statistics from gn_sums (as ctsi_gn_apply reads them),
     * so the residual tensor never goes to HBM.  mode 0 only, no colsum, no act; gn_x has y's shape and channel
     * stride and may alias y (in place).  The gather-GEMM kernel implements it (1x1x1 residual convs).           */
    const void* gn_x;        /* bf16 NDHWC, channel stride cout_stride, offset c_off                              */
    const double* gn_sums;   /* [n][groups][2] fp64 (sum, sumsq) of gn_x                                          */
    const float* gn_gamma;
    const float* gn_beta;
    int gn_groups;
    float gn_eps;
    long long gn_count;      /* elements per (sample, group) the statistics were taken over                       */
    int gn_silu;             /* 1: SiLU after the add                                                             */
    /* split-K plans (ctsi_conv_plan_workspace_bytes() > 0): device scratch of that size that belongs to THIS layer; zero it
     * once before the first launch (it holds the hand-off tickets), the kernel leaves it ready for the next launch      */
    void* workspace;
} ctsi_conv_out;

int ctsi_conv_plan_create(ctsi_conv_plan** plan, const ctsi_conv_desc* desc);
/* the weight tensor carries only `cin_w` (< c1+c2) input channels; the remaining activation
 * channels are layout padding (the 1-channel CT volume is stored with 8 channels).       */
int ctsi_conv_plan_set_weight_cin(ctsi_conv_plan* plan, int cin_w);
/* A 1x1x1 stride-1 layer that will run with the fused GroupNorm tail (ctsi_conv_out.gn_x: the ResBlock tails of
 * models/unet3d.py:102, 112-133) or as a plain bf16 conv + bias may take the streaming kernel (csrc/conv1_stream.hip: weights
 * resident in LDS, every voxel row read once straight into the MFMA layout).  Call it BEFORE ctsi_conv_plan_weight_bytes /
 * _pack_weights: the packed image differs.  on = 1 selects it where the layer qualifies (every source a multiple of 128
 * channels, K = 128 .. 512, 768 or 1024, cout in whole n-tiles) and is a no-op otherwise; ctsi_conv_plan_config reports
 * mode 10 when it is active.  Such a plan emits no column sums and no activation.                                  */
int ctsi_conv_plan_set_stream_tail(ctsi_conv_plan* plan, int on);
void ctsi_conv_plan_destroy(ctsi_conv_plan* plan);
/* output spatial size of the layer */
int ctsi_conv_plan_out_dims(const ctsi_conv_plan* plan, int* d_out, int* h_out, int* w_out);
/* bytes of the packed bf16 weight image this plan consumes */
size_t ctsi_conv_plan_weight_bytes(const ctsi_conv_plan* plan);
/* number of row tiles (all samples, all parity classes) and padded cout: sizes the colsum slab */
int ctsi_conv_plan_tiles(const ctsi_conv_plan* plan);
int ctsi_conv_plan_tiles_per_sample(const ctsi_conv_plan* plan);
int ctsi_conv_plan_cout_pad(const ctsi_conv_plan* plan);
/* bytes of ctsi_conv_out.workspace this plan needs (0 for most plans) */
size_t ctsi_conv_plan_workspace_bytes(const ctsi_conv_plan* plan);
/* algorithmic FLOPs (2*MAC, dense direct convolution) of one forward of this layer */
double ctsi_conv_plan_flops(const ctsi_conv_plan* plan);
/* which kernel variant the plan launches: MFMA tile (bm x bn) and staging mode
 * (0: general gather, 1: small-cin tap-packed K, 2: buffer-addressed whole-chunk gather)      */
int ctsi_conv_plan_config(const ctsi_conv_plan* plan, int* bm, int* bn, int* mode);
/* re-layout reference weights (fp32, PyTorch layout: Conv3d (cout,cin,kd,kh,kw),
 * ConvTranspose3d (cin,cout,kd,kh,kw)) into the kernel's bf16 [class][cout_pad][K] image. */
int ctsi_conv_plan_pack_weights(const ctsi_conv_plan* plan, const float* w_f32, void* packed,
                                void* stream);
/* y = conv(cat(x1,x2)) + bias, optional activation, optional GroupNorm column sums. */
int ctsi_conv_fwd(const ctsi_conv_plan* plan, const void* x1, const void* x2, const void* packed_w,
                  const float* bias, const ctsi_conv_out* out, void* stream);

/* GroupNorm (+SiLU, +time bias, +residual) ----------------------------------------------- *
 * nn.GroupNorm(eps=1e-5, affine) + SiLU + the ResBlock tail of models/unet3d.py:70-74,
 * 116-133 and models/vae.py:31-35, 50-56, 72-76, 93-97.
 * Statistics are two-stage: per-tile column sums (from the conv epilogue, or from
 * ctsi_gn_colsum for tensors that no conv produced) -> ctsi_gn_finalize -> (sum, sumsq)
 * per (sample, group) in fp64 -> ctsi_gn_apply.                                          */
int ctsi_gn_colsum(const void* x_bf16, float* colsum, int n, int c, int d, int h, int w,
                   int* tiles_per_sample, void* stream);
int ctsi_gn_colsum_tiles(int d, int h, int w);
/* sums[n][g][2] (double) = sum over tiles/columns: WRITTEN, not accumulated (no zeroing needed), by one block per
 * (sample, group) with a fixed-order reduce -- bit-identical from run to run (no atomics).
 * nclass > 1: tiles of class k of sample i start at (k*n + i)*tiles_per_sample.          */
/* accumulate != 0: sums += instead of sums = (a tensor produced by several conv launches -- the interior / boundary split
 * of a depth-sharded conv -- is finalised slab by slab; still deterministic: the launches are ordered on the stream). */
int ctsi_gn_finalize(const float* colsum, double* sums, int n, int c, int c_pad, int groups,
                     int tiles_per_sample, int nclass, int accumulate, void* stream);
/* y = [silu]( gn(x)*gamma+beta ) [+ tbias[n][c]] [+ residual] ; [silu] again if silu_post */
/* tbias row used for sample i: (step_ptr ? *step_ptr : 0) * n + i  (row length tbias_stride).
 * d_stat: depth the statistics in `sums` were accumulated over (== d on one GPU; the whole volume's
 * depth when the tensor is one depth slab of a volume sharded across GPUs and `sums` was all-reduced). */
int ctsi_gn_apply(const void* x_bf16, void* y_bf16, const double* sums, const float* gamma,
                  const float* beta, int n, int c, int d, int h, int w, int d_stat, int groups, float eps,
                  int silu_pre, const float* tbias, int tbias_stride, const int* step_ptr,
                  const void* residual_bf16, int silu_post, void* stream);

/* TemporalAttention, fused middle: P[n][pos][co] = bias[co] + sum_ci W[co][ci] * (gamma rstd (S - D mean) + D beta)[n][pos][ci]
 * -- ctsi_attn_normsum and the folded (proj_out . V-projection) 1x1x1 conv (models/unet3d.py:152-153, 181-192) in one
 * launch.  depthsum / sums / gamma / beta as for ctsi_attn_normsum; w_bf16 = bf16 [c][c] row-major (cout, cin); bias fp32 [c];
 * out bf16 [n][h*w][c].  ctsi_attn_pv_supported(c, groups) != 0 where the kernel applies (c in {128,256,512,1024}, 8 | c / groups);
 * other shapes use ctsi_attn_normsum + a 1x1x1 ctsi_conv_fwd. */
int ctsi_attn_pv_supported(int c, int groups);
int ctsi_attn_pv(const float* depthsum, const double* sums, const float* gamma, const float* beta, const void* w_bf16,
                 const float* bias, void* out, int n, int c, int d, int h, int w, int groups, float eps, void* stream);

/* TemporalAttention (models/unet3d.py:136-194) --------------------------------------------- *
 * The reference's second einsum 'bhqk,bhvc->bhqc' contracts k and v independently, so the
 * module equals proj_out(rowsum(softmax) * sum_t V_t) + x with rowsum(softmax) == 1.
 * mode 0 (fast) uses that identity; mode 1 (exact) additionally evaluates
 * rowsum(softmax(q k^T * hd^-0.5)) in fp32 per (position, head, query) and multiplies by it. */
/* pass 1: GroupNorm column sums of x and depth sums S[n][h][w][c] = sum_d x (fp32) */
int ctsi_attn_depthsum(const void* x_bf16, float* depthsum, float* colsum, int n, int c, int d,
                       int h, int w, void* stream);
int ctsi_attn_depthsum_tiles(int c, int h, int w);
/* pass 2: xhat_sum = gamma*rstd*(S - D*mean) + D*beta as bf16 (n, 1, h, w, c) */
int ctsi_attn_normsum(const float* depthsum, const double* sums, const float* gamma,
                      const float* beta, void* out_bf16, int n, int c, int d, int h, int w,
                      int groups, float eps, void* stream);
/* pass 3 (after a 1x1x1 conv of xhat_sum with W_proj*W_v): y = x + p[n][h][w][c] broadcast over d.
 * rowsum (exact mode, fp32 [n][d][h][w][heads]) may be NULL (== 1).                       */
int ctsi_attn_broadcast_add(const void* x_bf16, const void* p_bf16, const float* rowsum, int heads,
                            void* y_bf16, int n, int c, int d, int h, int w, void* stream);
/* exact mode helper: rowsum[n][d][h][w][head] = sum_k softmax_k(q.k * hd^-0.5) from
 * qk = conv1x1(gn(x)) restricted to the q and k thirds (bf16 NDHWC with 2*c channels). */
int ctsi_attn_softmax_rowsum(const void* qk_bf16, float* rowsum, int n, int c, int d, int h, int w,
                             int heads, void* stream);

/* time embedding (models/unet3d.py:18-48 and the per-block Linear of :88-91, 123-125) ------- *
 * temb = Linear2(SiLU(Linear1(sincos(t))));  tbias[r][o] = W_all[o] . SiLU(temb[r]) + b_all[o]
 * for the concatenation of every ResBlock's time_mlp.1 (`total_out` rows of W_all).
 * `t_rows` holds `rows` timestep values on the device (all steps x batch of a sampling loop are
 * embedded in one call, the schedule being known up front); scratch = rows*(dim+2*time_dim) floats. */
int ctsi_time_embed_fwd(const int* t_rows, int rows, int dim, int time_dim, const float* w1,
                        const float* b1, const float* w2, const float* b2, const float* w_all,
                        const float* b_all, int total_out, float* scratch, float* tbias_out,
                        void* stream);

/* trilinear depth upsample (F.interpolate(..., 'trilinear', align_corners=False) with h, w
 * unchanged: models/model.py:284-289, 191-196).  fp32 NCDHW in -> bf16 NDHWC channel slice
 * [c_off, c_off+c) of a c_total-channel tensor, plus an optional fp32 NCDHW copy.            */
int ctsi_trilinear_depth_fwd(const float* src_f32_ncdhw, void* dst_bf16_ndhwc, int n, int c,
                             int d_in, int d_out, int h, int w, int c_total, int c_off,
                             float* dst_f32_ncdhw, void* stream);

/* sampler updates --------------------------------------------------------------------------- *
 * DDIM (inference/sampler.py:294-334) and DDPM (models/diffusion.py:270-338) elementwise
 * updates with the reference's epsilons, clamps and nan_to_num guards folded in.
 * coef is a device table of 8 floats per step (see video-to-video-diffusion_amd/sampler.py);
 * the row used is coef[*step_ptr] (row 0 when step_ptr is NULL).  z (fp32 NDHWC) is updated in
 * place and a bf16 copy is written into channels [c_off, c_off+c) of the U-Net input tensor.
 * noise (fp32 NCDHW, the layout torch.randn_like(z) has) may be NULL.
 * ctsi_step_advance increments the device-side step counter (last node of a step graph).   */
/* nonfinite (DDIM only, may be NULL): int32 table [steps][6], row *step_ptr += {noise_pred NaN, Inf, z_0_pred NaN, Inf,
 * z-after-update NaN, Inf} -- what the reference's per-step guards report through logger.error (inference/sampler.py:
 * 288-292, 307-311, 331-334); the host reads it once after the loop (no sync inside the captured step).             */
int ctsi_ddim_step(float* z, const float* eps, const float* noise_ncdhw, void* zin_bf16,
                   int c_total, int c_off, const float* coef, const int* step_ptr, int n, int c,
                   int d, int h, int w, int* nonfinite, void* stream);
int ctsi_ddpm_step(float* z, const float* eps, const float* noise_ncdhw, void* zin_bf16,
                   int c_total, int c_off, const float* coef, const int* step_ptr, int n, int c,
                   int d, int h, int w, void* stream);
/* DDPM posterior with PER-SAMPLE timesteps on fp32 NCDHW tensors (replaces the arithmetic of GaussianDiffusion.
 * _predict_z_0_from_noise, p_mean_variance and p_sample: models/diffusion.py:249-268, 270-308, 310-338).  Sample b uses
 * coef[b*8 ..] = {sqrt(1-abar_t), sqrt(abar_t), posterior_mean_coef1, posterior_mean_coef2, [t != 0] exp(0.5 logvar)}.
 * z0_out (may be NULL) <- (z - c0 eps) / c1, clamped to [-1, 1] when clip != 0; out (may be NULL) <- c2 z0 + c3 z
 * (+ c4 noise when noise != NULL).  per_sample = C*D*H*W.                                                               */
int ctsi_ddpm_posterior(const float* z, const float* eps, const float* noise, float* z0_out, float* out,
                        const float* coef, int n, long long per_sample, int clip, void* stream);
int ctsi_step_advance(int* step_ptr, void* stream);
/* x <- nan_to_num(x, nan=0, posinf=1, neginf=-1) on a flat fp32 buffer (model.py:262-341) */
int ctsi_nan_to_num_f32(float* x, long long count, void* stream);
/* counts[0] += #NaN, counts[1] += #Inf of x; sanitize != 0 also applies nan_to_num (sampler.py:268-275 checkpoints) */
int ctsi_count_nonfinite_f32(float* x, long long count, int sanitize, int* counts, void* stream);

/* ---- training path (models/diffusion.py:81-247, models/model.py:158-228; backward = what autograd derives) ---- */
/* Weight gradient of Conv3d / ConvTranspose3d:  dw[cr*stride_r + cg*stride_g + t*stride_t] =
 *   scale * sum_v R[v][cr] * G[map(v,t)][cg],  map(v,t) = (n, d-pd+kd, h*sh-ph+kh, w*sw-pw+kw), zero outside G.
 * Conv3d: R = grad of the output, G = layer input (weight (cout,cin,kd,kh,kw): stride_r = cin*T, stride_g = T,
 * stride_t = 1).  ConvTranspose3d: R = layer input, G = grad of the output (weight (cin,cout,kd,kh,kw)).
 * bf16 NDHWC tensors, channel counts multiples of 8; a concatenated input is two calls with dw offset by the
 * first source's channels.  Deterministic (split-K partials in `workspace`, summed in a fixed order). */
typedef struct ctsi_wgrad_desc {
    int kd, kh, kw;
    int sh, sw;
    int pd, ph, pw;
    int n;
    int dr, hr, wr;        /* spatial size of R */
    int dg, hg, wg;        /* spatial size of G */
    int cr, cr_stride;     /* channels of R used / channels per voxel of R */
    int cg, cg_stride;
} ctsi_wgrad_desc;
size_t ctsi_wgrad_workspace_bytes(const ctsi_wgrad_desc* desc);
double ctsi_wgrad_flops(const ctsi_wgrad_desc* desc);
/* workspace_bytes: size of `workspace`; the call fails (nothing is launched) when it is smaller than what
 * ctsi_wgrad_workspace_bytes(desc) returns at launch time */
int ctsi_wgrad(const ctsi_wgrad_desc* desc, const void* r, const void* g, void* workspace, size_t workspace_bytes, float* dw,
               long long stride_r, long long stride_g, long long stride_t, float scale, void* stream);

/* out (ci_cnt, cout, T) with out[ci'][co][T-1-t] = w[co][ci_off+ci'][t]: the weight of the stride-1 'same' Conv3d
 * that computes the data gradient of a stride-1 'same' Conv3d with weight w (cout, cin, T) for the input-channel
 * slice [ci_off, ci_off+ci_cnt).  (Strided Conv3d <-> ConvTranspose3d data gradients use the weights as they are.) */
int ctsi_weight_dgrad_layout(const float* w, float* out, int cout, int cin, int taps, int ci_off, int ci_cnt,
                             void* stream);

/* Weight and bias gradients of MANY small pointwise layers (the proj_out / V layers of the TemporalAttention blocks, whose
 * operands are depth-summed tensors of a few hundred rows) in ONE launch.  entries: device array of
 *   { const bf16* x [rows][cin]; const bf16* dy [rows][cout]; float* dw [cout][dw_stride]; float* db or NULL;
 *     int rows, cin, cout, dw_stride; float b_scale; int pad }                                     (56 bytes each)
 * blocks: device array of { int entry, cout_tile, cin_tile, pad } -- one row per 64 x 64 tile of an entry's dW (16 bytes each).
 * dw[co][ci] = sum_r dy[r][co] * x[r][ci] (WRITTEN, not accumulated), db[co] = b_scale * sum_r dy[r][co]; deterministic. */
int ctsi_linear_wgrad_multi(const void* entries, const void* blocks, int n_blocks, void* stream);

/* Backward of ctsi_gn_apply (GroupNorm [+SiLU] [+time bias] [+residual] [+SiLU]).  x: the tensor that was normalised,
 * dy: gradient of the output (depth-broadcast (n,1,h,w,c) tensor when dy_bcast_d), sums: the forward's fp64 statistics,
 * residual: the forward's residual input (needed when silu_post).  Writes g_buf = gradient of the GroupNorm output
 * (== gradient of the residual when !silu_pre), dx (+ add when given), dgamma/dbeta (c floats) and, when dtbias is
 * given, dtbias[n][c] (row stride dtbias_stride) = per-sample channel sums of the gradient after the outer SiLU, and,
 * when dxsum is given, dxsum[c] = sum over samples and voxels of dx (before `add`) = the bias gradient of the
 * convolution that produced x, evaluated in fp32 from the statistics instead of re-reading dx.
 * g_buf may be NULL when there is neither a residual nor an outer SiLU (nothing but dx needs that gradient: the second pass
 * re-derives it from dy; same dx bit for bit, one tensor write less).
 * workspace: ctsi_gn_bwd_workspace_floats() floats (the statistics tile shrinks from 512 rows for small tensors, so that
 * function -- not ctsi_gn_bwd_tiles, the tile count at 512 rows -- sizes it). */
int ctsi_gn_bwd_tiles(int d, int h, int w);
size_t ctsi_gn_bwd_workspace_floats(int n, int c, int d, int h, int w, int groups);
int ctsi_gn_bwd(const void* x, const void* dy, int dy_bcast_d, const double* sums, const float* gamma,
                const float* beta, int n, int c, int d, int h, int w, int groups, float eps, int silu_pre,
                const void* residual, int silu_post, const void* add, void* g_buf, void* dx, float* workspace,
                float* dgamma, float* dbeta, float* dtbias, long long dtbias_stride, float* dxsum, void* stream);

/* out[c] = scale * sum over rows of x[row][c]  (bf16 rows of c_stride channels; conv bias gradients) */
size_t ctsi_channel_sum_workspace_floats(long long rows, int c);
int ctsi_channel_sum(const void* x, long long rows, int c, int c_stride, float* workspace, float* out, float scale,
                     void* stream);
int ctsi_add_bf16(void* a, const void* b, long long count, void* stream);        /* a += b */
int ctsi_f32_to_bf16(const float* src, void* dst, long long count, void* stream);

/* q_sample (models/diffusion.py:81-106): z_t = sqrt_alphas_cumprod[t_b]*z0 + sqrt_one_minus_alphas_cumprod[t_b]*noise,
 * fp32 NCDHW in, bf16 NDHWC channel slice out (the U-Net input tensor [z_t | cond]). */
int ctsi_q_sample(const float* z0, const float* noise, const float* sqrt_ac, const float* sqrt_1mac, const int* t,
                  void* dst, int n, int c, int d, int h, int w, int c_total, int c_off, void* stream);
/* Min-SNR weighted MSE (models/diffusion.py:147-203): loss_out[0] = sum_b norm[b] * sum mask*(pred-noise)^2,
 * loss_out[1+b] = per-sample sums.  pred fp32 NDHWC, noise fp32 NCDHW, mask fp32 (n,c,d) or NULL; the caller folds the
 * SNR weight and the normalisation of the reference's three cases into norm[b].  _bwd writes
 * d_pred = 2*norm[b]*mask*(pred-noise)*gscale[0] as bf16 NDHWC with c_stride channels per voxel (extra ones zero). */
size_t ctsi_mse_loss_workspace_doubles(int n);
int ctsi_mse_loss_fwd(const float* pred, const float* noise, const float* mask, const float* norm, int n, int c,
                      int d, int h, int w, double* workspace, float* loss_out, void* stream);
int ctsi_mse_loss_bwd(const float* pred, const float* noise, const float* mask, const float* norm,
                      const float* gscale, int n, int c, int d, int h, int w, void* dpred, int c_stride, void* stream);

/* time embedding for training: as ctsi_time_embed_fwd, but scratch = [sincos | first Linear PRE-activation | temb] */
int ctsi_time_embed_train_fwd(const int* t_rows, int rows, int dim, int time_dim, const float* w1, const float* b1,
                              const float* w2, const float* b2, const float* w_all, const float* b_all,
                              int total_out, float* scratch, float* tbias_out, void* stream);
/* backward of y = act(x) W^T + b (act = SiLU when silu_in, x then is the pre-activation); rows <= 64; any of
 * dw/db/dx may be NULL */
int ctsi_linear_bwd(const float* x, const float* w, const float* dy, int rows, int in_dim, int out_dim, int silu_in,
                    float* dw, float* db, float* dx, void* stream);

/* sliding-window stitching (inference/sampler.py:63-172, 338-453): Gaussian-weighted accumulation of one decoded
 * patch (fp32 NCDHW, nc = batch*channels planes) into the full-volume accumulator and weight map, and the final
 * acc / (wsum + 1e-8).  wd/wh/ww are the 1-D windows exp(-(x-(n-1)/2)^2 / (2 (n/6)^2)) on the device.          */
int ctsi_blend_accumulate(float* acc, float* wsum, const float* patch, const float* wd, const float* wh,
                          const float* ww, int nc, int pd, int ph, int pw, int d_full, int h_full, int w_full,
                          int d0, int h0, int w0, void* stream);
int ctsi_blend_normalize(float* acc, const float* wsum, long long count, void* stream);

/* validation metrics (utils/metrics.py:14-193) on device: for fp32 (n,c,d,h,w) volumes a, b writes, per depth
 * slice, out[slice][4] = (mean squared difference, mean SSIM, NaN count of a, NaN count of b) over (n,c,h,w).
 * SSIM = the reference's avg_pool2d box-window form (window odd <= 15, zeros counted at the borders, variances
 * clamped at 0, map clamped to [0,1]); PSNR = 20 log10(max_val / sqrt(max(mse, 1e-8))) is left to the caller. */
size_t ctsi_slice_metrics_workspace_doubles(int n, int c, int d, int h, int w);
int ctsi_slice_metrics(const float* a, const float* b, int n, int c, int d, int h, int w, int window, float max_val,
                       double* workspace, double* out, void* stream);

/* hipMemsetAsync on the engine stream (zeroing GroupNorm accumulators / padded channels);
 * capturable as a memset node.                                                              */
int ctsi_memset_async(void* ptr, int value, size_t bytes, void* stream);

/* ---- optimizer step on the device (training/train.py:172-212 Adam / AdamW over parameter groups; trainer.py:237-247) ----
 * ctsi_adamw_multi: ONE launch over device tables built by the host (optim.py):
 *   tensors[i] = { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; long long numel; int group; int pad }
 *   groups[j]  = { float lr, beta1, beta2, eps, weight_decay, step_size (= lr / (1 - beta1^t)), bc2_sqrt (= sqrt(1 - beta2^t)),
 *                  decay (= 1 - lr weight_decay), one_m_b1, one_m_b2, grad_scale; int decoupled (1 AdamW, 0 Adam + L2),
 *                  maximize, pad[3] }   (64 bytes; the derived constants are rounded from the host's doubles, as torch's are)
 *   chunks[b]  = { int tensor; int first_element / 4 }: block b updates ctsi_adamw_chunk_elems() elements of that tensor.
 * fp32 throughout, torch.optim.AdamW's single-tensor arithmetic operation by operation.
 * ctsi_copy_scale_multi: dst[i] = scale * src[i] over a table segs[k] = { const float* src; float* dst; long long n; float
 * scale; int pad }, pieces[b] = { int seg; int first_element / 4096 } -- every small fp32 operand of a program refreshed from
 * the parameters in one launch. */
int ctsi_adamw_chunk_elems(void);
int ctsi_adamw_multi(const void* tensors, const void* groups, const void* chunks, int nchunks, void* stream);
int ctsi_copy_scale_multi(const void* segs, const void* pieces, int npieces, void* stream);

/* Device-side errors recorded since the last call with reset != 0 (0 on a healthy run): today the only source is a split-K
 * conv block whose bounded wait for its partner's partial sums expired (csrc/conv3_halo_k32.hip): that tile's output is
 * then invalid, and this sticky count is what tells the host so.  *detail (may be NULL) = that tile's index.
 * SYNCHRONOUS 8-byte device-to-host read: the samplers call it once per sample(), where they read the non-finite table. */
int ctsi_device_error_status(unsigned int* count, unsigned int* detail, int reset);

/* hipGraph helpers (one captured graph per denoising step) -------------------------------- */
typedef struct ctsi_graph ctsi_graph;
int ctsi_graph_begin_capture(void* stream);
int ctsi_graph_end_capture(void* stream, ctsi_graph** graph);
int ctsi_graph_launch(ctsi_graph* graph, void* stream);
void ctsi_graph_destroy(ctsi_graph* graph);

/* timing helpers used by bench.py (HIP events on the engine's own stream) ----------------- */
typedef struct ctsi_event ctsi_event;
int ctsi_event_create(ctsi_event** ev);
int ctsi_event_record(ctsi_event* ev, void* stream);
/* device-side wait of `stream` for `ev` (no host sync; forks / joins a stream capture) */
int ctsi_stream_wait_event(void* stream, ctsi_event* ev);
int ctsi_event_elapsed_ms(ctsi_event* start, ctsi_event* stop, float* ms);
void ctsi_event_destroy(ctsi_event* ev);

/* depth-sharding collectives (RCCL over xGMI, one process per GPU) --------------------------- *
 * a volume's depth is cut into `world` slabs; before a depth-3 conv a slab needs one boundary
 * slice of each depth neighbour (models/unet3d.py:56,96,204-207,218-221; models/vae.py:27,65-69,86-90), GroupNorm needs
 * (sum, sumsq) over the whole depth (unet3d.py:59,97,151,329) and TemporalAttention the depth sum.  Each call below is
 * ONE sync point on `stream` (one RCCL group) and is stream-capture safe.  RCCL is bound at run time (dlopen; a librccl
 * already in the process is reused).
 *   host protocol: rank 0 calls ctsi_comm_unique_id, the 128 bytes are broadcast by the host (torch.distributed, MPI, a
 *   file ...), every rank calls ctsi_comm_init with the current HIP device set.                                        */
typedef struct ctsi_comm ctsi_comm;
int ctsi_comm_unique_id(void* id128);
/* world == 1 with id128 == NULL: no RCCL at all (every exchange degenerates to the volume-end zero fill). */
int ctsi_comm_init(ctsi_comm** comm, const void* id128, int rank, int world);
void ctsi_comm_destroy(ctsi_comm* comm);
int ctsi_comm_rank(const ctsi_comm* comm);
int ctsi_comm_world(const ctsi_comm* comm);
/* lo_halo <- rank-1's hi_own, hi_halo <- rank+1's lo_own (`bytes` each; zeros at the volume's ends). */
int ctsi_halo_exchange(ctsi_comm* comm, const void* lo_own, const void* hi_own, void* lo_halo, void* hi_halo,
                       size_t bytes, void* stream);
/* the same exchange plus, in the same sync point, sums[nsums] (fp64) and f32[nf32] summed over all ranks in place
 * (GroupNorm statistics travel with the boundary slices of the tensor they normalise); any part may be absent. */
int ctsi_halo_exchange_reduce(ctsi_comm* comm, const void* lo_own, const void* hi_own, void* lo_halo, void* hi_halo,
                              size_t bytes, double* sums, int nsums, float* f32, long long nf32, void* stream);
/* GroupNorm statistics (and optionally the TemporalAttention depth sum, fp32) summed over all ranks in place. */
int ctsi_gn_allreduce(ctsi_comm* comm, double* sums, int nsums, float* f32, long long nf32, void* stream);
/* recv = concatenation over ranks of `bytes` from each rank's send (result gather along depth). */
int ctsi_comm_allgather(ctsi_comm* comm, const void* send, void* recv, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTSI_H */
