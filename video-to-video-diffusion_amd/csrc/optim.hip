// Optimizer step of the training path on the device (reference: training/train.py:172-212 builds
// torch.optim.Adam / AdamW over per-module parameter groups with their own learning rates, training/trainer.py:237-247 steps
// it once per accumulation window).
//
// ONE launch for all tensors of all parameter groups ("multi-tensor apply"): a device table of tensors {param, grad, exp_avg,
// exp_avg_sq, numel, group} and a table of 64 Ki-element chunks {tensor, first element}; block b owns chunk b.  HBM-bound:
// 16 B read + 12 B written per element (fp32 master weights and both moments; 264.66 M parameters = 7.4 GB per step).  The
// arithmetic is torch.optim.AdamW's single-tensor form, operation by operation in fp32 (decoupled decay first, lerp, addcmul,
// bias corrections folded into step_size / denominator the way torch folds them), so parameters track torch's to rounding.
// The bf16 kernel images the conv kernels read are re-packed by the engine right behind this launch (engine.Program.
// fast_repack: one ctsi_copy_scale_multi for every small fp32 operand + one pack launch per conv image).
#include "ctsi_internal.h"

struct CtsiOptTensor {       // 48 bytes
    float* p;
    const float* g;
    float* m;
    float* v;
    long long numel;
    int group;
    int pad;
};
struct CtsiOptGroup {        // 64 bytes per hyper-parameter row, refreshed by the host before every step.  Everything torch derives
    float lr, beta1, beta2, eps, weight_decay;   // in Python doubles (1 - beta, 1 - lr wd, the bias corrections) arrives already
    float step_size;         // lr / (1 - beta1^t)            rounded from the double, so the kernel sees torch's constants
    float bc2_sqrt;          // sqrt(1 - beta2^t)
    float decay;             // 1 - lr * weight_decay
    float one_m_b1, one_m_b2;
    float grad_scale;        // gradients are multiplied by this first (1 / loss-scale when the caller unscales here; 1 otherwise)
    int decoupled;           // 1: AdamW (p *= 1 - lr wd); 0: Adam with L2 (g += wd p)
    int maximize;
    int pad[3];
};
static constexpr int OPT_CHUNK = 65536;

__global__ void __launch_bounds__(256)
adamw_multi_kernel(const CtsiOptTensor* __restrict__ tensors, const CtsiOptGroup* __restrict__ groups,
                   const int2* __restrict__ chunks /* {tensor, first element / 4} */) {
    const int2 ck = chunks[blockIdx.x];
    const CtsiOptTensor t = tensors[ck.x];
    const CtsiOptGroup h = groups[t.group];
    const long long e0 = (long long)ck.y * 4;
    long long e1 = e0 + OPT_CHUNK;
    if (e1 > t.numel) e1 = t.numel;
    // torch's kernels, operation by operation (each line = one torch op; the compiler may contract a * b + c into an fma
    // exactly where torch's pointwise functors get contracted):
    auto upd = [&](float& p, float g, float& m, float& v) {
        g *= h.grad_scale;
        if (h.maximize) g = -g;
        if (h.decoupled) p *= h.decay; else g = g + h.weight_decay * p;     // param.mul_(1 - lr wd) | grad.add(param, alpha=wd)
        m = m + h.one_m_b1 * (g - m);                        // exp_avg.lerp_(grad, 1 - beta1)
        const float vb = v * h.beta2;                        // exp_avg_sq.mul_(beta2)
        const float gg = g * g;
        v = vb + h.one_m_b2 * gg;                            //           .addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;   // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
        const float r = m / denom;
        p = p - h.step_size * r;                             // param.addcdiv_(exp_avg, denom, value=-step_size)
    };
    const bool vec = ((((unsigned long long)t.p | (unsigned long long)t.g | (unsigned long long)t.m | (unsigned long long)t.v) & 15ull) == 0);
    if (vec) {
        const long long q0 = e0 >> 2, q1 = e1 >> 2;          // whole float4 groups (e0 is a multiple of 4)
        for (long long q = q0 + threadIdx.x; q < q1; q += 256) {
            float4 p = reinterpret_cast<float4*>(t.p)[q];
            const float4 g = reinterpret_cast<const float4*>(t.g)[q];
            float4 m = reinterpret_cast<float4*>(t.m)[q];
            float4 v = reinterpret_cast<float4*>(t.v)[q];
            upd(p.x, g.x, m.x, v.x);
            upd(p.y, g.y, m.y, v.y);
            upd(p.z, g.z, m.z, v.z);
            upd(p.w, g.w, m.w, v.w);
            reinterpret_cast<float4*>(t.p)[q] = p;
            reinterpret_cast<float4*>(t.m)[q] = m;
            reinterpret_cast<float4*>(t.v)[q] = v;
        }
        for (long long e = (q1 << 2) + threadIdx.x; e < e1; e += 256) upd(t.p[e], t.g[e], t.m[e], t.v[e]);
    } else {
        for (long long e = e0 + threadIdx.x; e < e1; e += 256) upd(t.p[e], t.g[e], t.m[e], t.v[e]);
    }
}

extern "C" int ctsi_adamw_chunk_elems(void) { return OPT_CHUNK; }

// tensors / groups / chunks: device tables as laid out above (the host builds them: optim.py); nchunks blocks.
extern "C" int ctsi_adamw_multi(const void* tensors, const void* groups, const void* chunks, int nchunks, void* stream) {
    CTSI_CHECK_ARG(tensors && groups && chunks && nchunks >= 0, "ctsi_adamw_multi: bad arguments");
    if (nchunks == 0) return CTSI_OK;
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream,
                       (const CtsiOptTensor*)tensors, (const CtsiOptGroup*)groups, (const int2*)chunks);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// dst[i] = scale * src[i] for a table of fp32 segments {src, dst, n, scale}: every small fp32 operand of a program (biases,
// GroupNorm gamma / beta, the time-embedding matrices and their row-concatenations, scaled bias slices) refreshed from the
// parameters in ONE launch instead of one copy each.  One block per 4096-element piece: table2 = {segment, first element}.
struct CtsiCopySeg {
    const float* src;
    float* dst;
    long long n;
    float scale;
    int pad;
};
__global__ void __launch_bounds__(256)
copy_scale_multi_kernel(const CtsiCopySeg* __restrict__ segs, const int2* __restrict__ pieces) {
    const int2 pc = pieces[blockIdx.x];
    const CtsiCopySeg s = segs[pc.x];
    const long long e0 = (long long)pc.y * 4096;
    long long e1 = e0 + 4096;
    if (e1 > s.n) e1 = s.n;
    for (long long e = e0 + threadIdx.x; e < e1; e += 256) s.dst[e] = s.scale * s.src[e];
}
extern "C" int ctsi_copy_scale_multi(const void* segs, const void* pieces, int npieces, void* stream) {
    CTSI_CHECK_ARG(segs && pieces && npieces >= 0, "ctsi_copy_scale_multi: bad arguments");
    if (npieces == 0) return CTSI_OK;
    hipLaunchKernelGGL(copy_scale_multi_kernel, dim3((unsigned)npieces), dim3(256), 0, (hipStream_t)stream,
                       (const CtsiCopySeg*)segs, (const int2*)pieces);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}
