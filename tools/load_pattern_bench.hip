// Which input path should a read-once layer (the few-cout conv heads, csrc/conv3_head2.hip) use?  Every kernel here reads the
// same 3 GiB of 256-byte rows (C = 128 bf16 channels) exactly once, 16 rows (one MFMA A tile of 16 voxels x 128 channels) at a
// time, and feeds them to v_mfma_f32_16x16x32_bf16 so that the loads cannot be dropped:
//   stream      plain 16 B / lane coalesced read, no MFMA                                   (the copy-rate yardstick)
//   direct      global_load_dwordx4 straight into the MFMA A layout: lane -> row lane & 15, 16-byte piece 4 s + (lane >> 4):
//               consecutive lanes touch DIFFERENT 256-byte rows (16 x 64-byte segments per wave instruction)
//   dma         buffer_load ... lds (1 KiB contiguous per wave instruction, XOR swizzle on the source side) into a wave-private
//               ring, then 4 conflict-free ds_read_b128 per tile
// PF tiles of 4 KiB are kept in flight per wave.
//     hipcc -O3 --offload-arch=gfx950 tools/load_pattern_bench.hip -o gpurun_out/load_pattern_bench && gpurun_out/load_pattern_bench
#include "../video-to-video-diffusion_amd/csrc/conv3_halo_common.h"
#include <stdio.h>
#include <stdlib.h>

__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ x, long long n16, unsigned* sink) {
    unsigned acc = 0;
    const long long per = 4 * 256;                       // a block walks contiguous batches of 4 x 256 chunks
    for (long long b = blockIdx.x; b * per < n16; b += gridDim.x) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = x[b * per + u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345u) *sink = acc;
}

// tiles of 16 rows x 256 B; wave w of the grid takes tiles w, w + W, ...
template <int PF>
__global__ void __launch_bounds__(256) k_direct(const char* __restrict__ x, long long ntiles, float* sink) {
    const int lane = threadIdx.x & 63;
    const long long wv = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, W = ((long long)gridDim.x * 256) >> 6;
    const int r = lane & 15, kg = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bf16x8 fb;
#pragma unroll
    for (int k = 0; k < 8; ++k) fb[k] = (short)0x3f80;
    bf16x8 fa[PF][4];
    auto issue = [&](long long t, bf16x8* f) {
        const char* p = x + (t * 16 + r) * 256 + kg * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const bf16x8*>(p + s * 64);
    };
    long long t = wv;
#pragma unroll
    for (int i = 0; i < PF; ++i)
        if (t + i * W < ntiles) issue(t + i * W, fa[i]);
    for (; t < ntiles; t += PF * W) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            if (t + i * W >= ntiles) break;
            bf16x8 cur[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) cur[s] = fa[i][s];
            if (t + (i + PF) * W < ntiles) issue(t + (i + PF) * W, fa[i]);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[s], fb, acc, 0, 0, 0);
        }
    }
    if (acc[0] == 12345.f) *sink = acc[0];
}

template <int PF>
__global__ void __launch_bounds__(256) k_dma(const char* __restrict__ x, long long ntiles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int R = PF + 1;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const long long wv = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, W = ((long long)gridDim.x * 256) >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr3_t)smem + wave * R * 4096;
    char* mine = smem + wave * R * 4096;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bf16x8 fb;
#pragma unroll
    for (int k = 0; k < 8; ++k) fb[k] = (short)0x3f80;
    // DMA piece i of a tile: rows 4 i .. 4 i + 3; lane -> row 4 i + (lane >> 4), LDS piece lane & 15 <- source piece (lane & 15) ^ row
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * i + (lane >> 4);
        voff[i] = (unsigned)(row * 256 + (((lane & 15) ^ row) << 4));
    }
    auto issue = [&](long long t, int slot) {
        const v4i_t rs = h3_make_rsrc(x + t * 4096, 4096u);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            h3_dma16(rs, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + slot * 4096 + i * 1024)), voff[i], 0u);
    };
    long long t = wv;
    int q = 0;
    for (int i = 0; i < PF; ++i)
        if (t + i * W < ntiles) issue(t + i * W, i);
    for (; t < ntiles; t += W, ++q) {
        const bool more = t + PF * W < ntiles;
        if (more) issue(t + PF * W, (q + PF) % R);
        if (more) {
            if (PF == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (PF == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (PF == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            if (PF == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const char* slot = mine + (q % R) * 4096 + r * 256;
        bf16x8 cur[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) cur[s] = *reinterpret_cast<const bf16x8*>(slot + (((4 * s + kg) ^ r) << 4));
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[s], fb, acc, 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc[0] == 12345.f) *sink = acc[0];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const long long bytes = 3LL << 30;
    char* buf = nullptr;
    float* sink = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    const long long ntiles = bytes / 4096;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, 0);
            launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-28s %8.3f ms  %6.2f TB/s\n", name, best, (double)bytes / best * 1e-9);
        return 0;
    };
    hipFuncSetAttribute((const void*)k_dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_dma<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    timeit("stream (16 B/lane)", [&] { hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, (unsigned*)sink); });
    for (int blocks : {512, 1024, 2048}) {
        char nm[64];
        snprintf(nm, sizeof nm, "direct PF2, %d blocks", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL(k_direct<2>, dim3(blocks), dim3(256), 0, 0, buf, ntiles, sink); });
        snprintf(nm, sizeof nm, "direct PF4, %d blocks", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL(k_direct<4>, dim3(blocks), dim3(256), 0, 0, buf, ntiles, sink); });
        snprintf(nm, sizeof nm, "dma PF2, %d blocks", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL(k_dma<2>, dim3(blocks), dim3(256), 4 * 3 * 4096, 0, buf, ntiles, sink); });
        snprintf(nm, sizeof nm, "dma PF3, %d blocks", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL(k_dma<3>, dim3(blocks), dim3(256), 4 * 4 * 4096, 0, buf, ntiles, sink); });
        snprintf(nm, sizeof nm, "dma PF4, %d blocks", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL(k_dma<4>, dim3(blocks), dim3(256), 4 * 5 * 4096, 0, buf, ntiles, sink); });
    }
    CK(hipDeviceSynchronize());
    return 0;
}
