// Calibration of rocprofv3's FETCH_SIZE for the access shapes of this repo's kernels: every kernel
// here reads a KNOWN number of unique bytes exactly once from a buffer larger than the Infinity Cache, so
//     factor = bytes / (FETCH_SIZE [KB] * 1024)
// is the correction to apply to FETCH_SIZE for that shape (MI355X_MICROARCH.md prescribes x2 for wide coalesced reads and
// calls other widths uncalibrated).
//     hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o gpurun_out/fetch_calib
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/calib -o calib -- gpurun_out/fetch_calib
//     python tools/fetch_calib_report.py gpurun_out/calib/calib_counter_collection.csv profiles/r03_fetch_calibration.json
// Shapes:
//   stream16        global_load_dwordx4, 16 B per lane, fully coalesced (the guide's calibrated case: expect 2.0)
//   dma_rows32_one  buffer_load_dwordx4 ... lds as conv3_halo_k32_kernel's halo DMA issues it: a lane pair fetches one 32-byte
//                   piece (16 channels) of a voxel row, rows 256 B apart (C = 128 channels) -- ONE chunk of every row only,
//                   i.e. 32 of every 256 bytes
//   dma_rows32_all  the conv's real pattern: a block walks the 8 chunks of its 1224-row halo tile one after the other
//                   (chunk outer), so every byte of every row is read once, 32 B at a time
//   dma_rows32_c16  the same with 32-byte rows 32 B apart (C = 16: the U-Net stem) = a dense stream in 32-B pieces
#include "../video-to-video-diffusion_amd/csrc/conv3_halo_common.h"
#include <stdio.h>
#include <stdlib.h>

__global__ void __launch_bounds__(256) stream16(const uint4* __restrict__ x, long long n16, unsigned* sink) {
    unsigned acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256) {
        const uint4 v = x[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) *sink = acc;
}

// one block = one tile of ROWS rows; row r of the tile lives at (tile * ROWS + r) * row_bytes; chunk c = bytes [32 c, 32 c + 32)
template <int ROWS>
__global__ void __launch_bounds__(512) dma_rows32(const char* __restrict__ x, int row_bytes, int chunks, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const v4i_t rs = h3_make_rsrc(x + (long long)blockIdx.x * ROWS * row_bytes, 0x7fffffffu);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr3_t)smem;
    constexpr int INSTR = (ROWS + 31) / 32;
    unsigned acc = 0;
    for (int c = 0; c < chunks; ++c) {
        for (int j = wave; j < INSTR; j += 8) {
            const int v = j * 32 + (lane >> 1);
            const unsigned voff = v < ROWS ? (unsigned)v * row_bytes + (lane & 1) * 16 : 0x80000000u;
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + j * 1024));
            h3_dma16(rs, dst, voff, (unsigned)__builtin_amdgcn_readfirstlane(c * 32));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc ^= *reinterpret_cast<const unsigned*>(smem + (tid * 4) % (INSTR * 1024));
        __syncthreads();
    }
    if (acc == 0x12345u) *sink = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const long long bytes = 3LL << 30;   // 3 GiB >> 256 MiB Infinity Cache
    char* buf = nullptr;
    unsigned* sink = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes));
    CK(hipDeviceSynchronize());
    constexpr int ROWS = 1224;
    const int lds = ((ROWS + 31) / 32) * 1024;
    for (int rep = 0; rep < 2; ++rep) {
        // stream16: 2 GiB
        hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, (2LL << 30) / 16, sink);
        // 256-byte rows (C = 128): tiles of 1224 rows = 313,344 B; 8192 tiles = 2.39 GiB footprint
        hipLaunchKernelGGL(dma_rows32<ROWS>, dim3(8192), dim3(512), lds, 0, buf, 256, 1, sink);   // 8192 * 1224 * 32 B
        hipLaunchKernelGGL(dma_rows32<ROWS>, dim3(8192), dim3(512), lds, 0, buf, 256, 8, sink);   // 8192 * 1224 * 256 B
        // 32-byte rows (C = 16): 65536 tiles * 1224 * 32 B = 2.39 GiB
        hipLaunchKernelGGL(dma_rows32<ROWS>, dim3(65536), dim3(512), lds, 0, buf, 32, 1, sink);
        CK(hipDeviceSynchronize());
    }
    printf("expected_bytes stream16 %lld dma_rows32_one %lld dma_rows32_all %lld dma_rows32_c16 %lld\n", 2LL << 30,
           8192LL * ROWS * 32, 8192LL * ROWS * 256, 65536LL * ROWS * 32);
    return 0;
}
