// Helpers shared by the 3x3x3 halo-tile kernels (conv3_halo.hip, conv3_halo_persist.hip).
#pragma once
#include "ctsi_internal.h"

typedef __attribute__((address_space(3))) void* lptr3_t;

typedef int v4i_t __attribute__((ext_vector_type(4)));

// buffer resource words for a raw (stride 0) byte buffer; every input is wave-uniform
__device__ __forceinline__ v4i_t h3_make_rsrc(const void* ptr, unsigned num_bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    v4i_t r;
    r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull));
    r.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffull));
    r.z = (int)num_bytes;
    r.w = 0x00020000;
    return r;
}

// One LDS-DMA wave instruction (64 lanes x 16 B -> 1 KiB at `lds_addr`), issued from inline asm so that
// hipcc does not order later ds_reads behind it with a vmcnt(0): completion is waited for by hand
// (s_waitcnt vmcnt(0) + s_barrier before the data is read, see the main loop).
__device__ __forceinline__ void h3_dma16(const v4i_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
        : "memory");
}

// (m-tile, n-tile) of a block.  order 0: the n-tiles of one m-tile are neighbours (they share the halo in L2).
// order 1 (n_major): inside the contiguous range of logical ids an XCD owns, blocks walk all m-tiles of n-tile 0, then
// n-tile 1, ...: the CUs of an XCD then stream ONE n-tile's weight slab at a time (3.5 MB for 128 couts x 512 cin x 27,
// which fits the 4 MB L2), instead of all n-tiles' slabs at once (14 MB: every round re-fetches them from HBM).
// Needs mtiles % 8 == 0 (each XCD owns whole m-tiles x all n-tiles); otherwise order 0 is used.
__device__ __forceinline__ void h3_decode_tile(int bid, int mtiles, int ntiles_n, int n_major, int* mt, int* nt) {
    if (n_major && (mtiles & 7) == 0) {
        const int per_xcd = (mtiles >> 3) * ntiles_n;
        const int x = bid / per_xcd, local = bid - x * per_xcd;
        const int mcount = mtiles >> 3;
        *nt = local / mcount;
        *mt = x * mcount + (local - *nt * mcount);
    } else {
        *mt = bid / ntiles_n;
        *nt = bid - *mt * ntiles_n;
    }
}

__device__ __forceinline__ int xcd_remap_h(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (orig >> 3);
}
