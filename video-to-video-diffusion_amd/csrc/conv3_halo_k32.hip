// 3x3x3 stride-1 convolution, 512-voxel LDS halo tile x 128 couts on v_mfma_f32_16x16x32_bf16: the K = 32 of one MFMA is
// TWO TAPS x 16 channels.
//
// Why this form: the round-1 512-voxel kernel (conv3_halo32m_kernel, 32x32x16 MFMAs; now experiments/conv3_halo_m512.hip) is
// not cycle- but CLOCK-bound on real data -- the same
// launch on all-zero operands runs 26-33 % faster (1236 -> 1579 TFLOP/s for 384->128 @48x128^2; profiles/r02_notes.md), i.e. the
// chip lowers its clock under the MFMA load (MI355X_MICROARCH.md, DVFS give-back) and cycles saved in the issue stream come back
// only in part.  What raises the clock for the same FLOPs: the 16x16x32 MFMA shape (give-back item 7: ~1.12-1.15 x the FLOP/s
// of the 32x32x16 loop at equal cycles) and fewer VALU instructions per MFMA (rule 28).  This kernel keeps conv3_halo32m's
// data movement (16-channel chunks, 32-byte halo rows, double-buffered halo, weight ring fed by LDS-DMA, one barrier per
// step, 128 KB output tile through LDS) and changes the arithmetic:
//   * the (chunk, tap) pairs of a block form ONE linear stream q = chunk * 27 + tap; a *unit* is two consecutive entries
//     (2 x 16 channels = K 32), a *step* two units (4 taps, 16 KB of weights, one barrier).  Lanes 0-31 of an operand read
//     the unit's first tap, lanes 32-63 the second: the tap shift is a per-lane LDS address, so pairs may straddle (kd, kh)
//     rows and chunks (27 taps per chunk is odd); the stream is zero-padded to whole steps in the packed image;
//   * per wave 64 voxels x 128 couts = 4 x 8 tiles of 16x16: 32 MFMAs and 12 ds_read_b128 per unit -- the same LDS bytes per
//     FLOP as the 32x32x16 form -- with every fragment address = one per-lane base + a per-unit tap offset + an immediate
//     (4 VALU per 32 MFMAs instead of ~5 per fragment);
//   * no LDS swizzle: a 16-lane group of a ds_read_b128 covers 16 consecutive 32-byte rows, rows r and r + 8 (same banks)
//     always with DIFFERENT 16-byte halves (group lanes {0-3, 12-15} read half h, {4-11} read half 1 - h), conflict-free as laid
//     out by the DMA.
#include "conv3_halo_common.h"
#include <stdlib.h>
#include <type_traits>

// <TD, TH, TW, BN, UPS>: tile TD x TH x TW = 512 or 384 voxels, BN couts per block, UPS units per step (one barrier per step):
//   <4,4,32,128,2> / <4,8,16,128,2>: per wave 64 voxels x 128 couts (32 MFMAs, 12 ds_read_b128 per unit), steps of 4 entries;
//   <3,4,32,128,2> / <3,8,16,128,2>: 384 voxels, 48 per wave (24 MFMAs, 11 reads per unit) for levels the 512-voxel grid fills
//     badly (48 x 32 x 32 x 512 couts: 384 blocks of 512 x 128 = 1.5 rounds of the 256 CUs, 512 of 384 x 128 = 2 rounds);
//   <4,4,32, 64,4> / <4,8,16, 64,4>: per wave 64 x 64 (16 MFMAs, 8 reads per unit), steps of 8 entries -- the same 16 KB of
//     weights and 64 MFMAs per wave and barrier -- for levels where 128-cout blocks leave CUs idle (48 x 32 x 32 x 512 couts:
//     96 tiles x 4 = 384 blocks = 1.5 rounds of the 256 CUs; x 8 = 768 = 3 whole rounds).  Parity-tested, then measured
//     1.5-3 % SLOWER there than conv3_halo32_kernel's 768 blocks of 256 x 128 (1126-1159 vs 1148-1176 TFLOP/s; 0.5 LDS reads per
//     MFMA instead of 0.375 and twice the halo DMA per FLOP eat the MFMA-shape gain): NOT instantiated or dispatched
//     (profiles/r02_notes.md).
// Epilogue form (round 4), template parameter DIRECT.  DIRECT: packed row 16 j + r of an n-tile holds cout NJ r + j, so that lane
// r16 ends up with the NJ = 8 CONSECUTIVE couts NJ r16 .. NJ r16 + 7 of each of its 16 voxel rows and stores them as 16-byte pieces
// straight from the accumulators (a store instruction of a wave = 4 voxel rows x 256 contiguous bytes): no 128 KB tile through
// LDS, no 2-byte LDS writes, no row re-reads.  Staged (round 2): the bf16 tile goes through LDS, packed row = cout.  The two forms
// use different packed weight images (ctsi_conv3_halo_k32_pack's `direct`).  Same-box alternation of whole bench runs
// (profiles/r04_epilogue_ab.log): 512-voxel tiles -1.8 % (plain), -2.1 % (ConvTranspose), -0.7 % (split-K), the 384-voxel split-K
// form -4.3 %; the 24- / 12-wide 384-voxel tiles -2 % (25 stitching windows: 46.7 -> 45.8 ms over their 20 launches); the PLAIN
// 3x4x32 / 3x8x16 tiles +3 % / +1.5-2 % (slower) and their Downsample forms +-0: ctsi_conv3_halo_k32_direct() picks per form.
// cout (within the block's n-tile) of accumulator column r16 of cout tile j
template <int NJ, bool DIRECT>
__device__ __forceinline__ int hk_col(int j, int r16) { return DIRECT ? r16 * NJ + j : j * 16 + r16; }

template <int TD_, int TH_, int TW_ = 32, int BN_ = 128, int UPS_ = 2>
struct HkCfg {
    static constexpr int TD = TD_, TH = TH_, TW = TW_, UPS = UPS_;
    static constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
    static constexpr int HV = HD * HH * HW;                 // <4,4,32>: 1224 halo voxels
    static constexpr int HALO_INSTR = (HV + 31) / 32;       // wave-DMAs of 32 voxels x 32 B
    static constexpr int HALO_BYTES = HALO_INSTR * 1024;
    static constexpr int BM = TD * TH * TW;                 // 512
    static constexpr int BN = BN_;
    static constexpr int NJ = BN / 16;                      // 16-cout B tiles per wave
    static constexpr int TAP_BYTES = BN * 32;               // [BN couts][16 channels] bf16
    static constexpr int STEP_TAPS = 2 * UPS;
    static constexpr int WSLOT_BYTES = STEP_TAPS * TAP_BYTES;   // 16384
    static constexpr int NWS = 4;                           // three steps in flight + the one being read
    static constexpr int NWAVE = 8;
    static constexpr int MA = BM / 128;                     // 16-voxel A tiles per wave: 4 (512-voxel tile) or 3 (384)
    static constexpr int NTH = 64 * NWAVE;
    static constexpr int NPIECE = (HALO_INSTR + NWAVE - 1) / NWAVE;   // halo DMAs per wave and chunk
    static constexpr int OFF_W = 2 * HALO_BYTES;
    static constexpr int LOOP_END = OFF_W + NWS * WSLOT_BYTES;
    // output tile rows padded by 16 B: rows 4 apart (the k-groups of one ds_write_b16) fall 16 banks apart instead of on the
    // same banks (SQ_LDS_BANK_CONFLICT of the kernel 3.0 -> 0.8 %; the staging itself stayed at ~3.7 k cycles per tile: a
    // 2-way conflict fits under a 2-byte write's issue cost)
    static constexpr int BNP = BN + 8;
    static constexpr int OFF_ROW = LOOP_END > BM * BNP * 2 ? LOOP_END : BM * BNP * 2;
    static constexpr int OFF_CS = OFF_ROW + BM * 8;
    static constexpr int LDS_BYTES = OFF_CS + NWAVE * BN * 8;
    static constexpr int LPW = 64 / TW;                     // W-lines per wave
    // TW = 24 / 12 (round 4: the 24- and 12-wide planes of 192^2 patches, which 16- / 32-wide tiles cover at 75 %): a 16-row A tile
    // then STRADDLES W-lines, so every lane carries its own row offset per A tile instead of one wave-uniform offset (MA more
    // v_add per unit); nothing else in the kernel assumes that an A tile lies in one line
    static constexpr bool STRADDLE = TW % 16 != 0;
    static_assert(BM % 128 == 0 && (MA == 3 || (MA == 4 && TH % LPW == 0 && !STRADDLE)) && NPIECE <= 5 && LDS_BYTES <= 160 * 1024 &&
                      WSLOT_BYTES == 16384 && (UPS == 2 || UPS == 4), "unsupported tile");
    // A tile i (rows [16 i, 16 i + 16) of the wave's 64): halo-voxel offset from the wave's first voxel
    static constexpr int a_imm(int i) { return (((16 * i) / TW) * HW + (16 * i) % TW) * 32; }
};

__device__ __forceinline__ void hk_wait_vm(int allowed) {   // wave-uniform `allowed`
    switch (allowed) {
        case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); break;
    }
}

// TR = true: ConvTranspose3d k = (3,4,4), s = (1,2,2), p = 1 (models/unet3d.py:218-221 Upsample, models/vae.py decoder): output
// voxel (d, 2 h + py, 2 w + px) of parity class (py, px) is a 3 x 2 x 2-tap convolution on the INPUT grid (k_h in {1, 3} at
// input rows {h, h - 1} for py = 0, {2, 0} at {h, h + 1} for py = 1; same along w; i_d = o_d + 1 - k_d), i.e. the same halo
// tile as the 3x3x3 conv with 12 entries per chunk instead of 27.  One block = one (input tile, class, n-tile); the four
// classes of a tile are neighbours in the grid (they share the halo in L2).
//
// SK = true: 2-way split-K.  Levels with few voxels and many channels (48 x 16 x 16 x 512 -> 512: 32 tiles of 384 voxels x 4
// n-tiles = 128 blocks for 256 CUs) run every (tile, n-tile) as TWO blocks that each walk half of the input-channel chunks.
// The block that finishes first parks its fp32 accumulators in a workspace ([tile][register][512 threads]: coalesced, the
// partner has the same register <-> output mapping) and leaves; the second adds them to its own and runs the epilogue.  a + b
// = b + a in fp32, so the result does not depend on which block comes first: bit-stable without a second pass.  Hand-off per
// MI355X_MICROARCH.md (per-XCD L2s are not coherent), in its form WITHOUT cache maintenance: every store and load of the parked
// bytes and of the flag is an agent-scope (sc1) access; the producer drains them (vmcnt(0), workgroup barrier) before one lane
// raises the flag, the consumer polls with one lane, then a workgroup barrier (see the epilogue).  The consumer waits only for
// a block that has already taken its ticket, i.e. one that is in its epilogue: no deadlock; the spin is bounded anyway.
// DS = true: strided Conv3d k = (3,4,4), s = (1,2,2), p = 1 (models/unet3d.py:204-207 Downsample, models/vae.py encoder) -- the
// mirror of the TR form.  Split the INPUT into its four (h, w) parity sub-grids in_c[d, m, n] = in[d, 2 m + py, 2 n + px]:
// out[d, y, x] = sum over the 4 classes of a 3 x 2 x 2-tap stride-1 convolution on that sub-grid (k_h = 1 - py + 2 b reads
// sub-grid row y + b - py, b in {0, 1}; same along w), i.e. 48 taps = 4 "virtual chunks" of 12 entries per 16 input
// channels, every one on the 3x3x3 conv's halo tile of the OUTPUT grid.  The sub-grid is only an address pattern: the halo DMA
// of virtual chunk vc = 4 * chunk + class fetches voxel (2 m + py, 2 n + px) -- a wave-uniform add to the class-(0, 0)
// offsets (H_in, W_in are even, so the validity of a halo voxel does not depend on the class).  All classes accumulate into
// the same output tile: one block = one (output tile, n-tile), K = 48 * Cin like the gather kernel, which re-staged the slab
// per tap (505-850 TFLOP/s on the three Downsample layers of the U-Net; 27-48 % LDS bank conflicts in its 48-tap form).
//
// Sticky device-side error word ([0] = count, [1] = last tile): set when a split-K consumer's bounded wait expires (see the
// hand-off below); read and cleared by ctsi_device_error_status().
__device__ unsigned int g_hk_device_error[2] = {0u, 0u};
// timing-only instrumentation (CTSI_DEBUG_FLAGS & 4096): per block, wave 0 records s_memtime at 7 points + its HW_ID
#define HK_NSTAMP 4096
__device__ unsigned long long g_hk_stamps[HK_NSTAMP][8];
#define HK_STAMP(K)                                                                                              \
    if ((CTSI_DBG(p.dbg, 4096)) && tid == 0 && blockIdx.x < HK_NSTAMP) g_hk_stamps[blockIdx.x][K] = __builtin_amdgcn_s_memtime();

template <int TD_, int TH_, int TW_ = 32, int BN_ = 128, int UPS_ = 2, bool TR = false, bool SK = false, bool DS = false, bool DIRECT = false>
__global__ void __attribute__((amdgpu_flat_work_group_size(1, 512)))
conv3_halo_k32_kernel(const Conv3HaloParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using Cfg = HkCfg<TD_, TH_, TW_, BN_, UPS_>;
    constexpr int UPS = Cfg::UPS, NJ = Cfg::NJ, NJH = Cfg::NJ / 2, STEP_TAPS = Cfg::STEP_TAPS;
    constexpr int TAPS = (TR || DS) ? 12 : 27;               // entries per 16-channel chunk (DS: per virtual chunk)
    static_assert(!(TR && DS), "one form at a time");
    constexpr int MA = Cfg::MA;
    constexpr int TH = Cfg::TH, TW = Cfg::TW, HH = Cfg::HH, HW = Cfg::HW, HV = Cfg::HV;
    constexpr int HALO_INSTR = Cfg::HALO_INSTR, HALO_BYTES = Cfg::HALO_BYTES, BM = Cfg::BM, BN = Cfg::BN;
    constexpr int TAP_BYTES = Cfg::TAP_BYTES, WSLOT_BYTES = Cfg::WSLOT_BYTES, NWS = Cfg::NWS, NWAVE = Cfg::NWAVE;
    constexpr int NTH = Cfg::NTH, NPIECE = Cfg::NPIECE, OFF_W = Cfg::OFF_W, OFF_ROW = Cfg::OFF_ROW, OFF_CS = Cfg::OFF_CS;
    constexpr int BNP = Cfg::BNP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long* s_rowoff = reinterpret_cast<long long*>(smem + OFF_ROW);
    float* s_cs = reinterpret_cast<float*>(smem + OFF_CS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    HK_STAMP(0);
    if ((CTSI_DBG(p.dbg, 4096)) && tid == 0 && blockIdx.x < HK_NSTAMP)
        g_hk_stamps[blockIdx.x][7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32) |
                                     (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    int bid = xcd_remap_h(blockIdx.x, gridDim.x);
    const int khalf = SK ? (bid & 1) : 0;                   // the two halves of a tile are neighbours in the grid
    if (SK) bid >>= 1;
    int mt, nt, cls = 0;
    if (TR) {
        // siblings = the 4 ntiles_n (class, n-tile) blocks of one input tile: they share the halo, each streams its own weight
        // slab.  G = p.n_major siblings of a tile are neighbours in the grid (0: all of them), the sibling groups are walked
        // one after the other: G trades halo re-reads (x siblings / G) against the weight slabs an XCD's L2 holds at a time.
        const int ns = 4 * p.ntiles_n;
        const int G = (p.n_major > 0 && p.n_major < ns && ns % p.n_major == 0) ? p.n_major : ns;
        const int per_g = p.mtiles * G;
        const int sg = bid / per_g;
        const int rem = bid - sg * per_g;
        mt = rem / G;
        const int sib = sg * G + (rem - mt * G);
        cls = sib / p.ntiles_n;
        nt = sib - cls * p.ntiles_n;
    } else {
        h3_decode_tile(bid, p.mtiles, p.ntiles_n, p.n_major, &mt, &nt);
    }
    const int py = cls >> 1, px = cls & 1;
    const int n0 = nt * BN;
    const int nchunks = SK ? p.nchunks / 2 : p.nchunks;      // chunks this block walks, starting at chunk cbase
    const int cbase = khalf * nchunks;
    const int nb = mt / p.tps;
    int r0 = mt - nb * p.tps;
    int tD, tH, tW;
    if (p.tile_order == 1) {
        // (tH, tD, tW): the depth bands of one tile row innermost -- the two depth-halo slices a tile shares with the band above
        // are re-read a tile row later instead of a whole band later.  Measured on the VAE decoder (48 x 512^2, 48 x 256^2): no
        // change in time, FETCH_SIZE +8 % (profiles/r04_notes.md): kept as a switch, not selected
        tH = r0 / (p.tilesD * p.tilesW);
        r0 -= tH * p.tilesD * p.tilesW;
        tD = r0 / p.tilesW;
        tW = r0 - tD * p.tilesW;
    } else if (p.tile_order == 2 && (p.tilesW & 3) == 0 && (p.tilesH & 7) == 0) {
        // 2-D super-tiles of 8 (H) x 4 (W) tiles inside a depth band: the 32 blocks an XCD runs at a time share their H AND W halo
        // rows in its L2 (the plain order keeps two tile rows of 16 co-resident on 512-wide planes)
        tD = r0 / (p.tilesH * p.tilesW);
        r0 -= tD * p.tilesH * p.tilesW;
        const int st = r0 >> 5, in = r0 & 31, spr = p.tilesW >> 2;
        tW = (st % spr) * 4 + (in & 3);
        tH = (st / spr) * 8 + (in >> 2);
    } else {
        tD = r0 / (p.tilesH * p.tilesW);
        r0 -= tD * p.tilesH * p.tilesW;
        tH = r0 / p.tilesW;
        tW = r0 - tH * p.tilesW;
    }
    const int d0 = tD * Cfg::TD, h0 = tH * TH, w0 = tW * TW;

    int dlo = d0 + p.dshift - 1;
    dlo = dlo < 0 ? 0 : dlo;
    const long long basevox = ((long long)(nb * p.Di + dlo) * p.Hi) * p.Wi;
    const v4i_t rs1 = h3_make_rsrc(reinterpret_cast<const char*>(p.x1) + basevox * p.C1 * 2, 0x7fffffffu);
    const v4i_t rs2 = h3_make_rsrc(reinterpret_cast<const char*>(p.x2) + basevox * p.C2 * 2, 0x7fffffffu);
    // (TR: one packed image per class, each padded to whole steps)
    const long long w_class = (long long)((p.nchunks * TAPS + STEP_TAPS - 1) / STEP_TAPS) * STEP_TAPS * p.CoutPad * 32;
    // (SK: the second half starts at entry TAPS * cbase of the packed stream: a whole number of steps, checked on the host)
    const v4i_t rsw = h3_make_rsrc(reinterpret_cast<const char*>(p.w) + cls * w_class + (long long)n0 * 32 +
                                       (long long)cbase * TAPS * p.CoutPad * 32, 0x7fffffffu);
    const unsigned lds0 = (unsigned)(unsigned long long)(lptr3_t)smem;

    const int C1 = p.C1, C2 = p.C2, CoutPad = p.CoutPad;
    const int Q = nchunks * TAPS;                            // (chunk, tap) entries
    const int S = (Q + STEP_TAPS - 1) / STEP_TAPS;           // steps (the packed image is zero-padded to whole steps)
    // weights of step s = STEP_TAPS entries x TAP_BYTES = 16 pieces of 1 KB: wave w copies pieces w and 8 + w
    const unsigned w_voff = (unsigned)lane * 16u;
    auto issue_weights = [&](int s) {
        const unsigned slot = lds0 + OFF_W + (s % NWS) * WSLOT_BYTES;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = wave + NWAVE * k;
            constexpr int PPT = TAP_BYTES / 1024;            // pieces per entry
            const int e = piece / PPT, quarter = piece % PPT;
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(((s * STEP_TAPS + e) * CoutPad) * 32 + quarter * 1024);
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(slot + piece * 1024));
            h3_dma16(rsw, dst, w_voff, soff);
        }
    };

    // first of the prologue (see below): the bias row and the weights of step 0 need no per-lane halo arithmetic -- they fly while it runs
    const bool has_bias = p.bias != nullptr && khalf == 0;
    if (has_bias && wave == 0) {
        const int left = p.Cout - n0;
        const v4i_t rsb = h3_make_rsrc(p.bias + n0, (unsigned)(left > 0 ? left * 4 : 0));
        h3_dma16(rsb, (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + OFF_CS)), (unsigned)lane * 16u, 0u);
    }
    if (S > 0) issue_weights(0);

    // halo DMA: piece j = wave + NWAVE * i covers halo voxels 32 j .. 32 j + 31; lane -> voxel 32 j + lane / 2, 16-byte half lane & 1
    int hrel[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int j = wave + NWAVE * i;
        const int v = j * 32 + (lane >> 1);
        const int hd = v / (HH * HW), rem = v - hd * (HH * HW);
        const int hh = rem / HW, hw = rem - hh * HW;
        const int gd = d0 + p.dshift - 1 + hd, gh = h0 - 1 + hh, gw = w0 - 1 + hw;
        if (DS) {   // (gh, gw) index a parity sub-grid of Ho x Wo voxels: input voxel (2 gh + py, 2 gw + px), class added per chunk
            const bool ok = (j < HALO_INSTR) && (v < HV) && gd >= 0 && gd < p.Di && gh >= 0 && gh < p.Ho && gw >= 0 && gw < p.Wo;
            hrel[i] = ok ? ((gd - dlo) * p.Hi + 2 * gh) * p.Wi + 2 * gw : -1;
        } else {
            const bool ok = (j < HALO_INSTR) && (v < HV) && gd >= 0 && gd < p.Di && gh >= 0 && gh < p.Hi && gw >= 0 && gw < p.Wi;
            hrel[i] = ok ? ((gd - dlo) * p.Hi + gh) * p.Wi + gw : -1;
        }
    }
    const unsigned hq16 = (unsigned)((lane & 1) * 16);

    auto issue_halo = [&](int cc, int i) -> int {
        const int j = wave + NWAVE * i;
        if (j >= HALO_INSTR) return 0;
        const int vc = cbase + cc;                           // DS: virtual chunk = 4 * channel chunk + parity class
        const int ch0 = (DS ? (vc >> 2) : vc) * 16;
        const int cadd = DS ? ((vc >> 1) & 1) * p.Wi + (vc & 1) : 0;   // sub-grid origin (py, px) in input voxels
        const bool second = ch0 >= C1;
        const unsigned cbytes = (unsigned)((second ? C2 : C1) * 2);
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((second ? ch0 - C1 : ch0) * 2);
        int hsel = hrel[0];
#pragma unroll
        for (int q = 1; q < NPIECE; ++q) {   // (the empty asm keeps hrel[] in registers: hipcc otherwise turns the select chain
            int cand = hrel[q];              //  into a dynamically indexed scratch array, and a scratch load counts in vmcnt)
            asm("" : "+v"(cand));
            hsel = (i == q) ? cand : hsel;
        }
        const unsigned voff = hsel >= 0 ? (unsigned)(hsel + cadd) * cbytes + hq16 : 0x80000000u;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (cc & 1) * HALO_BYTES + j * 1024));
        if (second)
            h3_dma16(rs2, dst, voff, soff);
        else
            h3_dma16(rs1, dst, voff, soff);
        return 1;
    };
    // fragment addressing: lane -> row r16 = lane & 15 of the 16-row operand tile, k group kg = lane >> 4:
    // kg >> 1 = which of the unit's two taps, kg & 1 = which 8 of the tap's 16 channels
    const int r16 = lane & 15, kg = lane >> 4;
    const bool tap1 = kg >= 2;
    // Which of an A tile's 16 voxels MFMA row rho computes is free to choose.  ConvTranspose form with the direct epilogue: voxel =
    // 4 (rho & 3) + (rho >> 2), i.e. the four lane groups of one store instruction hold four CONSECUTIVE input voxels: -3.7 % on
    // those launches (1.88 -> 1.81 ms for the U-Net's two).  The same permutation costs the plain and split-K forms 0-3 % (the lane
    // order of the A fragments' ds_read_b128 changes): they keep voxel = rho (profiles/r04_notes.md).
    constexpr bool VPERM = DIRECT && TR;
    const int v16 = VPERM ? 4 * (r16 & 3) + (r16 >> 2) : r16;
    int a_lane, b_lane;
    int aoff[MA];                                            // byte offset of A tile i from the wave's first voxel: immediates for
    {                                                        // the 512-voxel tiles, wave-uniform registers for 384 (48 rows per
        auto hv = [&](int row) {                             // wave = 1.5 W-lines of 32: the split depends on the wave's parity)
            const int line = row / TW, wofs = row % TW;
            return ((line / TH) * HH + (line % TH)) * HW + wofs;
        };
        const int row0 = wave * 16 * MA;
        const int vbase = hv(row0);
        if constexpr (Cfg::STRADDLE) {                       // per-lane offsets: row r16 of A tile i may lie on the next W-line
            const int v0 = hv(row0 + v16);
#pragma unroll
            for (int i = 0; i < MA; ++i) aoff[i] = (hv(row0 + 16 * i + v16) - v0) * 32;
            a_lane = v0 * 32 + (kg & 1) * 16;
        } else {
#pragma unroll
            for (int i = 0; i < MA; ++i) {
                if constexpr (MA == 4)
                    aoff[i] = Cfg::a_imm(i);
                else
                    aoff[i] = (hv(row0 + 16 * i) - vbase) * 32;
            }
            a_lane = (vbase + v16) * 32 + (kg & 1) * 16;
        }
        b_lane = OFF_W + (kg >> 1) * TAP_BYTES + r16 * 32 + (kg & 1) * 16;
    }
    // LDS byte offset of entry q's tap in its halo buffer (wave-uniform); entries past the end repeat the last one (their
    // weights are zero in the packed image; a repeated REAL tap keeps 0 x value finite wherever the real product is)
    auto tap_off = [&](int q) -> int {
        q = q < Q ? q : Q - 1;
        const int cc = q / TAPS, t = q - cc * TAPS;
        int kd, kh, kw;                                      // halo coordinates of the tap (0..2 each)
        if (TR) {
            kd = t >> 2;                                     // (entry t holds kernel tap k_d = 2 - (t >> 2): halo slices ascending)
            kh = ((t >> 1) & 1) ? (py ? 2 : 0) : 1;
            kw = (t & 1) ? (px ? 2 : 0) : 1;
        } else if (DS) {                                     // class of the virtual chunk: (cbase + cc) & 3 (cbase % 4 == 0)
            kd = t >> 2;
            kh = ((t >> 1) & 1) + 1 - ((cc >> 1) & 1);
            kw = (t & 1) + 1 - (cc & 1);
        } else {
            kd = t / 9;
            const int t2 = t - kd * 9;
            kh = t2 / 3;
            kw = t2 - kh * 3;
        }
        return (cc & 1) * HALO_BYTES + ((kd * HH + kh) * HW + kw) * 32;
    };

    f32x4 acc[MA][NJ];

    bf16x8 fa0[MA], fa1[MA], fbl[NJH], fbh[NJH];

    // A fragments i0, i0 + 1 of a unit into FA; B fragments j0 .. j0 + NJH - 1 of unit `uu` of the step at BADDR into FB
#define HK_LOAD_A(FA, I0, AADDR)                                                                               \
    {                                                                                                          \
        _Pragma("unroll") for (int i_ = (I0); i_ < ((I0) == 0 ? 2 : MA); ++i_)                                 \
            FA[i_] = *reinterpret_cast<const bf16x8*>(smem + (AADDR) + aoff[i_]);                              \
    }
#define HK_LOAD_B(FB, J0, BADDR, UU)                                                                           \
    {                                                                                                          \
        _Pragma("unroll") for (int j_ = 0; j_ < NJH; ++j_)                                                     \
            FB[j_] = *reinterpret_cast<const bf16x8*>(smem + (BADDR) + (UU) * 2 * TAP_BYTES + ((J0) + j_) * 512); \
    }
#define HK_MFMA(FA, FB, J0)                                                                                    \
    {                                                                                                          \
        _Pragma("unroll") for (int j_ = 0; j_ < NJH; ++j_) _Pragma("unroll") for (int i_ = 0; i_ < MA; ++i_)    \
            acc[i_][(J0) + j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA[i_], FB[j_], acc[i_][(J0) + j_], 0, 0, 0); \
    }
    // a phase: 4 NJH MFMAs with its NJH + 2 ds_read_b128 in the first gaps (they feed the NEXT phase)
#define HK_SCHED()                                                                                             \
    {                                                                                                          \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                                                     \
        _Pragma("unroll") for (int q_ = 0; q_ < NJH + 2; ++q_) {                                               \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                 \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                 \
        }                                                                                                      \
        __builtin_amdgcn_sched_group_barrier(0x008, MA * NJH - (NJH + 2), 0);                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
    }

    // prologue: the block's 128 bias values (one 512-byte LDS-DMA piece of wave 0 into the column-sum scratch; couts past Cout
    // read as 0 through the buffer's range check), weights of step 0, halo of chunk 0 -- and BEHIND them the weights of steps
    // 1 and 2, which may stay in flight when the loop starts: every CU runs its prologue at the same time and that burst is
    // HBM-bound (13-14 k cycles per tile with all 87 KB awaited, tools/k32_stamps.py), so the loop starts on the first 55 KB.
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) issue_halo(0, i);
    int n_trail = 0;                 // pieces of steps 1 and 2 (the youngest of this wave)
#pragma unroll
    for (int s = 1; s < NWS - 1; ++s)
        if (s < S) {
            issue_weights(s);
            n_trail += 2;
        }
    if (tid < BM) {   // output row offsets (read in the epilogue only): computed while the DMAs fly; row = line * TW + m, line = ld * TH + lh
        const int mm = tid % TW, line = tid / TW;
        const int d = d0 + line / TH, h = h0 + line % TH, w = w0 + mm;
        long long off = -1;
        if (TR) {
            if (d < p.Do && 2 * h < p.Ho && 2 * w < p.Wo)
                off = ((((long long)nb * p.Do + d) * p.Ho + 2 * h + py) * p.Wo + 2 * w + px) * p.cout_stride + p.c_off;
        } else if (d < p.Do && h < p.Ho && w < p.Wo) {
            off = ((((long long)nb * p.Do + d) * p.Ho + h) * p.Wo + w) * p.cout_stride + p.c_off;
        }
        s_rowoff[tid] = off;
    }

    // Every form orders its entries by halo depth slice: step 0 and the fragment loads it runs ahead read slice offsets 0 only, i.e. halo depth slices 0 .. TD - 1;
    // the pieces that hold slices TD, TD + 1 (the youngest halo pieces of a wave, issued before the weights of steps 1-2) may
    // stay in flight as well: B_0 waits for everything but the weights of step 2, and slice TD is first read behind B_1.
    int n_late = 0;
    {
        constexpr int NEED = (Cfg::TD * HH * HW + 31) / 32;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) n_late += (wave + NWAVE * i >= NEED && wave + NWAVE * i < HALO_INSTR) ? 1 : 0;
    }
    hk_wait_vm(n_trail == 4 ? n_trail + n_late : n_trail);
    __syncthreads();
    HK_STAMP(1);
    // the accumulators start at the bias of their cout (split-K: in the first half only): no bias pass in the epilogue
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float bv = has_bias ? s_cs[hk_col<NJ, DIRECT>(j, r16)] : 0.0f;
#pragma unroll
        for (int i = 0; i < MA; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][j][q] = bv;
    }
    {
        const int oa = tap_off(0), ob = tap_off(1);
        const int aaddr = a_lane + (tap1 ? ob : oa);
        HK_LOAD_A(fa0, 0, aaddr);
        HK_LOAD_A(fa0, 2, aaddr);
        HK_LOAD_B(fbl, 0, b_lane, 0);
    }
    if (CTSI_DBG(p.dbg, 128)) {      // timing-only (ablation builds): the second B half is never loaded -- 0.25 LDS reads per MFMA
#pragma unroll
        for (int j = 0; j < NJH; ++j) fbh[j] = fbl[j];
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- main loop ------------------------------------------------------------------------------------------------------------
    // Units u = UPS s + uu.  Unit u's fragments are loaded during unit u - 1 (its second B half during its own first phase).
    // Barrier B_s stands before the LAST unit of step s: everything step s + 1 reads has landed and is visible (the loads of
    // step s + 1's first unit run in that last unit); every wave has drained weight slot (s - 1) % NWS and any halo chunk whose
    // last entry lies in step s.  Issue group s (behind B_s) = weights of step s + 3 into that slot + halo pieces of the next
    // chunk.  At B_t a wave waits for all its pieces but those of group t - 1 (counted vmcnt); when halo pieces of group t - 1
    // are needed by step t + 1 (UPS = 4: a chunk lasts 3.4 steps), they were issued BEFORE the group's weights and only those
    // weight pieces may stay in flight.
    int n_prev = n_trail == 4 ? 2 : 0;   // pieces that may stay in flight at the next barrier: those of the most recent issue
                                         // group; at B_0 the weights of step 2 (step 1's are read behind B_0)
    int hc = 1, hp = 0, hfree = 0;   // next halo chunk to fetch, its next piece, the first step whose group may issue it
    // halo pieces per group: a chunk lasts 6.75 steps (27 entries, UPS 2: 2 + 1 + 1 + 1), 3.4 (UPS 4) or 3 (TR: 12 entries): 3 + 2
    constexpr int H_FIRST = (UPS == 2 && !TR && !DS) ? 2 : 3, H_LATER = (UPS == 2 && !TR && !DS) ? 1 : 3;
    auto issue_group = [&](int s) {
        int n_halo = 0, n_w = 0;
        bool urgent = false;
        if (hc < nchunks && s >= hfree && !(CTSI_DBG(p.dbg, 1))) {
            // the chunk's first entry TAPS hc lies in unit (TAPS hc) >> 1 = step S': it is first read behind B_{S' - 1}
            urgent = s + 2 >= ((TAPS * hc) >> 1) / UPS;
            const int cnt = hp == 0 ? H_FIRST : H_LATER;
#pragma unroll
            for (int k = 0; k < (H_FIRST > H_LATER ? H_FIRST : H_LATER); ++k)
                if (k < cnt && hp < NPIECE) {
                    n_halo += issue_halo(hc, hp);
                    ++hp;
                }
            if (hp >= NPIECE) {
                hp = 0;
                hfree = ((TAPS * hc - 1) >> 1) / UPS;        // step of the unit that holds chunk hc - 1's last entry
                ++hc;
            }
        }
        if (s + NWS - 1 < S && !(CTSI_DBG(p.dbg, 2))) {
            issue_weights(s + NWS - 1);
            n_w = 2;
        }
        n_prev = urgent ? n_w : n_w + n_halo;
        __builtin_amdgcn_sched_barrier(0);
    };
    // one unit: FAc = this unit's A fragments (fbl holds its first B half), FAn receives the next unit's
#define HK_UNIT(FAc, FAn, BADDR, UU, BADDR_N, UU_N, AADDR_N, MID)                                              \
    {                                                                                                          \
        if (!(CTSI_DBG(p.dbg, 128))) HK_LOAD_B(fbh, NJH, BADDR, UU);   /* (128, ablation builds: 8 instead of 12 reads per unit) */ \
        HK_LOAD_A(FAn, 0, AADDR_N);                                                                            \
        HK_MFMA(FAc, fbl, 0);                                                                                  \
        HK_SCHED();                                                                                            \
        MID;                                                                                                   \
        HK_LOAD_A(FAn, 2, AADDR_N);                                                                            \
        HK_LOAD_B(fbl, 0, BADDR_N, UU_N);                                                                      \
        HK_MFMA(FAc, fbh, NJH);                                                                                \
        HK_SCHED();                                                                                            \
    }
    for (int s = 0; s < S; ++s) {
        const int baddr = b_lane + (s % NWS) * WSLOT_BYTES;
        const int baddr_n = b_lane + ((s + 1) % NWS) * WSLOT_BYTES;
        const int q0 = STEP_TAPS * s;
        int an[UPS];                                             // A address of units UPS s + 1 .. UPS s + UPS (the next step's first)
#pragma unroll
        for (int k = 0; k < UPS; ++k) {
            const int oa = tap_off(q0 + 2 * k + 2), ob = tap_off(q0 + 2 * k + 3);
            an[k] = a_lane + (tap1 ? ob : oa);
        }
        HK_UNIT(fa0, fa1, baddr, 0, baddr, 1, an[0], );
        if constexpr (UPS == 4) {
            HK_UNIT(fa1, fa0, baddr, 1, baddr, 2, an[1], );
            HK_UNIT(fa0, fa1, baddr, 2, baddr, 3, an[2], );
        }
        hk_wait_vm(n_prev);
        if (!(CTSI_DBG(p.dbg, 64))) __builtin_amdgcn_s_barrier();      // (64: timing-only ablation)
        __builtin_amdgcn_sched_barrier(0);
        // SIMD partners (waves w and w + 4) issue their pieces at DIFFERENT phase boundaries: a piece costs its wave ~60-100 issue
        // cycles during which it feeds no MFMAs, so waves 0-3 issue behind the first MFMA phase after the barrier -- while waves
        // 4-7 run their second phase on the matrix pipe -- and waves 4-7 behind that second phase (p.dbg & 16: all waves at the
        // first boundary, for A/B timing: 5-6 % slower on real data, 12-14 % on zeros)
        HK_UNIT(fa1, fa0, baddr, UPS - 1, baddr_n, 0, an[UPS - 1], if (wave < 4 || (CTSI_DBG(p.dbg, 16))) issue_group(s));
        if (wave >= 4 && !(CTSI_DBG(p.dbg, 16))) issue_group(s);
    }
#undef HK_UNIT
#undef HK_LOAD_A
#undef HK_LOAD_B
#undef HK_MFMA
#undef HK_SCHED
    HK_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // staged / split-K forms overwrite or hand over what other waves may still be reading; the direct form touches only its own
    // accumulators, the read-only row offsets and its own slots of the column-sum scratch until the barrier before the cross-wave
    // sums, so its waves run into the epilogue as they finish (waves 0-3 leave the last step ~900 cycles before their partners)
    if (!(DIRECT && !SK) || (p.dbg_epi_barrier != 0)) __syncthreads();
    HK_STAMP(3);

    // ---- epilogue: bias, GroupNorm column sums, the 512 x 128 bf16 tile through LDS (128 KB), 16-byte row stores -----
    // accumulator (i, j)[q]: row 16 i + 4 kg + q of the wave's 64, cout 16 j + r16
    if (CTSI_DBG(p.dbg, 8)) return;
    if (SK) {
        const int tile = mt * p.ntiles_n + nt;
        constexpr int NREG = MA * NJ * 4;
        float* wsl = p.sk_ws + (size_t)tile * NREG * NTH + tid;
        int* s_role = reinterpret_cast<int*>(s_cs);          // (s_cs is first written after the barriers below)
        if (tid == 0) {
            const int tk = __hip_atomic_fetch_add(p.sk_sync + 2 * tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // A ticket >= 2 means an EARLIER launch left this tile's hand-off words set (its consumer's wait expired, see below):
            // nothing computed on them can be trusted, so this launch raises the sticky error as well -- the condition stays
            // visible to every ctsi_device_error_status() until the host has re-zeroed the workspace (engine.check_device_errors
            // does so for every live program) -- and the two blocks still take complementary roles by parity: no block waits
            // on a partner that also waits.
            if (tk >= 2) {
                atomicAdd(&g_hk_device_error[0], 1u);
                g_hk_device_error[1] = (unsigned)tile;
            }
            *s_role = tk & 1;
        }
        __syncthreads();
        const int role = *s_role;
        __syncthreads();
        // Hand-off without cache maintenance (MI355X_MICROARCH.md, second valid form): EVERY store and load of the handed-off
        // bytes is an agent-scope (sc1) access, drained before the flag is raised; a release / acquire pair would write back
        // and invalidate the XCD's whole L2 -- 128 times per launch, under the weight streams of the blocks still computing.
        if (role == 0) {                                     // first to finish: park the partial sums and leave
#pragma unroll
            for (int i = 0; i < MA; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        __hip_atomic_store(wsl + (size_t)((i * NJ + j) * 4 + q) * NTH, acc[i][j][q], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(p.sk_sync + 2 * tile + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(p.sk_sync + 2 * tile + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && spins < (1 << 22)) {
                __builtin_amdgcn_s_sleep(4);
                ++spins;
            }
            if (spins >= (1 << 22)) {
                // The partner holds its ticket, i.e. it is in its epilogue: this cannot happen on a healthy device.  If it
                // does, the result must not pass as valid: the sticky device error word makes the next
                // ctsi_device_error_status() (the samplers read it once per sample()) raise, and ticket / flag are left as
                // they are (a late partner must not meet reset flags it would then corrupt).  No other work in this path:
                // a second barrier + a select over the accumulators here cost the split-K layers 5-17 % (r03 notes).
                atomicAdd(&g_hk_device_error[0], 1u);
                g_hk_device_error[1] = (unsigned)tile;
            } else {   // ready for the next launch on this stream
                __hip_atomic_store(p.sk_sync + 2 * tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(p.sk_sync + 2 * tile + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MA; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[i][j][q] += __hip_atomic_load(wsl + (size_t)((i * NJ + j) * 4 + q) * NTH, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
    }
    bf16_t* s_tile = reinterpret_cast<bf16_t*>(smem);  // [BM][BN] bf16
    const bool want_sums = p.colsum != nullptr;
    unsigned vbits = 0;
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            vbits |= (unsigned)(s_rowoff[wave * 16 * MA + 16 * i + (VPERM ? 4 * q + kg : 4 * kg + q)] >= 0) << (4 * i + q);
    // VALU per accumulator: half a v_cvt_pk_bf16_f32 (rows q, q + 1 of one cout share it), one 2-byte LDS write, and for the
    // column sums one add + one fma; the validity select only in waves that own rows outside the volume (ragged tiles).
    auto tile_out = [&](auto masked_tag, auto sums_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value, SUMS = decltype(sums_tag)::value;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = hk_col<NJ, DIRECT>(j, r16);
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int i = 0; i < MA; ++i) {
                bf16_t* trow = s_tile + (wave * 16 * MA + 16 * i + 4 * kg) * BNP + col;
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    const float v0 = acc[i][j][q], v1 = acc[i][j][q + 1];
                    const uint32_t pk = pack_bf16x2_v(f32x2_t{v0, v1});
                    trow[q * BNP] = (bf16_t)(pk & 0xffffu);
                    trow[(q + 1) * BNP] = (bf16_t)(pk >> 16);
                    if (SUMS) {
                        const float m0 = (!MASKED || ((vbits >> (4 * i + q)) & 1u)) ? v0 : 0.0f;
                        const float m1 = (!MASKED || ((vbits >> (4 * i + q + 1)) & 1u)) ? v1 : 0.0f;
                        s1 += m0;
                        s2 = __builtin_fmaf(m0, m0, s2);
                        s1 += m1;
                        s2 = __builtin_fmaf(m1, m1, s2);
                    }
                }
            }
            if (SUMS) {
                s1 += __shfl_xor(s1, 16);
                s2 += __shfl_xor(s2, 16);
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (kg == 0) {
                    s_cs[(wave * BN + col) * 2 + 0] = s1;
                    s_cs[(wave * BN + col) * 2 + 1] = s2;
                }
            }
        }
    };
    bf16_t* y = reinterpret_cast<bf16_t*>(p.y);
    if constexpr (DIRECT) {
        // direct form: lane (kg, r16) owns couts NJ r16 .. NJ r16 + NJ - 1 of voxel rows 16 i + 4 kg + q: one 16-byte store per
        // (i, q) straight from the accumulators; the column sums of the lane's NJ couts are taken in the same sweep (the
        // accumulators of a row die with its store: a separate sums pass keeps all 128 alive next to the store addresses and
        // made hipcc spill)
        static_assert(NJ == 8, "a lane's couts of one voxel are one 16-byte piece");
        const int co = n0 + r16 * NJ;
        const bool co_ok = co < p.Cout;
        auto direct_out = [&](auto masked_tag, auto sums_tag) {
            constexpr bool MASKED = decltype(masked_tag)::value, SUMS = decltype(sums_tag)::value;
            float s1[NJ], s2[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) s1[j] = s2[j] = 0.0f;
#pragma unroll
            for (int i = 0; i < MA; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long long off = s_rowoff[wave * 16 * MA + 16 * i + (VPERM ? 4 * q + kg : 4 * kg + q)];
                    if (SUMS) {
                        const bool live = !MASKED || ((vbits >> (4 * i + q)) & 1u);
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            const float m = live ? acc[i][j][q] : 0.0f;
                            s1[j] += m;
                            s2[j] = __builtin_fmaf(m, m, s2[j]);
                        }
                    }
                    if (off >= 0 && co_ok && !(CTSI_DBG(p.dbg, 4))) {
                        typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
                        u4_t w4;
                        w4.x = pack_bf16x2_v(f32x2_t{acc[i][0][q], acc[i][1][q]});
                        w4.y = pack_bf16x2_v(f32x2_t{acc[i][2][q], acc[i][3][q]});
                        w4.z = pack_bf16x2_v(f32x2_t{acc[i][4][q], acc[i][5][q]});
                        w4.w = pack_bf16x2_v(f32x2_t{acc[i][6][q], acc[i][7][q]});
                        if (p.nt_store)       // streaming (non-temporal) stores: see ctsi_conv_fwd
                            __builtin_nontemporal_store(w4, reinterpret_cast<u4_t*>(y + off + co));
                        else
                            *reinterpret_cast<u4_t*>(y + off + co) = w4;
                    }
                }
            if (SUMS) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float a1 = s1[j], a2 = s2[j];
                    a1 += __shfl_xor(a1, 16);
                    a2 += __shfl_xor(a2, 16);
                    a1 += __shfl_xor(a1, 32);
                    a2 += __shfl_xor(a2, 32);
                    if (kg == 0) {
                        s_cs[(wave * BN + r16 * NJ + j) * 2 + 0] = a1;
                        s_cs[(wave * BN + r16 * NJ + j) * 2 + 1] = a2;
                    }
                }
            }
        };
        using T = std::true_type;
        using F = std::false_type;
        const bool ragged = __builtin_amdgcn_ballot_w64(vbits != (1u << (4 * MA)) - 1u) != 0ull;   // wave-uniform
        if (!want_sums) direct_out(F{}, F{});
        else if (ragged) direct_out(T{}, T{});
        else direct_out(F{}, T{});
    } else {
        using T = std::true_type;
        using F = std::false_type;
        const bool ragged = __builtin_amdgcn_ballot_w64(vbits != (1u << (4 * MA)) - 1u) != 0ull;   // wave-uniform
        if (!want_sums) tile_out(F{}, F{});
        else if (ragged) tile_out(T{}, T{});
        else tile_out(F{}, T{});
    }
    HK_STAMP(4);
    // Every wave streams out ITS OWN 16 MA rows of the tile (it wrote exactly those: no barrier before), then one barrier and the
    // cross-wave column sums while the stores drain.  (The other order -- barrier and sums first, nothing behind the last store --
    // measured 1.8 k cycles per tile slower: the store drain then sits in the hand-over to the next block.)
    constexpr int CPR = BN / 8;
    if constexpr (!DIRECT) {
        constexpr int RPW = 16 * MA;
#pragma unroll 4
        for (int k = 0; k < RPW * CPR / 64; ++k) {
            const int c = k * 64 + lane;
            const int row = wave * RPW + c / CPR, ch = c % CPR;
            const long long off = s_rowoff[row];
            const int co = n0 + ch * 8;
            if (off >= 0 && co < p.Cout && !(CTSI_DBG(p.dbg, 4))) {
                const uint4 v = *reinterpret_cast<const uint4*>(s_tile + row * BNP + ch * 8);
                if (p.nt_store) {     // streaming (non-temporal) stores: see ctsi_conv_fwd
                    typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
                    const u4_t w4 = {v.x, v.y, v.z, v.w};
                    __builtin_nontemporal_store(w4, reinterpret_cast<u4_t*>(y + off + co));
                } else {
                    *reinterpret_cast<uint4*>(y + off + co) = v;
                }
            }
        }
    }
    HK_STAMP(5);
    if (want_sums) {
        __syncthreads();
        if (tid < BN) {
            const int col = tid;
            float t1 = 0.0f, t2 = 0.0f;
#pragma unroll
            for (int q = 0; q < NWAVE; ++q) {
                t1 += s_cs[(q * BN + col) * 2 + 0];
                t2 += s_cs[(q * BN + col) * 2 + 1];
            }
            const long long slab = (long long)(TR ? 4 : 1) * p.mtiles * CoutPad;   // [class][m-tile][cout_pad], as the gather kernel's
            const long long tg = (long long)cls * p.mtiles + mt;
            p.colsum[tg * CoutPad + n0 + col] = t1;
            p.colsum[slab + tg * CoutPad + n0 + col] = t2;
        }
    }
    HK_STAMP(6);
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- weight packing: fp32 (cout, cin, 3,3,3) -> bf16 [entry q = chunk16 * 27 + tap][cout_pad][16], zero entries up to whole steps;
//      ConvTranspose3d (cin, cout, 3,4,4) -> [class][entry q = chunk16 * 12 + t][cout_pad][16], t = (a * 2 + b) * 2 + c with
//      kernel taps k_d = 2 - a (halo slices ascending), k_h = (py ? {2, 0} : {1, 3})[b], k_w likewise (the tap order conv3_halo_k32_kernel<TR> walks)
__global__ void conv3_halo_k32_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Cout, int CoutPad,
                                           int CinW, int nchunks, long long per_class, int form, int bn, int direct) {
    // form 0: 3x3x3; 1: ConvTranspose3d (3,4,4)/(1,2,2) (4 class images); 2: Conv3d (3,4,4)/(1,2,2) (nchunks = 4 virtual chunks
    // per 16 input channels: vc = 4 * chunk + class, entry t = (a * 2 + b) * 2 + c <-> k_d = a, k_h = 2 b + 1 - py, k_w = 2 c + 1 - px)
    const bool transposed = form == 1;
    const int taps = form ? 12 : 27;
    const int Q = nchunks * taps;
    const long long total = per_class * (transposed ? 4 : 1);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int cls = (int)(idx / per_class);
        const long long in_class = idx - cls * per_class;
        const int e = (int)(in_class & 15);
        const long long row = in_class >> 4;           // q * CoutPad + cout
        const int co_p = (int)(row % CoutPad);           // packed row
        // direct epilogue form: packed row 16 j + r of a bn-cout n-tile holds cout (bn / 16) r + j (see hk_col)
        const int co = direct ? (co_p / bn) * bn + (co_p % 16) * (bn / 16) + (co_p % bn) / 16 : co_p;
        const long long q = row / CoutPad;
        float v = 0.0f;
        if (q < Q) {
            const int t = (int)(q % taps), cc = (int)(q / taps);
            const int ci = (form == 2 ? (cc >> 2) : cc) * 16 + e;
            if (co < Cout && ci < CinW) {
                if (form == 2) {
                    const int py = (cc >> 1) & 1, px = cc & 1;
                    const int a = t >> 2, b = (t >> 1) & 1, c = t & 1;
                    v = w[((long long)co * CinW + ci) * 48 + (a * 4 + 2 * b + 1 - py) * 4 + 2 * c + 1 - px];
                } else if (transposed) {
                    const int py = cls >> 1, px = cls & 1;
                    const int a = 2 - (t >> 2), b = (t >> 1) & 1, c = t & 1;   // i_d = o_d + 1 - k_d: k_d descending = halo slices ascending
                    const int ky = py ? (b ? 0 : 2) : (b ? 3 : 1), kx = px ? (c ? 0 : 2) : (c ? 3 : 1);
                    v = w[((long long)ci * Cout + co) * 48 + (a * 4 + ky) * 4 + kx];
                } else {
                    v = w[((long long)co * CinW + ci) * 27 + t];
                }
            }
        }
        out[idx] = f32_to_bf16(v);
    }
}

// form: 0 = 3x3x3, 1 = ConvTranspose3d (3,4,4)/(1,2,2), 2 = strided Conv3d (3,4,4)/(1,2,2)
extern "C" size_t ctsi_conv3_halo_k32_weight_bytes(int cin, int cout_pad, int bn, int form) {
    const long long q = (long long)(cin / 16) * (form == 2 ? 48 : form == 1 ? 12 : 27), step = bn == 64 ? 8 : 4;   // entries per step: 2 x UPS
    return (size_t)(((q + step - 1) / step) * step * cout_pad * 32) * (form == 1 ? 4 : 1);
}

extern "C" int ctsi_conv3_halo_k32_pack(const float* w, void* packed, int cout, int cout_pad, int cin, int cin_w, int bn,
                                        int form, int direct, void* stream) {
    CTSI_CHECK_ARG(w && packed && cin % 16 == 0 && (bn == 64 || bn == 128) && cout_pad % bn == 0 && (form != 1 || cin_w == cin) &&
                       form >= 0 && form <= 2,
                   "ctsi_conv3_halo_k32_pack: bad arguments");
    const int nchunks = (cin / 16) * (form == 2 ? 4 : 1);
    const int transposed = form;
    const long long total = (long long)ctsi_conv3_halo_k32_weight_bytes(cin, cout_pad, bn, form) / 2;
    const long long per_class = form == 1 ? total / 4 : total;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(conv3_halo_k32_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)packed,
                       cout, cout_pad, cin_w, nchunks, per_class, transposed, bn, direct);
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// Which forms store straight from the accumulators (measured per form, see hk_col): the 512-voxel tiles and the 24- / 12-wide
// 384-voxel tiles in every form, and the 384-voxel split-K form of the plain conv; the other 384-voxel forms keep the staged epilogue.
// (-DHK_STAGED_EPILOGUE, `make staged`: the staged epilogue everywhere, for A/B timing through CTSI_LIB.)
extern "C" int ctsi_conv3_halo_k32_direct(int tile, int ksplit, int ds) {
#ifdef HK_STAGED_EPILOGUE
    return 0;
#else
    const bool t512 = tile == 0 || tile == 2, narrow = tile == 6 || tile == 7;
    return (t512 || narrow || (tile == 5 && ksplit == 2 && !ds)) ? 1 : 0;
#endif
}

template <int TD, int TH, int TW, int BN, int UPS, bool TR, bool SK = false, bool DS = false>
static void hk_launch(const Conv3HaloParams* hp, hipStream_t stream) {
    using Cfg = HkCfg<TD, TH, TW, BN, UPS>;
#ifdef HK_STAGED_EPILOGUE
    constexpr bool DIRECT = false;
#else
    constexpr bool DIRECT = TD * TH * TW == 512 || TW == 24 || TW == 12 || (SK && !DS);
#endif
    auto k = conv3_halo_k32_kernel<TD, TH, TW, BN, UPS, TR, SK, DS, DIRECT>;
    static CtsiPerDeviceOnce attr_once;
    if (attr_once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3((TR ? 4 : 1) * (SK ? 2 : 1) * hp->mtiles * hp->ntiles_n), dim3(Cfg::NTH), Cfg::LDS_BYTES, stream,
                       *hp);
}

extern "C" size_t ctsi_conv3_halo_k32_splitk_bytes(int tiles) {   // [2 ints per tile: ticket, flag | pad to 256 B][partials]
    const size_t sync = ((size_t)tiles * 8 + 255) / 256 * 256;
    return sync + (size_t)tiles * (4 * 8 * 4) * 512 * sizeof(float);   // sized for the 512-voxel tile (128 accumulators per lane)
}

extern "C" int ctsi_conv3_halo_k32_launch(const Conv3HaloParams* hp, int tile /* 0: 4x4x32, 2: 4x8x16, 3: 3x4x32, 5: 3x8x16, 6: 4x4x24, 7: 8x4x12 */,
                                          int bn, void* stream) {
    CTSI_CHECK_ARG(bn == 128 && (tile == 0 || tile == 2 || tile == 3 || tile == 5 || tile == 6 || tile == 7),
                   "ctsi_conv3_halo_k32_launch: bad BN %d / tile %d", bn, tile);
    if (tile == 6 || tile == 7) {   // 384-voxel tiles for 24- / 12-wide planes (2-way split-K for the plain conv only)
        CTSI_CHECK_ARG(!(hp->ds && hp->tr) && !(hp->ksplit == 2 && (hp->ds || hp->tr)),
                       "ctsi_conv3_halo_k32_launch: tile %d has no split-K form for strided / transposed layers", tile);
        if (hp->ksplit == 2) {
            CTSI_CHECK_ARG(hp->sk_ws && hp->sk_sync && hp->nchunks % 8 == 0, "ctsi_conv3_halo_k32_launch: split-K needs its workspace");
            if (tile == 6)
                hk_launch<4, 4, 24, 128, 2, false, true>(hp, (hipStream_t)stream);
            else
                hk_launch<8, 4, 12, 128, 2, false, true>(hp, (hipStream_t)stream);
        } else if (hp->ds) {
            CTSI_CHECK_ARG(hp->Hi == 2 * hp->Ho && hp->Wi == 2 * hp->Wo && hp->C2 == 0 && hp->nchunks % 4 == 0,
                           "ctsi_conv3_halo_k32_launch: the Downsample form needs even input planes and one source");
            if (tile == 6)
                hk_launch<4, 4, 24, 128, 2, false, false, true>(hp, (hipStream_t)stream);
            else
                hk_launch<8, 4, 12, 128, 2, false, false, true>(hp, (hipStream_t)stream);
        } else if (hp->tr) {
            if (tile == 6)
                hk_launch<4, 4, 24, 128, 2, true>(hp, (hipStream_t)stream);
            else
                hk_launch<8, 4, 12, 128, 2, true>(hp, (hipStream_t)stream);
        } else if (tile == 6) {
            hk_launch<4, 4, 24, 128, 2, false>(hp, (hipStream_t)stream);
        } else {
            hk_launch<8, 4, 12, 128, 2, false>(hp, (hipStream_t)stream);
        }
        CTSI_LAUNCH_CHECK();
        return CTSI_OK;
    }
    if (hp->ds) {   // strided Conv3d (3,4,4)/(1,2,2): nchunks counts virtual chunks (4 per 16 input channels)
        CTSI_CHECK_ARG(!hp->tr && hp->Hi == 2 * hp->Ho && hp->Wi == 2 * hp->Wo && hp->C2 == 0 && hp->nchunks % 4 == 0,
                       "ctsi_conv3_halo_k32_launch: the Downsample form needs even input planes and one source");
        if (hp->ksplit == 2) {
            CTSI_CHECK_ARG(tile == 5 && hp->sk_ws && hp->sk_sync && hp->nchunks % 8 == 0, "ctsi_conv3_halo_k32_launch: split-K needs its workspace");
            hk_launch<3, 8, 16, 128, 2, false, true, true>(hp, (hipStream_t)stream);
        } else if (tile == 5) {
            hk_launch<3, 8, 16, 128, 2, false, false, true>(hp, (hipStream_t)stream);
        } else if (tile == 3) {
            hk_launch<3, 4, 32, 128, 2, false, false, true>(hp, (hipStream_t)stream);
        } else if (tile == 2) {
            hk_launch<4, 8, 16, 128, 2, false, false, true>(hp, (hipStream_t)stream);
        } else {
            hk_launch<4, 4, 32, 128, 2, false, false, true>(hp, (hipStream_t)stream);
        }
    } else if (hp->tr && tile == 5) {
        hk_launch<3, 8, 16, 128, 2, true>(hp, (hipStream_t)stream);
    } else if (hp->tr && tile == 3) {
        hk_launch<3, 4, 32, 128, 2, true>(hp, (hipStream_t)stream);
    } else if (tile == 5) {          // 3x8x16 = 384 voxels for 16-wide levels, plain or 2-way split-K
        if (hp->ksplit == 2) {
            CTSI_CHECK_ARG(hp->sk_ws && hp->sk_sync && hp->nchunks % 8 == 0, "ctsi_conv3_halo_k32_launch: split-K needs its workspace");
            hk_launch<3, 8, 16, 128, 2, false, true>(hp, (hipStream_t)stream);
        } else {
            hk_launch<3, 8, 16, 128, 2, false, false>(hp, (hipStream_t)stream);
        }
    } else if (tile == 3) {
        hk_launch<3, 4, 32, 128, 2, false>(hp, (hipStream_t)stream);
    } else if (tile == 0 && hp->ksplit == 2) {   // 4x4x32 with 2-way split-K
        CTSI_CHECK_ARG(hp->sk_ws && hp->sk_sync && hp->nchunks % 8 == 0, "ctsi_conv3_halo_k32_launch: split-K needs its workspace");
        hk_launch<4, 4, 32, 128, 2, false, true>(hp, (hipStream_t)stream);
    } else if (hp->tr) {
        if (tile == 2)
            hk_launch<4, 8, 16, 128, 2, true>(hp, (hipStream_t)stream);
        else
            hk_launch<4, 4, 32, 128, 2, true>(hp, (hipStream_t)stream);
    } else if (tile == 2) {
        hk_launch<4, 8, 16, 128, 2, false>(hp, (hipStream_t)stream);
    } else {
        hk_launch<4, 4, 32, 128, 2, false>(hp, (hipStream_t)stream);
    }
    CTSI_LAUNCH_CHECK();
    return CTSI_OK;
}

// Number of device-side errors recorded since the last call with reset != 0 (0 on a healthy run), and the split-K tile of the
// last one.  SYNCHRONOUS (a 8-byte device-to-host copy): call it where the host reads results anyway.
extern "C" int ctsi_debug_k32_stamps(unsigned long long* out, int nblocks) {   // timing-only (CTSI_DEBUG_FLAGS & 4096)
    if (nblocks > HK_NSTAMP) nblocks = HK_NSTAMP;
    CTSI_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hk_stamps), (size_t)nblocks * 8 * sizeof(unsigned long long)));
    return CTSI_OK;
}

extern "C" int ctsi_device_error_status(unsigned int* count, unsigned int* detail, int reset) {
    CTSI_CHECK_ARG(count, "ctsi_device_error_status: null argument");
    unsigned int h[2] = {0u, 0u};
    CTSI_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_hk_device_error), sizeof(h)));
    *count = h[0];
    if (detail) *detail = h[1];
    if (reset && h[0]) {
        const unsigned int z[2] = {0u, 0u};
        CTSI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_hk_device_error), z, sizeof(z)));
    }
    return CTSI_OK;
}
