// fp16 3x3 convolution on the gfx950 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulation) -- the first piece of
// the fp16 path (BASELINE configs 3-4).  Replaces nn.Conv2d(3x3) + LeakyReLU(0.1) (reference models/PWCNet.py:26-33)
// for half-precision activations.
//
// Activation layout: channel-blocked  [B][Cg = ceil(C/8)][H][W][8] halves  ("c8"), so that the 8 consecutive k a lane
// feeds to one MFMA operand are ONE 16-byte LDS read and one pixel is one 16-byte LDS-DMA piece (no alignment cases:
// halo, zero padding and ragged edges are per-piece range checks).  Channels past C inside the last group are zero
// and are written as zero.  Only the batch stride is free, so a tensor may be a channel-group slice of an arena.
//
// GEMM view, per (tap, pair of channel groups): D[cout 32][pixel 32] += A[cout][k 16] * B[k 16][pixel], k = 8*kh + j
//   <-> channel 8*(2*cgp + kh) + j;  lane l: row/col = l & 31, kh = l >> 5.
//   packed filters [cgp][tap][kh][CoutP][8]  -> A fragment = one ds_read_b128, 32 lanes x 16 B contiguous
//   staged input   [kh][row][col][8]         -> B fragment = one ds_read_b128, 32 lanes x 16 B contiguous
// Workgroup = 4 MFMA waves + 1 loader wave (the fp32 kernels showed that a wave cannot issue LDS-DMA and MFMA
// back to back; here the bytes per MFMA cycle are 8x higher).  The loader runs a 3-slot ring of 16-channel chunks
// (input halo tile + filter slab; 2 slots for the wide halos of dilation 8 / 16) under a counted vmcnt, one barrier per
// chunk.  Tile = (8 rows x 32 cols) x 32*MT couts.
// Epilogue: bias (fp32), LeakyReLU, round to half, 8-byte stores that interleave to 512 contiguous bytes per wave.
#include <stdlib.h>

#include "pwc_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBlock = 320;          // waves 0..3: MFMA, wave 4: loader
constexpr int kModeLeaky = 1, kModeOutF32 = 2, kModeSplitW = 4;

constexpr int kTileW = 32;
constexpr unsigned kOOB = 0x80000000u;

// R = ring depth (3 where the LDS allows it, 2 for the wide halos of dilation 8 / 16).  Dilation 16 stages the three
// ky row-sets separately (3 x 8 rows instead of 40).
// NT = rows per MFMA wave: tile = 4*NT rows x 32 cols.  NT = 4 (16-row tiles, 2-slot ring) halves the filter bytes a
// workgroup pulls from L2 per flop -- at ~1 PFLOP/s the 8-row tile moves 5.6 TB/s from L2, mostly filter slabs.
template <int MT, int NT, int S, int D, int R>
struct G16 {
    static_assert(R == 2 || R == 3, "ring depth");
    static constexpr int kRing = R;
    static constexpr int kTileH = 4 * NT;
    static constexpr bool kRowSep = (D >= 16);
    static constexpr int kInH = kRowSep ? 3 * kTileH : (kTileH - 1) * S + 2 * D + 1;
    static constexpr int kInW = (kTileW - 1) * S + 2 * D + 1;
    static constexpr int kInPieces = 2 * kInH * kInW;             // [kh][row][col] 16-byte pieces per chunk
    static constexpr int kInInstr = (kInPieces + 63) / 64;
    static constexpr int kCoutT = 32 * MT;
    static constexpr int kWPieces = 18 * kCoutT;                   // [tap][kh][cout]
    static constexpr int kWInstr = (kWPieces + 63) / 64;
    static constexpr int kInstr = kInInstr + kWInstr;              // LDS-DMA instructions per chunk (1 KiB each)
    static constexpr int kWOffBytes = kInInstr * 1024;
    static constexpr int kSlotBytes = kInstr * 1024;
    static constexpr int kSmem = R * kSlotBytes;
    static constexpr bool kValid = kSmem <= 160 * 1024 && (!kRowSep || S == 1);
};

// One-dimensional grid of nblk tiles x ngroups cout groups (ngroups = cout tiles of the layer / MT, recomputed from CoutP).
// Workgroups i, i+8, i+16, ... run on the same XCD: each XCD gets a contiguous run of tiles (the halo rows / columns re-read by
// neighbouring tiles hit in its L2) and -- round 3 -- runs the cout groups of one tile BACK TO BACK, so that the second (third, fourth:
// split filters) group finds the input tile in that L2 instead of fetching it from HBM again: the 8-wave kernel's two 64-cout
// groups cost 1.90x the algorithmic traffic with the groups in blockIdx.y (all tiles of group 0, then all of group 1).
// Speed only; any mapping is correct.  -DPWC_F16_NO_XCD_MAP: plain order.
__device__ __forceinline__ void block_to_tile(int ngroups, int &bid, int &g) {
    const int nblk = (int)gridDim.x / ngroups;
    const int id = (int)blockIdx.x;
#ifndef PWC_F16_NO_XCD_MAP
    if ((nblk & 7) == 0) {
        const int slot = id >> 3;
        g = slot % ngroups;
        bid = (id & 7) * (nblk >> 3) + slot / ngroups;
        return;
    }
#endif
    g = id / nblk;
    bid = id % nblk;
}

// loader wave: start the LDS-DMA of one 16-channel chunk (input halo tile, then the filter slab) into its ring slot
template <class G>
__device__ __forceinline__ void issue_f16(const _Float16 *xb, const _Float16 *wp, int chunk, int Cg, int plane, int CoutP,
                                          unsigned char *smem, const unsigned *off) {
    const int cgv = min(2, Cg - 2 * chunk);               // ragged last pair: kh = 1 is range-checked to zero
    const pwc::v4i32 rin = pwc::make_rsrc(xb + (int64_t)chunk * 2 * plane * 8, cgv * plane * 16);
    const pwc::v4i32 rw = pwc::make_rsrc(wp + (int64_t)chunk * 18 * CoutP * 8, 18 * CoutP * 16);
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)pwc::lds_addr(smem + (chunk % G::kRing) * G::kSlotBytes));
#pragma unroll
    for (int i = 0; i < G::kInInstr; ++i) pwc::dma_b128(rin, base + i * 1024, off[i]);
#pragma unroll
    for (int i = 0; i < G::kWInstr; ++i) pwc::dma_b128(rw, base + G::kWOffBytes + i * 1024, off[G::kInInstr + i]);
}

template <int MT, int NT, int S, int D, int R>
__global__ void __launch_bounds__(kBlock)
conv3x3_f16_kernel(const _Float16 *__restrict__ x, const _Float16 *__restrict__ wp, const float *__restrict__ bias,
                   void *__restrict__ yv, int Cg, int H, int W, int Cout, int CoutP, int Ho, int Wo,
                   int tiles_x, int tiles_y, int64_t bsx, int64_t bsy, float slope, int mode) {
    using G = G16<MT, NT, S, D, R>;
    constexpr int kNT = NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    int bid, g;
    block_to_tile((CoutP / 32 + MT - 1) / MT, bid, g);
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int ox0 = tx * kTileW;
    const int oy0 = ty * G::kTileH;
    const int plane = H * W;
    const int nchunks = (Cg + 1) / 2;
    const _Float16 *xb = x + (int64_t)b * bsx;

    if (wave == 4) {
        // ================= loader wave =====================================================================
        __builtin_amdgcn_s_setprio(3);
        unsigned off[G::kInstr];
#pragma unroll
        for (int i = 0; i < G::kInInstr; ++i) {
            const int p = i * 64 + lane;
            const int kh = p / (G::kInH * G::kInW);
            const int rem = p % (G::kInH * G::kInW);
            const int r = rem / G::kInW;
            const int iy = G::kRowSep ? oy0 + (r % G::kTileH) - D + (r / G::kTileH) * D : oy0 * S - D + r;
            const int ix = ox0 * S - D + rem % G::kInW;
            const bool ok = (p < G::kInPieces) && (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);
            off[i] = ok ? (unsigned)(kh * plane + iy * W + ix) * 16u : kOOB;
        }
#pragma unroll
        for (int i = 0; i < G::kWInstr; ++i) {
            const int q = i * 64 + lane;
            const int row = q / G::kCoutT;                    // tap*2 + kh
            const int co = g * G::kCoutT + q % G::kCoutT;
            off[G::kInInstr + i] = (q < G::kWPieces && co < CoutP) ? (unsigned)(row * CoutP + co) * 16u : kOOB;
        }
        issue_f16<G>(xb, wp, 0, Cg, plane, CoutP, smem, off);
        if (R == 3 && nchunks > 1) issue_f16<G>(xb, wp, 1, Cg, plane, CoutP, smem, off);
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            // chunk has landed; with R == 3 the next one may stay in flight
            if (R == 3 && chunk + 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(G::kInstr) : "memory");
            else                               asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // consumers may read slot chunk%R; they are done with (chunk-1)%R
            if (chunk + R - 1 < nchunks) issue_f16<G>(xb, wp, chunk + R - 1, Cg, plane, CoutP, smem, off);
        }
        return;
    }

    // ================= MFMA waves: wave w owns rows 2w, 2w+1 of the tile ======================================
    const int col = lane & 31;
    const int kh = lane >> 5;
    const bool do_leaky = mode & kModeLeaky, out_f32 = mode & kModeOutF32, split_w = mode & kModeSplitW;
    f32x16 acc[MT][kNT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            // split filters: every 32-row block carries 16 couts -- rows 0..15 the filters rounded to half, rows 16..31 their
            // rounding residuals (no bias there)
            const int co = split_w ? (g * G::kCoutT + mt * 32) / 2 + (j & 3) + 8 * ((j >> 2) & 1) + 4 * kh
                                   : g * G::kCoutT + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh;
            float bv = bias[min(co, Cout - 1)];
            if (split_w && j >= 8) bv = 0.f;
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) acc[mt][nt][j] = bv;
        }
    }
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned char *cur = smem + (chunk % R) * G::kSlotBytes;
        const h8 *in = reinterpret_cast<const h8 *>(cur) + (kh * G::kInH + wave * kNT * (G::kRowSep ? 1 : S)) * G::kInW + col * S;
        const h8 *ws = reinterpret_cast<const h8 *>(cur + G::kWOffBytes) + kh * G::kCoutT + col;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            h8 a[MT], bv[kNT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = ws[tap * 2 * G::kCoutT + mt * 32];
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt)
                bv[nt] = in[(G::kRowSep ? nt + ky * G::kTileH : nt * S + ky * D) * G::kInW + kx * D];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < kNT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], bv[nt], acc[mt][nt], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // LDS reads retired before the slot can be refilled
    }

    // ---- epilogue: register quad q of accumulator tile mt = channels 4*kh .. 4*kh+3 of output group (..)/8 + q ------
    const int ox = ox0 + col;
    const int64_t oplane = (int64_t)Ho * Wo;
    const int cg_out = (Cout + 7) / 8;
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt) {
        const int oy = oy0 + wave * kNT + nt;
        if (oy >= Ho || ox >= Wo) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (split_w && q >= 2) continue;                       // 16 couts = 2 output groups per 32-row block
                const int cg = split_w ? (g * G::kCoutT + mt * 32) / 16 + q : (g * G::kCoutT + mt * 32) / 8 + q;
                if (cg >= cg_out) continue;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[mt][nt][4 * q + i];
                    // split filters: row r + 16 = the sum over the filters' low halves scaled by 2^11 (same lane, register + 8)
                    if (split_w) v[i] += acc[mt][nt][(4 * q + i + 8) & 15] * (1.0f / 2048.0f);
                    if (do_leaky) v[i] = pwc::leaky(v[i], slope);
                    if (cg * 8 + 4 * kh + i >= Cout) v[i] = 0.f;
                }
                const int64_t at = (int64_t)b * bsy + ((int64_t)cg * oplane + (int64_t)oy * Wo + ox) * 8 + 4 * kh;
                if (out_f32) {
                    *reinterpret_cast<float4 *>(static_cast<float *>(yv) + at) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    h4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = pwc::sat_half(v[i]);
                    *reinterpret_cast<h4 *>(static_cast<_Float16 *>(yv) + at) = o;
                }
            }
        }
    }
}

// ---- 8-wave variant for the large stride-1 layers (levels 2 and 3, context network) ----------------------------------
// Same LDS images and MFMA loop as above, other residency: a workgroup is EIGHT MFMA waves on a 16-row x 32-col tile (wave w
// owns rows 2w, 2w+1) x 32*MT couts, with a plain double buffer that every wave fills (each issues its 1/8 of the next
// chunk's LDS-DMA right after the barrier, as the fp32 kernel does) instead of a dedicated loader wave.  With MT <= 2 the
// workgroup needs <= 80 KiB of LDS and <= 128 registers, so TWO workgroups share a CU = four MFMA waves per SIMD: one
// wave's barrier / LDS latency / DMA issue is covered by the other three (PMC on the 5-wave kernel, profiles/
// r02_f16_pmc_diag.txt: its MFMA waves are parked 25 % of their life and the matrix pipe is busy 62 %).  The taller tile
// also halves the filter bytes pulled from L2 per flop.
// (A v_mfma_f32_16x16x32_f16 form of this kernel was built and measured 2-10 % slower: profiles/r02_f16_mfma_shape_experiment.txt.)
template <int MT, int D, int SH = 0>
struct G8 {
    static constexpr int kTileH = 16;
    static constexpr int kInH = kTileH + 2 * D;
    static constexpr int kInW = kTileW + 2 * D;
    static constexpr int kInPieces = 2 * kInH * kInW;              // [kh][row][col]
    static constexpr int kInWI = (kInPieces + 63) / 64;            // wave-instructions (1 KiB each)
    static constexpr int kCoutT = 32 * MT;
    static constexpr int kWRows = SH ? 20 : 18;                    // [tap][kh] rows of kCoutT pieces (+ 2 zero rows for SH)
    static constexpr int kWWI = kWRows * kCoutT / 64;
    static constexpr int kWI = kInWI + kWWI;
    static constexpr int kPer = (kWI + 7) / 8;                     // per wave and chunk
    static constexpr int kWOffBytes = kInWI * 1024;
    static constexpr int kSlotBytes = kWI * 1024;
    static constexpr int kSmem = 2 * kSlotBytes;
    static constexpr int kWavesPerSimd = (kSmem <= 80 * 1024 && MT <= 2) ? 4 : 2;
    static constexpr bool kValid = kSmem <= 160 * 1024 && (kWRows * kCoutT) % 64 == 0 && !(SH && MT > 3);
};

template <int MT, int D, int SH>
__global__ void __launch_bounds__(512, (G8<MT, D, SH>::kWavesPerSimd))
conv3x3_f16w8_kernel(const _Float16 *__restrict__ x, const _Float16 *__restrict__ wp, const float *__restrict__ bias,
                     void *__restrict__ yv, int Cg, int H, int W, int Cout, int CoutP,
                     int tiles_x, int tiles_y, int64_t bsx, int64_t bsy, float slope, int mode) {
    using G = G8<MT, D, SH>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    int bid, g;
    block_to_tile((CoutP / 32 + MT - 1) / MT, bid, g);
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int ox0 = tx * kTileW;
    const int oy0 = ty * G::kTileH;
    const int plane = H * W;
    const int nchunks = (Cg + 1) / 2;
    const _Float16 *xb = x + (int64_t)b * bsx;

    // wave-instruction ids of this wave: wave, wave + 8, ...; ids < kInWI read the input, the others the filter slab
    unsigned off[G::kPer];
#pragma unroll
    for (int j = 0; j < G::kPer; ++j) {
        const int id = j * 8 + wave;
        if (id < G::kInWI) {
            const int p = id * 64 + lane;
            const int kh = p / (G::kInH * G::kInW);
            const int rem = p % (G::kInH * G::kInW);
            const int iy = oy0 - D + rem / G::kInW;
            const int ix = ox0 - D + rem % G::kInW;
            const bool ok = (p < G::kInPieces) && (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);
            off[j] = ok ? (unsigned)(kh * plane + iy * W + ix) * 16u : kOOB;
        } else {
            const int q = (id - G::kInWI) * 64 + lane;
            const int row = q / G::kCoutT;                    // tap*2 + kh (18, 19: the zero rows of the 16x16x32 form)
            const int co = g * G::kCoutT + q % G::kCoutT;
            off[j] = (id < G::kWI && row < 18 && co < CoutP) ? (unsigned)(row * CoutP + co) * 16u : kOOB;
        }
    }
    auto issue = [&](int chunk) {
        const int cgv = min(2, Cg - 2 * chunk);
        const pwc::v4i32 rin = pwc::make_rsrc(xb + (int64_t)chunk * 2 * plane * 8, cgv * plane * 16);
        const pwc::v4i32 rw = pwc::make_rsrc(wp + (int64_t)chunk * 18 * CoutP * 8, 18 * CoutP * 16);
        const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)pwc::lds_addr(smem + (chunk & 1) * G::kSlotBytes));
#pragma unroll
        for (int j = 0; j < G::kPer; ++j) {
            const int id = j * 8 + wave;                       // wave-uniform
            if (id < G::kInWI)     pwc::dma_b128(rin, base + id * 1024, off[j]);
            else if (id < G::kWI)  pwc::dma_b128(rw, base + id * 1024, off[j]);
        }
    };

    const bool do_leaky = mode & kModeLeaky, out_f32 = mode & kModeOutF32, split_w = mode & kModeSplitW;
    const int64_t oplane = (int64_t)H * W;
    const int cg_out = (Cout + 7) / 8;

    {
        const int col = lane & 31;
        const int kh = lane >> 5;
        f32x16 acc[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = split_w ? (g * G::kCoutT + mt * 32) / 2 + (j & 3) + 8 * ((j >> 2) & 1) + 4 * kh
                                       : g * G::kCoutT + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh;
                const float bv = (split_w && j >= 8) ? 0.f : bias[min(co, Cout - 1)];     // split filters: see conv3x3_f16_kernel
                acc[mt][0][j] = bv;
                acc[mt][1][j] = bv;
            }
        }
        issue(0);
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of `chunk` has landed
            __builtin_amdgcn_s_barrier();                          // ... everyone's has; the other slot is no longer read
            asm volatile("" ::: "memory");
            if (chunk + 1 < nchunks) issue(chunk + 1);
            const unsigned char *cur = smem + (chunk & 1) * G::kSlotBytes;
            const h8 *in = reinterpret_cast<const h8 *>(cur) + (kh * G::kInH + wave * 2) * G::kInW + col;
            const h8 *ws = reinterpret_cast<const h8 *>(cur + G::kWOffBytes) + kh * G::kCoutT + col;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                h8 a[MT], bv[2];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = ws[tap * 2 * G::kCoutT + mt * 32];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) bv[nt] = in[(nt + ky * D) * G::kInW + kx * D];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt], bv[nt], acc[mt][nt], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // LDS reads retired before the next barrier frees the slot
        }

        const int ox = ox0 + col;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int oy = oy0 + wave * 2 + nt;
            if (oy >= H || ox >= W) continue;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (split_w && q >= 2) continue;
                    const int cg = split_w ? (g * G::kCoutT + mt * 32) / 16 + q : (g * G::kCoutT + mt * 32) / 8 + q;
                    if (cg >= cg_out) continue;
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[i] = acc[mt][nt][4 * q + i];
                        if (split_w) v[i] += acc[mt][nt][(4 * q + i + 8) & 15] * (1.0f / 2048.0f);
                        if (do_leaky) v[i] = pwc::leaky(v[i], slope);
                        if (cg * 8 + 4 * kh + i >= Cout) v[i] = 0.f;
                    }
                    const int64_t at = (int64_t)b * bsy + ((int64_t)cg * oplane + (int64_t)oy * W + ox) * 8 + 4 * kh;
                    if (out_f32) {
                        *reinterpret_cast<float4 *>(static_cast<float *>(yv) + at) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        h4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = pwc::sat_half(v[i]);
                        *reinterpret_cast<h4 *>(static_cast<_Float16 *>(yv) + at) = o;
                    }
                }
            }
        }
    }
}

// wp[cgp][tap][kh][CoutP][8] <- w[co][ci = 8*(2*cgp + kh) + j][tap]   (zero outside Cin / Cout)
// split = 1 (CoutP = 32 * ceil(Cout / 16)): every 32-row block carries 16 couts -- row r < 16 holds filter 16*block + r rounded
// to half, row r + 16 its rounding residual times 2^11 (exact scaling; keeps it out of the subnormal range), so that
// hi + lo / 2^11 carries ~22 bits of the filter; the two partial sums of a cout sit in the same lane of the MFMA tile
__global__ void __launch_bounds__(256)
pack3x3_f16_kernel(const float *__restrict__ w, _Float16 *__restrict__ wp, int Cin, int Cout, int CoutP, int64_t total, int split) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 7);
    int64_t t = i >> 3;
    const int co = (int)(t % CoutP);
    t /= CoutP;
    const int kh = (int)(t & 1);
    t >>= 1;
    const int tap = (int)(t % 9);
    const int cgp = (int)(t / 9);
    const int ci = 8 * (2 * cgp + kh) + j;
    float v = 0.f;
    const int cs = split ? (co >> 5) * 16 + (co & 15) : co;
    if (cs < Cout && ci < Cin) v = w[((int64_t)cs * Cin + ci) * 9 + tap];
    const _Float16 hi = pwc::sat_half(v);
    wp[i] = (split && (co & 16)) ? (_Float16)((v - (float)hi) * 2048.0f) : hi;
}

// [B][C][H][W] f32 -> [B][Cg][H][W][8] f16 (zero channel padding) and back
__global__ void __launch_bounds__(256)
nchw_to_c8_kernel(const float *__restrict__ x, _Float16 *__restrict__ y, int C, int Cg, int64_t plane, int64_t total,
                  int64_t bsx, int64_t bsy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per output pixel-group
    if (i >= total) return;
    const int64_t pix = i % plane;
    int64_t t = i / plane;
    const int cg = (int)(t % Cg);
    const int64_t b = t / Cg;
    h8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        o[j] = (c < C) ? pwc::sat_half(x[b * bsx + (int64_t)c * plane + pix]) : (_Float16)0.f;
    }
    *reinterpret_cast<h8 *>(y + b * bsy + ((int64_t)cg * plane + pix) * 8) = o;
}

// the same with the rounding residual kept: hi = half(x) into y_hi, lo = half(x - hi) into y_lo (two c8 tensors with the same
// group count) -- a consumer that gives both channel sets the same filters sees ~22-bit activations (strict mode hand-over)
__global__ void __launch_bounds__(256)
nchw_to_c8_hilo_kernel(const float *__restrict__ x, _Float16 *__restrict__ y_hi, _Float16 *__restrict__ y_lo, int C, int Cg,
                       int64_t plane, int64_t total, int64_t bsx, int64_t bs_hi, int64_t bs_lo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t pix = i % plane;
    int64_t t = i / plane;
    const int cg = (int)(t % Cg);
    const int64_t b = t / Cg;
    h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        const float v = (c < C) ? x[b * bsx + (int64_t)c * plane + pix] : 0.f;
        hi[j] = pwc::sat_half(v);
        const float r = v - (float)hi[j];                    // exact in fp32; inf - inf cannot occur (hi saturates)
        lo[j] = (r == r) ? pwc::sat_half(r) : (_Float16)0.f;  // a NaN input is already carried by hi
    }
    *reinterpret_cast<h8 *>(y_hi + b * bs_hi + ((int64_t)cg * plane + pix) * 8) = hi;
    *reinterpret_cast<h8 *>(y_lo + b * bs_lo + ((int64_t)cg * plane + pix) * 8) = lo;
}

__global__ void __launch_bounds__(256)
c8_to_nchw_kernel(const _Float16 *__restrict__ x, float *__restrict__ y, int C, int Cg, int64_t plane, int64_t total,
                  int64_t bsx, int64_t bsy) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t pix = i % plane;
    int64_t t = i / plane;
    const int cg = (int)(t % Cg);
    const int64_t b = t / Cg;
    const h8 v = *reinterpret_cast<const h8 *>(x + b * bsx + ((int64_t)cg * plane + pix) * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        if (c < C) y[b * bsy + (int64_t)c * plane + pix] = (float)v[j];
    }
}

struct Args16 {
    const _Float16 *x, *wp;
    const float *bias;
    void *y;
    int B, Cg, H, W, Cout, CoutP, Ho, Wo;
    int64_t bsx, bsy;
    float slope;
    int mode;
    hipStream_t stream;
};

template <int MT, int NT, int S, int D, int R>
int launch16(const Args16 &a) {
    using G = G16<MT, NT, S, D, R>;
    if constexpr (!G::kValid) {
        PWC_FAIL(PWC_EINVAL, "pwc_conv2d_f16_fwd: internal: MT=%d does not fit for stride %d dilation %d", MT, S, D);
    } else {
        const int tiles_x = (a.Wo + kTileW - 1) / kTileW;
        const int tiles_y = (a.Ho + G::kTileH - 1) / G::kTileH;
        const int64_t nblk = (int64_t)a.B * tiles_x * tiles_y;
        const int groups = (a.CoutP / 32 + MT - 1) / MT;
        if (nblk * groups > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_f16_fwd: grid too large");
        auto kern = conv3x3_f16_kernel<MT, NT, S, D, R>;
        static pwc::LdsAttrOnce attr;
        if (const int rc = pwc::ensure_lds_attr(attr, reinterpret_cast<const void *>(kern), G::kSmem, "pwc_conv2d_f16_fwd")) return rc;
        // short-K layers (pyramid level 1: one or two chunks) only need as many ring slots as they have chunks: the
        // smaller LDS footprint lets several workgroups share a CU and cover each other's single DMA round trip
        const int nchunks = (a.Cg + 1) / 2;
        const int smem = (nchunks < R ? nchunks : R) * G::kSlotBytes;
        hipLaunchKernelGGL(kern, dim3((unsigned)(nblk * groups)), dim3(kBlock), smem, a.stream,
                           a.x, a.wp, a.bias, a.y, a.Cg, a.H, a.W, a.Cout, a.CoutP, a.Ho, a.Wo, tiles_x, tiles_y,
                           a.bsx, a.bsy, a.slope, a.mode);
        pwc::note_kernel("conv3x3_f16_kernel", MT, NT, S, D, R, 0);
        return pwc::check_launch("conv3x3_f16_kernel");
    }
}

template <int MT, int D, int SH>
int launch16w8s(const Args16 &a);

template <int MT, int D>
int launch16w8(const Args16 &a) { return launch16w8s<MT, D, 0>(a); }

template <int MT, int D, int SH>
int launch16w8s(const Args16 &a) {
    using G = G8<MT, D, SH>;
    static_assert(G::kValid, "8-wave tile does not fit the LDS");
    const int tiles_x = (a.Wo + kTileW - 1) / kTileW;
    const int tiles_y = (a.Ho + G::kTileH - 1) / G::kTileH;
    const int64_t nblk = (int64_t)a.B * tiles_x * tiles_y;
    const int groups = (a.CoutP / 32 + MT - 1) / MT;
    if (nblk * groups > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_f16_fwd: grid too large");
    auto kern = conv3x3_f16w8_kernel<MT, D, SH>;
    static pwc::LdsAttrOnce attr;
    if (const int rc = pwc::ensure_lds_attr(attr, reinterpret_cast<const void *>(kern), G::kSmem, "pwc_conv2d_f16_fwd")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nblk * groups)), dim3(512), G::kSmem, a.stream,
                       a.x, a.wp, a.bias, a.y, a.Cg, a.H, a.W, a.Cout, a.CoutP, tiles_x, tiles_y,
                       a.bsx, a.bsy, a.slope, a.mode);
    pwc::note_kernel("conv3x3_f16w8_kernel", MT, 2, 1, D, 2, SH ? 16 : 32);
    return pwc::check_launch("conv3x3_f16w8_kernel");
}

// Tile choice: widest cout tile that fits the LDS, 3-slot ring if possible, else 2 slots; 16-row tiles (NT = 4) for
// stride-1 layers with >= 64 couts once the grid is large enough to keep every CU busy with them.
// PWC_CONV16F_MT / PWC_CONV16F_RING / PWC_CONV16F_NT override (tuning).
inline bool skip_uneven() {      // PWC_CONV16F_UNEVEN=1 keeps the 96 + 32 split (A/B runs)
    static const bool on = [] { const char *e = getenv("PWC_CONV16F_UNEVEN"); return !(e && e[0] == '1'); }();
    return on;
}

template <int S, int D>
int dispatch16(const Args16 &a) {
    static const int forced_mt = [] { const char *e = getenv("PWC_CONV16F_MT"); return (e && *e) ? atoi(e) : 0; }();
    static const int forced_r = [] { const char *e = getenv("PWC_CONV16F_RING"); return (e && *e) ? atoi(e) : 0; }();
    static const int forced_nt = [] { const char *e = getenv("PWC_CONV16F_NT"); return (e && *e) ? atoi(e) : 0; }();
    const int t32 = a.CoutP / 32;
    // 8-wave kernel (stride 1, dilation <= 8) where it measured faster than the 5-wave one (profiles/r02_f16_w8_sweep.txt,
    // batch 16): dilation 1 with >= 64 couts (128 couts run as two 64-cout groups so that two workgroups share a CU:
    // conv2_0 -18 %, conv2_1 -16 %, dc_conv1 -4 %, level 3 -12..-17 %), 32-cout layers with a long K on large maps
    // (conv2_4 -4 %), and the 96-cout dilation-8 layer (dc_conv4 -23 %).  Dilation 2 / 4 need > 80 KiB per workgroup for
    // the double buffer (one workgroup per CU: no gain) and stay on the 5-wave kernel.
    // PWC_CONV16F_W8: unset = this rule, 0 = off, N = force cout tiles of 32*N (read per call: tests flip it).
    if constexpr (S == 1 && D <= 8) {
        const char *e = getenv("PWC_CONV16F_W8");
        const int w8 = (e && *e) ? atoi(e) : -1;
        const int64_t tiles16w = (int64_t)a.B * ((a.Wo + kTileW - 1) / kTileW) * ((a.Ho + 15) / 16);
        if (w8 != 0 && !((a.mode & kModeSplitW) && a.Cout <= 16) && tiles16w >= 256) {
            int mt = 0;
            if (w8 > 0) {
                mt = min(w8, t32 == 3 ? 3 : t32);
            } else if (D == 1) {
                if (t32 >= 4 || t32 == 2) mt = 2;
                else if (t32 == 3) mt = 3;
                else if (tiles16w >= 512 && (a.Cg + 1) / 2 >= 8) mt = 1;
            } else if (D == 8 && t32 == 3) {
                mt = 3;
            }
            if (mt >= 4 && t32 >= 4) { if constexpr (G8<4, D>::kValid) return launch16w8<4, D>(a); }
            if (mt == 3) { if constexpr (G8<3, D>::kValid) return launch16w8<3, D>(a); }
            if (mt >= 2 && t32 >= 2) { if constexpr (G8<2, D>::kValid) return launch16w8<2, D>(a); }
            if (mt >= 1) { if constexpr (G8<1, D>::kValid) return launch16w8<1, D>(a); }
        }
    }
    int want = forced_mt > 0 ? min(forced_mt, t32) : min(t32, 4);
    if ((a.mode & kModeSplitW) && a.Cout <= 16) want = 1;
    // small grids (levels 6-4, batch-1 inference): narrower cout tiles = more workgroups; a workgroup's K loop is then
    // bound by its DMA round trips instead of MT x as many MFMAs, and the tiny input is simply re-read per cout group
    const int64_t tiles8 = (int64_t)a.B * ((a.Wo + kTileW - 1) / kTileW) * ((a.Ho + 7) / 8);
    static const int kFillBlocks = [] { const char *e = getenv("PWC_CONV16F_FILL"); return (e && *e) ? atoi(e) : 256; }();
    static const int kFillBlocksD1 = [] { const char *e = getenv("PWC_CONV16F_FILL_D1"); return (e && *e) ? atoi(e) : 1024; }();   // dilation-1 layers: narrow tiles up to 1024 workgroups (measured below)
    if (forced_mt <= 0)
        while (want > 1 && tiles8 * ((t32 + want - 1) / want) < ((D == 1 && S == 1) ? kFillBlocksD1 : kFillBlocks)) {
            --want;
            if (want == 3 && t32 == 4 && skip_uneven()) want = 2;      // 128 couts as 96 + 32: the wide group finishes last
        }
    // short K (<= 12 chunks = 192 input channels: conv2_0, dc_conv2/3, the pyramid): a workgroup's prologue and epilogue
    // are a large share of its life, so prefer 64-cout tiles with a 2-slot ring -- two workgroups then share a CU and
    // cover each other (conv2_0 177 -> 142 us, dc_conv2 190 -> 150 us at batch 16; long-K layers lose with it)
    static const int short_k = [] { const char *e = getenv("PWC_CONV16F_SHORTK"); return (e && *e) ? atoi(e) : 12; }();
    const bool two_per_cu = forced_mt <= 0 && forced_r == 0 && (a.Cg + 1) / 2 <= short_k && t32 >= 2 && tiles8 >= 1024 && D <= 4;   // wide halos (dilation 8, 16): re-reading the input per 64-cout group costs more than it gains (dc_conv4 151 -> 252 us)
    if (two_per_cu && want > 2) want = 2;
    const int64_t tiles16 = (int64_t)a.B * ((a.Wo + kTileW - 1) / kTileW) * ((a.Ho + 15) / 16);
    // measured (batch 16, level 2): 96 couts 368 -> 336 us, 64 couts 312 -> 290 us; 128 couts (2 groups of 64) no gain
    const bool tall = forced_nt ? forced_nt == 4 : (S == 1 && D <= 4 && (t32 == 2 || t32 == 3) && tiles16 >= 512);
#define PWC_TRY(MT_, NT_, R_)                                                                                  \
    if (mt == MT_ && (forced_r == 0 || forced_r == R_)) {                                                      \
        if constexpr (G16<MT_, NT_, S, D, R_>::kValid) return launch16<MT_, NT_, S, D, R_>(a);                 \
    }
    if constexpr (S == 1 && D <= 4) {
        if (tall) {
            // (MT, NT) = (4, 4) would need 256 accumulator registers and spills with five waves per workgroup: use 2 x 4
            for (int mt = (want == 4 ? 2 : want); mt >= 2; --mt) { PWC_TRY(3, 4, 2) PWC_TRY(2, 4, 2) }
        }
    }
    if (two_per_cu) {
        for (int mt = want; mt >= 1; --mt) { PWC_TRY(2, 2, 2) PWC_TRY(1, 2, 2) }
    }
    for (int mt = want; mt >= 1; --mt) {
        PWC_TRY(4, 2, 3) PWC_TRY(3, 2, 3) PWC_TRY(2, 2, 3) PWC_TRY(1, 2, 3)
        PWC_TRY(4, 2, 2) PWC_TRY(3, 2, 2) PWC_TRY(2, 2, 2) PWC_TRY(1, 2, 2)
    }
#undef PWC_TRY
    PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv2d_f16_fwd: no tile fits the LDS for stride %d dilation %d", S, D);
}

inline int cout_padded(int Cout) { return (Cout + 31) / 32 * 32; }
inline int cout_padded_split(int Cout) { return (Cout + 15) / 16 * 32; }      // 16 couts (hi + lo rows) per 32-row block

}  // namespace

extern "C" int64_t pwc_conv3x3_f16_packed_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return -1;
    const int cg = (Cin + 7) / 8;
    return (int64_t)((cg + 1) / 2) * 18 * cout_padded(Cout) * 16;
}

extern "C" int64_t pwc_conv3x3_f16_packed_bytes_split(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return -1;
    const int cg = (Cin + 7) / 8;
    return (int64_t)((cg + 1) / 2) * 18 * cout_padded_split(Cout) * 16;
}

static int pack_f16(const void *w, void *wp, int Cin, int Cout, void *stream, int split, const char *who) {
    if (!w || !wp) PWC_FAIL(PWC_EINVAL, "%s: null pointer", who);
    if (Cin <= 0 || Cout <= 0) PWC_FAIL(PWC_EINVAL, "%s: bad shape", who);
    if (!pwc::aligned16(wp)) PWC_FAIL(PWC_EALIGN, "%s: packed buffer must be 16-byte aligned", who);
    const int64_t total = (split ? pwc_conv3x3_f16_packed_bytes_split(Cin, Cout) : pwc_conv3x3_f16_packed_bytes(Cin, Cout)) / 2;
    hipLaunchKernelGGL(pack3x3_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(w), static_cast<_Float16 *>(wp), Cin, Cout,
                       split ? cout_padded_split(Cout) : cout_padded(Cout), total, split);
    return pwc::check_launch("pack3x3_f16_kernel");
}

extern "C" int pwc_conv3x3_f16_pack(const void *w, void *wp, int Cin, int Cout, void *stream) {
    return pack_f16(w, wp, Cin, Cout, stream, 0, "pwc_conv3x3_f16_pack");
}

extern "C" int pwc_conv3x3_f16_pack_split(const void *w, void *wp, int Cin, int Cout, void *stream) {
    return pack_f16(w, wp, Cin, Cout, stream, 1, "pwc_conv3x3_f16_pack_split");
}

extern "C" int pwc_nchw_to_c8_f16(const void *x, void *y, int B, int C, int H, int W, int64_t x_bstride, int64_t y_bstride,
                                  void *stream) {
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_nchw_to_c8_f16: bad argument");
    if (!pwc::aligned16(y) || (y_bstride % 8)) PWC_FAIL(PWC_EALIGN, "pwc_nchw_to_c8_f16: output must be 16-byte aligned");
    const int cg = (C + 7) / 8;
    const int64_t plane = (int64_t)H * W, total = (int64_t)B * cg * plane;
    hipLaunchKernelGGL(nchw_to_c8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(x), static_cast<_Float16 *>(y), C, cg, plane, total, x_bstride, y_bstride);
    return pwc::check_launch("nchw_to_c8_kernel");
}

extern "C" int pwc_nchw_to_c8_f16_hilo(const void *x, void *y_hi, void *y_lo, int B, int C, int H, int W, int64_t x_bstride,
                                       int64_t hi_bstride, int64_t lo_bstride, void *stream) {
    if (!x || !y_hi || !y_lo || B <= 0 || C <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_nchw_to_c8_f16_hilo: bad argument");
    if (!pwc::aligned16(y_hi) || !pwc::aligned16(y_lo) || (hi_bstride % 8) || (lo_bstride % 8))
        PWC_FAIL(PWC_EALIGN, "pwc_nchw_to_c8_f16_hilo: outputs must be 16-byte aligned");
    const int cg = (C + 7) / 8;
    const int64_t plane = (int64_t)H * W, total = (int64_t)B * cg * plane;
    if ((total + 255) / 256 > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_nchw_to_c8_f16_hilo: grid too large");
    hipLaunchKernelGGL(nchw_to_c8_hilo_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(x), static_cast<_Float16 *>(y_hi), static_cast<_Float16 *>(y_lo), C, cg, plane, total,
                       x_bstride, hi_bstride, lo_bstride);
    return pwc::check_launch("nchw_to_c8_hilo_kernel");
}

extern "C" int pwc_c8_f16_to_nchw(const void *x, void *y, int B, int C, int H, int W, int64_t x_bstride, int64_t y_bstride,
                                  void *stream) {
    if (!x || !y || B <= 0 || C <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_c8_f16_to_nchw: bad argument");
    if (!pwc::aligned16(x) || (x_bstride % 8)) PWC_FAIL(PWC_EALIGN, "pwc_c8_f16_to_nchw: input must be 16-byte aligned");
    const int cg = (C + 7) / 8;
    const int64_t plane = (int64_t)H * W, total = (int64_t)B * cg * plane;
    hipLaunchKernelGGL(c8_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const _Float16 *>(x), static_cast<float *>(y), C, cg, plane, total, x_bstride, y_bstride);
    return pwc::check_launch("c8_to_nchw_kernel");
}

extern "C" int pwc_conv2d_f16_fwd(const void *x, const void *wp, const void *bias, void *y,
                                  int B, int Cin, int H, int W, int Cout, int stride, int dilation,
                                  unsigned flags, float leaky_slope, int64_t x_bstride, int64_t y_bstride, void *stream) {
    if (!x || !wp || !bias || !y) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_f16_fwd: null pointer");
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_f16_fwd: bad shape");
    if (!pwc::aligned16(x) || !pwc::aligned16(y) || !pwc::aligned16(wp) || (x_bstride % 8) || (y_bstride % 8))
        PWC_FAIL(PWC_EALIGN, "pwc_conv2d_f16_fwd: tensors must be 16-byte aligned with batch strides that are multiples of 8");
    if (flags & PWC_CONV_RESIDUAL) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv2d_f16_fwd: residual is not implemented for fp16");
    const int64_t plane = (int64_t)H * W;
    const int cg = (Cin + 7) / 8;
    if (x_bstride < (int64_t)cg * plane * 8) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_f16_fwd: x batch stride < Cg*H*W*8");
    if (plane * 32 >= 0x7fffffffLL) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv2d_f16_fwd: image plane too large for 32-bit DMA offsets");
    Args16 a;
    a.x = static_cast<const _Float16 *>(x);
    a.wp = static_cast<const _Float16 *>(wp);
    a.bias = static_cast<const float *>(bias);
    a.y = y;
    a.B = B; a.Cg = cg; a.H = H; a.W = W; a.Cout = Cout;
    a.CoutP = (flags & PWC_CONV_SPLIT_W) ? cout_padded_split(Cout) : cout_padded(Cout);
    a.Ho = (H - 1) / stride + 1;
    a.Wo = (W - 1) / stride + 1;
    a.bsx = x_bstride; a.bsy = y_bstride;
    a.slope = leaky_slope;
    a.mode = ((flags & PWC_ACT_LEAKY) ? kModeLeaky : 0) | ((flags & PWC_CONV_OUT_F32) ? kModeOutF32 : 0) |
             ((flags & PWC_CONV_SPLIT_W) ? kModeSplitW : 0);
    a.stream = static_cast<hipStream_t>(stream);
    if (stride == 1) {
        switch (dilation) {
            case 1: return dispatch16<1, 1>(a);
            case 2: return dispatch16<1, 2>(a);
            case 4: return dispatch16<1, 4>(a);
            case 8: return dispatch16<1, 8>(a);
            case 16: return dispatch16<1, 16>(a);
        }
    } else if (stride == 2 && dilation == 1) {
        return dispatch16<2, 1>(a);
    }
    PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv2d_f16_fwd: stride %d dilation %d has no fp16 kernel (stride 1: dilation 1,2,4,8,16; stride 2: dilation 1)",
             stride, dilation);
}
