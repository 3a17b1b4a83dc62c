// 3x3 convolution as an im2col-free implicit GEMM on the gfx950 matrix cores -- kernel template.
// Included by one translation unit per (stride, dilation) so the instantiations build in parallel.
//
// GEMM view (fp32 in, fp32 accumulate; v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain):
//     Y[cout, pixel] = bias[cout] + sum_{cin,ky,kx} Wt[cout,(cin,ky,kx)] * X[cin, y*s + ky*d - d, x*s + kx*d - d]
//   A operand = weights (lane l holds A[cout = l&31][k = l>>5]),  B operand = input (B[k = l>>5][pixel = l&31]),
//   D = 32 cout x 32 pixel with the pixel on the lane -> every store instruction writes 128-byte row
//   segments of the NCHW output.  The two k of one MFMA are the channel pair (2cp, 2cp+1) at one tap, so
//   both operands are ds_read_b32 of 32 consecutive dwords per half-wave (conflict-free).
//
// Workgroup = 4 waves (one per SIMD).  Tile = (4*NT rows x 32 cols) pixels x (32*MT) couts; wave w owns rows
// w*NT..w*NT+NT-1.  Cin is consumed in chunks of CK channels through a DOUBLE-BUFFERED LDS image
//     input tile  [CK][rows+halo][cols+halo]   flat, dword granular     (tap = address offset: no im2col)
//     weights     [CK][9 taps][32*MT couts]                              (from the packed [cin][tap][CoutP])
// filled ONLY by buffer_load ... lds (LDS-DMA): no staging VGPRs, no VALU in the loop.  Each lane's source
// offset is computed once per workgroup; zero padding, ragged image edges and the ragged last channel
// chunk all come from the buffer range check (out-of-range lanes deliver 0 to LDS).  Chunk k+1 streams in
// while chunk k's MFMAs run; one barrier per chunk.
//
// Two residency variants per tile shape (TWO):
//   TWO=0: CK=8, one workgroup per CU, accumulators may use the whole 512-entry register file (MT*NT <= 16);
//   TWO=1: CK=4, <=256 registers and <=80 KiB LDS so that TWO workgroups share a CU (MT*NT <= 8): the second
//          workgroup's MFMAs cover the first one's barrier / DMA issue / prologue / epilogue.  Measured
//          (profiles/r01_conv_notes.md): with one workgroup per CU ~13-30 us per workgroup are uncovered,
//          which costs short-K layers (conv2_0: 15 chunks) 29 % of the MFMA peak.
#pragma once
#include <stdlib.h>

#include "pwc_common.h"

// output store policy of the MFMA kernels: -DPWC_CONV_NT_STORE streams the activations past L2 (experiment)
#ifdef PWC_CONV_NT_STORE
#define PWC_CONV_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define PWC_CONV_STORE(ptr, val) (*(ptr) = (val))
#endif

namespace pwc_conv {

using pwc::leaky;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int kPackCK = 8;        // packed weights are zero-padded to a multiple of 8 input channels
constexpr int kThreads = 256;     // 4 waves
constexpr int kTileW = 32;        // = MFMA N
constexpr unsigned kOOB = 0x80000000u;   // voffset that always fails the range check -> LDS gets 0
constexpr int kRsrcFlags = 0x00020000;

__device__ __forceinline__ void *uniform_ptr(const void *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<void *>(((uint64_t)hi << 32) | lo);
}

// FOLD (maps of at most 16 columns: pyramid / decoder level 6 at 448x1024): the 32 MFMA columns are 16 pixel columns x two groups of
// four rows -- an 8 x 16 pixel tile -- instead of 4 x 32: a 7x16 map is ONE tile at 7/8 use instead of two tiles at 7/16.  Only the lane
// -> pixel mapping changes (column = lane % 16, row += 4 * (lane / 16 % 2)); dilation 1, MT = NT = 1 (stride 1 and 2).
template <int MT, int NT, int S, int D, int TWO, int CKO = 0, int FOLD = 0>
struct Geom {
    static_assert(!FOLD || (MT == 1 && NT == 1 && D == 1), "folded tile: 1x1 tiles, dilation 1");
    static constexpr int kCK = CKO ? CKO : (TWO ? 4 : 8);           // input channels per chunk
    static constexpr int kWPS = TWO ? 2 : 1;                        // waves per SIMD the register budget must allow
    static constexpr int kTW = FOLD ? 16 : 32;                      // pixel columns of a tile
    static constexpr int kTileH = FOLD ? 8 : 4 * NT;
    static constexpr bool kRowSep = (D >= 16);                      // stage the three ky row-sets separately
    // A staged row starts kPadL >= D columns left of the tile, at a multiple of four input columns, and is a whole number of 16-byte
    // pieces long: with W % 4 == 0 and 16-byte aligned tensors the tile then arrives as 16-byte LDS-DMA pieces (round 3: a third of the
    // instructions of the dword form -- each costs the issuing wave 60-180 cycles -- which remains for other widths / alignments).
    static constexpr int kPadL = (D + 3) / 4 * 4;
    static constexpr int kInW = (kPadL + (kTW - 1) * S + 1 + D + 3) / 4 * 4;       // row pitch (floats)
    static constexpr int kCol0 = kPadL - D;                                        // staged index of the tile's first window column
    static constexpr int kInH = kRowSep ? 3 * kTileH : (kTileH - 1) * S + 2 * D + 1;
    static constexpr int kCH = kInH * kInW;                          // floats per staged channel (flat)
    static constexpr int kKyStride = kRowSep ? kTileH * kInW : D * kInW;
    static constexpr int kInElems = kCK * kCH;
    static constexpr int kInSlots = (kInElems + kThreads - 1) / kThreads;       // dword DMAs per thread per chunk
    static constexpr int kInSlots16 = (kInElems / 4 + kThreads - 1) / kThreads;   // 16-byte DMAs per thread per chunk
    // floats: whole instructions of either form (the last 16-byte instruction of a chunk writes up to 1 KiB x 4 waves past kInElems)
    static constexpr int kInRegion = kInSlots * kThreads > kInSlots16 * kThreads * 4 ? kInSlots * kThreads : kInSlots16 * kThreads * 4;
    static constexpr int kCoutT = 32 * MT;
    static constexpr int kWPieces = kCK * 9 * kCoutT / 4;                         // 16-byte pieces per chunk
    static constexpr int kWSlots = (kWPieces + kThreads - 1) / kThreads;
    static constexpr int kWRegion = kWSlots * kThreads * 4;                       // floats
    static constexpr int kBufFloats = kInRegion + kWRegion;
    static constexpr int kSmemBytes = 2 * kBufFloats * 4;
    // does this variant exist?  TWO needs two workgroups' LDS and accumulators + operands within 256 registers
    static constexpr bool kValid = TWO ? (kSmemBytes <= 80 * 1024 && MT * NT <= 8)
                                       : (kSmemBytes <= 160 * 1024 && MT * NT <= 16);
};

namespace {   // kernels and launchers have internal linkage: each translation unit owns its instantiations

// Start the LDS-DMA of one chunk (input tile + weight slab) into `buf`.
template <class G>
__device__ __forceinline__ void issue_chunk(const float *xb, const float *wg, int chunk, int Cin, int plane,
                                            int64_t wchunk, unsigned wbytes, int wave, float *buf,
                                            const unsigned *in_off, const unsigned *w_off, bool p16) {
    const int c0 = chunk * G::kCK;
    const int cvalid = min(G::kCK, Cin - c0);
    // descriptors built from readfirstlane'd words so hipcc can prove them wave-uniform
    // (otherwise every DMA is wrapped in a waterfall loop)
    __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(xb + (int64_t)c0 * plane), 0, __builtin_amdgcn_readfirstlane(cvalid * plane * 4), kRsrcFlags);
    if (p16) {                                  // wave-uniform: 16-byte pieces, piece j * 256 + tid lands at float 4 * (j * 256 + tid)
        float *dst_in = buf + wave * 256;
#pragma unroll
        for (int j = 0; j < G::kInSlots16; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(dst_in + j * kThreads * 4), 16, in_off[j], 0, 0, 0);
    } else {
        float *dst_in = buf + wave * 64;
#pragma unroll
        for (int j = 0; j < G::kInSlots; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void *)(dst_in + j * kThreads), 4, in_off[j], 0, 0, 0);
    }
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(wg + (int64_t)chunk * wchunk), 0, __builtin_amdgcn_readfirstlane((int)wbytes), kRsrcFlags);
    float *dst_w = buf + G::kInRegion + wave * 256;
#pragma unroll
    for (int j = 0; j < G::kWSlots; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void *)(dst_w + j * kThreads * 4), 16, w_off[j], 0, 0, 0);
}

// SPLIT=1 (split-K): blockIdx.z owns the chunk range [z*cps, (z+1)*cps) and writes its raw partial sums to
// y = workspace [z][b][Cout][Ho][Wo] (bsy = Cout*Ho*Wo, zstride = B*bsy); bias / activation / residual are applied by
// splitk_reduce_kernel, which adds the partials in fixed z order (deterministic).  Used when a layer has too
// few output tiles to occupy 256 CUs and a long Cin (levels 6-4, batch-1 inference).
template <int MT, int NT, int S, int D, int TWO, int SPLIT, int FOLD = 0>
__global__ void __launch_bounds__(kThreads, (TWO ? 2 : 1))
conv3x3_mfma_kernel(const float *__restrict__ x, const float *__restrict__ wp, const float *__restrict__ bias,
                    const float *__restrict__ residual, float *__restrict__ y,
                    int Cin, int H, int W, int Cout, int CoutP, int Ho, int Wo, int tiles_x, int tiles_y,
                    int64_t bsx, int64_t bsy, int64_t bsr, float slope, int do_leaky, int cps, int64_t zstride, int p16i) {
    using G = Geom<MT, NT, S, D, TWO, 0, FOLD>;
    const bool p16 = __builtin_amdgcn_readfirstlane(p16i) != 0;
    constexpr int CK = G::kCK;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int col = FOLD ? (lane & 15) : (lane & 31);               // pixel column inside the tile
    const int frow = FOLD ? ((lane >> 4) & 1) * 4 : 0;              // folded tile: MFMA columns 16..31 are rows 4..7
    const int kh = lane >> 5;

    int bid = blockIdx.x;
#ifndef PWC_CONV_NO_XCD_MAP
    // workgroups i, i+8, ... share an XCD: give each XCD a contiguous run of tiles so that halo re-reads hit its L2
    if ((gridDim.x & 7u) == 0) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
#endif
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int g = blockIdx.y;
    const int ox0 = tx * G::kTW;
    const int oy0 = ty * G::kTileH;
    const int plane = H * W;

    // ---- per-lane DMA source offsets, computed once --------------------------------------------------
    unsigned in_off[G::kInSlots];
#pragma unroll
    for (int j = 0; j < G::kInSlots; ++j) {
        // element (dword form) or first element of the piece (16-byte form) number j * 256 + tid of the staged chunk [c][row][kInW]
        const int i = p16 ? 4 * (j * kThreads + tid) : j * kThreads + tid;
        const int c = i / G::kCH;
        const int rem = i % G::kCH;
        const int r = rem / G::kInW;
        const int xx = rem % G::kInW;
        int iy;
        if constexpr (G::kRowSep) {
            iy = oy0 + (r % G::kTileH) - D + (r / G::kTileH) * D;
        } else {
            iy = oy0 * S - D + r;
        }
        const int ix = ox0 * S - G::kPadL + xx;                 // 16-byte form: ix and W are multiples of 4 -> a piece is all-in or all-out
        const bool ok = (i < G::kInElems) && (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);
        in_off[j] = ok ? (unsigned)(c * plane + iy * W + ix) * 4u : kOOB;
    }
    unsigned w_off[G::kWSlots];
#pragma unroll
    for (int j = 0; j < G::kWSlots; ++j) {
        const int p = j * kThreads + tid;                       // 16-byte piece index inside the chunk image
        const int row = p / (G::kCoutT / 4);                    // (c*9 + tap)
        const int q = p % (G::kCoutT / 4);
        w_off[j] = (p < G::kWPieces) ? (unsigned)(row * CoutP + q * 4) * 4u : kOOB;
    }

    // ---- accumulators start at the bias: D row (cout) of register j is (j&3) + 8*(j>>2) + 4*kh --------
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int co = g * G::kCoutT + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh;
            const float bv = SPLIT ? 0.f : bias[min(co, Cout - 1)];          // rows >= Cout are never stored
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt][j] = bv;
        }
    }

    const float *xb = x + (int64_t)b * bsx;
    const int nchunks = (Cin + CK - 1) / CK;
    const int64_t wchunk = (int64_t)CK * 9 * CoutP;             // floats per chunk of the packed [cin][tap][CoutP]
    const float *wg = wp + g * G::kCoutT;                       // this workgroup's cout columns
    const unsigned wbytes = (unsigned)(wchunk - g * G::kCoutT) * 4u;

    int chunk_lo = 0, chunk_hi = nchunks;
    if constexpr (SPLIT) {
        chunk_lo = (int)blockIdx.z * cps;                       // host guarantees chunk_lo < nchunks
        chunk_hi = min(nchunks, chunk_lo + cps);
    }
    issue_chunk<G>(xb, wg, chunk_lo, Cin, plane, wchunk, wbytes, wave, smem + (chunk_lo & 1) * G::kBufFloats, in_off, w_off, p16);
    for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
        float *cur = smem + (chunk & 1) * G::kBufFloats;
        // This wave's DMA of `chunk` has landed (explicit wait: hipcc does not reliably keep its own
        // vmcnt(0) in front of the in-loop barrier for LDS-DMA), then the barrier makes every wave's
        // part visible and guarantees the other buffer is no longer being read.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (chunk + 1 < chunk_hi)
            issue_chunk<G>(xb, wg, chunk + 1, Cin, plane, wchunk, wbytes, wave, smem + ((chunk + 1) & 1) * G::kBufFloats,
                           in_off, w_off, p16);

        const float *rd_in = cur + kh * G::kCH + (wave * NT + frow) * S * G::kInW + col * S + G::kCol0;
        const float *rd_w = cur + G::kInRegion + kh * 9 * G::kCoutT + (lane & 31);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int cp = 0; cp < CK / 2; ++cp) {
                float a[MT], bv[NT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = rd_w[(cp * 2 * 9 + tap) * G::kCoutT + mt * 32];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    bv[nt] = rd_in[cp * 2 * G::kCH + ky * G::kKyStride + kx * D + nt * S * G::kInW];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], bv[nt], acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: LeakyReLU / residual, 128-byte row-segment stores ----------------------------------
    const int ox = ox0 + col;
    const int64_t oplane = (int64_t)Ho * Wo;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int oy = oy0 + wave * NT + nt + frow;
        if (oy >= Ho || ox >= Wo) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = g * G::kCoutT + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh;
                if (co >= Cout) continue;
                float v = acc[mt][nt][j];
                const int64_t off = (int64_t)co * oplane + (int64_t)oy * Wo + ox;
                if constexpr (SPLIT) {
                    y[(int64_t)blockIdx.z * zstride + (int64_t)b * bsy + off] = v;
                } else {
                    if (do_leaky) v = leaky(v, slope);
                    if (residual) v += residual[(int64_t)b * bsr + off];
                    PWC_CONV_STORE(y + (int64_t)b * bsy + off, v);
                }
            }
        }
    }
}

// ---- Cout <= 16 (conv1aa / conv1b: 16 -> 16 at half resolution) -----------------------------------------------
// With the 32x32x2 MFMA half of the 32 cout rows would be padding.  v_mfma_f32_16x16x4_f32 has the same FLOP rate
// on a 16 cout x 16 pixel x 4 channel step (A[l&15][k=l>>4], B[k=l>>4][l&15], D row = 4*(l>>4)+reg, col = l&15), so
// the same LDS image (stride 1, dilation 1 geometry of Geom<1,NT,1,1,..>) is consumed with the four lane quarters
// on four consecutive channels of one tap and two MFMAs per 32-pixel row segment: no wasted rows.
// Measured on conv1aa (16 -> 16 @224x512, 32 images): 338 us with the 32x32x2 kernel, 229 us here with 16-row tiles
// and 4-channel chunks; staging the whole Cin = 16 as one chunk (no pipelining inside a workgroup) is slower
// (275-292 us), see profiles/r01_conv_notes.md.
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT, int CKT>
__global__ void __launch_bounds__(kThreads, 2)
conv3x3_mfma16_kernel(const float *__restrict__ x, const float *__restrict__ wp, const float *__restrict__ bias,
                      const float *__restrict__ residual, float *__restrict__ y,
                      int Cin, int H, int W, int Cout, int CoutP, int tiles_x, int tiles_y,
                      int64_t bsx, int64_t bsy, int64_t bsr, float slope, int do_leaky, int p16i) {
    using G = Geom<1, NT, 1, 1, 1, CKT>;
    const bool p16 = __builtin_amdgcn_readfirstlane(p16i) != 0;
    constexpr int CK = G::kCK;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int acol = lane & 15;         // A: cout row / B, D: pixel column inside a 16-wide half
    const int kg = lane >> 4;           // A, B: channel inside the group of 4 / D: cout rows 4*kg .. 4*kg+3

    int bid = blockIdx.x;
#ifndef PWC_CONV_NO_XCD_MAP
    // workgroups i, i+8, ... share an XCD: give each XCD a contiguous run of tiles so that halo re-reads hit its L2
    if ((gridDim.x & 7u) == 0) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
#endif
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int ox0 = tx * kTileW;
    const int oy0 = ty * G::kTileH;
    const int plane = H * W;

    unsigned in_off[G::kInSlots];
#pragma unroll
    for (int j = 0; j < G::kInSlots; ++j) {
        const int i = p16 ? 4 * (j * kThreads + tid) : j * kThreads + tid;      // see conv3x3_mfma_kernel
        const int c = i / G::kCH;
        const int rem = i % G::kCH;
        const int iy = oy0 - 1 + rem / G::kInW;
        const int ix = ox0 - G::kPadL + rem % G::kInW;
        const bool ok = (i < G::kInElems) && (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);
        in_off[j] = ok ? (unsigned)(c * plane + iy * W + ix) * 4u : kOOB;
    }
    unsigned w_off[G::kWSlots];
#pragma unroll
    for (int j = 0; j < G::kWSlots; ++j) {
        const int p = j * kThreads + tid;
        const int row = p / (G::kCoutT / 4);
        const int q = p % (G::kCoutT / 4);
        w_off[j] = (p < G::kWPieces) ? (unsigned)(row * CoutP + q * 4) * 4u : kOOB;
    }

    f32x4 acc[NT][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float bv = bias[min(kg * 4 + j, Cout - 1)];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { acc[nt][0][j] = bv; acc[nt][1][j] = bv; }
    }

    const float *xb = x + (int64_t)b * bsx;
    const int nchunks = (Cin + CK - 1) / CK;
    const int64_t wchunk = (int64_t)CK * 9 * CoutP;
    const unsigned wbytes = (unsigned)wchunk * 4u;

    issue_chunk<G>(xb, wp, 0, Cin, plane, wchunk, wbytes, wave, smem, in_off, w_off, p16);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        float *cur = smem + (chunk & 1) * G::kBufFloats;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // see conv3x3_mfma_kernel
        __syncthreads();
        if (chunk + 1 < nchunks)
            issue_chunk<G>(xb, wp, chunk + 1, Cin, plane, wchunk, wbytes, wave, smem + ((chunk + 1) & 1) * G::kBufFloats,
                           in_off, w_off, p16);
        const float *rd_in = cur + kg * G::kCH + (wave * NT) * G::kInW + acol + G::kCol0;
        const float *rd_w = cur + G::kInRegion + kg * 9 * G::kCoutT + acol;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int kq = 0; kq < CK / 4; ++kq) {
                const float a = rd_w[(kq * 4 * 9 + tap) * G::kCoutT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float *p = rd_in + kq * 4 * G::kCH + (nt + ky) * G::kInW + kx;
                    acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, p[0], acc[nt][0], 0, 0, 0);
                    acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, p[16], acc[nt][1], 0, 0, 0);
                }
            }
        }
    }

    const int64_t oplane = (int64_t)H * W;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int oy = oy0 + wave * NT + nt;
        if (oy >= H) continue;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ox = ox0 + 16 * h + acol;
            if (ox >= W) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = kg * 4 + j;
                if (co >= Cout) continue;
                float v = acc[nt][h][j];
                if (do_leaky) v = leaky(v, slope);
                const int64_t off = (int64_t)co * oplane + (int64_t)oy * W + ox;
                if (residual) v += residual[(int64_t)b * bsr + off];
                PWC_CONV_STORE(y + (int64_t)b * bsy + off, v);
            }
        }
    }
}

}  // namespace (anonymous)

struct ConvArgs {
    const float *x, *wp, *bias, *residual;
    float *y;
    int B, Cin, H, W, Cout, CoutP, Ho, Wo;
    int64_t bsx, bsy, bsr;
    float slope;
    int do_leaky;
    hipStream_t stream;
    // split-K (set by pwc_conv2d_fwd when a workspace is supplied and the layer qualifies): partial sums go to
    // `partial` [ksplit][B][Cout][Ho][Wo], `cps` chunks of 8 input channels per split
    float *partial = nullptr;
    int ksplit = 1, cps = 0;
    // the input tile as 16-byte LDS-DMA pieces (W % 4 == 0, 16-byte aligned x, batch stride a multiple of 4); else dwords.
    // PWC_CONV_P16=0 forces the dword form (A/B runs)
    int p16() const {
        static const bool on = [] { const char *e = getenv("PWC_CONV_P16"); return !(e && e[0] == '0'); }();
        return on && (W % 4 == 0) && (reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (bsx % 4 == 0);
    }
};

// Split-K plan shared by pwc_conv2d_fwd and pwc_conv2d_workspace_bytes: ksplit (1 = do not split) and the
// 8-channel chunks per split, for a stride-1 dilation-1 layer run with the 4x32-pixel x 32-cout tile.
struct SplitPlan { int ksplit, cps; };
// maps of at most 16 columns and more than one 4-row strip take the folded 8 x 16 tile (Geom FOLD); stride 1, dilation 1 only
inline bool fold_tile(int Ho, int Wo) {
    static const bool on = [] { const char *e = getenv("PWC_CONV_FOLD"); return !(e && e[0] == '0'); }();
    return on && Wo <= 16 && Ho > 4;
}
inline SplitPlan plan_split(int B, int Cin, int Ho, int Wo, int CoutP) {
    static const int knob = [] { const char *e = getenv("PWC_CONV_SPLIT"); return (e && *e) ? atoi(e) : -1; }();
    const int64_t blocks = fold_tile(Ho, Wo) ? (int64_t)B * ((Ho + 7) / 8) * (CoutP / 32)
                                             : (int64_t)B * ((Wo + kTileW - 1) / kTileW) * ((Ho + 3) / 4) * (CoutP / 32);
    const int nchunks = (Cin + 7) / 8;
    // measured (tools/sweep_split.sh): a short K only pays when the grid is nearly empty
    if (knob == 0 || blocks > 256 || nchunks < (blocks <= 64 ? 8 : 16)) return {1, nchunks};
    // aim at ~512 workgroups (two per CU) but keep at least 3 chunks per split
    int ks = (int)((512 + blocks - 1) / blocks);
    if (knob > 0) ks = knob;
    ks = min(ks, nchunks / 3);
    if (ks < 2) return {1, nchunks};
    const int cps = (nchunks + ks - 1) / ks;
    ks = (nchunks + cps - 1) / cps;                     // no empty split
    return {ks, cps};
}

namespace {

template <int MT, int NT, int S, int D, int TWO, int FOLD = 0>
int launch(const ConvArgs &a) {
    using G = Geom<MT, NT, S, D, TWO, 0, FOLD>;
    if constexpr (!G::kValid) {
        PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: internal: tile %dx%d two=%d does not exist", MT, NT, TWO);
    } else {
        const int tiles_x = (a.Wo + G::kTW - 1) / G::kTW;
        const int tiles_y = (a.Ho + G::kTileH - 1) / G::kTileH;
        const int64_t nblk = (int64_t)a.B * tiles_x * tiles_y;
        const int groups = (a.CoutP / 32 + MT - 1) / MT;
        if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: grid too large");
        auto kern = conv3x3_mfma_kernel<MT, NT, S, D, TWO, 0, FOLD>;
        static pwc::LdsAttrOnce attr;   // one per instantiation, tracked per device
        if (const int rc = pwc::ensure_lds_attr(attr, reinterpret_cast<const void *>(kern), G::kSmemBytes, "pwc_conv2d_fwd"))
            return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)groups), dim3(kThreads), G::kSmemBytes, a.stream,
                           a.x, a.wp, a.bias, a.residual, a.y, a.Cin, a.H, a.W, a.Cout, a.CoutP, a.Ho, a.Wo,
                           tiles_x, tiles_y, a.bsx, a.bsy, a.bsr, a.slope, a.do_leaky, 0, (int64_t)0, a.p16());
        pwc::note_kernel("conv3x3_mfma_kernel", MT, NT, S, D, TWO, FOLD ? 16 : 0);        // (last field: 16 = folded 8 x 16 tile)
        return pwc::check_launch("conv3x3_mfma_kernel");
    }
}

// split-K launch of the 4x32 x 32-cout tile (CK = 8): raw partials into a.partial
template <int S, int D, int FOLD = 0>
int launch_split(const ConvArgs &a) {
    using G = Geom<1, 1, S, D, 0, 0, FOLD>;
    const int tiles_x = (a.Wo + G::kTW - 1) / G::kTW;
    const int tiles_y = (a.Ho + G::kTileH - 1) / G::kTileH;
    const int64_t nblk = (int64_t)a.B * tiles_x * tiles_y;
    const int groups = a.CoutP / 32;
    auto kern = conv3x3_mfma_kernel<1, 1, S, D, 0, 1, FOLD>;
    static pwc::LdsAttrOnce attr;
    if (const int rc = pwc::ensure_lds_attr(attr, reinterpret_cast<const void *>(kern), G::kSmemBytes, "pwc_conv2d_fwd"))
        return rc;
    const int64_t bsp = (int64_t)a.Cout * a.Ho * a.Wo;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)groups, (unsigned)a.ksplit), dim3(kThreads), G::kSmemBytes,
                       a.stream, a.x, a.wp, a.bias, (const float *)nullptr, a.partial, a.Cin, a.H, a.W, a.Cout, a.CoutP,
                       a.Ho, a.Wo, tiles_x, tiles_y, a.bsx, bsp, (int64_t)0, 0.f, 0, a.cps, (int64_t)a.B * bsp, a.p16());
    return pwc::check_launch("conv3x3_mfma_kernel<split>");
}

// Cout <= 16, stride 1, dilation 1: the 16x16x4 kernel.  NT (tile rows / 4) from PWC_CONV16_TILE or the rule below.
struct TileChoice16 { int nt; };

template <int NT, int CKT>
int launch16(const ConvArgs &a) {
    using G = Geom<1, NT, 1, 1, 1, CKT>;
    static_assert(G::kSmemBytes <= 160 * 1024, "tile does not exist");
    const int tiles_x = (a.Wo + kTileW - 1) / kTileW;
    const int tiles_y = (a.Ho + G::kTileH - 1) / G::kTileH;
    const int64_t nblk = (int64_t)a.B * tiles_x * tiles_y;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: grid too large");
    auto kern = conv3x3_mfma16_kernel<NT, CKT>;
    static pwc::LdsAttrOnce attr;
    if (const int rc = pwc::ensure_lds_attr(attr, reinterpret_cast<const void *>(kern), G::kSmemBytes, "pwc_conv2d_fwd"))
        return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(kThreads), G::kSmemBytes, a.stream,
                       a.x, a.wp, a.bias, a.residual, a.y, a.Cin, a.H, a.W, a.Cout, a.CoutP, tiles_x, tiles_y,
                       a.bsx, a.bsy, a.bsr, a.slope, a.do_leaky, a.p16());
    return pwc::check_launch("conv3x3_mfma16_kernel");
}

inline int dispatch16(const ConvArgs &a) {
    static const TileChoice16 forced = [] {
        TileChoice16 f{0};
        const char *e = getenv("PWC_CONV16_TILE");
        if (e) sscanf(e, "%d", &f.nt);
        return f;
    }();
    TileChoice16 t = forced;
    if (t.nt == 0) {
        // rows per tile: 4 while the grid is small, 16 once there are plenty of tiles (measured: 248 / 234 / 229 us
        // for 4 / 8 / 16 rows on conv1aa at batch 16; 17.9 us vs 23 us for 4 vs 8 rows at batch 1)
        const int64_t blocks1 = (int64_t)a.B * ((a.Wo + kTileW - 1) / kTileW) * ((a.Ho + 3) / 4);
        t.nt = blocks1 <= 2048 ? 1 : (blocks1 <= 8192 ? 2 : 4);
    }
    if (t.nt == 1) return launch16<1, 4>(a);
    if (t.nt == 2) return launch16<2, 4>(a);
    if (t.nt == 4) return launch16<4, 4>(a);
    PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: PWC_CONV16_TILE must be 1, 2 or 4");
}

// (MT, NT, TWO) choice by a cost model fitted to measurements on MI355X (time in units of one MFMA
// tile-step over 8 channels ~ 1 us):
//   one workgroup per CU  : ceil(blocks/256) * (chunks8 * (MT*NT + 0.45) + 20)
//   two workgroups per CU : (blocks/512 + 0.25) * (chunks8 * 2*MT*NT + 5)        (blocks > 512)
// (sweep of 13 variants x 7 level-2 layers at batch 16: profiles/r01_conv_notes.md)
struct TileChoice { int mt, nt, two; };

template <int S, int D>
constexpr bool variant_valid(int mt, int nt, int two) {
#define PWC_V(MT_, NT_) if (mt == MT_ && nt == NT_) return two ? Geom<MT_, NT_, S, D, 1>::kValid : Geom<MT_, NT_, S, D, 0>::kValid;
    PWC_V(1, 1) PWC_V(2, 1) PWC_V(3, 1) PWC_V(4, 1) PWC_V(1, 2) PWC_V(2, 2) PWC_V(3, 2) PWC_V(4, 2)
    PWC_V(1, 4) PWC_V(2, 4) PWC_V(3, 4) PWC_V(4, 4)
#undef PWC_V
    return false;
}

inline bool tune_l2() {        // PWC_CONV_NO_TUNE_L2=1: the cost model without the measured level-2 residuals (A/B runs)
    static const bool on = [] { const char *e = getenv("PWC_CONV_NO_TUNE_L2"); return !(e && e[0] == '1'); }();
    return on;
}

template <int S, int D>
inline TileChoice choose_tile(int B, int Cin, int Ho, int Wo, int CoutP, int max_nt, int max_mt, int force_two) {
    const int tiles32 = CoutP / 32;
    const int tiles_x = (Wo + kTileW - 1) / kTileW;
    const double chunks8 = (Cin + 7) / 8;
    TileChoice best{1, 1, 0};
    double best_cost = 1e300;
    for (int two = 0; two <= 1; ++two) {
        if (force_two >= 0 && two != force_two) continue;
        for (int mt = 1; mt <= max_mt; ++mt) {
            if (mt > tiles32) break;
            const int groups = (tiles32 + mt - 1) / mt;
            for (int nt = 1; nt <= max_nt; nt *= 2) {
                if (!variant_valid<S, D>(mt, nt, two)) continue;
                const int tiles_y = (Ho + 4 * nt - 1) / (4 * nt);
                const double blocks = (double)B * tiles_x * tiles_y * groups;
                double cost;
                if (!two) {
                    // equal-length workgroups, one per CU: discrete rounds
                    const double rounds = (double)(int64_t)((blocks + 255.0) / 256.0);
                    cost = rounds * (chunks8 * (mt * nt + 0.45) + 20.0);
                } else if (blocks <= 256.0) {
                    cost = chunks8 * (mt * nt + 0.75) + 20.0;      // alone on its CU, twice the barriers (CK=4)
                } else if (blocks <= 512.0) {
                    cost = chunks8 * 2.0 * mt * nt + 5.0;           // the busiest CU holds two
                } else {
                    // workgroups retire and start independently: work / throughput + a quarter-round tail
                    cost = (blocks / 512.0 + 0.25) * (chunks8 * 2.0 * mt * nt + 5.0);
                    if (mt == 3) cost *= 1.12;                      // measured: the 96-wide variant under-performs here
                }
                cost *= 1.0 + 0.02 * (groups - 1);               // mild penalty: input re-read per cout group
                if (S == 1 && two && blocks > 512.0 && tune_l2()) {
                    // measured residuals of the model on the level-2 layers at batch 16 (19-variant sweep after the
                    // XCD-aware tile order, profiles/r01_conv_notes.md): 8-row tiles halve the halo share of the
                    // staging for 32-cout layers and for dilation 4; with dilation 16 the 4-row tile wins
                    if (D == 1 && tiles32 == 1 && nt == 1) cost *= 1.045;
                    if (D == 4 && nt == 1) cost *= 1.04;
                    if (D == 16 && nt > 1) cost *= 1.05;
                }
                if (cost < best_cost) { best_cost = cost; best = {mt, nt, two}; }
            }
        }
    }
    return best;
}

// one of these per translation unit
template <int S, int D, int MAXNT, int MAXMT>
int dispatch(const ConvArgs &a) {
    static const int force_two = [] { const char *e = getenv("PWC_CONV_TWO"); return (e && *e) ? atoi(e) : -1; }();
    TileChoice t = choose_tile<S, D>(a.B, a.Cin, a.Ho, a.Wo, a.CoutP, MAXNT, MAXMT, force_two);
    if (force_two >= 0 && !variant_valid<S, D>(t.mt, t.nt, t.two))
        t = choose_tile<S, D>(a.B, a.Cin, a.Ho, a.Wo, a.CoutP, MAXNT, MAXMT, -1);
    // tuning knob: PWC_CONV_TILE="mt,nt,two" overrides the model when that variant exists for this layer
    static const TileChoice forced = [] {
        TileChoice f{0, 0, 0};
        const char *e = getenv("PWC_CONV_TILE");
        if (e) sscanf(e, "%d,%d,%d", &f.mt, &f.nt, &f.two);
        return f;
    }();
    if (forced.mt > 0 && forced.mt <= MAXMT && forced.nt <= MAXNT && forced.mt * 32 <= a.CoutP + 31 &&
        variant_valid<S, D>(forced.mt, forced.nt, forced.two))
        t = forced;
#define PWC_TILE(MT_, NT_)                                                                       \
    if (t.mt == MT_ && t.nt == NT_) return t.two ? launch<MT_, NT_, S, D, 1>(a) : launch<MT_, NT_, S, D, 0>(a);
    PWC_TILE(1, 1) PWC_TILE(2, 1) PWC_TILE(3, 1)
    PWC_TILE(1, 2) PWC_TILE(2, 2) PWC_TILE(3, 2)
    if constexpr (MAXMT >= 4) { PWC_TILE(4, 1) PWC_TILE(4, 2) }
    if constexpr (MAXNT >= 4) { PWC_TILE(1, 4) PWC_TILE(2, 4) PWC_TILE(3, 4) PWC_TILE(4, 4) }
#undef PWC_TILE
    PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: internal tile choice %dx%d", t.mt, t.nt);
}

}  // namespace (anonymous)

}  // namespace pwc_conv
