// Cost-volume correlation for gfx950 (MI355X).
//
// Replaces the reference's channels_first + correlation_forward CUDA kernels
// (models/correlation_package/correlation_cuda_kernel.cu:46-147) and their launcher
// (:336-427).  Semantics kept: output channel order tc = (dy+r)*D + (dx+r), zero padding,
// fp32 accumulation; NOT kept: the NHWC scratch round trip, the one-pixel-per-32-thread-block
// layout and the 81 serial shuffle reductions.
//
// Fast path (kernel_size 1, max_disp 4, stride1 = stride2 = 1, pad 4 -- the only
// configuration PWCDCNet instantiates, PWCNet.py:71):
//   * one workgroup = 9 wavefronts = one 8 x 32 pixel tile; wave w owns displacement row
//     dy = w - 4, so dy is wave-uniform and the 81-neighbour product needs no cross-lane
//     reduction: every lane owns 4 consecutive pixels x 9 dx = 36 fp32 accumulators;
//   * both feature maps are read NCHW with 16-byte coalesced loads along W (rows are
//     contiguous in W -- no transpose needed) and staged per 8-channel chunk in LDS: the in1
//     tile (8 x 32) and the in2 displacement tile with its +-4 halo (16 x 40);
//   * per channel a lane issues 4 ds_read_b128 (1 for in1, 3 for the 12-wide in2 window) and
//     36 v_fma; the lane -> (row, column-group) map is chosen so that each ds_read_b128 lane
//     group covers rows {k, k+4}, which with a 40-float row pitch touches all 64 banks once;
//   * epilogue fuses the scale (corr_multiply or 1/C) and LeakyReLU and stores 128-byte row
//     segments straight into the caller's [B,81,H,W] slot (batch stride free, so the slot can
//     live inside the decoder's concat arena).
// Generic path (any pad/kernel/stride): one thread per output element, used only by callers
// other than PWCDCNet.
#include <stdlib.h>

#include "pwc_common.h"
#include "pwc_corr_pipe.h"
#include "pwc_warp_taps.h"

// Cache policy (profiles/r01_corr_ablation.md): the 81-channel output is written once and not read again by this
// kernel -> non-temporal stores (64.8 -> 59.8 us at level 2, batch 16; whole forward unchanged).  The inputs are NOT
// streamed non-temporally: the 2.5x halo re-reads must hit in L2 (nt loads: 78 us; nt on in1 only: +8 %).  -DPWC_CORR_NT_LOAD /
// -DPWC_CORR_TEMPORAL_STORE rebuild the other variants.
#ifndef PWC_CORR_TEMPORAL_STORE
#define PWC_CORR_NT_STORE 1
#endif
#ifdef PWC_CORR_NT_LOAD
#define PWC_CORR_DMA pwc::dma_b128_nt
#else
#define PWC_CORR_DMA pwc::dma_b128
#endif
typedef float f32x4v __attribute__((ext_vector_type(4)));

namespace {

using pwc::from_f32;
using pwc::leaky;
using pwc::to_f32;

constexpr int kD = 4;               // max displacement of the fast path
constexpr int kND = 2 * kD + 1;     // 9
constexpr int kPX = 4;              // pixels per lane
constexpr int kTH = 8;              // tile rows
constexpr int kTG = 8;              // 4-pixel groups per tile row
constexpr int kTW = kTG * kPX;      // 32 tile columns
constexpr int kPitch = 40;          // floats per LDS row (10 x 16 B): 4 rows apart == 8 slots mod 16
constexpr int kS2Rows = kTH + 2 * kD;            // 16
constexpr int kS2Quads = (kTW + 2 * kD) / 4;     // 10 float4 per in2 row
constexpr int kThreads = 64 * kND;  // 576

template <int CK>
struct CorrSmem {
    float s1[CK][kTH][kPitch];
    float s2[CK][kS2Rows][kPitch];
};

// ds_read_b128 services a wave in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} and the
// same +32 (MI355X microarch guide, LDS table).  Give group k the tile rows {k, k+4}.
__device__ __forceinline__ void lane_to_rg(int lane, int &r, int &g) {
    const int l = lane & 31;
    int grp, pos;
    if (l < 4)       { grp = 0; pos = l; }
    else if (l < 12) { grp = 1; pos = l - 4; }
    else if (l < 16) { grp = 0; pos = l - 8; }
    else if (l < 20) { grp = 1; pos = l - 8; }
    else if (l < 28) { grp = 0; pos = l - 12; }
    else             { grp = 1; pos = l - 16; }
    grp += (lane >> 5) * 2;
    r = grp + 4 * (pos >> 3);
    g = pos & 7;
}

template <typename T>
__device__ __forceinline__ float4 load_quad(const T *__restrict__ row, int x, int W, bool row_ok, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!row_ok) return v;
    if (vec) {
        if (x >= 0 && x < W) {
            if constexpr (sizeof(T) == 4) {
                v = *reinterpret_cast<const float4 *>(row + x);
            } else {
                const uint2 raw = *reinterpret_cast<const uint2 *>(row + x);
                const __half2 lo = *reinterpret_cast<const __half2 *>(&raw.x);
                const __half2 hi = *reinterpret_cast<const __half2 *>(&raw.y);
                v = make_float4(__low2float(lo), __high2float(lo), __low2float(hi), __high2float(hi));
            }
        }
    } else {
        if (x >= 0 && x < W) v.x = to_f32<T>(row[x]);
        if (x + 1 >= 0 && x + 1 < W) v.y = to_f32<T>(row[x + 1]);
        if (x + 2 >= 0 && x + 2 < W) v.z = to_f32<T>(row[x + 2]);
        if (x + 3 >= 0 && x + 3 < W) v.w = to_f32<T>(row[x + 3]);
    }
    return v;
}

template <typename T, int CK>
__global__ void __launch_bounds__(kThreads)
corr81_kernel(const T *__restrict__ in1, const T *__restrict__ in2, T *__restrict__ out,
              int C, int H, int W, int tiles_x, int tiles_y,
              int64_t bs1, int64_t bs2, int64_t bso,
              float scale, float slope, int do_leaky, int vec) {
    __shared__ __attribute__((aligned(16))) CorrSmem<CK> sm;

    const int tid = threadIdx.x;
    const int wave = tid >> 6;            // displacement row index dyi = dy + 4 (wave-uniform)
    const int lane = tid & 63;
    int r, g;
    lane_to_rg(lane, r, g);

    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx * kTW;
    const int y0 = ty * kTH;

    const int64_t plane = (int64_t)H * W;
    const T *p1 = in1 + (int64_t)b * bs1;
    const T *p2 = in2 + (int64_t)b * bs2;

    float acc[kND][kPX];
#pragma unroll
    for (int dx = 0; dx < kND; ++dx)
#pragma unroll
        for (int p = 0; p < kPX; ++p) acc[dx][p] = 0.f;

    constexpr int kQ1 = CK * kTH * kTG;             // float4 slots of the in1 chunk
    constexpr int kQ2 = CK * kS2Rows * kS2Quads;    // float4 slots of the in2 chunk

    for (int c0 = 0; c0 < C; c0 += CK) {
        if (c0) __syncthreads();                    // previous chunk fully consumed
        for (int i = tid; i < kQ1 + kQ2; i += kThreads) {
            if (i < kQ1) {
                const int c = i / (kTH * kTG);
                const int row = (i / kTG) % kTH;
                const int q = i % kTG;
                const int y = y0 + row;
                const bool ok = (c0 + c < C) && (y < H);
                const float4 v = load_quad<T>(p1 + (int64_t)(c0 + c) * plane + (int64_t)y * W, x0 + 4 * q, W, ok, vec);
                *reinterpret_cast<float4 *>(&sm.s1[c][row][4 * q]) = v;
            } else {
                const int j = i - kQ1;
                const int c = j / (kS2Rows * kS2Quads);
                const int rem = j % (kS2Rows * kS2Quads);
                const int row = rem / kS2Quads;
                const int q = rem % kS2Quads;
                const int y = y0 + row - kD;
                const bool ok = (c0 + c < C) && (y >= 0) && (y < H);
                const float4 v = load_quad<T>(p2 + (int64_t)(c0 + c) * plane + (int64_t)y * W, x0 + 4 * q - kD, W, ok, vec);
                *reinterpret_cast<float4 *>(&sm.s2[c][row][4 * q]) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CK; ++c) {
            const float4 a4 = *reinterpret_cast<const float4 *>(&sm.s1[c][r][4 * g]);
            const float4 w0 = *reinterpret_cast<const float4 *>(&sm.s2[c][r + wave][4 * g]);
            const float4 w1 = *reinterpret_cast<const float4 *>(&sm.s2[c][r + wave][4 * g + 4]);
            const float4 w2 = *reinterpret_cast<const float4 *>(&sm.s2[c][r + wave][4 * g + 8]);
            const float a[kPX] = {a4.x, a4.y, a4.z, a4.w};
            const float w[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
#pragma unroll
            for (int dx = 0; dx < kND; ++dx)
#pragma unroll
                for (int p = 0; p < kPX; ++p) acc[dx][p] = fmaf(a[p], w[p + dx], acc[dx][p]);
        }
    }

    const int y = y0 + r;
    const int x = x0 + 4 * g;
    if (y >= H || x >= W) return;
    T *po = out + (int64_t)b * bso + (int64_t)(wave * kND) * plane + (int64_t)y * W + x;
#pragma unroll
    for (int dx = 0; dx < kND; ++dx) {
        float v[kPX];
#pragma unroll
        for (int p = 0; p < kPX; ++p) {
            v[p] = acc[dx][p] * scale;
            if (do_leaky) v[p] = leaky(v[p], slope);
        }
        T *q = po + (int64_t)dx * plane;
        if (vec) {
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4 *>(q) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                __half2 lo = __floats2half2_rn(v[0], v[1]);
                __half2 hi = __floats2half2_rn(v[2], v[3]);
                uint2 raw;
                raw.x = *reinterpret_cast<unsigned *>(&lo);
                raw.y = *reinterpret_cast<unsigned *>(&hi);
                *reinterpret_cast<uint2 *>(q) = raw;
            }
        } else {
#pragma unroll
            for (int p = 0; p < kPX; ++p)
                if (x + p < W) q[p] = from_f32<T>(v[p]);
        }
    }
}

// ---- fp32 fast path v3 -------------------------------------------------------------------------------------
// Same tile / lane roles as corr81_kernel (wave = dy, lane = 4 pixels x 9 dx), restructured around what the
// profiles showed on MI355X (profiles/r01_corr_ablation.md):
//   * PRODUCER / CONSUMER waves.  A wave's LDS-DMA issue stalls while its older LDS-DMAs are still in flight,
//     so when the nine fma waves issued the ring's DMA themselves every phase serialised (v2: shell + fma +
//     loads + stores ~ total).  Now a tenth wave does nothing but run the ring: it issues the 15
//     buffer_load_dwordx4 ... lds of chunk s+2, waits (counted vmcnt) for chunk s+1, and meets the fma waves
//     at one barrier per chunk; the fma waves touch VMEM only for their 16-byte output stores.
//   * PERSISTENT workgroups walk a strided list of tiles and the 3-slot ring runs ahead ACROSS tile
//     boundaries, so the stores of tile t drain while tile t+1 is already being multiplied.
//   * v_pk_fma_f32 without repacking: for pixel p the accumulators are paired over dx so that the in2
//     operand pair (w[p+dx], w[p+dx+1]) starts at an EVEN window index (= an aligned VGPR pair straight
//     out of ds_read_b128) and in1[p] is broadcast: 4 packed + 1 scalar fma per pixel instead of 9.
//   * zero padding, ragged edges and the ragged last channel chunk come from the buffer range check.
// Needs W % 4 == 0 and 16-byte aligned operands (the launcher falls back to corr81_kernel otherwise).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef PWC_CORR_EXP
#define PWC_CORR_EXP 0       // timing experiments on corr81_dma_kernel (results invalid): 1 = the fma waves skip the arithmetic, 2 = every fetch reads
#endif                       // channel chunk 0 of tile 0 (cache-resident), 4 = no LDS-DMA inside the loop, 8 = no output stores
constexpr int kCKd = 4;                                   // channels per chunk
#ifndef PWC_CORR_RING
#define PWC_CORR_RING 3
#endif
constexpr int kRing = PWC_CORR_RING;                      // LDS slots: one consumed, two in flight (2: 76 us, 3: 65 us, 4: 75 us)
constexpr int kS2Floats = kCKd * kS2Rows * kPitch;        // 2560: in2 halo tile [c][16][40]
constexpr int kS1Floats = kCKd * kTH * kPitch;            // 1280: in1 tile      [c][8][40] (cols 32..39 unused)
constexpr int kS2Instr = kS2Floats / 4 / 64;              // 10 wave-instructions of 64 x 16 B
constexpr int kS1Instr = kS1Floats / 4 / 64;              // 5
constexpr int kDmaInstr = kS2Instr + kS1Instr;            // 15 per chunk, all issued by the loader wave
constexpr int kBufFloats = kS2Floats + kS1Floats;         // 3840 floats = 15 KiB per slot
constexpr unsigned kOOBv = 0x80000000u;
constexpr int kLoaderWave = kND;                          // wave 9 (a second loader wave measured slower: 72 vs 65 us)
constexpr int kThreadsDma = 64 * (kND + 1);               // 640
// Fused warp + correlation (PWCNet.py:212-213, 226-227, 240-241, 256-257: warp(c2, up_flow*s) is consumed by the correlation
// only): five more waves per workgroup PRODUCE the in2 halo tile of every chunk -- each lane owns two of the 640 halo pixels,
// computes their sample taps once per tile and per chunk gathers 2 x 8 bytes per channel, blends and writes the warped value
// into the ring slot with ds_write -- while the loader wave keeps streaming in1 by LDS-DMA and the nine fma waves run
// unchanged.  The warped tensor never exists in HBM (one write + one read of c2 per level less) and the sample taps are
// computed 2.5x per pixel (halo) instead of once per 8-channel block of every pixel.
#ifndef PWC_WARPCORR_WPE
#define PWC_WARPCORR_WPE 8          // waves per SIMD the fused kernel's register budget must allow (8: two workgroups per CU)
#endif
constexpr int kProducers = 5;
constexpr int kThreadsWarp = 64 * (kND + 1 + kProducers);  // 960
static_assert(kProducers * 64 * 2 == kS2Rows * 40, "two halo pixels per producer lane");

struct WarpArgs {
    const float *flo;          // [B,2,H,W] (u, v); nullptr = plain correlation
    int64_t bsf;
    float flow_scale, thr;
    int align_corners;
};
#ifndef PWC_CORR_PER_CU
#define PWC_CORR_PER_CU 2
#endif
constexpr int kPersistentPerCU = PWC_CORR_PER_CU;

struct TileXY { int b, x0, y0; };

__device__ __forceinline__ TileXY tile_of(int t, int nblk, int tiles_x, int tiles_y) {
    // tiles are dealt so that each XCD (blocks i, i+8, ... share one) owns a contiguous run: the +-4 halo
    // re-read by neighbouring tiles then hits in that XCD's L2 (speed only, any mapping is correct)
    if ((nblk & 7) == 0) t = (t & 7) * (nblk >> 3) + (t >> 3);
    TileXY r;
    r.x0 = (t % tiles_x) * kTW;
    t /= tiles_x;
    r.y0 = (t % tiles_y) * kTH;
    r.b = t / tiles_y;
    return r;
}

// per-lane source offsets of the 15 DMA instructions of one chunk (loader wave only)
__device__ __forceinline__ void corr_offsets(unsigned (&off)[kDmaInstr], int lane, const TileXY &t, int H, int W, int plane) {
#pragma unroll
    for (int k = 0; k < kDmaInstr; ++k) {
        int c, row, q, iy, ix;
        bool ok;
        if (k < kS2Instr) {
            const int p = k * 64 + lane;
            c = p / (kS2Rows * 10); row = (p / 10) % kS2Rows; q = p % 10;
            iy = t.y0 + row - kD; ix = t.x0 + 4 * q - kD;
            ok = true;
        } else {
            const int p = (k - kS2Instr) * 64 + lane;
            c = p / (kTH * 10); row = (p / 10) % kTH; q = p % 10;
            iy = t.y0 + row; ix = t.x0 + 4 * q;
            ok = (q < kTG);
        }
        ok = ok && (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);     // W % 4 == 0: a piece is all-in or all-out
        off[k] = ok ? (unsigned)(c * plane + iy * W + ix) * 4u : kOOBv;
    }
}

template <bool WARP>
__device__ __forceinline__ void corr_issue(const float *p1, const float *p2, int c0, int C, int plane, float *buf,
                                           const unsigned (&off)[kDmaInstr]) {
    const int nbytes = min(kCKd, C - c0) * plane * 4;
    const pwc::v4i32 r2 = pwc::make_rsrc(p2 + (int64_t)c0 * plane, nbytes);
    const pwc::v4i32 r1 = pwc::make_rsrc(p1 + (int64_t)c0 * plane, nbytes);
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)pwc::lds_addr(buf));
    if constexpr (!WARP) {                      // fused kernel: the in2 tile is produced by the warp waves
#pragma unroll
        for (int k = 0; k < kS2Instr; ++k) PWC_CORR_DMA(r2, base + k * 1024, off[k]);
    }
#pragma unroll
    for (int k = kS2Instr; k < kDmaInstr; ++k) PWC_CORR_DMA(r1, base + k * 1024, off[k]);
}

// (fused: 2 workgroups x 15 waves per CU = 7.5 per SIMD -> at most 64 registers)
template <bool WARP>
__global__ void __launch_bounds__(WARP ? kThreadsWarp : kThreadsDma, WARP ? PWC_WARPCORR_WPE : 1)
corr81_dma_kernel(const float *__restrict__ in1, const float *__restrict__ in2, float *__restrict__ out,
                  int C, int H, int W, int tiles_x, int tiles_y, int nblk,
                  int64_t bs1, int64_t bs2, int64_t bso, float scale, float slope, int do_leaky, WarpArgs wa) {
    __shared__ __attribute__((aligned(16))) float smem[kRing * kBufFloats];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..8: dyi = dy + 4;  9: loader
    const int lane = tid & 63;
    const int plane = H * W;
    const int nchunks = (C + kCKd - 1) / kCKd;
    const int stride = gridDim.x;
    const int my_tiles = (nblk - (int)blockIdx.x + stride - 1) / stride;      // tiles blockIdx.x, +stride, ...
    const int nsteps = my_tiles * nchunks;

    if (wave == kLoaderWave) {
        // ================= producer: keeps two chunks in flight ahead of the consumers =====================
        // the ring must never wait for issue slots behind nine fma waves (57.5 -> 56.5 us, -DPWC_CORR_NO_PRIO to compare)
#ifndef PWC_CORR_NO_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        int is_tile = blockIdx.x, is_chunk = 0, is_step = 0;
        unsigned off[kDmaInstr];
        const float *ip1 = nullptr, *ip2 = nullptr;
        auto issue_next = [&]() {
            if (is_step >= nsteps) return;
            if (is_chunk == 0 && (!(PWC_CORR_EXP & 2) || is_step == 0)) {
                const TileXY t = tile_of(is_tile, nblk, tiles_x, tiles_y);
                corr_offsets(off, lane, t, H, W, plane);
                ip1 = in1 + (int64_t)t.b * bs1;
                ip2 = in2 + (int64_t)t.b * bs2;
            }
            if (!(PWC_CORR_EXP & 4) || is_step < kRing - 1)
                corr_issue<WARP>(ip1, ip2, (PWC_CORR_EXP & 2) ? 0 : is_chunk * kCKd, C, plane, smem + (is_step % kRing) * kBufFloats, off);
            ++is_step;
            if (++is_chunk == nchunks) { is_chunk = 0; is_tile += stride; }
        };
#pragma unroll
        for (int k = 0; k < kRing - 1; ++k) issue_next();
        for (int s = 0; s < nsteps; ++s) {
            // chunk s has landed; the chunks issued after it (up to kRing-2 of them) may stay in flight
            const int ahead = min(kRing - 2, nsteps - 1 - s);
            constexpr int kI = WARP ? kS1Instr : kDmaInstr;        // LDS-DMA instructions this wave issues per chunk
            if (ahead >= 2)      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * kI) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kI) : "memory");
            else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();      // B_s: consumers may read slot s%R; they have finished slot (s-1)%R
            issue_next();                      // chunk s+R-1 -> slot (s-1)%R
        }
        return;
    }

    if constexpr (WARP) {
        if (wave > kLoaderWave) {
            // ================= warp producers: chunk s+2 is written while the fma waves consume chunk s ===========
            const int pw = wave - (kLoaderWave + 1);
#ifndef PWC_CORR_PROD_PRIO
#define PWC_CORR_PROD_PRIO 2
#endif
            // issue priority above the fma waves these waves share their SIMDs with (the loader runs at 3): the producers set the pace
            if (PWC_CORR_PROD_PRIO) __builtin_amdgcn_s_setprio(PWC_CORR_PROD_PRIO);
            // (A variant that software-pipelined the two halo pixels of a lane as half-chunks -- one half's gathers in flight
            // while the other is blended -- measured SLOWER, 131 vs 119 us at level 2: the producers are bound by instruction
            // issue next to nine fma waves, not by gather latency; they have two ring steps of slack per chunk anyway.)
            int ldsoff[2], otop[2], obot[2];
            float wgt[2][4];
            const float *src = nullptr;
            int pr_tile = blockIdx.x, pr_chunk = 0, pr_step = 0;
            auto produce = [&]() {
                if (pr_step >= nsteps) return;
                if (pr_chunk == 0) {
                    const TileXY t = tile_of(pr_tile, nblk, tiles_x, tiles_y);
                    const float *fu = wa.flo + (int64_t)t.b * wa.bsf;
                    src = in2 + (int64_t)t.b * bs2;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int hp = (pw * 2 + k) * 64 + lane;              // consecutive lanes = consecutive halo columns
                        const int row = hp / 40, col = hp - row * 40;
                        const int gy = t.y0 - kD + row, gx = t.x0 - kD + col;
                        const bool inimg = (gy >= 0) && (gy < H) && (gx >= 0) && (gx < W);
                        const int gyc = min(max(gy, 0), H - 1), gxc = min(max(gx, 0), W - 1);
                        const float u = fu[(int64_t)gyc * W + gxc] * wa.flow_scale;
                        const float v = fu[(int64_t)plane + (int64_t)gyc * W + gxc] * wa.flow_scale;
                        const pwc_warp::PairTaps pt = pwc_warp::make_pair_taps((float)gx + u, (float)gy + v, H, W, wa.align_corners, wa.thr);
                        otop[k] = pt.otop * 4;
                        obot[k] = pt.obot * 4;
                        // a halo pixel outside the image is the correlation's zero padding: all four weights zero
                        wgt[k][0] = inimg ? pt.wa : 0.f; wgt[k][1] = inimg ? pt.wb : 0.f;
                        wgt[k][2] = inimg ? pt.wc : 0.f; wgt[k][3] = inimg ? pt.wd : 0.f;
                        ldsoff[k] = row * kPitch + col;
                    }
                }
                const int c0 = pr_chunk * kCKd;
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    pwc::uniform_ptr(src + (int64_t)c0 * plane), 0, __builtin_amdgcn_readfirstlane(min(kCKd, C - c0) * plane * 4), 0x00020000);
                float *slot = smem + (pr_step % kRing) * kBufFloats;
#ifdef PWC_WC_SEQ
                // one halo pixel at a time: 8 gathers in flight instead of 16, 16 fewer live registers
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    f32x2 top[kCKd], bot[kCKd];
#pragma unroll
                    for (int c = 0; c < kCKd; ++c) {
                        top[c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, otop[k], c * plane * 4, 0));
                        bot[c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, obot[k], c * plane * 4, 0));
                    }
#pragma unroll
                    for (int c = 0; c < kCKd; ++c)
                        slot[c * kS2Rows * kPitch + ldsoff[k]] =
                            pwc_warp::blend4(top[c][0], top[c][1], bot[c][0], bot[c][1], wgt[k][0], wgt[k][1], wgt[k][2], wgt[k][3]);
                    asm volatile("" ::: "memory");
                }
#else
                f32x2 top[2][kCKd], bot[2][kCKd];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int c = 0; c < kCKd; ++c) {       // channels past C fail the range check and read as 0 (ragged last chunk)
                        top[k][c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, otop[k], c * plane * 4, 0));
                        bot[k][c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, obot[k], c * plane * 4, 0));
                    }
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int c = 0; c < kCKd; ++c)
                        slot[c * kS2Rows * kPitch + ldsoff[k]] =
                            pwc_warp::blend4(top[k][c][0], top[k][c][1], bot[k][c][0], bot[k][c][1], wgt[k][0], wgt[k][1], wgt[k][2], wgt[k][3]);
#endif
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the tile is in LDS before this wave reaches the barrier
                ++pr_step;
                if (++pr_chunk == nchunks) { pr_chunk = 0; pr_tile += stride; }
            };
#pragma unroll
            for (int k = 0; k < kRing - 1; ++k) produce();
            for (int s = 0; s < nsteps; ++s) {
                __builtin_amdgcn_s_barrier();      // B_s: chunk s is readable, slot (s-1)%R is free
                produce();                         // chunk s+R-1 -> slot (s-1)%R
            }
            return;
        }
    }

    // ================= consumers: wave = displacement row dy, lane = 4 pixels x 9 dx ======================
    int r, g;
    lane_to_rg(lane, r, g);
    // pixel p even: pairs dx = (0,1)(2,3)(4,5)(6,7) + single dx 8;  p odd: pairs (1,2)(3,4)(5,6)(7,8) + single dx 0
    f32x2 acc2[kPX][4];
    float acc1[kPX];
#pragma unroll
    for (int p = 0; p < kPX; ++p) {
        acc1[p] = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) acc2[p][m] = (f32x2){0.f, 0.f};
    }
    int tile = blockIdx.x, chunk = 0;
    for (int s = 0; s < nsteps; ++s) {
        __builtin_amdgcn_s_barrier();          // B_s
        asm volatile("" ::: "memory");
        const float *cur = smem + (s % kRing) * kBufFloats;
        const float *s2 = cur + (r + wave) * kPitch + 4 * g;
        const float *s1 = cur + kS2Floats + r * kPitch + 4 * g;
#pragma unroll 2
        for (int c = 0; c < ((PWC_CORR_EXP & 1) ? 0 : kCKd); ++c) {
            const float4 a4 = *reinterpret_cast<const float4 *>(s1 + c * kTH * kPitch);
            const float4 w0 = *reinterpret_cast<const float4 *>(s2 + c * kS2Rows * kPitch);
            const float4 w1 = *reinterpret_cast<const float4 *>(s2 + c * kS2Rows * kPitch + 4);
            const float4 w2 = *reinterpret_cast<const float4 *>(s2 + c * kS2Rows * kPitch + 8);
            const float a[kPX] = {a4.x, a4.y, a4.z, a4.w};
            const f32x2 wp[6] = {{w0.x, w0.y}, {w0.z, w0.w}, {w1.x, w1.y}, {w1.z, w1.w}, {w2.x, w2.y}, {w2.z, w2.w}};
            const float ws[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
#pragma unroll
            for (int p = 0; p < kPX; ++p) {
                const f32x2 ap = {a[p], a[p]};
                const int m0 = (p + 1) / 2;                 // first aligned window pair: index p (p even) / p+1 (p odd)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc2[p][m] = __builtin_elementwise_fma(ap, wp[m0 + m], acc2[p][m]);
                acc1[p] = fmaf(a[p], (p & 1) ? ws[p] : ws[p + 8], acc1[p]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS reads done before the next barrier frees the slot
        if (++chunk == nchunks) {
            // ---- tile finished: scale / LeakyReLU / 16-byte stores (they drain while the next tile runs)
            const TileXY t = tile_of(tile, nblk, tiles_x, tiles_y);
            const int y = t.y0 + r;
            const int x = t.x0 + 4 * g;
            if (y < H && x < W && !(PWC_CORR_EXP & 8)) {
                float *po = out + (int64_t)t.b * bso + (int64_t)(wave * kND) * plane + (int64_t)y * W + x;
#pragma unroll
                for (int dx = 0; dx < kND; ++dx) {
                    float v[kPX];
#pragma unroll
                    for (int p = 0; p < kPX; ++p) {
                        float q;
                        if (p & 1) q = (dx == 0) ? acc1[p] : acc2[p][(dx - 1) / 2][(dx - 1) & 1];
                        else       q = (dx == 8) ? acc1[p] : acc2[p][dx / 2][dx & 1];
                        q *= scale;
                        v[p] = do_leaky ? leaky(q, slope) : q;
                    }
#ifdef PWC_CORR_NT_STORE
                    __builtin_nontemporal_store(f32x4v{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4v *>(po + (int64_t)dx * plane));
#else
                    *reinterpret_cast<float4 *>(po + (int64_t)dx * plane) = make_float4(v[0], v[1], v[2], v[3]);
#endif
                }
            }
#pragma unroll
            for (int p = 0; p < kPX; ++p) {
                acc1[p] = 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m) acc2[p][m] = (f32x2){0.f, 0.f};
            }
            chunk = 0;
            tile += stride;
        }
    }
}

// Any (pad, kernel, max_disp, stride1, stride2): one thread per output element.
template <typename T>
__global__ void __launch_bounds__(256)
corr_generic_kernel(const T *__restrict__ in1, const T *__restrict__ in2, T *__restrict__ out,
                    int C, int H, int W, int nch, int oh, int ow,
                    int pad, int krad, int max_disp, int s1, int s2, int drad,
                    int64_t bs1, int64_t bs2, int64_t bso, int64_t total,
                    float scale, float slope, int do_leaky) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % ow);
    int64_t t = idx / ow;
    const int oy = (int)(t % oh);
    t /= oh;
    const int tc = (int)(t % nch);
    const int b = (int)(t / nch);
    const int Dd = 2 * drad + 1;
    const int tj = tc / Dd - drad;
    const int ti = tc % Dd - drad;
    // coordinates in the padded frame, then back to the unpadded one
    const int y1 = oy * s1 + max_disp - pad;
    const int x1 = ox * s1 + max_disp - pad;
    const int y2 = y1 + tj * s2;
    const int x2 = x1 + ti * s2;
    const int64_t plane = (int64_t)H * W;
    const T *p1 = in1 + (int64_t)b * bs1;
    const T *p2 = in2 + (int64_t)b * bs2;
    float acc = 0.f;
    for (int j = -krad; j <= krad; ++j) {
        for (int i = -krad; i <= krad; ++i) {
            const int ya = y1 + j, xa = x1 + i, yb = y2 + j, xb = x2 + i;
            if (ya < 0 || ya >= H || xa < 0 || xa >= W || yb < 0 || yb >= H || xb < 0 || xb >= W) continue;
            const T *qa = p1 + (int64_t)ya * W + xa;
            const T *qb = p2 + (int64_t)yb * W + xb;
            for (int c = 0; c < C; ++c) acc = fmaf(to_f32<T>(qa[c * plane]), to_f32<T>(qb[c * plane]), acc);
        }
    }
    float v = acc * scale;
    if (do_leaky) v = leaky(v, slope);
    out[(int64_t)b * bso + (int64_t)tc * oh * ow + (int64_t)oy * ow + ox] = from_f32<T>(v);
}

// ---- backward ------------------------------------------------------------------------------------------------
// Gradients of the cost volume w.r.t. both inputs (reference correlation_cuda_kernel.cu:150-334: one kernel per input
// and per batch item, scatter-free because it loops over the outputs that touched an input element).  Both kernels
// here are GATHERS with a fixed summation order: deterministic, no atomics.
//
// Generic form, any (pad, kernel_size, max_disp, stride1, stride2): one thread per input element (b, c, y, x).
//   forward: out[tc, oy, ox] += in1[c, y1+j, x1+i] * in2[c, y1+tj*s2+j, x1+ti*s2+i],  y1 = oy*s1 + max_disp - pad
//   d in1[c, ya, xa] = sum_{tc, j, i} gout[tc, (ya - j - max_disp + pad)/s1, ...] * in2[c, ya + tj*s2, xa + ti*s2]
//   d in2[c, yb, xb] = sum_{tc, j, i} gout[tc, (yb - tj*s2 - j - max_disp + pad)/s1, ...] * in1[c, yb - tj*s2, xb - ti*s2]
// (terms whose output index is fractional / out of range or whose partner pixel lies outside the image vanish).
template <typename T>
__global__ void __launch_bounds__(256)
corr_bwd_generic_kernel(const T *__restrict__ in1, const T *__restrict__ in2, const T *__restrict__ gout,
                        T *__restrict__ g1, T *__restrict__ g2,
                        int C, int H, int W, int oh, int ow, int pad, int krad, int max_disp, int s1, int s2, int drad,
                        int64_t total, float scale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % W);
    int64_t t = idx / W;
    const int y = (int)(t % H);
    t /= H;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const int Dd = 2 * drad + 1;
    const int64_t plane = (int64_t)H * W, oplane = (int64_t)oh * ow;
    const T *go = gout + (int64_t)b * Dd * Dd * oplane;
    const T *a = in1 + ((int64_t)b * C + c) * plane;
    const T *bb = in2 + ((int64_t)b * C + c) * plane;
    float acc1 = 0.f, acc2 = 0.f;
    for (int tj = -drad; tj <= drad; ++tj) {
        for (int ti = -drad; ti <= drad; ++ti) {
            const T *gc = go + (int64_t)((tj + drad) * Dd + (ti + drad)) * oplane;
            const int dy = tj * s2, dx = ti * s2;
            for (int j = -krad; j <= krad; ++j) {
                for (int i = -krad; i <= krad; ++i) {
                    {   // this element as in1[.., ya = y, xa = x]
                        const int ny = y - j - max_disp + pad, nx = x - i - max_disp + pad;
                        const int yb = y + dy, xb = x + dx;
                        if (ny >= 0 && nx >= 0 && ny % s1 == 0 && nx % s1 == 0 && ny / s1 < oh && nx / s1 < ow &&
                            yb >= 0 && yb < H && xb >= 0 && xb < W)
                            acc1 = fmaf(to_f32<T>(gc[(int64_t)(ny / s1) * ow + nx / s1]), to_f32<T>(bb[(int64_t)yb * W + xb]), acc1);
                    }
                    {   // this element as in2[.., yb = y, xb = x]
                        const int ny = y - dy - j - max_disp + pad, nx = x - dx - i - max_disp + pad;
                        const int ya = y - dy, xa = x - dx;
                        if (ny >= 0 && nx >= 0 && ny % s1 == 0 && nx % s1 == 0 && ny / s1 < oh && nx / s1 < ow &&
                            ya >= 0 && ya < H && xa >= 0 && xa < W)
                            acc2 = fmaf(to_f32<T>(gc[(int64_t)(ny / s1) * ow + nx / s1]), to_f32<T>(a[(int64_t)ya * W + xa]), acc2);
                    }
                }
            }
        }
    }
    g1[idx] = from_f32<T>(acc1 * scale);
    g2[idx] = from_f32<T>(acc2 * scale);
}

// PWC-Net's configuration (pad 4, kernel 1, max displacement 4, strides 1), fp32: tiled like the forward.  One thread owns one
// pixel of an 8 x 32 tile and keeps its 81 + 81 gradient-of-output values in registers -- g1[d] = gout[d, p] for d in1 and
// g2[d] = gout[d, p - d] for d in2, loaded once -- while the input maps stream through LDS in chunks of 8 channels WITH their
// +-4 halos (coalesced 16-byte row loads, zero outside the image); per channel the thread does 2 x 81 fma on conflict-free
// ds_read_b32 (consecutive lanes = consecutive columns).  HBM traffic = the algorithmic (4*C + 81) * H * W * 4 bytes plus the
// 2.5x halo re-reads, which hit in L2.
constexpr int kBwdCK = 8;
__global__ void __launch_bounds__(256)
corr81_bwd_kernel(const float *__restrict__ in1, const float *__restrict__ in2, const float *__restrict__ gout,
                  float *__restrict__ g1o, float *__restrict__ g2o, int C, int H, int W, int tiles_x, int tiles_y, float scale) {
    __shared__ __attribute__((aligned(16))) float s1[kBwdCK][kS2Rows][kPitch];     // in1 halo tile
    __shared__ __attribute__((aligned(16))) float s2[kBwdCK][kS2Rows][kPitch];     // in2 halo tile
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx * kTW, y0 = ty * kTH;
    const int px = tid & 31, py = tid >> 5;
    const int x = x0 + px, y = y0 + py;
    const bool inside = (x < W) && (y < H);
    const int64_t plane = (int64_t)H * W;
    const float *go = gout + (int64_t)b * 81 * plane;
    float ga[81], gb[81];
#pragma unroll
    for (int d = 0; d < 81; ++d) {
        const int dy = d / 9 - kD, dx = d % 9 - kD;
        ga[d] = inside ? go[(int64_t)d * plane + (int64_t)y * W + x] : 0.f;
        const int yy = y - dy, xx = x - dx;
        gb[d] = (inside && yy >= 0 && yy < H && xx >= 0 && xx < W) ? go[(int64_t)d * plane + (int64_t)yy * W + xx] : 0.f;
    }
    const float *p1 = in1 + (int64_t)b * C * plane;
    const float *p2 = in2 + (int64_t)b * C * plane;
    const bool vec = (W % 4 == 0);
    for (int c0 = 0; c0 < C; c0 += kBwdCK) {
        if (c0) __syncthreads();
        // stage both halo tiles: kBwdCK x 16 rows x 10 quads each
        for (int i = tid; i < 2 * kBwdCK * kS2Rows * kS2Quads; i += 256) {
            const int which = i / (kBwdCK * kS2Rows * kS2Quads);
            const int rem = i % (kBwdCK * kS2Rows * kS2Quads);
            const int c = rem / (kS2Rows * kS2Quads);
            const int row = (rem / kS2Quads) % kS2Rows;
            const int q = rem % kS2Quads;
            const int yy = y0 + row - kD;
            const bool ok = (c0 + c < C) && (yy >= 0) && (yy < H);
            const float *src = (which ? p2 : p1) + (int64_t)(c0 + c) * plane + (int64_t)yy * W;
            const float4 v = load_quad<float>(src, x0 + 4 * q - kD, W, ok, vec);
            *reinterpret_cast<float4 *>(which ? &s2[c][row][4 * q] : &s1[c][row][4 * q]) = v;
        }
        __syncthreads();
        const int cn = min(kBwdCK, C - c0);
        for (int c = 0; c < cn; ++c) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int d = 0; d < 81; ++d) {
                const int dyi = d / 9, dxi = d % 9;
                a1 = fmaf(ga[d], s2[c][py + dyi][px + dxi], a1);                 // in2[c, p + d]
                a2 = fmaf(gb[d], s1[c][py + 8 - dyi][px + 8 - dxi], a2);         // in1[c, p - d]
            }
            if (inside) {
                const int64_t o = ((int64_t)b * C + c0 + c) * plane + (int64_t)y * W + x;
                g1o[o] = a1 * scale;
                g2o[o] = a2 * scale;
            }
        }
    }
}

// ---- small maps (pyramid levels 6-4, batch-1 inference) ----------------------------------------------------------------------------
// The tiled kernels above give a workgroup an 8x32-pixel tile and walk the channels in 4-channel chunks through an LDS ring: with
// a handful of tiles per launch (level 6: ONE 7x16 map per image, 196 channels = 49 chunks) the launch is a chain of DMA round
// trips on a few CUs -- 35 us at every batch size up to 16.  Here the parallelism comes from the displacements and the channels
// instead: a workgroup = 64 consecutive pixels x one displacement row (9 outputs per pixel), its four waves take every fourth
// channel straight from L2 (buffer loads: the zero padding is the range check, the channel a wave-uniform offset), and add up
// through LDS in a fixed order.  Same operator, another summation order than the tiled kernels (a chain per channel quarter instead
// of one over all channels): equal to them within fp32 rounding, bit-repeatable, independent of the batch slot.
constexpr int kSmallWaves = 4;
__global__ void __launch_bounds__(64 * kSmallWaves)
corr81_small_kernel(const float *__restrict__ in1, const float *__restrict__ in2, float *__restrict__ out, int C, int H, int W, int npb,
                    int64_t bs1, int64_t bs2, int64_t bso, float scale, float slope, int do_leaky) {
    __shared__ float red[kSmallWaves][kND][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = blockIdx.x;
    const int dy = bid % kND;
    bid /= kND;
    const int pb = bid % npb, b = bid / npb;
    const int plane = H * W;
    const int p = pb * 64 + lane;
    const int py = p / W, px = p - py * W;
    const int y2 = py + dy - kD;
    const bool rowok = (p < plane) && (y2 >= 0) && (y2 < H);
    unsigned off2[kND];
#pragma unroll
    for (int dx = 0; dx < kND; ++dx) {
        const int x2 = px + dx - kD;
        off2[dx] = (rowok && x2 >= 0 && x2 < W) ? (unsigned)(y2 * W + x2) * 4u : kOOBv;
    }
    const unsigned off1 = (p < plane) ? (unsigned)p * 4u : kOOBv;
    const int nbytes = __builtin_amdgcn_readfirstlane(C * plane * 4);
    __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(pwc::uniform_ptr(in1 + (int64_t)b * bs1), 0, nbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(pwc::uniform_ptr(in2 + (int64_t)b * bs2), 0, nbytes, 0x00020000);
    float acc[kND];
#pragma unroll
    for (int dx = 0; dx < kND; ++dx) acc[dx] = 0.f;
    auto ld1 = [&](int c) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, off1, __builtin_amdgcn_readfirstlane(c * plane * 4), 0)); };
    auto ld2 = [&](int c, int dx) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2, off2[dx], __builtin_amdgcn_readfirstlane(c * plane * 4), 0)); };
    int c = wave;
    for (; c + 3 * kSmallWaves < C; c += 4 * kSmallWaves) {          // four channels' 40 loads in flight per wave
        float a[4], v[4][kND];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = ld1(c + u * kSmallWaves);
#pragma unroll
            for (int dx = 0; dx < kND; ++dx) v[u][dx] = ld2(c + u * kSmallWaves, dx);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int dx = 0; dx < kND; ++dx) acc[dx] = fmaf(a[u], v[u][dx], acc[dx]);
    }
    for (; c < C; c += kSmallWaves) {
        const float a = ld1(c);
#pragma unroll
        for (int dx = 0; dx < kND; ++dx) acc[dx] = fmaf(a, ld2(c, dx), acc[dx]);
    }
#pragma unroll
    for (int dx = 0; dx < kND; ++dx) red[wave][dx][lane] = acc[dx];
    __syncthreads();
    float *po = out + (int64_t)b * bso + (int64_t)dy * kND * plane;
    for (int i = tid; i < kND * 64; i += 64 * kSmallWaves) {
        const int dx = i >> 6, l = i & 63, pp = pb * 64 + l;
        if (pp >= plane) continue;
        float v = ((red[0][dx][l] + red[1][dx][l]) + red[2][dx][l]) + red[3][dx][l];
        v *= scale;
        if (do_leaky) v = pwc::leaky(v, slope);
        po[(int64_t)dx * plane + pp] = v;
    }
}

template <typename T>
int launch_corr(const void *in1, const void *in2, void *out, int B, int C, int H, int W,
                int pad, int ksz, int max_disp, int s1, int s2, float scale, unsigned flags, float slope,
                int64_t bs1, int64_t bs2, int64_t bso, hipStream_t st) {
    const int krad = (ksz - 1) / 2;
    const int drad = max_disp / s2;
    const int Dd = 2 * drad + 1;
    const int nch = Dd * Dd;
    const int border = krad + max_disp;
    const int oh = (H + 2 * pad - 2 * border + s1 - 1) / s1;
    const int ow = (W + 2 * pad - 2 * border + s1 - 1) / s1;
    if (oh <= 0 || ow <= 0) PWC_FAIL(PWC_EINVAL, "pwc_corr_fwd: empty output (%d x %d)", oh, ow);
    const int do_leaky = (flags & PWC_ACT_LEAKY) ? 1 : 0;
    const T *a = static_cast<const T *>(in1);
    const T *b = static_cast<const T *>(in2);
    T *o = static_cast<T *>(out);

    if (ksz == 1 && max_disp == kD && pad == kD && s1 == 1 && s2 == 1) {
        const int tiles_x = (W + kTW - 1) / kTW;
        const int tiles_y = (H + kTH - 1) / kTH;
        const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
        if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_corr_fwd: grid too large");
        const int elt = (int)sizeof(T);
        const int va = 4;  // elements per vector access
        const bool ptr_ok = ((reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(in2) |
                              reinterpret_cast<uintptr_t>(out)) & (uintptr_t)(va * elt - 1)) == 0;
        const int vec = (W % 4 == 0) && ptr_ok && (bs1 % 4 == 0) && (bs2 % 4 == 0) && (bso % 4 == 0);
        if constexpr (sizeof(T) == 4) {
            if (nblk <= pwc::option(pwc::OPT_CORR_SMALL_TILES) && (int64_t)C * H * W * 4 < 0x7fffffffLL) {      // small maps: see corr81_small_kernel
                const int npb = (H * W + 63) / 64;
                hipLaunchKernelGGL(corr81_small_kernel, dim3((unsigned)(B * npb * kND)), dim3(64 * kSmallWaves), 0, st,
                                   a, b, o, C, H, W, npb, bs1, bs2, bso, scale, slope, do_leaky);
                return pwc::check_launch("corr81_small_kernel");
            }
            if (vec && pwc::corr81_pipe_enabled() && pwc::corr81_pipe_fits(B, C, H, W))      // large levels: pwc_corr_pipe.hip
                return pwc::launch_corr81_pipe(a, b, o, B, C, H, W, bs1, bs2, bso, scale, slope, do_leaky, st);
            if (vec && (int64_t)H * W * kCKd * 4 < 0x7fffffffLL) {
                const int grid = (int)((nblk < kPersistentPerCU * 256) ? nblk : kPersistentPerCU * 256);
                hipLaunchKernelGGL(corr81_dma_kernel<false>, dim3((unsigned)grid), dim3(kThreadsDma), 0, st,
                                   a, b, o, C, H, W, tiles_x, tiles_y, (int)nblk, bs1, bs2, bso, scale, slope, do_leaky, WarpArgs{});
                return pwc::check_launch("corr81_dma_kernel");
            }
        }
        hipLaunchKernelGGL((corr81_kernel<T, 8>), dim3((unsigned)nblk), dim3(kThreads), 0, st,
                           a, b, o, C, H, W, tiles_x, tiles_y, bs1, bs2, bso, scale, slope, do_leaky, vec);
        return pwc::check_launch("corr81_kernel");
    }
    const int64_t total = (int64_t)B * nch * oh * ow;
    const int64_t nblk = (total + 255) / 256;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_corr_fwd: grid too large");
    hipLaunchKernelGGL((corr_generic_kernel<T>), dim3((unsigned)nblk), dim3(256), 0, st,
                       a, b, o, C, H, W, nch, oh, ow, pad, krad, max_disp, s1, s2, drad,
                       bs1, bs2, bso, total, scale, slope, do_leaky);
    return pwc::check_launch("corr_generic_kernel");
}

}  // namespace

extern "C" int pwc_corr_fwd(const void *in1, const void *in2, void *out,
                            int B, int C, int H, int W,
                            int pad_size, int kernel_size, int max_disp, int stride1, int stride2,
                            float corr_multiply, int dtype, unsigned flags, float leaky_slope,
                            int64_t in1_bstride, int64_t in2_bstride, int64_t out_bstride,
                            void *stream) {
    if (!in1 || !in2 || !out) PWC_FAIL(PWC_EINVAL, "pwc_corr_fwd: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_corr_fwd: bad shape %dx%dx%dx%d", B, C, H, W);
    if (kernel_size < 1 || (kernel_size & 1) == 0 || max_disp < 0 || stride1 < 1 || stride2 < 1 || pad_size < 0)
        PWC_FAIL(PWC_EINVAL, "pwc_corr_fwd: bad parameters pad=%d k=%d d=%d s1=%d s2=%d", pad_size, kernel_size,
                 max_disp, stride1, stride2);
    const int64_t chw = (int64_t)C * H * W;
    if (in1_bstride < chw || in2_bstride < chw) PWC_FAIL(PWC_EINVAL, "pwc_corr_fwd: input batch stride < C*H*W");
    const float scale = (flags & PWC_CORR_NORMALIZE) ? 1.0f / (float)(kernel_size * kernel_size * C) : corr_multiply;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case PWC_F32:
            return launch_corr<float>(in1, in2, out, B, C, H, W, pad_size, kernel_size, max_disp, stride1, stride2,
                                      scale, flags, leaky_slope, in1_bstride, in2_bstride, out_bstride, st);
        case PWC_F16:
            return launch_corr<__half>(in1, in2, out, B, C, H, W, pad_size, kernel_size, max_disp, stride1, stride2,
                                       scale, flags, leaky_slope, in1_bstride, in2_bstride, out_bstride, st);
        default:
            PWC_FAIL(PWC_EUNSUPPORTED, "pwc_corr_fwd: dtype %d", dtype);
    }
}

extern "C" int pwc_warp_corr81_fwd(const void *in1, const void *x2, const void *flo, void *out, int B, int C, int H, int W,
                                   float flow_scale, int align_corners, float mask_threshold,
                                   float corr_multiply, unsigned flags, float leaky_slope,
                                   int64_t in1_bstride, int64_t x2_bstride, int64_t flo_bstride, int64_t out_bstride, void *stream) {
    if (!in1 || !x2 || !flo || !out) PWC_FAIL(PWC_EINVAL, "pwc_warp_corr81_fwd: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_warp_corr81_fwd: bad shape %dx%dx%dx%d", B, C, H, W);
    const int64_t plane = (int64_t)H * W;
    if (in1_bstride < C * plane || x2_bstride < C * plane || flo_bstride < 2 * plane || out_bstride < 81 * plane)
        PWC_FAIL(PWC_EINVAL, "pwc_warp_corr81_fwd: batch stride smaller than the tensor");
    const uintptr_t al = reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(x2) | reinterpret_cast<uintptr_t>(out);
    if ((W % 4) || (al & 15u) || (in1_bstride % 4) || (x2_bstride % 4) || (out_bstride % 4) || (reinterpret_cast<uintptr_t>(flo) & 3u) ||
        plane * kCKd * 4 >= 0x7fffffffLL) {
        pwc::set_error("pwc_warp_corr81_fwd: needs W %% 4 == 0, 16-byte aligned operands and H*W*16 < 2^31 (call pwc_warp_fwd + pwc_corr_fwd instead)");
        return PWC_EUNSUPPORTED;
    }
    const int tiles_x = (W + kTW - 1) / kTW, tiles_y = (H + kTH - 1) / kTH;
    const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_warp_corr81_fwd: grid too large");
    const float scale = (flags & PWC_CORR_NORMALIZE) ? 1.0f / (float)C : corr_multiply;
    if (pwc::warp_corr81_pipe_fits(B, C, H, W))       // levels 2 and 3: the LDS-window kernel of pwc_corr_pipe.hip
        return pwc::launch_warp_corr81_pipe(static_cast<const float *>(in1), static_cast<const float *>(x2), static_cast<const float *>(flo),
                                            static_cast<float *>(out), B, C, H, W, in1_bstride, x2_bstride, flo_bstride, out_bstride,
                                            flow_scale, align_corners, mask_threshold, scale, leaky_slope, (flags & PWC_ACT_LEAKY) ? 1 : 0,
                                            static_cast<hipStream_t>(stream));
    static const int per_cu = [] { const char *e = getenv("PWC_WARPCORR_PER_CU"); return (e && *e) ? atoi(e) : kPersistentPerCU; }();
    const int grid = (int)((nblk < per_cu * 256) ? nblk : per_cu * 256);
    const WarpArgs wa{static_cast<const float *>(flo), flo_bstride, flow_scale, mask_threshold, align_corners};
    hipLaunchKernelGGL(corr81_dma_kernel<true>, dim3((unsigned)grid), dim3(kThreadsWarp), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(in1), static_cast<const float *>(x2), static_cast<float *>(out), C, H, W,
                       tiles_x, tiles_y, (int)nblk, in1_bstride, x2_bstride, out_bstride, scale, leaky_slope,
                       (flags & PWC_ACT_LEAKY) ? 1 : 0, wa);
    return pwc::check_launch("corr81_dma_kernel<warp>");
}

// Is the fused kernel the faster way to warp + correlate this geometry?  Not for maps of a few tiles: there pwc_warp_fwd followed by
// pwc_corr_fwd (its small-map kernel) wins by 3x, launch included (profiles/r04_corr_notes.md section 6).
extern "C" int pwc_warp_corr81_preferred(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t nblk = (int64_t)B * ((W + kTW - 1) / kTW) * ((H + kTH - 1) / kTH);
    return nblk > pwc::option(pwc::OPT_CORR_SMALL_TILES) ? 1 : 0;
}

extern "C" int pwc_corr_bwd(const void *in1, const void *in2, const void *grad_out, void *grad_in1, void *grad_in2,
                            int B, int C, int H, int W,
                            int pad_size, int kernel_size, int max_disp, int stride1, int stride2,
                            float corr_multiply, int dtype, unsigned flags, void *stream) {
    if (!in1 || !in2 || !grad_out || !grad_in1 || !grad_in2) PWC_FAIL(PWC_EINVAL, "pwc_corr_bwd: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_corr_bwd: bad shape");
    if (kernel_size < 1 || (kernel_size & 1) == 0 || max_disp < 0 || stride1 < 1 || stride2 < 1 || pad_size < 0)
        PWC_FAIL(PWC_EINVAL, "pwc_corr_bwd: bad parameters pad=%d k=%d d=%d s1=%d s2=%d", pad_size, kernel_size, max_disp,
                 stride1, stride2);
    const int krad = (kernel_size - 1) / 2;
    const int drad = max_disp / stride2;
    const int border = krad + max_disp;
    const int oh = (H + 2 * pad_size - 2 * border + stride1 - 1) / stride1;
    const int ow = (W + 2 * pad_size - 2 * border + stride1 - 1) / stride1;
    if (oh <= 0 || ow <= 0) PWC_FAIL(PWC_EINVAL, "pwc_corr_bwd: empty output (%d x %d)", oh, ow);
    const float scale = (flags & PWC_CORR_NORMALIZE) ? 1.0f / (float)(kernel_size * kernel_size * C) : corr_multiply;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == PWC_F32 && kernel_size == 1 && max_disp == kD && pad_size == kD && stride1 == 1 && stride2 == 1 &&
        !((reinterpret_cast<uintptr_t>(in1) | reinterpret_cast<uintptr_t>(in2)) & 15u)) {
        const int tiles_x = (W + kTW - 1) / kTW, tiles_y = (H + kTH - 1) / kTH;
        const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
        if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_corr_bwd: grid too large");
        hipLaunchKernelGGL(corr81_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, st,
                           (const float *)in1, (const float *)in2, (const float *)grad_out, (float *)grad_in1,
                           (float *)grad_in2, C, H, W, tiles_x, tiles_y, scale);
        return pwc::check_launch("corr81_bwd_kernel");
    }
    const int64_t total = (int64_t)B * C * H * W;
    const int64_t nblk = (total + 255) / 256;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_corr_bwd: grid too large");
    if (dtype == PWC_F32) {
        hipLaunchKernelGGL((corr_bwd_generic_kernel<float>), dim3((unsigned)nblk), dim3(256), 0, st,
                           (const float *)in1, (const float *)in2, (const float *)grad_out, (float *)grad_in1,
                           (float *)grad_in2, C, H, W, oh, ow, pad_size, krad, max_disp, stride1, stride2, drad, total, scale);
    } else if (dtype == PWC_F16) {
        hipLaunchKernelGGL((corr_bwd_generic_kernel<__half>), dim3((unsigned)nblk), dim3(256), 0, st,
                           (const __half *)in1, (const __half *)in2, (const __half *)grad_out, (__half *)grad_in1,
                           (__half *)grad_in2, C, H, W, oh, ow, pad_size, krad, max_disp, stride1, stride2, drad, total, scale);
    } else {
        PWC_FAIL(PWC_EUNSUPPORTED, "pwc_corr_bwd: dtype %d", dtype);
    }
    return pwc::check_launch("corr_bwd_generic_kernel");
}

// timing-experiment mask this translation unit was built with (0 in the product; pwc_experiment_mask)
namespace pwc { int exp_mask_corr() { return PWC_CORR_EXP; } }
