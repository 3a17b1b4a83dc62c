// Cost-volume correlation for gfx950, round-4 form: ONE workgroup per CU, a deep LDS-DMA ring, the output drained through LDS.
//
// Same operator and tile roles as corr81_dma_kernel in pwc_corr.hip (reference semantics: correlation_cuda_kernel.cu:73-147 /
// correlation.py:12-40; PWC configuration pad 4, kernel 1, max displacement 4, strides 1; fused form PWCNet.py:141-177,212-214):
// nine fma waves own one displacement row dy each of an 8 x 32 pixel tile (lane = 4 pixels x 9 dx, v_pk_fma_f32, operands by
// conflict-free ds_read_b128 out of [chunk of 4 channels][row][40] LDS images filled by 16-byte LDS-DMA).  What changed is
// everything AROUND the arithmetic.  Facts it is built on (profiles/r04_corr_notes.md, all measured on MI355X this round):
//   * cold operands (three sets in rotation, > 256 MiB): the round-2 kernel reads at 3.4 TB/s with its stores off -- two
//     3-slot rings per CU keep ~32 KB of HBM bytes in flight, a 2.4 us round trip; its 81-instruction store burst per tile
//     then adds its full length on top (the CU's vector-memory pipeline is in order: fetches queue behind the burst);
//   * v_pk_fma_f32 costs 4.3 cycles of its SIMD (5.25 for a wave alone), v_fma_f32 2.6: the 20 736 x C fmas of a tile are 19 us
//     of VALU time per CU at level 2 -- the arithmetic is a third of the kernel, not hidden "for free" (round 3's ablation
//     had let the compiler delete it together with the stores);
//   * a 16-byte store costs the issuing wave ~80 cycles (5 source registers x 64 lanes to the address/data path), an LDS-DMA
//     piece ~25; a workgroup barrier of 12 waves ~300 cycles; 80 scalar instructions per step in the fma waves cost as much
//     as their fmas (a wave issues in order).
// Hence:
//   * ONE workgroup per CU, ring of R = 8 chunks: two loader waves keep six chunks (90 KB, ~50 KB of them HBM bytes) in flight.
//   * THE OUTPUT NEVER LEAVES FROM THE FMA WAVES.  A finished tile's 36 values per lane move to a second register set
//     (finalised: scale, LeakyReLU); during the NEXT tile's ring steps the fma waves drop them piecewise (81 planes spread
//     evenly over the steps) into a small double-buffered LDS stage, and a DRAINER wave reads the stage back and issues the
//     16-byte stores: ~10 store instructions per ring step, never a burst, from a wave whose stalls stop nobody's arithmetic.
//   * Every step is STATIC code: the kernel is a template of the chunk count (8: level 2, 16: level 3), the tile loop's body is
//     the unrolled sequence of its steps -- ring slot, stage share and plane numbers are immediates; the fma waves execute
//     ~10 scalar instructions per step.
//   * The fma waves read half a step ahead (operands of the next chunk's first two channels are requested between the two
//     halves of this one and stay in flight across the barrier): with three fma waves on a SIMD nothing else covers the LDS
//     latency behind a barrier.
//   * (fused) the warped second operand is sampled from an LDS window: see the WARP section below.
// Numerics are those of the round-2 kernels to the bit: same per-pixel fma chain over ascending channels, same blend.
#include <stdlib.h>

#include "pwc_common.h"
#include "pwc_corr_pipe.h"
#include "pwc_warp_taps.h"

#ifndef PWC_PIPE_EXP
#define PWC_PIPE_EXP 0      // timing experiments (results invalid): 1 = no arithmetic (the compiler then drops the operand reads too),
                            // 2 = stores go out with out-of-range offsets, 4 = no LDS-DMA after the prologue, 8 = the drainer only keeps the barriers,
                            // 16 = every batch item is written over item 0 (cache-resident output), 32 = every batch item reads item 0,
                            // 64 = default-policy stores instead of nt, 128 = sc0 sc1 (write-through) stores, 256 = only half of the in2 rows are fetched, 512 = in1 is not fetched,
                            // 1024 / 2048 = in1 is read / the output is written as 1 KB contiguous per channel (plane) and tile
#endif

namespace {

using pwc::leaky;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kD = 4;
constexpr int kND = 2 * kD + 1;                           // 9 displacement rows = 9 fma waves
constexpr int kPX = 4;                                    // pixels per lane
constexpr int kTH = 8, kTG = 8, kTW = kTG * kPX;          // tile: 8 rows x 32 columns
constexpr int kPitch = 40;                                // floats per LDS row: rows 4 apart land on the other half of the banks
constexpr int kS2Rows = kTH + 2 * kD;                     // 16
constexpr int kCK = 4;                                    // channels per ring step
constexpr int kS2F = kCK * kS2Rows * kPitch;              // 2560 floats: in2 halo tile [c][16][40]
constexpr int kS1F = kCK * kTH * kPitch;                  // 1280 floats: in1 tile [c][8][40] (columns 32..39 unused)
constexpr int kS1I = kS1F / 256, kS2I = kS2F / 256;       // 5 + 10 LDS-DMA wave-instructions (64 x 16 B) per chunk
constexpr int kDmaI = kS1I + kS2I;                        // instruction k < 5: in1, else in2
constexpr int kR = 8;                                     // ring slots (the chunk count of a tile is a multiple: slots are static)
constexpr int kLoadSplit = 7;                             // loader wave 0 issues instructions 0..6, loader wave 1 instructions 7..14
constexpr int kMaxPieces = 11;                            // ceil(81 / 8) planes leave per ring step at most
constexpr int kStageF = kMaxPieces * 256;                 // one stage buffer: 11 pieces of 64 lanes x 16 B
constexpr unsigned kOOBv = 0x80000000u;
static_assert((kR - 1) * (kDmaI - kLoadSplit) <= 63 && (kR - 1) * kLoadSplit <= 63, "vmcnt is a 6-bit counter");

// wave roles (plain kernel): 0..8 fma, 9..10 loaders, 11 drainer -> 768 threads, three waves per SIMD, 168 registers
constexpr int kWaveLoad0 = kND, kWaveDrain = kND + 2;
constexpr int kThreadsPlain = 64 * (kND + 3);

constexpr int kLdsPlain = (2 * kStageF + kR * (kS1F + kS2F)) * 4;
static_assert(kLdsPlain <= 160 * 1024, "LDS");

// ds_read_b128 services a wave in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32: group k gets
// tile rows {k, k+4}, which with the 40-float pitch touch all 64 banks once
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

struct TileXY { int b, x0, y0; };

// tiles are dealt so that each XCD (workgroups i, i+8, ... share one) owns a contiguous run: the +-4 halo re-read by
// neighbouring tiles then hits in that XCD's L2 (speed only, any mapping is correct)
__device__ __forceinline__ TileXY tile_of(int t, int nblk, int tiles_x, int tiles_y) {
    if ((nblk & 7) == 0) t = (t & 7) * (nblk >> 3) + (t >> 3);
    TileXY r;
    r.x0 = (t % tiles_x) * kTW;
    t /= tiles_x;
    r.y0 = (t % tiles_y) * kTH;
    r.b = t / tiles_y;
    return r;
}

template <int I>
__device__ __forceinline__ void wait_chunks_in_flight(int n) {       // s_waitcnt vmcnt(n * I), n wave-uniform in [0, kR-2]
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1 * I) : "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * I) : "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * I <= 63 ? 3 * I : 63) : "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * I <= 63 ? 4 * I : 63) : "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(5 * I <= 63 ? 5 * I : 63) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6 * I <= 63 ? 6 * I : 63) : "memory"); break;
    }
}

struct PipeArgs {
    const float *in1, *in2;
    float *out;
    int C, H, W, tiles_x, tiles_y, nblk;
    int64_t bs1, bs2, bso;
    float scale, slope;
    int do_leaky;
};

// The 81 output planes of a tile leave during the NCH ring steps of the next one: step K takes the pieces q in
// [81 K / NCH, 81 (K+1) / NCH), piece q = plane (q % 9) * 9 + q / 9 -- displacement column dx = q / 9 of fma wave q % 9, so that
// every wave stages one piece (sometimes two) per step.  All of it is known at compile time.
template <int NCH, int K> struct Share {
    static constexpr int q0 = 81 * K / NCH, q1 = 81 * (K + 1) / NCH;
    static_assert(q1 - q0 <= kMaxPieces, "stage buffer");
};

// ---- fma waves ------------------------------------------------------------------------------------------------------
struct HalfOps { float4 a[2], w0[2], w1[2], w2[2]; };     // the operands of two channels

__device__ __forceinline__ void load_half(HalfOps &h, const float *s1, const float *s2, int c0) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        h.a[c] = *reinterpret_cast<const float4 *>(s1 + (c0 + c) * kTH * kPitch);
        h.w0[c] = *reinterpret_cast<const float4 *>(s2 + (c0 + c) * kS2Rows * kPitch);
        h.w1[c] = *reinterpret_cast<const float4 *>(s2 + (c0 + c) * kS2Rows * kPitch + 4);
        h.w2[c] = *reinterpret_cast<const float4 *>(s2 + (c0 + c) * kS2Rows * kPitch + 8);
    }
}

// pixel p even: pairs dx = (0,1)(2,3)(4,5)(6,7) + single dx 8;  p odd: pairs (1,2)(3,4)(5,6)(7,8) + single dx 0 -- so that every
// in2 operand pair starts at an even window index (an aligned register pair straight out of ds_read_b128)
__device__ __forceinline__ void fma_half(const HalfOps &h, f32x2 (&acc2)[kPX][4], float (&acc1)[kPX]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float av[kPX] = {h.a[c].x, h.a[c].y, h.a[c].z, h.a[c].w};
        const f32x2 wp[6] = {{h.w0[c].x, h.w0[c].y}, {h.w0[c].z, h.w0[c].w}, {h.w1[c].x, h.w1[c].y},
                             {h.w1[c].z, h.w1[c].w}, {h.w2[c].x, h.w2[c].y}, {h.w2[c].z, h.w2[c].w}};
        const float ws[12] = {h.w0[c].x, h.w0[c].y, h.w0[c].z, h.w0[c].w, h.w1[c].x, h.w1[c].y, h.w1[c].z, h.w1[c].w,
                              h.w2[c].x, h.w2[c].y, h.w2[c].z, h.w2[c].w};
#pragma unroll
        for (int p = 0; p < kPX; ++p) {
            const f32x2 ap = {av[p], av[p]};
            const int m0 = (p + 1) / 2;                 // first aligned window pair: index p (p even) / p+1 (p odd)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc2[p][m] = __builtin_elementwise_fma(ap, wp[m0 + m], acc2[p][m]);
            acc1[p] = fmaf(av[p], (p & 1) ? ws[p] : ws[p + 8], acc1[p]);
        }
    }
}

struct FmaState {
    f32x2 acc2[kPX][4];
    float acc1[kPX];
    float done[kND][kPX];        // the finished tile, finalised, waiting for its turn in the stage
    HalfOps ha, hb;
};

// this wave's share of step K's pieces -> stage buffer (K & 1)
template <int NCH, int K>
__device__ __forceinline__ void stage_share(const FmaState &st, float *stage, int wave, int lane) {
    constexpr int q0 = Share<NCH, K>::q0, q1 = Share<NCH, K>::q1;
    float *sb = stage + (K & 1) * kStageF + lane * 4;
#pragma unroll
    for (int dx = q0 / 9; dx <= (q1 - 1) / 9; ++dx) {
        const int lo = q0 - 9 * dx > 0 ? q0 - 9 * dx : 0, hi = q1 - 9 * dx < 9 ? q1 - 9 * dx : 9;      // waves whose piece 9 dx + w is in [q0, q1)
        if (wave >= lo && wave < hi)
            *reinterpret_cast<f32x4v *>(sb + (9 * dx + wave - q0) * 256) = (f32x4v){st.done[dx][0], st.done[dx][1], st.done[dx][2], st.done[dx][3]};
    }
}

// One ring step of the fma waves.  Half-step software pipeline: the operands of channels 0-1 of this chunk were requested
// during the previous step (the loaders promise chunk s+1 at barrier B_s), so the fmas start right behind the barrier while
// the reads of channels 2-3 are in flight; the reads of the NEXT chunk's first half go out between the two halves and stay in
// flight across the barrier.  Straight-line code from the first read on: the compiler's counted lgkmcnt waits are exact.
template <int NCH, int K>
__device__ __forceinline__ void fma_step(FmaState &st, const PipeArgs &a, float *stage, const float *s1l, const float *s2l,
                                         int wave, int lane, bool have_prev) {
    if (have_prev) stage_share<NCH, K>(st, stage, wave, lane);
    asm volatile("" ::: "memory");              // the stage writes stay AHEAD of the step's operand reads (see the barrier note)
    constexpr int slot = K % kR, nslot = (K + 1) % kR;
    load_half(st.hb, s1l + slot * kS1F, s2l + slot * kS2F, 2);
    __builtin_amdgcn_sched_barrier(0);          // (hipcc otherwise sinks these reads below the first half's fmas)
    if (!(PWC_PIPE_EXP & 1)) fma_half(st.ha, st.acc2, st.acc1);
    __builtin_amdgcn_sched_barrier(0);
    load_half(st.ha, s1l + nslot * kS1F, s2l + nslot * kS2F, 0);      // (behind the last chunk: a slot nobody needs)
    __builtin_amdgcn_sched_barrier(0);
    if (!(PWC_PIPE_EXP & 1)) fma_half(st.hb, st.acc2, st.acc1);
    // The fmas are register-only code: nothing ties them to the barrier, and with the steps unrolled hipcc moved ALL of a tile's
    // fmas behind its last barrier (operands spilled to scratch meanwhile).  An empty asm that reads and writes the accumulators
    // pins this step's fmas above this point and the next step's below it.
    asm volatile("" : "+v"(st.acc2[0][0]), "+v"(st.acc2[0][1]), "+v"(st.acc2[0][2]), "+v"(st.acc2[0][3]),
                      "+v"(st.acc2[1][0]), "+v"(st.acc2[1][1]), "+v"(st.acc2[1][2]), "+v"(st.acc2[1][3]),
                      "+v"(st.acc2[2][0]), "+v"(st.acc2[2][1]), "+v"(st.acc2[2][2]), "+v"(st.acc2[2][3]),
                      "+v"(st.acc2[3][0]), "+v"(st.acc2[3][1]), "+v"(st.acc2[3][2]), "+v"(st.acc2[3][3]),
                      "+v"(st.acc1[0]), "+v"(st.acc1[1]), "+v"(st.acc1[2]), "+v"(st.acc1[3]));
    if constexpr (K == NCH - 1) {
        // ---- tile finished: scale / LeakyReLU into the second register set; it leaves during the next tile's steps
#pragma unroll
        for (int dx = 0; dx < kND; ++dx)
#pragma unroll
            for (int p = 0; p < kPX; ++p) {
                float q;
                if (p & 1) q = (dx == 0) ? st.acc1[p] : st.acc2[p][(dx - 1) / 2][(dx - 1) & 1];
                else       q = (dx == 8) ? st.acc1[p] : st.acc2[p][dx / 2][dx & 1];
                q *= a.scale;
                st.done[dx][p] = a.do_leaky ? leaky(q, a.slope) : q;
            }
#pragma unroll
        for (int p = 0; p < kPX; ++p) {
            st.acc1[p] = 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m) st.acc2[p][m] = (f32x2){0.f, 0.f};
        }
    }
    // Nothing to wait for before the barrier: the stage writes are OLDER than reads whose data the fmas above have consumed, and
    // the LDS operations of a wave complete in order; the look-ahead reads stay in flight across the barrier.
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int NCH, int K>
__device__ __forceinline__ void fma_steps(FmaState &st, const PipeArgs &a, float *stage, const float *s1l, const float *s2l,
                                          int wave, int lane, bool have_prev) {
    if constexpr (K < NCH) {
        fma_step<NCH, K>(st, a, stage, s1l, s2l, wave, lane, have_prev);
        fma_steps<NCH, K + 1>(st, a, stage, s1l, s2l, wave, lane, have_prev);
    }
}

template <int NCH, int K>
__device__ __forceinline__ void tail_steps(const FmaState &st, float *stage, int wave, int lane) {       // drain-only steps of the last tile
    if constexpr (K < NCH) {
        stage_share<NCH, K>(st, stage, wave, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the stage writes are done before the barrier
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        tail_steps<NCH, K + 1>(st, stage, wave, lane);
    }
}

template <int NCH>
__device__ __forceinline__ void fma_wave(const PipeArgs &a, float *stage, const float *s1ring, const float *s2ring,
                                         int wave, int lane, int my_tiles) {
    int r, g;
    lane_to_rg(lane, r, g);
    FmaState st;
#pragma unroll
    for (int p = 0; p < kPX; ++p) {
        st.acc1[p] = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) st.acc2[p][m] = (f32x2){0.f, 0.f};
    }
#pragma unroll
    for (int dx = 0; dx < kND; ++dx)
#pragma unroll
        for (int p = 0; p < kPX; ++p) st.done[dx][p] = 0.f;
    const float *s1l = s1ring + r * kPitch + 4 * g, *s2l = s2ring + (r + wave) * kPitch + 4 * g;
    __builtin_amdgcn_s_barrier();              // B_0: chunks 0 and 1 are readable
    asm volatile("" ::: "memory");
    load_half(st.ha, s1l, s2l, 0);
#pragma unroll 1
    for (int t = 0; t < my_tiles; ++t) fma_steps<NCH, 0>(st, a, stage, s1l, s2l, wave, lane, t > 0);
    tail_steps<NCH, 0>(st, stage, wave, lane);
}

// ---- loader wave: instructions K0..K1 of the chunk's 15 LDS-DMA instructions --------------------------------------------
template <int K0, int K1>
struct Loader {
    static constexpr int I = K1 - K0;
    unsigned off[I];
    const float *ip1, *ip2;

    __device__ __forceinline__ void new_tile(const PipeArgs &a, int tile, int lane, int plane) {
        const TileXY t = tile_of(tile, a.nblk, a.tiles_x, a.tiles_y);
#pragma unroll
        for (int k = K0; k < K1; ++k) {
            int c, row, q, iy, ix;
            bool ok;
            if (k < kS1I) {
                const int p = k * 64 + lane;
                c = p / (kTH * 10); row = (p / 10) % kTH; q = p % 10;
                iy = t.y0 + row; ix = t.x0 + 4 * q;
                ok = (q < kTG) && !(PWC_PIPE_EXP & 512);
            } else {
                const int p = (k - kS1I) * 64 + lane;
                c = p / (kS2Rows * 10); row = (p / 10) % kS2Rows; q = p % 10;
                iy = t.y0 + row - kD; ix = t.x0 + 4 * q - kD;
                ok = !(PWC_PIPE_EXP & 256) || row >= 8;
            }
            ok = ok && (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W);     // W % 4 == 0: a piece is all-in or all-out
            off[k - K0] = ok ? (unsigned)(c * plane + iy * a.W + ix) * 4u : kOOBv;
            if ((PWC_PIPE_EXP & 1024) && k < kS1I && ok) {        // experiment: in1 read as contiguous 1 KB per channel and tile
                const int tl = (t.y0 / kTH) * a.tiles_x + t.x0 / kTW;
                off[k - K0] = (unsigned)(c * plane + tl * 256 + row * 32 + 4 * q) * 4u;
            }
        }
        ip1 = a.in1 + (int64_t)((PWC_PIPE_EXP & 32) ? 0 : t.b) * a.bs1;
        ip2 = a.in2 + (int64_t)((PWC_PIPE_EXP & 32) ? 0 : t.b) * a.bs2;
    }

    __device__ __forceinline__ void issue(const PipeArgs &a, int chunk, int slot, float *s1ring, float *s2ring, int plane) {
        const int c0 = chunk * kCK;
        const int nbytes = min(kCK, a.C - c0) * plane * 4;        // channels past C fail the range check: zeros
        const pwc::v4i32 r1 = pwc::make_rsrc(ip1 + (int64_t)c0 * plane, nbytes);
        const pwc::v4i32 r2 = pwc::make_rsrc(ip2 + (int64_t)c0 * plane, nbytes);
        const unsigned b1 = (unsigned)__builtin_amdgcn_readfirstlane((int)pwc::lds_addr(s1ring + slot * kS1F));
        const unsigned b2 = (unsigned)__builtin_amdgcn_readfirstlane((int)pwc::lds_addr(s2ring + slot * kS2F));
#pragma unroll
        for (int k = K0; k < K1; ++k) {
            if (k < kS1I) pwc::dma_b128(r1, b1 + k * 1024, off[k - K0]);
            else          pwc::dma_b128(r2, b2 + (k - kS1I) * 1024, off[k - K0]);
        }
    }
};

// step K of tile t (global step s = t NCH + K), behind barrier B_s: chunk s+R-1 goes into the slot chunk s-1 has left, then
// the wave waits until chunk s+2 has landed (B_{s+1} promises chunks s+1 and s+2: the fma waves read half a step ahead)
template <int NCH, int K, int K0, int K1>
__device__ __forceinline__ void loader_steps(Loader<K0, K1> &ld, const PipeArgs &a, float *s1ring, float *s2ring, int lane, int plane,
                                             int t, int nsteps, int stride) {
    if constexpr (K < NCH) {
        constexpr int I = K1 - K0;
        __builtin_amdgcn_s_barrier();              // B_s
        const int s = t * NCH + K;
        constexpr int kc = (K + kR - 1) % NCH;     // chunk index inside its tile of the chunk issued now
        if (s + kR - 1 < nsteps && (!(PWC_PIPE_EXP & 4))) {
            if constexpr (kc == 0) ld.new_tile(a, (int)blockIdx.x + (t + (K + kR - 1) / NCH) * stride, lane, plane);
            ld.issue(a, kc, (K + kR - 1) % kR, s1ring, s2ring, plane);
        }
#ifdef PWC_PIPE_AHEAD        // experiment: wait until only this many chunks are outstanding (the promise needs <= kR - 3)
        if (s + kR - 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PWC_PIPE_AHEAD * I) : "memory");
#else
        if (s + kR - 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((kR - 3) * I) : "memory");     // steady state
#endif
        else if (s + 1 < nsteps) wait_chunks_in_flight<I>(max(nsteps - 1 - (s + 2), 0));
        loader_steps<NCH, K + 1, K0, K1>(ld, a, s1ring, s2ring, lane, plane, t, nsteps, stride);
    }
}

template <int NCH, int K0, int K1>
__device__ __forceinline__ void loader_wave(const PipeArgs &a, float *s1ring, float *s2ring, int lane, int my_tiles) {
    constexpr int I = K1 - K0;
    __builtin_amdgcn_s_setprio(3);                  // the ring never waits for issue slots behind the fma waves
    const int plane = a.H * a.W;
    const int stride = gridDim.x;
    const int nsteps = my_tiles * NCH;
    Loader<K0, K1> ld;
    ld.new_tile(a, blockIdx.x, lane, plane);
    static_assert(kR - 1 <= 8, "prologue inside the first tile");
#pragma unroll
    for (int k = 0; k < kR - 1; ++k)                // chunks 0 .. R-2 of the first tile (NCH >= 8 > R-2)
        if (k < nsteps) ld.issue(a, k, k, s1ring, s2ring, plane);
    wait_chunks_in_flight<I>(max(min(kR - 1, nsteps) - 2, 0));        // B_0 promises chunks 0 and 1
#pragma unroll 1
    for (int t = 0; t <= my_tiles; ++t) loader_steps<NCH, 0, K0, K1>(ld, a, s1ring, s2ring, lane, plane, t, nsteps, stride);
    __builtin_amdgcn_s_barrier();                   // the drainer's last share
}

// ---- drainer wave: stage -> global, 81 / NCH store instructions per ring step -----------------------------------------
template <int NCH, int K>
__device__ __forceinline__ void drain_steps(const float *stage, int lane, bool have_prev, __amdgpu_buffer_rsrc_t rs, unsigned voff, int plane4) {
    if constexpr (K < NCH) {
        __builtin_amdgcn_s_barrier();              // the pieces staged during step K are in stage[K & 1]
        asm volatile("" ::: "memory");             // (s_barrier is IntrNoMem to the compiler: keep the stage reads below it)
        if (have_prev && !(PWC_PIPE_EXP & 8)) {
            constexpr int q0 = Share<NCH, K>::q0, q1 = Share<NCH, K>::q1;
            const float *sb = stage + (K & 1) * kStageF + lane * 4;
            f32x4v v[q1 - q0];
#pragma unroll
            for (int j = 0; j < q1 - q0; ++j) v[j] = *reinterpret_cast<const f32x4v *>(sb + j * 256);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // stage reads done before the barrier that frees the buffer
#pragma unroll
            for (int j = 0; j < q1 - q0; ++j) {
                const int q = q0 + j, ch = (q % 9) * 9 + q / 9;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pwc::v4i32, v[j]), rs, (PWC_PIPE_EXP & 2) ? kOOBv : voff,
                                                       ch * plane4, (PWC_PIPE_EXP & 64) ? 0 : (PWC_PIPE_EXP & 128) ? 17 : 2 /* nt */);
            }
        }
        drain_steps<NCH, K + 1>(stage, lane, have_prev, rs, voff, plane4);
    }
}

template <int NCH>
__device__ __forceinline__ void drainer_wave(const PipeArgs &a, const float *stage, int lane, int my_tiles) {
    const int plane4 = a.H * a.W * 4;
    const int stride = gridDim.x;
    int r, g;
    lane_to_rg(lane, r, g);
    unsigned voff = kOOBv;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)nullptr, 0, 0, 0x00020000);
    __builtin_amdgcn_s_barrier();                  // B_0
#pragma unroll 1
    for (int t = 0; t <= my_tiles; ++t) {          // during tile slot t (my_tiles = the drain-only tail) the planes of tile t-1 leave
        if (t >= 1) {
            const TileXY tl = tile_of((int)blockIdx.x + (t - 1) * stride, a.nblk, a.tiles_x, a.tiles_y);
            const int y = tl.y0 + r, x = tl.x0 + 4 * g;
            voff = (y < a.H && x < a.W) ? (unsigned)(y * a.W + x) * 4u : kOOBv;
            if (PWC_PIPE_EXP & 2048) voff = (unsigned)(((tl.y0 / kTH) * a.tiles_x + tl.x0 / kTW) * 256 + r * 32 + 4 * g) * 4u;   // experiment: 1 KB contiguous per plane and tile
            // (the SGPR offset takes part in the range check: num_records spans all 81 planes)
            rs = __builtin_amdgcn_make_buffer_rsrc(pwc::uniform_ptr(a.out + (int64_t)((PWC_PIPE_EXP & 16) ? 0 : tl.b) * a.bso), 0,
                                                   __builtin_amdgcn_readfirstlane(81 * plane4), 0x00020000);
        }
        drain_steps<NCH, 0>(stage, lane, t >= 1, rs, voff, plane4);
    }
}

// =====================================================================================================================
template <int NCH>
__global__ void __launch_bounds__(kThreadsPlain, 3)
corr81_pipe_kernel(PipeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *stage = smem;                             // [2][kStageF]
    float *s1ring = smem + 2 * kStageF;              // [kR][kS1F]
    float *s2ring = s1ring + kR * kS1F;              // [kR][kS2F]
    static_assert(NCH % kR == 0, "static ring slots");

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int stride = gridDim.x;
    const int my_tiles = (a.nblk - (int)blockIdx.x + stride - 1) / stride;
    // barriers of every wave: B_0, one per ring step of its tiles, NCH drain-only steps = my_tiles NCH + NCH + 1

    if (wave == kWaveLoad0)          loader_wave<NCH, 0, kLoadSplit>(a, s1ring, s2ring, lane, my_tiles);
    else if (wave == kWaveLoad0 + 1) loader_wave<NCH, kLoadSplit, kDmaI>(a, s1ring, s2ring, lane, my_tiles);
    else if (wave == kWaveDrain)     drainer_wave<NCH>(a, stage, lane, my_tiles);
    else                             fma_wave<NCH>(a, stage, s1ring, s2ring, wave, lane, my_tiles);
}

pwc::LdsAttrOnce g_lds_plain8, g_lds_plain16;

}  // namespace

namespace pwc {

bool corr81_pipe_enabled() { return option(OPT_CORR_PIPE) != 0; }

bool corr81_pipe_fits(int B, int C, int H, int W) {
    const int64_t nblk = (int64_t)B * ((W + kTW - 1) / kTW) * ((H + kTH - 1) / kTH);
    const int nch = (C + kCK - 1) / kCK;
    return (nch == 8 || nch == 16) && nblk >= option(OPT_CORR_PIPE_MIN_TILES) && nblk <= 0x7fffffffLL &&
           (int64_t)H * W * 81 * 4 < 0x7fffffffLL;
}

int launch_corr81_pipe(const float *in1, const float *in2, float *out, int B, int C, int H, int W,
                       int64_t bs1, int64_t bs2, int64_t bso, float scale, float slope, int do_leaky, hipStream_t st) {
    const int tiles_x = (W + kTW - 1) / kTW, tiles_y = (H + kTH - 1) / kTH;
    const int nblk = B * tiles_x * tiles_y;
    const int nch = (C + kCK - 1) / kCK;
    const int grid = nblk < 256 ? nblk : 256;       // one workgroup per CU (its LDS does not admit two)
    const PipeArgs a{in1, in2, out, C, H, W, tiles_x, tiles_y, nblk, bs1, bs2, bso, scale, slope, do_leaky};
    if (nch == 8) {
        int rc = ensure_lds_attr(g_lds_plain8, reinterpret_cast<const void *>(corr81_pipe_kernel<8>), kLdsPlain, "corr81_pipe_kernel<8>");
        if (rc != PWC_OK) return rc;
        hipLaunchKernelGGL(corr81_pipe_kernel<8>, dim3((unsigned)grid), dim3(kThreadsPlain), kLdsPlain, st, a);
    } else {
        int rc = ensure_lds_attr(g_lds_plain16, reinterpret_cast<const void *>(corr81_pipe_kernel<16>), kLdsPlain, "corr81_pipe_kernel<16>");
        if (rc != PWC_OK) return rc;
        hipLaunchKernelGGL(corr81_pipe_kernel<16>, dim3((unsigned)grid), dim3(kThreadsPlain), kLdsPlain, st, a);
    }
    return check_launch("corr81_pipe_kernel");
}

}  // namespace pwc
