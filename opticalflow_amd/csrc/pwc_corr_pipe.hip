// Cost-volume correlation for gfx950, round-4 form: ONE workgroup per CU, a deep LDS-DMA ring, the output leaving plane by plane.
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
//   * THE OUTPUT NEVER LEAVES AS A BURST.  A finished tile's 36 values per lane move to a second register set (finalised: scale,
//     LeakyReLU); during the NEXT tile's ring steps each fma wave stores them plane by plane (81 planes spread evenly over the
//     steps: about one 16-byte store instruction per wave and step, behind its fmas).  The first form dropped them into a small
//     double-buffered LDS stage that a DRAINER wave read back and stored (VERDICT r3's proposal): 68.7 us against 67.1 us for the
//     direct stores -- one wave issuing ten stores per step pays 80 cycles for each, in order -- so the stage went.
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
                            // 1024 = in1 is read as 1 KB contiguous per channel and tile, 2048 = (fused) every pixel takes the gather path,
                            // 4096 = (fused) the producers only keep the barriers, 8192 = (fused) no pixel ever counts as outside the window,
                            // 16384 = (fused) the outside pixels' gathers are neither issued nor blended, 32768 = issued, not blended
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
constexpr unsigned kOOBv = 0x80000000u;
static_assert((kR - 1) * (kDmaI - kLoadSplit) <= 63 && (kR - 1) * kLoadSplit <= 63, "vmcnt is a 6-bit counter");

// wave roles (plain kernel): 0..8 fma, 9..10 loaders -> 704 threads, at most three waves per SIMD, 168 registers
constexpr int kWaveLoad0 = kND;
#ifdef PWC_PIPE_LOADERS4        // experiment: four loader waves (13 waves: 128 registers -- the fma waves spill; timing of the memory side only)
constexpr int kThreadsPlain = 64 * (kND + 4);
#else
constexpr int kThreadsPlain = 64 * (kND + 2);
#endif

constexpr int kLdsPlain = kR * (kS1F + kS2F) * 4;
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
// the same from the divisors' reciprocals (q = mulhi(t, ceil(2^32 / d)) is exact while t * d < 2^32: fill_magic checks): every wave
// role calls this once per tile and a runtime integer division is ~35 instructions on gfx950 (measured: no difference here, the
// divisions were not on the critical path -- kept because it is simply less code to execute)
struct PipeArgs;
__device__ __forceinline__ TileXY tile_of(int t, const PipeArgs &a);

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
    int seg_len, segs_per_strip, nseg;      // rolling kernel: vertical runs of tiles (see TileSeq)
    const float *flo;                       // fused kernel: [B,2,H,W] flow (u, v), its batch stride, scale, mask threshold, sampling mode
    int64_t bsf;
    float flow_scale, thr;
    int align_corners;
    unsigned magic_tx, magic_ty;            // ceil(2^32 / tiles_x), ceil(2^32 / tiles_y): tile_of() without integer division (set by fill_magic)
};

__device__ __forceinline__ TileXY tile_of(int t, const PipeArgs &a) {
    if ((a.nblk & 7) == 0) t = (t & 7) * (a.nblk >> 3) + (t >> 3);
    const unsigned q1 = a.tiles_x == 1 ? (unsigned)t : __umulhi((unsigned)t, a.magic_tx);       // (ceil(2^32 / 1) does not fit a word)
    const unsigned q2 = a.tiles_y == 1 ? q1 : __umulhi(q1, a.magic_ty);
    TileXY r;
    r.x0 = (int)((unsigned)t - q1 * (unsigned)a.tiles_x) * kTW;
    r.y0 = (int)(q1 - q2 * (unsigned)a.tiles_y) * kTH;
    r.b = (int)q2;
    return r;
}

// The 81 output planes of a tile leave during the NCH ring steps of the next one: step K takes the pieces q in
// [81 K / NCH, 81 (K+1) / NCH), piece q = plane (q % 9) * 9 + q / 9 -- displacement column dx = q / 9 of fma wave q % 9, so that
// every wave stores one piece (sometimes two) per step.  All of it is known at compile time.
template <int NCH, int K> struct Share {
    static constexpr int q0 = 81 * K / NCH, q1 = 81 * (K + 1) / NCH;
};

// ---- fma waves ------------------------------------------------------------------------------------------------------
template <int LA> struct Ops { float4 a[LA], w0[LA], w1[LA], w2[LA]; };     // the operands of LA channels

template <int LA, int S2C>                      // S2C: floats between two channels of the in2 image (640: [c][16][40], 320: [c][8][40] halves)
__device__ __forceinline__ void load_ops(Ops<LA> &h, const float *s1, const float *s2, int c0) {
#pragma unroll
    for (int c = 0; c < LA; ++c) {
        h.a[c] = *reinterpret_cast<const float4 *>(s1 + (c0 + c) * kTH * kPitch);
        h.w0[c] = *reinterpret_cast<const float4 *>(s2 + (c0 + c) * S2C);
        h.w1[c] = *reinterpret_cast<const float4 *>(s2 + (c0 + c) * S2C + 4);
        h.w2[c] = *reinterpret_cast<const float4 *>(s2 + (c0 + c) * S2C + 8);
    }
}

// pixel p even: pairs dx = (0,1)(2,3)(4,5)(6,7) + single dx 8;  p odd: pairs (1,2)(3,4)(5,6)(7,8) + single dx 0 -- so that every
// in2 operand pair starts at an even window index (an aligned register pair straight out of ds_read_b128)
template <int LA>
__device__ __forceinline__ void fma_ops(const Ops<LA> &h, f32x2 (&acc2)[kPX][4], float (&acc1)[kPX]) {
#pragma unroll
    for (int c = 0; c < LA; ++c) {
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

template <int LA>
struct FmaState {
    f32x2 acc2[kPX][4];
    float acc1[kPX];
    float done[kND][kPX];        // the finished tile, finalised, leaving plane by plane
    Ops<LA> ha, hb;
    // where the finished tile goes: descriptor of its batch item's 81 planes, this lane's pixel offset, this wave's first plane
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff;
    int sbase, plane4;
};

// this wave's share of step K's pieces: 16-byte stores straight from the second register set.  (A store costs its wave ~80 cycles
// of issue -- 5 source registers x 64 lanes -- during which the other fma waves of the SIMD run; the nine waves' address/data paths
// work in parallel, which one drainer wave's could not: staged through LDS the same stores took 880 cycles of every step.)
template <int NCH, int K, int LA>
__device__ __forceinline__ void store_share(const FmaState<LA> &st, int wave) {
    constexpr int q0 = Share<NCH, K>::q0, q1 = Share<NCH, K>::q1;
#pragma unroll
    for (int dx = q0 / 9; dx <= (q1 - 1) / 9; ++dx) {
        const int lo = q0 - 9 * dx > 0 ? q0 - 9 * dx : 0, hi = q1 - 9 * dx < 9 ? q1 - 9 * dx : 9;      // waves whose piece 9 dx + w is in [q0, q1)
        if (wave >= lo && wave < hi)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pwc::v4i32, (f32x4v){st.done[dx][0], st.done[dx][1], st.done[dx][2], st.done[dx][3]}),
                                                   st.rs, (PWC_PIPE_EXP & 2) ? kOOBv : st.voff, st.sbase + dx * st.plane4, 2 /* nt */);
    }
}

template <int LA>
__device__ __forceinline__ void pin_acc(FmaState<LA> &st) {
    // The fmas are register-only code: nothing ties them to the barrier, and with the steps unrolled hipcc moved ALL of a tile's
    // fmas behind its last barrier (operands spilled to scratch meanwhile).  An empty asm that reads and writes the accumulators
    // pins the fmas issued so far above this point and the later ones below it.
    asm volatile("" : "+v"(st.acc2[0][0]), "+v"(st.acc2[0][1]), "+v"(st.acc2[0][2]), "+v"(st.acc2[0][3]),
                      "+v"(st.acc2[1][0]), "+v"(st.acc2[1][1]), "+v"(st.acc2[1][2]), "+v"(st.acc2[1][3]),
                      "+v"(st.acc2[2][0]), "+v"(st.acc2[2][1]), "+v"(st.acc2[2][2]), "+v"(st.acc2[2][3]),
                      "+v"(st.acc2[3][0]), "+v"(st.acc2[3][1]), "+v"(st.acc2[3][2]), "+v"(st.acc2[3][3]),
                      "+v"(st.acc1[0]), "+v"(st.acc1[1]), "+v"(st.acc1[2]), "+v"(st.acc1[3]));
}

// One ring step of the fma waves, software-pipelined over groups of LA channels (LA = 2 at 168 registers, 1 at 128): the operands
// of this chunk's first group were requested during the previous step (the loaders promise chunk s+1 at barrier B_s), so the fmas
// start right behind the barrier; each group's fmas run while the next group's reads -- the last group: the NEXT chunk's first --
// are in flight, and that look-ahead stays in flight across the barrier.  Straight-line code from the first read on: the
// compiler's counted lgkmcnt waits are exact.  (With three fma waves on a SIMD the LDS phase and the fma phase of a step
// otherwise add up.)
template <int NCH, int K, int LA, int S2C, int S2R = kR>
__device__ __forceinline__ void fma_step(FmaState<LA> &st, const PipeArgs &a, const float *s1l, const float *s2l, const float *s2n, int wave, bool have_prev) {
    if (have_prev) store_share<NCH, K, LA>(st, wave);
    constexpr int slot = K % kR, nslot = (K + 1) % kR;
    constexpr int slot2 = K % S2R, nslot2 = (K + 1) % S2R;       // the in2 image may live in a shorter ring (fused kernel)
    constexpr int G = kCK / LA;                 // groups per chunk (even): group g lives in ha (g even) / hb (g odd)
#pragma unroll
    for (int g = 0; g < G; ++g) {
        Ops<LA> &cur = (g & 1) ? st.hb : st.ha, &nxt = (g & 1) ? st.ha : st.hb;
        if (g + 1 < G) load_ops<LA, S2C>(nxt, s1l + slot * kS1F, s2l + slot2 * kS2F, (g + 1) * LA);
        else           load_ops<LA, S2C>(nxt, s1l + nslot * kS1F, (K == NCH - 1 ? s2n : s2l) + nslot2 * kS2F, 0);   // (behind the last chunk: a slot nobody needs)
        __builtin_amdgcn_sched_barrier(0);      // (hipcc otherwise sinks the reads below the fmas they should overlap)
        if (!(PWC_PIPE_EXP & 1)) fma_ops<LA>(cur, st.acc2, st.acc1);
        __builtin_amdgcn_sched_barrier(0);
    }
    pin_acc(st);
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
    __builtin_amdgcn_s_barrier();               // (the look-ahead reads stay in flight across it)
    asm volatile("" ::: "memory");
}

template <int NCH, int K, int LA, int S2C, int S2R = kR>
__device__ __forceinline__ void fma_steps(FmaState<LA> &st, const PipeArgs &a, const float *s1l, const float *s2l, const float *s2n, int wave, bool have_prev) {
    if constexpr (K < NCH) {
        fma_step<NCH, K, LA, S2C, S2R>(st, a, s1l, s2l, s2n, wave, have_prev);
        fma_steps<NCH, K + 1, LA, S2C, S2R>(st, a, s1l, s2l, s2n, wave, have_prev);
    }
}

template <int NCH, int K, int LA>
__device__ __forceinline__ void tail_stores(const FmaState<LA> &st, int wave) {       // the last tile's planes
    if constexpr (K < NCH) {
        store_share<NCH, K, LA>(st, wave);
        tail_stores<NCH, K + 1, LA>(st, wave);
    }
}

// where the tile the wave has just finished goes (its planes leave during the next tile's steps)
template <int LA>
__device__ __forceinline__ void set_destination_xy(FmaState<LA> &st, const PipeArgs &a, const TileXY &tl, int r, int g);

template <int LA>
__device__ __forceinline__ void set_destination(FmaState<LA> &st, const PipeArgs &a, int tile, int r, int g) {
    set_destination_xy(st, a, tile_of(tile, a), r, g);
}

template <int LA>
__device__ __forceinline__ void set_destination_xy(FmaState<LA> &st, const PipeArgs &a, const TileXY &tl, int r, int g) {
    const int y = tl.y0 + r, x = tl.x0 + 4 * g;
    st.voff = (y < a.H && x < a.W) ? (unsigned)(y * a.W + x) * 4u : kOOBv;
    // (the SGPR offset takes part in the range check: num_records spans all 81 planes)
    st.rs = __builtin_amdgcn_make_buffer_rsrc(pwc::uniform_ptr(a.out + (int64_t)((PWC_PIPE_EXP & 16) ? 0 : tl.b) * a.bso), 0,
                                              __builtin_amdgcn_readfirstlane(81 * st.plane4), 0x00020000);
}

template <int NCH, int LA>
__device__ __forceinline__ void fma_wave(const PipeArgs &a, const float *s1ring, const float *s2ring, int wave, int lane, int my_tiles) {
    int r, g;
    lane_to_rg(lane, r, g);
    FmaState<LA> st;
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
    st.plane4 = a.H * a.W * 4;
    st.sbase = wave * kND * st.plane4;
    st.voff = kOOBv;
    st.rs = __builtin_amdgcn_make_buffer_rsrc((void *)nullptr, 0, 0, 0x00020000);
    const int stride = gridDim.x;
    const float *s1l = s1ring + r * kPitch + 4 * g, *s2l = s2ring + (r + wave) * kPitch + 4 * g;
    __builtin_amdgcn_s_barrier();              // B_0: chunks 0 and 1 are readable
    asm volatile("" ::: "memory");
    load_ops<LA, kS2Rows * kPitch>(st.ha, s1l, s2l, 0);
#pragma unroll 1
    for (int t = 0; t < my_tiles; ++t) {
        if (t > 0) set_destination(st, a, (int)blockIdx.x + (t - 1) * stride, r, g);
        fma_steps<NCH, 0, LA, kS2Rows * kPitch>(st, a, s1l, s2l, s2l, wave, t > 0);
    }
    set_destination(st, a, (int)blockIdx.x + (my_tiles - 1) * stride, r, g);
    tail_stores<NCH, 0, LA>(st, wave);
}

// ---- loader wave: instructions K0..K1 of the chunk's 15 LDS-DMA instructions --------------------------------------------
template <int K0, int K1>
struct Loader {
    static constexpr int I = K1 - K0;
    unsigned off[I];
    const float *ip1, *ip2;

    __device__ __forceinline__ void new_tile(const PipeArgs &a, int tile, int lane, int plane) {
        const TileXY t = tile_of(tile, a);
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
        const unsigned b2 = (K1 > kS1I) ? (unsigned)__builtin_amdgcn_readfirstlane((int)pwc::lds_addr(s2ring + slot * kS2F)) : 0u;
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
    for (int t = 0; t < my_tiles; ++t) loader_steps<NCH, 0, K0, K1>(ld, a, s1ring, s2ring, lane, plane, t, nsteps, stride);
    __builtin_amdgcn_s_barrier();                   // the barrier that ends the last step
}

// =====================================================================================================================
// ROLLING form (C <= 32: the in2 halo tiles of all eight chunks fit in LDS together).
// What bounds the ring form is not HBM but the CU's vector-memory pipeline: it moves ~12 bytes per clock, LDS-DMA fetches and
// stores one after the other, and L2 hits cost the same as misses (profiles/r04_corr_notes.md: 120 KB of LDS-DMA + 81 KB of stores
// per tile = 7.8 us, x 7 tiles).  80 of those 120 KB are the in2 halo tile, and half of its rows were in LDS a moment ago: the
// tile above needed them.  So a workgroup walks DOWN a column of tiles and keeps the in2 rows it has:
//   * in2 lives as [chunk k][half][4 channels][8 rows][40]; halo row i of the run's tile number ti is in half ((i >> 3) + ti) & 1:
//     the lower half of a tile is the upper half of the next, and only 8 new rows per channel are fetched (5 LDS-DMA instructions
//     per chunk instead of 10; the first tile of a run fetches both halves);
//   * chunk k of tile n+1 (in1 and in2) is fetched into exactly what chunk k of tile n has vacated, right after the step that
//     multiplied it: every fetch is issued NCH - 1 = 7 steps before it is needed -- the ring IS one whole tile.
struct TileSeq {            // this workgroup's tiles: runs ("segments") of seg_len tiles down one column of one image
    int seg, ti, cnt;       // current segment, tile number inside it, tiles in it
    TileXY xy;
    __device__ __forceinline__ void load_segment(const PipeArgs &a) {
        int sg = seg;
        if ((a.nseg & 7) == 0) sg = (sg & 7) * (a.nseg >> 3) + (sg >> 3);     // segments running together on one XCD: neighbouring columns
        const int tx = sg % a.tiles_x;
        sg /= a.tiles_x;
        const int part = sg % a.segs_per_strip;
        xy.b = sg / a.segs_per_strip;
        xy.x0 = tx * kTW;
        xy.y0 = part * a.seg_len * kTH;
        cnt = min(a.seg_len, a.tiles_y - part * a.seg_len);
        ti = 0;
    }
    __device__ __forceinline__ void start(const PipeArgs &a) { seg = blockIdx.x; load_segment(a); }
    __device__ __forceinline__ bool valid(const PipeArgs &a) const { return seg < a.nseg; }
    __device__ __forceinline__ void advance(const PipeArgs &a) {
        if (++ti < cnt) { xy.y0 += kTH; return; }
        seg += gridDim.x;
        if (seg < a.nseg) load_segment(a);
    }
};

__device__ __forceinline__ int roll_my_tiles(const PipeArgs &a) {
    int n = 0;
    for (int sg = blockIdx.x; sg < a.nseg; sg += gridDim.x) {
        int s2 = sg;
        if ((a.nseg & 7) == 0) s2 = (s2 & 7) * (a.nseg >> 3) + (s2 >> 3);
        const int part = (s2 / a.tiles_x) % a.segs_per_strip;
        n += min(a.seg_len, a.tiles_y - part * a.seg_len);
    }
    return n;
}

constexpr int kHalfF = kCK * 8 * kPitch;          // 1280 floats: four channels x eight rows of the in2 halo tile
constexpr int kHalfI = kHalfF / 256;              // 5 LDS-DMA instructions
constexpr int kRollNCH = 8;
constexpr int kLdsRoll = kRollNCH * (kS1F + 2 * kHalfF) * 4;       // 120 KB
constexpr int kThreadsRoll = 64 * (kND + 2);      // nine fma waves, loader A (in1 + upper in2 half of a run's first tile), loader B (new in2 rows)

// Loader A: WHICH = 0 -> in1 (8 rows of the tile) + E = the upper half (halo rows 0-7) when the tile starts a run;
// loader B: WHICH = 1 -> the lower half (halo rows 8-15): the 8 new rows.
template <int WHICH>
struct RollLoader {
    unsigned off[kHalfI], offE[kHalfI];
    const float *ip;          // in1 (A) / in2 (B) of the tile's batch item
    const float *ipE;         // in2 for E (A only)
    bool first;               // the tile being fetched starts a run
    int half;                 // in2 half the new rows go to (B); E goes to the other one (A)

    __device__ __forceinline__ void new_tile(const PipeArgs &a, const TileSeq &sq, int lane, int plane) {
        first = (sq.ti == 0);
        half = (1 + sq.ti) & 1;
#pragma unroll
        for (int k = 0; k < kHalfI; ++k) {
            const int p = k * 64 + lane;
            const int c = p / 80, row = (p / 10) % 8, q = p % 10;
            if (WHICH == 0) {
                const int iy = sq.xy.y0 + row, ix = sq.xy.x0 + 4 * q;
                const bool ok = (q < kTG) && (iy < a.H) && (ix < a.W);
                off[k] = ok ? (unsigned)(c * plane + iy * a.W + ix) * 4u : kOOBv;
                const int ey = sq.xy.y0 - kD + row, ex = sq.xy.x0 - kD + 4 * q;
                const bool eok = (ey >= 0) && (ey < a.H) && (ex >= 0) && (ex < a.W);
                offE[k] = eok ? (unsigned)(c * plane + ey * a.W + ex) * 4u : kOOBv;
            } else {
                const int iy = sq.xy.y0 - kD + 8 + row, ix = sq.xy.x0 - kD + 4 * q;
                const bool ok = (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W);
                off[k] = ok ? (unsigned)(c * plane + iy * a.W + ix) * 4u : kOOBv;
            }
        }
        const int b = (PWC_PIPE_EXP & 32) ? 0 : sq.xy.b;
        ip = (WHICH == 0 ? a.in1 + (int64_t)b * a.bs1 : a.in2 + (int64_t)b * a.bs2);
        ipE = a.in2 + (int64_t)b * a.bs2;
    }

    // chunk k of the tile described by new_tile()
    __device__ __forceinline__ void issue(const PipeArgs &a, int k, float *s1ring, float *in2buf, int plane) {
        const int c0 = k * kCK;
        const int nbytes = min(kCK, a.C - c0) * plane * 4;        // channels past C fail the range check: zeros
        const pwc::v4i32 rs = pwc::make_rsrc(ip + (int64_t)c0 * plane, nbytes);
        float *dst = (WHICH == 0) ? s1ring + k * kS1F : in2buf + k * 2 * kHalfF + half * kHalfF;
        const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)pwc::lds_addr(dst));
#pragma unroll
        for (int i = 0; i < kHalfI; ++i) pwc::dma_b128(rs, base + i * 1024, off[i]);
        if (WHICH == 0 && first) {
            const pwc::v4i32 rsE = pwc::make_rsrc(ipE + (int64_t)c0 * plane, nbytes);
            const unsigned baseE = (unsigned)__builtin_amdgcn_readfirstlane((int)pwc::lds_addr(in2buf + k * 2 * kHalfF + (1 - half) * kHalfF));
#pragma unroll
            for (int i = 0; i < kHalfI; ++i) pwc::dma_b128(rsE, baseE + i * 1024, offE[i]);
        }
    }
};

template <int K, int WHICH>
__device__ __forceinline__ void roll_loader_steps(RollLoader<WHICH> &ld, const PipeArgs &a, float *s1ring, float *in2buf, int plane,
                                                  bool have_next, int steps_left) {
    if constexpr (K < kRollNCH) {
        __builtin_amdgcn_s_barrier();              // B_s: step s = (tile n, chunk K) begins; chunk K-1's place is free
        constexpr int kv = (K + kRollNCH - 1) % kRollNCH;       // the chunk vacated by the step that has just ended
        // (K = 0: chunk 7 of the PREVIOUS tile was vacated -- its successor is chunk 7 of THIS tile, fetched from the previous
        // tile's loop with that tile's new_tile(); so the fetches of this loop are chunks 0..6 of tile n+1 at K = 1..7, and chunk 7
        // of tile n+1 at K = 0 of the next loop: see roll_loader_wave)
        if (K >= 1 && have_next && !(PWC_PIPE_EXP & 4)) ld.issue(a, kv, s1ring, in2buf, plane);
        // the fetch needed at B_{s+1} (chunk of step s+2) has landed: at most NCH-3 younger fetches of 5 instructions are outstanding
        // (a run's first tile makes loader A's 10 -- then this waits for a little more than it has to)
        const int younger = min(kRollNCH - 3, steps_left - 3 - K);        // fetches issued after the one for step s+2
        wait_chunks_in_flight<kHalfI>(max(younger, 0));
        roll_loader_steps<K + 1, WHICH>(ld, a, s1ring, in2buf, plane, have_next, steps_left);
    }
}

template <int WHICH>
__device__ __forceinline__ void roll_loader_wave(const PipeArgs &a, float *s1ring, float *in2buf, int lane, int my_tiles) {
    __builtin_amdgcn_s_setprio(3);
    const int plane = a.H * a.W;
    TileSeq sq;
    sq.start(a);
    RollLoader<WHICH> ld;
    ld.new_tile(a, sq, lane, plane);
    // prologue: the whole first tile.  Loader A issues 8 x 10 instructions: six chunks, a wait, the other two (vmcnt is a 6-bit counter)
#pragma unroll
    for (int k = 0; k < 6; ++k) ld.issue(a, k, s1ring, in2buf, plane);
    if (WHICH == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * 2 * kHalfI) : "memory");        // chunks 0, 1 have landed
    ld.issue(a, 6, s1ring, in2buf, plane);
    ld.issue(a, 7, s1ring, in2buf, plane);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6 * kHalfI) : "memory");      // B_0 promises chunks 0 and 1 (loader B: 6 x 5 younger; A: more than needed)
    // tile loop: during tile n the chunks 0..6 of tile n+1 are fetched at K = 1..7 and chunk 7 of tile n at K = 0 (vacated by tile n-1)
    bool pending7 = false;      // chunk 7 of the tile `ld` describes is still to be fetched (at K = 0 of the next loop)
#pragma unroll 1
    for (int n = 0; n < my_tiles; ++n) {
        // K = 0 belongs to the tile described by the loader state of the previous loop: fetch its chunk 7 first
        __builtin_amdgcn_s_barrier();              // B_s, s = (n, 0)
        if (pending7 && !(PWC_PIPE_EXP & 4)) ld.issue(a, kRollNCH - 1, s1ring, in2buf, plane);
        const int steps_left = (my_tiles - n) * kRollNCH;               // ring steps from (n, 0) to the end
        {
            const int younger = min(kRollNCH - 3, steps_left - 3);
            wait_chunks_in_flight<kHalfI>(max(younger, 0));
        }
        const bool have_next = (n + 1 < my_tiles);
        if (have_next) { sq.advance(a); ld.new_tile(a, sq, lane, plane); }
        pending7 = have_next;
        roll_loader_steps<1, WHICH>(ld, a, s1ring, in2buf, plane, have_next, steps_left);
    }
    __builtin_amdgcn_s_barrier();                   // the barrier that ends the last step
}

template <int LA>
__device__ __forceinline__ void roll_fma_wave(const PipeArgs &a, const float *s1ring, const float *in2buf, int wave, int lane, int my_tiles) {
    int r, g;
    lane_to_rg(lane, r, g);
    FmaState<LA> st;
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
    st.plane4 = a.H * a.W * 4;
    st.sbase = wave * kND * st.plane4;
    st.voff = kOOBv;
    st.rs = __builtin_amdgcn_make_buffer_rsrc((void *)nullptr, 0, 0, 0x00020000);
    const float *s1l = s1ring + r * kPitch + 4 * g;
    // this lane's halo row i = r + wave lives in half ((i >> 3) + ti) & 1 of every chunk's in2 block
    const int i = r + wave;
    const float *s2e = in2buf + (i >> 3) * kHalfF + (i & 7) * kPitch + 4 * g;            // run tile number even
    const float *s2o = in2buf + (1 - (i >> 3)) * kHalfF + (i & 7) * kPitch + 4 * g;      // odd
    TileSeq sq;
    sq.start(a);
    TileXY prev = sq.xy;
    __builtin_amdgcn_s_barrier();              // B_0: chunks 0 and 1 are readable
    asm volatile("" ::: "memory");
    load_ops<LA, 8 * kPitch>(st.ha, s1l, s2e, 0);
#pragma unroll 1
    for (int n = 0; n < my_tiles; ++n) {
        if (n > 0) set_destination_xy(st, a, prev, r, g);
        prev = sq.xy;
        const float *s2l = (sq.ti & 1) ? s2o : s2e;
        sq.advance(a);                                                  // (past the last tile: unused)
        const float *s2n = (sq.ti & 1) ? s2o : s2e;
        fma_steps<kRollNCH, 0, LA, 8 * kPitch>(st, a, s1l, s2l, s2n, wave, n > 0);
    }
    set_destination_xy(st, a, prev, r, g);
    tail_stores<kRollNCH, 0, LA>(st, wave);
}

__global__ void __launch_bounds__(kThreadsRoll, 3)
corr81_roll_kernel(PipeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s1ring = smem;                            // [8 chunks][kS1F]
    float *in2buf = s1ring + kRollNCH * kS1F;        // [8 chunks][2 halves][kHalfF]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int my_tiles = roll_my_tiles(a);
    if (wave == kND)          roll_loader_wave<0>(a, s1ring, in2buf, lane, my_tiles);
    else if (wave == kND + 1) roll_loader_wave<1>(a, s1ring, in2buf, lane, my_tiles);
    else                      roll_fma_wave<2>(a, s1ring, in2buf, wave, lane, my_tiles);
}

// =====================================================================================================================
// WARP: fused warp + correlation (PWCNet.py:212-213, 226-227, 240-241, 256-257: corr(c1, warp(c2, up_flow * s)); the warped tensor
// has no other consumer).  The round-2 form had five waves gather the bilinear taps of the 640 halo pixels from global memory --
// 640 eight-byte gather instructions per tile, 1.46 M per launch, which alone paced the kernel (profiles/r03_pmc_summary.txt).
// Here the taps come out of LDS:
//   * A SOURCE WINDOW of c2 -- 23 rows x 52 columns per channel, 4 channels per ring step -- is staged by 16-byte LDS-DMA like
//     every other operand (19 coalesced instructions per chunk; neighbouring tiles' windows overlap, so most of it is an L2 hit,
//     which the measurements of this round price at next to nothing).  Its origin follows the flow: the sample position of the
//     tile's centre pixel, computed from one scalar load of the flow by every wave that needs it (loaders and producers run
//     the same arithmetic on the same two numbers -- nothing is handed over).  The window leaves the halo tile a margin of
//     4 columns / 3 rows of flow VARIATION inside the tile on every side (the flow itself may be anything).
//   * Four PRODUCER waves own three halo pixels per lane: taps, weights and mask once per tile (pwc_warp_taps.h, the same code as
//     every warp kernel), then per ring step two ds_read2_b32 + blend4 + one ds_write_b32 per pixel and channel into the in2 image
//     the fma waves read, two steps ahead of them.
//   * A pixel whose taps leave the window (wild flow, motion boundary) takes the round-2 path for that pixel only: 8-byte
//     gathers from global memory, same blend -- bit-identical either way (test_warp_correlation_fused_equals_two_kernels).
constexpr int kWR = 23, kWQ = 13, kWC = 4 * kWQ;              // window rows, 16-byte pieces per row, columns
constexpr int kWMy = 3, kWMx = 4;                             // margins (rows / columns) around the halo tile's own extent
constexpr int kWinF = kCK * kWR * kWC;                        // 4784 floats per ring slot
constexpr int kWinI = (kCK * kWR * kWQ + 63) / 64;            // 19 LDS-DMA instructions per chunk (the last one partly out of range)
constexpr int kWinSlots = 4, kS2Slots = 4;
// Sixteen waves is all a workgroup gets, and the producers' chain per ring step is the kernel's critical path: the in1 stream (five
// LDS-DMA instructions per step) rides on the second window loader wave, which frees a wave for a FIFTH producer -- two halo
// pixels per lane instead of three (5 x 64 x 2 = 640 exactly): 104.9 -> 100.0 us on one box.  PWC_PIPE_PROD4 = the first form
// (own in1 loader wave, 4 x 3 pixels).
#ifdef PWC_PIPE_PROD4
constexpr int kProducers = 4, kProdPx = 3, kLoaderWaves = 3;  // 4 x 64 x 3 = 768 >= 640 halo pixels
#else
constexpr int kProducers = 5, kProdPx = 2, kLoaderWaves = 2;
#endif
static_assert(kProducers * 64 * kProdPx >= kS2Rows * kPitch, "halo pixels");
constexpr int kWaveWin0 = kND + kLoaderWaves - 2, kWaveProd0 = kND + kLoaderWaves;    // waves: 0-8 fma, [9 in1 loader,] two window loaders, producers
constexpr int kThreadsWarp = 64 * (kND + kLoaderWaves + kProducers);     // 1024: four waves per SIMD, 128 registers
static_assert(kThreadsWarp == 1024, "sixteen waves");
constexpr int kLdsWarp = (kR * kS1F + kS2Slots * kS2F + kWinSlots * kWinF) * 4;
static_assert(kLdsWarp <= 160 * 1024, "LDS");
static_assert(kWR == kS2Rows + 2 * kWMy + 1 && kWC >= kPitch + 2 * kWMx + 1 + 3, "window covers the halo tile + margins + tap + alignment");

__device__ __forceinline__ float scalar_load(const float *p) {      // wave-uniform address; not counted in vmcnt (the loaders count by hand)
    float v;
    asm volatile("s_nop 4\n\ts_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// two wave-uniform loads in flight at once (u and v of the window origin's flow sample); like scalar_load, outside vmcnt
__device__ __forceinline__ void scalar_load2(const float *p0, const float *p1, float &v0, float &v1) {
    asm volatile("s_nop 4\n\ts_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(v0), "=&s"(v1) : "s"(p0), "s"(p1) : "memory");
}

// window origin of a tile: from the sample position of its centre pixel (halo row 7, halo column 19), the same for every wave.
// (The centre pixel's position displaced by the MEAN of the flow at four pixels of the tile leaves 1.1 % instead of 1.8 % of the
// benchmark's halo pixels outside the window -- tests/flow_window_stats.py "mean4" -- and ran 3.5 us SLOWER: what a tile step can
// least afford is more work between two tiles, profiles/r04_corr_notes.md section 7.)
struct WinOrg { int wx0, wy0; };
struct WinCentre { int gxc, gyc; const float *fu; };       // the centre pixel and the address of its u (v one plane behind)
__device__ __forceinline__ WinCentre window_centre(const PipeArgs &a, const TileXY &t) {
    WinCentre c;
    c.gxc = min(t.x0 + 15, a.W - 1); c.gyc = min(t.y0 + 3, a.H - 1);
    c.fu = a.flo + (int64_t)t.b * a.bsf + (int64_t)c.gyc * a.W + c.gxc;
    return c;
}
// u_raw / v_raw: the flow at the centre pixel as stored (every wave applies the same arithmetic to the same two numbers)
__device__ __forceinline__ WinOrg window_origin_from(const PipeArgs &a, const TileXY &t, const WinCentre &wc, float u_raw, float v_raw) {
    const int gxc = wc.gxc, gyc = wc.gyc;
    const float u = u_raw * a.flow_scale, v = v_raw * a.flow_scale;
    const float px = (float)gxc + u, py = (float)gyc + v;
    const float gx = 2.0f * px / (float)max(a.W - 1, 1) - 1.0f, gy = 2.0f * py / (float)max(a.H - 1, 1) - 1.0f;
    float ix, iy;
    if (a.align_corners) { ix = (gx + 1.0f) / 2.0f * (float)(a.W - 1); iy = (gy + 1.0f) / 2.0f * (float)(a.H - 1); }
    else                 { ix = ((gx + 1.0f) * (float)a.W - 1.0f) / 2.0f; iy = ((gy + 1.0f) * (float)a.H - 1.0f) / 2.0f; }
    // anywhere is correct (pixels outside take the gather path); keep the integers small.  !(x > lo) also catches NaN
    ix = !(ix > -64.0f) ? -64.0f : fminf(ix, (float)a.W + 64.0f);
    iy = !(iy > -64.0f) ? -64.0f : fminf(iy, (float)a.H + 64.0f);
    const int cx = (int)floorf(ix) - (gxc - t.x0 - 15), cy = (int)floorf(iy) - (gyc - t.y0 - 3);     // (undo the clamp of the centre pixel)
    WinOrg o;
    o.wx0 = (cx - 19 - kWMx) & ~3;            // 16-byte pieces: a multiple of 4 at or left of the leftmost tap (two's complement: floors)
    o.wy0 = cy - 7 - kWMy;
    return o;
}
// loader waves: scalar loads (their vmcnt is counted by hand), u and v behind ONE wait
__device__ __forceinline__ WinOrg window_origin(const PipeArgs &a, const TileXY &t) {
    const WinCentre wc = window_centre(a, t);
    const float *fu = (const float *)pwc::uniform_ptr(wc.fu);
    float u, v;
#ifdef PWC_PIPE_SLOAD1
    u = scalar_load(fu); v = scalar_load(fu + (int64_t)a.H * a.W);
#else
    scalar_load2(fu, fu + (int64_t)a.H * a.W, u, v);
#endif
    return window_origin_from(a, t, wc, u, v);
}

// A workgroup's tiles are known when it starts (blockIdx.x + n gridDim.x), so every wave that needs window origins computes them
// ALL up front -- lane n: the origin of the workgroup's n-th tile, one round trip of ordinary loads for the whole launch -- and a
// tile step reads its two integers with v_readlane instead of waiting for two dependent scalar loads between two tiles (the tile
// step is what the whole workgroup waits for: the two loads behind one wait were worth 2 us of 97, profiles/r04_corr_notes.md
// section 7).  Tiles past the 64th of a workgroup (batches above ~140 pairs at level 2) take the scalar loads.
struct OrgTable {
    int wx, wy;
    __device__ __forceinline__ void fill(const PipeArgs &a, int lane, int my_tiles) {
        wx = 0; wy = 0;
#ifndef PWC_PIPE_NO_ORGTABLE
        if (lane < my_tiles) {
            const TileXY t = tile_of((int)blockIdx.x + lane * (int)gridDim.x, a);
            const WinCentre wc = window_centre(a, t);
            const WinOrg o = window_origin_from(a, t, wc, wc.fu[0], wc.fu[(int64_t)a.H * a.W]);
            wx = o.wx0; wy = o.wy0;
        }
#endif
    }
    __device__ __forceinline__ WinOrg get(const PipeArgs &a, const TileXY &t, int n) const {      // n wave-uniform
#ifndef PWC_PIPE_NO_ORGTABLE
        if (n < 64) {
            WinOrg o;
            o.wx0 = __builtin_amdgcn_readlane(wx, n);
            o.wy0 = __builtin_amdgcn_readlane(wy, n);
            return o;
        }
#endif
        return window_origin(a, t);
    }
};

// ---- window loader wave WHICH (0 / 1): instructions WHICH, WHICH + 2, ... of the chunk's 19 --------------------------------
template <int WHICH>
struct WinLoader {
    static constexpr int I = (kWinI - WHICH + 1) / 2;       // 10 / 9
    unsigned off[I];
    const float *ip;
    OrgTable tab;
    // the workgroup's n-th tile
    __device__ __forceinline__ void new_tile(const PipeArgs &a, int n, int stride, int lane, int plane) {
        const TileXY t = tile_of((int)blockIdx.x + n * stride, a);
        const WinOrg o = tab.get(a, t, n);
#pragma unroll
        for (int j = 0; j < I; ++j) {
            const int p = (WHICH + 2 * j) * 64 + lane;
            const int c = p / (kWR * kWQ), rem = p % (kWR * kWQ), row = rem / kWQ, q = rem % kWQ;
            const int iy = o.wy0 + row, ix = o.wx0 + 4 * q;
            const bool ok = (c < kCK) && (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W);       // W % 4 == 0: a piece is all-in or all-out
            off[j] = ok ? (unsigned)(c * plane + iy * a.W + ix) * 4u : kOOBv;
        }
        ip = a.in2 + (int64_t)((PWC_PIPE_EXP & 32) ? 0 : t.b) * a.bs2;
    }
    __device__ __forceinline__ void issue(const PipeArgs &a, int chunk, int slot, float *win, int plane) {
        const int c0 = chunk * kCK;
        const int nbytes = min(kCK, a.C - c0) * plane * 4;        // channels past C fail the range check: zeros
        const pwc::v4i32 rs = pwc::make_rsrc(ip + (int64_t)c0 * plane, nbytes);
        const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)pwc::lds_addr(win + slot * kWinF));
#pragma unroll
        for (int j = 0; j < I; ++j) pwc::dma_b128(rs, base + (WHICH + 2 * j) * 1024, off[j]);
    }
};

// step s = t NCH + K: window chunk s+4 goes into the slot chunk s has left (the producers read chunk s two steps ago); then the
// wave waits until chunk s+3 has landed (the barrier that ends the step promises it to the producers)
template <int NCH, int K, int WHICH>
__device__ __forceinline__ void win_loader_steps(WinLoader<WHICH> &ld, const PipeArgs &a, float *win, int lane, int plane, int t, int nsteps, int stride) {
    if constexpr (K < NCH) {
        constexpr int I = WinLoader<WHICH>::I;
        const int s = t * NCH + K;
        constexpr int kc = (K + kWinSlots) % NCH;
        if (s + kWinSlots < nsteps && !(PWC_PIPE_EXP & 4)) {
            if constexpr (kc == 0) ld.new_tile(a, t + (K + kWinSlots) / NCH, stride, lane, plane);
            ld.issue(a, kc, K % kWinSlots, win, plane);
        }
        if (s + kWinSlots < nsteps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(I) : "memory");      // chunk s+3 landed, chunk s+4 may fly
        else                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        win_loader_steps<NCH, K + 1, WHICH>(ld, a, win, lane, plane, t, nsteps, stride);
    }
}

template <int NCH, int WHICH>
__device__ __forceinline__ void win_loader_wave(const PipeArgs &a, float *win, int lane, int my_tiles) {
    constexpr int I = WinLoader<WHICH>::I;
    __builtin_amdgcn_s_setprio(3);
    const int plane = a.H * a.W, stride = gridDim.x, nsteps = my_tiles * NCH;
    WinLoader<WHICH> ld;
    ld.tab.fill(a, lane, my_tiles);
    ld.new_tile(a, 0, stride, lane, plane);
#pragma unroll
    for (int k = 0; k < kWinSlots; ++k) ld.issue(a, k, k, win, plane);      // chunks 0..3 of the first tile (NCH >= 8)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(I) : "memory");               // chunks 0, 1, 2 have landed
    __builtin_amdgcn_s_barrier();                   // B_pre: the producers make the first two in2 chunks
    __builtin_amdgcn_s_barrier();                   // B_0
#pragma unroll 1
    for (int t = 0; t < my_tiles; ++t) win_loader_steps<NCH, 0, WHICH>(ld, a, win, lane, plane, t, nsteps, stride);
}

// ---- in1 loader wave of the fused kernel: the ring form's loader restricted to in1 --------------------------------------
template <int NCH, int K>
__device__ __forceinline__ void in1_loader_steps(Loader<0, kS1I> &ld, const PipeArgs &a, float *s1ring, int lane, int plane, int t, int nsteps, int stride) {
    if constexpr (K < NCH) {
        const int s = t * NCH + K;
        constexpr int kc = (K + kR - 1) % NCH;
        if (s + kR - 1 < nsteps && !(PWC_PIPE_EXP & 4)) {
            if constexpr (kc == 0) ld.new_tile(a, (int)blockIdx.x + (t + (K + kR - 1) / NCH) * stride, lane, plane);
            ld.issue(a, kc, (K + kR - 1) % kR, s1ring, nullptr, plane);
        }
        if (s + kR - 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((kR - 3) * kS1I) : "memory");
        else if (s + 1 < nsteps) wait_chunks_in_flight<kS1I>(max(nsteps - 1 - (s + 2), 0));
        __builtin_amdgcn_s_barrier();
        in1_loader_steps<NCH, K + 1>(ld, a, s1ring, lane, plane, t, nsteps, stride);
    }
}

template <int NCH>
__device__ __forceinline__ void in1_loader_wave(const PipeArgs &a, float *s1ring, int lane, int my_tiles) {
    __builtin_amdgcn_s_setprio(3);
    const int plane = a.H * a.W, stride = gridDim.x, nsteps = my_tiles * NCH;
    Loader<0, kS1I> ld;
    ld.new_tile(a, blockIdx.x, lane, plane);
#pragma unroll
    for (int k = 0; k < kR - 1; ++k)
        if (k < nsteps) ld.issue(a, k, k, s1ring, nullptr, plane);
    wait_chunks_in_flight<kS1I>(max(min(kR - 1, nsteps) - 2, 0));     // chunks 0 and 1
    __builtin_amdgcn_s_barrier();                   // B_pre
    __builtin_amdgcn_s_barrier();                   // B_0
#pragma unroll 1
    for (int t = 0; t < my_tiles; ++t) in1_loader_steps<NCH, 0>(ld, a, s1ring, lane, plane, t, nsteps, stride);
}

// ---- window loader 1 + in1 loader in ONE wave ------------------------------------------------------------------------
// Per step, in issue order: in1 chunk s+kR-1 (kS1I instructions), then window chunk s+4 (I).  vmcnt retires in order, so "window
// chunk s+3 has landed" = at most the kS1I + I instructions issued after it are still in flight -- and every in1 chunk the fma
// waves can want (issued six steps earlier) has landed with it.
template <int NCH, int K>
__device__ __forceinline__ void win_in1_loader_steps(WinLoader<1> &lw, Loader<0, kS1I> &l1, const PipeArgs &a, float *s1ring, float *win,
                                                     int lane, int plane, int t, int nsteps, int stride) {
    if constexpr (K < NCH) {
        constexpr int I = WinLoader<1>::I;
        const int s = t * NCH + K;
        constexpr int kc1 = (K + kR - 1) % NCH, kcw = (K + kWinSlots) % NCH;
        const bool go1 = s + kR - 1 < nsteps && !(PWC_PIPE_EXP & 4), gow = s + kWinSlots < nsteps && !(PWC_PIPE_EXP & 4);
        if (go1) {
            if constexpr (kc1 == 0) l1.new_tile(a, (int)blockIdx.x + (t + (K + kR - 1) / NCH) * stride, lane, plane);
            l1.issue(a, kc1, (K + kR - 1) % kR, s1ring, nullptr, plane);
        }
        if (gow) {
            if constexpr (kcw == 0) lw.new_tile(a, t + (K + kWinSlots) / NCH, stride, lane, plane);
            lw.issue(a, kcw, K % kWinSlots, win, plane);
        }
        if (go1)      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kS1I + I) : "memory");     // (go1 implies gow: kR - 1 > kWinSlots)
        else if (gow) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(I) : "memory");
        else          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        win_in1_loader_steps<NCH, K + 1>(lw, l1, a, s1ring, win, lane, plane, t, nsteps, stride);
    }
}

template <int NCH>
__device__ __forceinline__ void win_in1_loader_wave(const PipeArgs &a, float *s1ring, float *win, int lane, int my_tiles) {
    constexpr int I = WinLoader<1>::I;
    static_assert(kR - 1 > kWinSlots && (kR - 1) * kS1I + (kWinSlots - 1) * I <= 63, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_setprio(3);
    const int plane = a.H * a.W, stride = gridDim.x, nsteps = my_tiles * NCH;
    WinLoader<1> lw;
    Loader<0, kS1I> l1;
    lw.tab.fill(a, lane, my_tiles);
    lw.new_tile(a, 0, stride, lane, plane);
    l1.new_tile(a, blockIdx.x, lane, plane);
#pragma unroll
    for (int k = 0; k < kR - 1; ++k)
        if (k < nsteps) l1.issue(a, k, k, s1ring, nullptr, plane);          // in1 chunks 0..6
#pragma unroll
    for (int k = 0; k < kWinSlots - 1; ++k) lw.issue(a, k, k, win, plane);   // window chunks 0, 1, 2 (NCH >= 8)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((kWinSlots - 1) * I) : "memory");      // the in1 chunks have landed (keeps the counter under 64)
    lw.issue(a, kWinSlots - 1, kWinSlots - 1, win, plane);                  // window chunk 3
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(I) : "memory");               // window chunks 0, 1, 2 have landed
    __builtin_amdgcn_s_barrier();                   // B_pre: the producers make the first two in2 chunks
    __builtin_amdgcn_s_barrier();                   // B_0
#pragma unroll 1
    for (int t = 0; t < my_tiles; ++t) win_in1_loader_steps<NCH, 0>(lw, l1, a, s1ring, win, lane, plane, t, nsteps, stride);
}

// ---- producer waves ---------------------------------------------------------------------------------------------------
struct ProdPx {                  // one halo pixel of this lane
    int dst;                     // float index inside an in2 chunk image [c][16][40]; < 0: no pixel (slot past the 640)
    int wt, wb;                  // float index of the top / bottom tap pair inside a window chunk image [c][23][52]; wt < 0: outside
    int otop, obot;              // byte offsets of the pairs inside a plane (gather path)
    float wa, wb_, wc, wd;
};

struct ProdFlow { float u[kProdPx], v[kProdPx], cu, cv; WinCentre wc; WinOrg org; TileXY t; int n; };      // what prod_setup needs from global memory, requested early

// the flow at this lane's halo pixels and the window origin of `tile`: issued at the START of the last step of the tile before,
// used at its end (a dependent global load behind a barrier would cost the producers ~1 us per tile)
__device__ __forceinline__ void prod_fetch_flow(ProdFlow &pf, const PipeArgs &a, int n, int stride, int pw, int lane) {
    pf.n = n;
    pf.t = tile_of((int)blockIdx.x + n * stride, a);
    const int plane = a.H * a.W;
    const float *fu = a.flo + (int64_t)pf.t.b * a.bsf;
#pragma unroll
    for (int k = 0; k < kProdPx; ++k) {
        const int hp = (pw * kProdPx + k) * 64 + lane;              // consecutive lanes = consecutive halo columns
        const int row = hp / kPitch, col = hp - row * kPitch;
        const int gyc = min(max(pf.t.y0 - kD + row, 0), a.H - 1), gxc = min(max(pf.t.x0 - kD + col, 0), a.W - 1);
        pf.u[k] = fu[(int64_t)gyc * a.W + gxc];
        pf.v[k] = fu[(int64_t)plane + (int64_t)gyc * a.W + gxc];
    }
#if defined(PWC_PIPE_PROD_SLOAD)
    pf.org = window_origin(a, pf.t);
#elif defined(PWC_PIPE_PROD_VLOAD)
    // the centre pixel's flow like the pixels' own: ordinary loads of a wave-uniform address, waited for where prod_setup uses them
    // (the scalar loads of window_origin() block the wave twice per tile, inside the step the whole workgroup waits for)
    pf.wc = window_centre(a, pf.t);
    pf.cu = pf.wc.fu[0];
    pf.cv = pf.wc.fu[plane];
#endif
}

__device__ __forceinline__ void prod_setup(ProdPx (&px)[kProdPx], const PipeArgs &a, ProdFlow &pf, const OrgTable &tab, int pw, int lane) {
#if defined(PWC_PIPE_PROD_VLOAD)
    pf.org = window_origin_from(a, pf.t, pf.wc, pf.cu, pf.cv);
#elif !defined(PWC_PIPE_PROD_SLOAD)
    pf.org = tab.get(a, pf.t, pf.n);          // (the origin's arithmetic is out of the step the whole workgroup waits for)
#endif
#pragma unroll
    for (int k = 0; k < kProdPx; ++k) {
        const int hp = (pw * kProdPx + k) * 64 + lane;
        const int row = hp / kPitch, col = hp - row * kPitch;
        const int gy = pf.t.y0 - kD + row, gx = pf.t.x0 - kD + col;
        const bool inimg = (gy >= 0) && (gy < a.H) && (gx >= 0) && (gx < a.W);
        const float u = pf.u[k] * a.flow_scale, v = pf.v[k] * a.flow_scale;
        const pwc_warp::PairTaps pt = pwc_warp::make_pair_taps((float)gx + u, (float)gy + v, a.H, a.W, a.align_corners, a.thr);
        // a halo pixel outside the image is the correlation's zero padding: all four weights zero
        px[k].wa = inimg ? pt.wa : 0.f; px[k].wb_ = inimg ? pt.wb : 0.f; px[k].wc = inimg ? pt.wc : 0.f; px[k].wd = inimg ? pt.wd : 0.f;
        px[k].otop = pt.otop * 4;
        px[k].obot = pt.obot * 4;
        // rows and column straight from the taps: no runtime integer division (~35 instructions each on gfx950; four per pixel here and
        // in make_pair_taps cost 1.7 us of the kernel's 100 -- this is the step the whole workgroup waits for)
        const int rt = pt.rtop, xb = pt.xb, rb = pt.rbot;
        const int wr = rt - pf.org.wy0, wr2 = rb - pf.org.wy0, wc = xb - pf.org.wx0;
        const bool inside = ((wr >= 0) && (wr2 < kWR) && (wr2 >= wr) && (wc >= 0) && (wc + 1 < kWC) && !(PWC_PIPE_EXP & 2048)) || (PWC_PIPE_EXP & 8192);
        px[k].wt = inside ? ((PWC_PIPE_EXP & 8192) ? min(max(wr, 0), kWR - 2) * kWC + min(max(wc, 0), kWC - 2) : wr * kWC + wc) : -1;
        px[k].wb = (PWC_PIPE_EXP & 8192) ? px[k].wt + kWC : wr2 * kWC + wc;
        px[k].dst = (hp < kS2Rows * kPitch) ? row * kPitch + col : -1;
    }
}

struct ProdState {
    ProdPx px[kProdPx];
    OrgTable tab;                // window origins of the workgroup's tiles
    const float *src;            // c2 of the tile's batch item
    f32x2 gt[kProdPx][kCK], gb[kProdPx][kCK];     // taps of the pixels outside the window, gathered one step ahead
};

// pixels inside the window: two ds_read2_b32 + blend4 + one ds_write_b32 per pixel and channel.  (Requesting the reads of two or
// three pixels before the first blend was tried: 16 / 32 more live registers next to the gathered taps spill at 128, 123 vs 109 us;
// with two pixels per lane nothing spills and nothing is gained: 100.0 vs 100.4 us.)
__device__ __forceinline__ void prod_sample(const ProdState &ps, float *dst, const float *win) {
#pragma unroll
    for (int k = 0; k < kProdPx; ++k) {
        if (ps.px[k].dst >= 0 && ps.px[k].wt >= 0) {
            f32x2 top[kCK], bot[kCK];
#pragma unroll
            for (int c = 0; c < kCK; ++c) {
                const float *wt = win + c * kWR * kWC + ps.px[k].wt, *wb = win + c * kWR * kWC + ps.px[k].wb;
                top[c] = (f32x2){wt[0], wt[1]};
                bot[c] = (f32x2){wb[0], wb[1]};
            }
#pragma unroll
            for (int c = 0; c < kCK; ++c)
                dst[c * kS2Rows * kPitch + ps.px[k].dst] =
                    pwc_warp::blend4(top[c][0], top[c][1], bot[c][0], bot[c][1], ps.px[k].wa, ps.px[k].wb_, ps.px[k].wc, ps.px[k].wd);
        }
    }
}

// pixels outside the window (wild flow, motion boundary): the round-2 path for those pixels only -- 8-byte gathers from global
// memory, ISSUED one ring step before they are blended so that their round trip is behind a whole step of other work
__device__ __forceinline__ void prod_gather_issue(ProdState &ps, const PipeArgs &a, int chunk) {
    const int plane = a.H * a.W, c0 = chunk * kCK;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pwc::uniform_ptr(ps.src + (int64_t)c0 * plane), 0,
                                                                  __builtin_amdgcn_readfirstlane(min(kCK, a.C - c0) * plane * 4), 0x00020000);
#pragma unroll
    for (int k = 0; k < kProdPx; ++k) {
        const bool out = ps.px[k].dst >= 0 && ps.px[k].wt < 0 && !(PWC_PIPE_EXP & 16384);
        if (__builtin_amdgcn_ballot_w64(out)) {                 // wave-uniform: some pixel of this slot has left the window
            if (out) {
#pragma unroll
                for (int c = 0; c < kCK; ++c) {                 // channels past C fail the range check and read as 0 (ragged last chunk)
                    ps.gt[k][c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, ps.px[k].otop, c * plane * 4, 0));
                    ps.gb[k][c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, ps.px[k].obot, c * plane * 4, 0));
                }
            }
        }
    }
}

__device__ __forceinline__ void prod_gather_finish(const ProdState &ps, float *dst) {
#pragma unroll
    for (int k = 0; k < kProdPx; ++k) {
        const bool out = ps.px[k].dst >= 0 && ps.px[k].wt < 0 && !(PWC_PIPE_EXP & (16384 | 32768));
        if (__builtin_amdgcn_ballot_w64(out)) {
            if (out) {
#pragma unroll
                for (int c = 0; c < kCK; ++c)
                    dst[c * kS2Rows * kPitch + ps.px[k].dst] = pwc_warp::blend4(ps.gt[k][c][0], ps.gt[k][c][1], ps.gb[k][c][0], ps.gb[k][c][1],
                                                                                 ps.px[k].wa, ps.px[k].wb_, ps.px[k].wc, ps.px[k].wd);
            }
        }
    }
}

// step s = t NCH + K: the in2 image of chunk s+2 from window chunk s+2, two steps ahead of the fma waves (they read one chunk ahead):
// sample the pixels inside the window, finish the gathers issued during the previous step, then issue the gathers of chunk s+3 --
// after moving on to the next tile's taps when chunk s+2 was the last of its tile (its flow was requested at the start of the step).
template <int NCH, int K>
__device__ __forceinline__ void prod_steps(ProdState &ps, const PipeArgs &a, float *s2ring, const float *win, int pw, int lane, int t, int nsteps, int stride) {
    if constexpr (K < NCH) {
        const int s = t * NCH + K;
        constexpr int kc = (K + 2) % NCH;               // chunk (inside its tile) produced during this step
        if (s + 2 < nsteps && !(PWC_PIPE_EXP & 4096)) {
            ProdFlow pf;
            const bool next_tile = (kc == NCH - 1) && (s + 3 < nsteps);
            if constexpr (kc == NCH - 1)
                if (next_tile) prod_fetch_flow(pf, a, t + (K + 3) / NCH, stride, pw, lane);
            float *dst = s2ring + ((K + 2) % kS2Slots) * kS2F;
            prod_sample(ps, dst, win + ((K + 2) % kWinSlots) * kWinF);
            prod_gather_finish(ps, dst);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the image is in LDS before this wave reaches the barrier
            if (s + 3 < nsteps) {
                if constexpr (kc == NCH - 1) {
                    prod_setup(ps.px, a, pf, ps.tab, pw, lane);
                    ps.src = a.in2 + (int64_t)((PWC_PIPE_EXP & 32) ? 0 : pf.t.b) * a.bs2;
                }
                prod_gather_issue(ps, a, (kc + 1) % NCH);
            }
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        prod_steps<NCH, K + 1>(ps, a, s2ring, win, pw, lane, t, nsteps, stride);
    }
}

template <int NCH>
__device__ __forceinline__ void producer_wave(const PipeArgs &a, float *s2ring, const float *win, int pw, int lane, int my_tiles) {
    // the producers' chain is the kernel's critical path and they share their SIMDs with fma waves: issue priority above those
    // (loaders: 3).  Interleaved same-box pairs: 100.8 / 98.8 us without, 96.2-96.7 with priority 1 or 2, 99.1 with 3.
    __builtin_amdgcn_s_setprio(2);
    const int stride = gridDim.x, nsteps = my_tiles * NCH;
    ProdState ps;
    {
        ProdFlow pf;
        ps.tab.fill(a, lane, my_tiles);
        prod_fetch_flow(pf, a, 0, stride, pw, lane);
        prod_setup(ps.px, a, pf, ps.tab, pw, lane);
        ps.src = a.in2 + (int64_t)((PWC_PIPE_EXP & 32) ? 0 : pf.t.b) * a.bs2;
    }
    prod_gather_issue(ps, a, 0);
    __builtin_amdgcn_s_barrier();                   // B_pre: window chunks 0, 1, 2 have landed
    asm volatile("" ::: "memory");
    prod_sample(ps, s2ring, win);
    prod_gather_finish(ps, s2ring);
    if (nsteps > 1) {
        prod_gather_issue(ps, a, 1);
        prod_sample(ps, s2ring + kS2F, win + kWinF);
        prod_gather_finish(ps, s2ring + kS2F);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (nsteps > 2) prod_gather_issue(ps, a, 2);    // (NCH >= 8: chunk 2 is in the first tile)
    __builtin_amdgcn_s_barrier();                   // B_0
    asm volatile("" ::: "memory");
#pragma unroll 1
    for (int t = 0; t < my_tiles; ++t) prod_steps<NCH, 0>(ps, a, s2ring, win, pw, lane, t, nsteps, stride);
}

template <int NCH>
__device__ __forceinline__ void warp_fma_wave(const PipeArgs &a, const float *s1ring, const float *s2ring, int wave, int lane, int my_tiles) {
    int r, g;
    lane_to_rg(lane, r, g);
    FmaState<1> st;
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
    st.plane4 = a.H * a.W * 4;
    st.sbase = wave * kND * st.plane4;
    st.voff = kOOBv;
    st.rs = __builtin_amdgcn_make_buffer_rsrc((void *)nullptr, 0, 0, 0x00020000);
    const int stride = gridDim.x;
    const float *s1l = s1ring + r * kPitch + 4 * g, *s2l = s2ring + (r + wave) * kPitch + 4 * g;
    __builtin_amdgcn_s_barrier();              // B_pre
    __builtin_amdgcn_s_barrier();              // B_0: in1 chunks 0, 1 and the in2 images of chunks 0, 1 are readable
    asm volatile("" ::: "memory");
    load_ops<1, kS2Rows * kPitch>(st.ha, s1l, s2l, 0);
#pragma unroll 1
    for (int t = 0; t < my_tiles; ++t) {
        if (t > 0) set_destination(st, a, (int)blockIdx.x + (t - 1) * stride, r, g);
        fma_steps<NCH, 0, 1, kS2Rows * kPitch, kS2Slots>(st, a, s1l, s2l, s2l, wave, t > 0);
    }
    set_destination(st, a, (int)blockIdx.x + (my_tiles - 1) * stride, r, g);
    tail_stores<NCH, 0, 1>(st, wave);
}

template <int NCH>
__global__ void __launch_bounds__(kThreadsWarp, 4)
warp_corr81_pipe_kernel(PipeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s1ring = smem;                            // [kR][kS1F]
    float *s2ring = s1ring + kR * kS1F;              // [kS2Slots][kS2F]: produced
    float *win = s2ring + kS2Slots * kS2F;           // [kWinSlots][kWinF]
    static_assert(NCH % kR == 0 && NCH % kWinSlots == 0 && NCH % kS2Slots == 0, "static ring slots");
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int stride = gridDim.x;
    const int my_tiles = (a.nblk - (int)blockIdx.x + stride - 1) / stride;
    // barriers of every wave: B_pre, B_0 and one at the end of every ring step
#ifdef PWC_PIPE_PROD4
    if (wave == kND)                 in1_loader_wave<NCH>(a, s1ring, lane, my_tiles);
    else if (wave == kWaveWin0)      win_loader_wave<NCH, 0>(a, win, lane, my_tiles);
    else if (wave == kWaveWin0 + 1)  win_loader_wave<NCH, 1>(a, win, lane, my_tiles);
#else
    if (wave == kWaveWin0)           win_loader_wave<NCH, 0>(a, win, lane, my_tiles);
    else if (wave == kWaveWin0 + 1)  win_in1_loader_wave<NCH>(a, s1ring, win, lane, my_tiles);
#endif
    else if (wave >= kWaveProd0)     producer_wave<NCH>(a, s2ring, win, wave - kWaveProd0, lane, my_tiles);
    else                             warp_fma_wave<NCH>(a, s1ring, s2ring, wave, lane, my_tiles);
}

// =====================================================================================================================
template <int NCH>
#ifdef PWC_PIPE_LOADERS4
__global__ void __launch_bounds__(kThreadsPlain, 4)
#else
__global__ void __launch_bounds__(kThreadsPlain, 3)
#endif
corr81_pipe_kernel(PipeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s1ring = smem;                            // [kR][kS1F]
    float *s2ring = s1ring + kR * kS1F;              // [kR][kS2F]
    static_assert(NCH % kR == 0, "static ring slots");

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int stride = gridDim.x;
    const int my_tiles = (a.nblk - (int)blockIdx.x + stride - 1) / stride;
    // barriers of every wave: B_0 and one at the end of every ring step of its tiles

#ifdef PWC_PIPE_LOADERS4
    if (wave == kWaveLoad0)          loader_wave<NCH, 0, 4>(a, s1ring, s2ring, lane, my_tiles);
    else if (wave == kWaveLoad0 + 1) loader_wave<NCH, 4, 8>(a, s1ring, s2ring, lane, my_tiles);
    else if (wave == kWaveLoad0 + 2) loader_wave<NCH, 8, 12>(a, s1ring, s2ring, lane, my_tiles);
    else if (wave == kWaveLoad0 + 3) loader_wave<NCH, 12, kDmaI>(a, s1ring, s2ring, lane, my_tiles);
#else
    if (wave == kWaveLoad0)          loader_wave<NCH, 0, kLoadSplit>(a, s1ring, s2ring, lane, my_tiles);
    else if (wave == kWaveLoad0 + 1) loader_wave<NCH, kLoadSplit, kDmaI>(a, s1ring, s2ring, lane, my_tiles);
#endif
    else                             fma_wave<NCH, 2>(a, s1ring, s2ring, wave, lane, my_tiles);
}

pwc::LdsAttrOnce g_lds_plain8, g_lds_plain16, g_lds_roll, g_lds_warp8, g_lds_warp16;

}  // namespace

namespace pwc {

bool corr81_pipe_enabled() { return option(OPT_CORR_PIPE) != 0; }

bool corr81_pipe_fits(int B, int C, int H, int W) {
    const int64_t nblk = (int64_t)B * ((W + kTW - 1) / kTW) * ((H + kTH - 1) / kTH);
    const int nch = (C + kCK - 1) / kCK;
    return (nch == 8 || nch == 16) && nblk >= option(OPT_CORR_PIPE_MIN_TILES) && nblk <= 0x7fffffffLL &&
           (int64_t)H * W * 81 * 4 < 0x7fffffffLL;
}

// the fused window kernel: every launch the fused entry gets with C in (28, 32] or (60, 64] -- levels 2 and 3 (option
// "warpcorr_window": 2 = both, 1 = C <= 32 only, 0 = the round-2 kernel).  It wins at every tile count the engine sends there
// (pwc_warp_corr81_preferred keeps launches of <= 48 tiles on warp + small-map correlation): level 2 / level 3 of 2, 4, 8, 12, 16, 32
// pairs 18.0 / 22.6, 29.2 / 23.0, 51.0 / 24.6, 73.8 / 40.4, 93 / 43, 180 / 83.5 us against 20.0 / 29.5, 31.6 / 29.6, 57.0 / 29.8,
// 96.9 / 44.4, 128 / 47, 246 / 104 for the round-2 kernel (profiles/r04_corr_notes.md, section 8), so "corr_pipe_min_tiles" -- the
// threshold of the plain ring / rolling kernels -- does not apply to it.  The tile_of() reciprocals need nblk * tiles < 2^32.
bool warp_corr81_pipe_fits(int B, int C, int H, int W) {
    const int nch = (C + kCK - 1) / kCK, mode = option(OPT_WARPCORR_WINDOW);
    const int64_t tiles_x = (W + kTW - 1) / kTW, tiles_y = (H + kTH - 1) / kTH, nblk = (int64_t)B * tiles_x * tiles_y;
    return mode > 0 && (nch == 8 || (nch == 16 && mode >= 2)) && W >= 2 && nblk >= 1 && nblk <= 0x7fffffffLL &&
           nblk * tiles_x < (1ll << 32) && nblk * tiles_y < (1ll << 32) && (int64_t)H * W * 81 * 4 < 0x7fffffffLL;
}

static bool fill_magic(PipeArgs &a) {
    if (a.tiles_x < 1 || a.tiles_y < 1 || (int64_t)a.nblk * a.tiles_x >= (1ll << 32) || (int64_t)a.nblk * a.tiles_y >= (1ll << 32)) return false;
    a.magic_tx = a.tiles_x == 1 ? 0u : (unsigned)(((1ull << 32) + a.tiles_x - 1) / a.tiles_x);       // (1: not used)
    a.magic_ty = a.tiles_y == 1 ? 0u : (unsigned)(((1ull << 32) + a.tiles_y - 1) / a.tiles_y);
    return true;
}

int launch_corr81_pipe(const float *in1, const float *in2, float *out, int B, int C, int H, int W,
                       int64_t bs1, int64_t bs2, int64_t bso, float scale, float slope, int do_leaky, hipStream_t st) {
    const int tiles_x = (W + kTW - 1) / kTW, tiles_y = (H + kTH - 1) / kTH;
    const int nblk = B * tiles_x * tiles_y;
    const int nch = (C + kCK - 1) / kCK;
    const int grid = nblk < 256 ? nblk : 256;       // one workgroup per CU (its LDS does not admit two)
    PipeArgs a{in1, in2, out, C, H, W, tiles_x, tiles_y, nblk, bs1, bs2, bso, scale, slope, do_leaky, 0, 0, 0, nullptr, 0, 0.f, 0.f, 0, 0u, 0u};
    if (!fill_magic(a)) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_corr_fwd: too many tiles for the pipelined kernels");
    if (nch == 8 && option(OPT_CORR_ROLL)) {
        // runs of seg_len tiles down a column: as long as possible while the runs still cover the chip about once
        int seg_len = nblk / 256;
        if (seg_len < 1) seg_len = 1;
        if (seg_len > tiles_y) seg_len = tiles_y;
        a.seg_len = seg_len;
        a.segs_per_strip = (tiles_y + seg_len - 1) / seg_len;
        a.nseg = B * tiles_x * a.segs_per_strip;
        const int groll = a.nseg < 256 ? a.nseg : 256;
        int rc = ensure_lds_attr(g_lds_roll, reinterpret_cast<const void *>(corr81_roll_kernel), kLdsRoll, "corr81_roll_kernel");
        if (rc != PWC_OK) return rc;
        hipLaunchKernelGGL(corr81_roll_kernel, dim3((unsigned)groll), dim3(kThreadsRoll), kLdsRoll, st, a);
        return check_launch("corr81_roll_kernel");
    }
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

int launch_warp_corr81_pipe(const float *in1, const float *x2, const float *flo, float *out, int B, int C, int H, int W,
                            int64_t bs1, int64_t bs2, int64_t bsf, int64_t bso, float flow_scale, int align_corners, float thr,
                            float scale, float slope, int do_leaky, hipStream_t st) {
    const int tiles_x = (W + kTW - 1) / kTW, tiles_y = (H + kTH - 1) / kTH;
    const int nblk = B * tiles_x * tiles_y;
    const int nch = (C + kCK - 1) / kCK;
    const int grid = nblk < 256 ? nblk : 256;
    PipeArgs a{in1, x2, out, C, H, W, tiles_x, tiles_y, nblk, bs1, bs2, bso, scale, slope, do_leaky, 0, 0, 0,
               flo, bsf, flow_scale, thr, align_corners, 0u, 0u};
    if (!fill_magic(a)) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_warp_corr81_fwd: too many tiles for the window kernel");
    if (nch == 8) {
        int rc = ensure_lds_attr(g_lds_warp8, reinterpret_cast<const void *>(warp_corr81_pipe_kernel<8>), kLdsWarp, "warp_corr81_pipe_kernel<8>");
        if (rc != PWC_OK) return rc;
        hipLaunchKernelGGL(warp_corr81_pipe_kernel<8>, dim3((unsigned)grid), dim3(kThreadsWarp), kLdsWarp, st, a);
    } else {
        int rc = ensure_lds_attr(g_lds_warp16, reinterpret_cast<const void *>(warp_corr81_pipe_kernel<16>), kLdsWarp, "warp_corr81_pipe_kernel<16>");
        if (rc != PWC_OK) return rc;
        hipLaunchKernelGGL(warp_corr81_pipe_kernel<16>, dim3((unsigned)grid), dim3(kThreadsWarp), kLdsWarp, st, a);
    }
    return check_launch("warp_corr81_pipe_kernel");
}

}  // namespace pwc

// timing-experiment mask this translation unit was built with (0 in the product; pwc_experiment_mask)
namespace pwc { int exp_mask_corr_pipe() { return PWC_PIPE_EXP; } }
