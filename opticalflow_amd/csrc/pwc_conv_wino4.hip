// 3x3 / stride 1 convolution by Winograd F(4x4,3x3) on the gfx950 matrix cores, fp32 throughout (round 3).
// Same operator as pwc_conv_wino.hip (nn.Conv2d(3x3, padding = dilation) + LeakyReLU, reference models/PWCNet.py:26-33) with
// 36 multiplications per 4x4 outputs instead of 64 (F(2x2)) or 144 (direct): 1.78x fewer MFMA passes than F(2x2,3x3).
//
//     Y(4x4) = At [ sum_cin (G g Gt) (.) (Bt d B) ] A        d = 6x6 input patch, g = 3x3 filter, points 0, +-3/4, +-3/2, inf
//
// The price is accuracy: the Winograd-domain products are large and mostly cancel in At M A, so the fp32 accumulation over Cin
// leaves ~3x (rms) / ~5x (max) the error of F(2x2) against an fp64 convolution (with the textbook points 0, +-1, +-2 it is 6x / 17x).
// The route is therefore GATED: pwc_conv3x3_wino4_preferred() only says yes for the large level-2 / level-3 layers, the per-layer
// error stays inside the tests' 3e-6*sqrt(9 Cin) bound with margin, and the whole-forward EPE against the oracle is asserted at 1e-4.
//
// Design -- the B operand never passes through LDS: a lane makes it from its own patch.
//   v_mfma_f32_16x16x4_f32: A = U_p[16 couts][4 cin], B = V_p[4 cin][16 tiles], D = 16 couts x 16 tiles = 4 accumulator registers
//   per lane and position; a wave owns two 16-cout blocks x one 16-tile group x 18 of the 36 positions = 144 accumulator registers.
//   Lane (n = lane % 16, k = lane / 16): channel k of the 4-channel chunk, tile n of the group.
//     B operand: the lane reads ITS 6x6 patch of channel k (6 x 2 ds_read_b128) and makes the 18 values of Bt d B its wave needs
//                itself -- rows first (two patch rows per phase, one chunk ahead), then one column per phase right before the
//                twelve MFMAs that consume it (each value feeds both cout blocks);
//     A operand: U in LDS as [position / 4][cin][cout][position % 4]: four positions per ds_read_b128, conflict-free;
//     epilogue : At M A for the lane's couts x 1 tile in registers (the two waves of a pair exchange half of their partial sums
//                through LDS once), bias, LeakyReLU, four 16-byte stores per cout.
//   Workgroup = 8 waves = 4 pairs = CB/2 32-cout groups x TG tile groups (<4,2>: 64 couts x 8 rows x 64 columns; <2,4>: 32 couts x
//   16 x 64); a tile group is one row of sixteen 4x4 tiles.  Cin runs in chunks of 4 channels through LDS rings of three slots filled by
//   LDS-DMA in 16-byte pieces: raw input tile [4][4 TG + 2 rows][72] (global columns ox0 - 4 .. ox0 + 67; the range check
//   supplies padding, edges and the ragged last chunk), U [9][4][16 CB][4].  Dilation 1 only.  One barrier per chunk:
//     iteration k:  wait {raw(k+1), U(k)} | 36 MFMAs per wave on U(k), V(k) | issue raw(k+3), U(k+2) | rows of raw(k+1) -> w(k+1)
#include <stdlib.h>

#include <type_traits>

#include "pwc_common.h"

#ifndef PWC_W4_EXP
#define PWC_W4_EXP 0          // timing experiments, tools/wino4_exp.sh (results invalid): 1 = every U fetch reads chunk 0, 2 = every raw fetch reads
#endif                        // chunk 0, 4 = no LDS-DMA inside the loop, 8 = no input transforms, 16 = no barrier / counted wait, 32 = no patch-row reads

namespace {

using pwc::leaky;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 512;
constexpr int kCK = 4;               // input channels per chunk = K of one MFMA
// A tile group is sixteen 4x4 tiles: one row of 16 (GW = 64 output columns x 4 rows) or, for narrow maps (the 14x32 lattice images of
// dc_conv4), two rows of 8 (GW = 32 columns x 8 rows).  A staged row holds global columns ox0 - 4 .. ox0 + GW + 3 as 16-byte pieces and
// lands ONE float to the right of a 16-byte LDS boundary, so that global column ox0 - 1 (the first patch column) sits at index 4:
// every patch row is 16-byte aligned.
constexpr unsigned kOOB = 0x80000000u;

template <int CB, int TG, int GW = 64>
struct Geo4 {
    static_assert(CB * TG == 8, "eight waves");
    static_assert(GW == 64 || GW == 32, "tile-group width");
    static constexpr int kGW = GW;
    static constexpr int kTC = GW / 4;                                // tiles per tile row of a group (16 or 8)
    static constexpr int kGH = 4 * (16 / kTC);                        // output rows of a group (4 or 8)
    static constexpr int kRowP = (GW + 8) / 4;                        // 16-byte pieces per staged row (18 or 10)
    static constexpr int kRawW = 4 * kRowP;                           // 72 or 40 floats
    static constexpr int kRows = kGH * TG + 2;
    static constexpr int kPP = (kRows * kRowP + 15) / 16 * 16;        // pieces per staged channel, padded: planes are multiples of 64 floats
    static constexpr int kPlane = 4 * kPP;
    static constexpr int kRawPieces = kCK * kPP;
    static constexpr int kRawFloats = kCK * kPlane + 4;               // + the one-float shift (kept 16-byte granular)
    static constexpr int kCoutT = 16 * CB;
    static constexpr int kUFloats = 36 * kCK * kCoutT;                // [9][4 cin][cout][4 positions]
    static constexpr int kUPieces = kUFloats / 4;                     // 16-byte pieces
    static constexpr int kSlot = kRawFloats + kUFloats;
    static constexpr int kSmemBytes = 3 * kSlot * 4;
    // LDS-DMA instructions per thread and chunk; the last one of each stream is issued by the first kTail / 64 waves only
    static constexpr int kRS = (kRawPieces + kThreads - 1) / kThreads, kRawTail = kRawPieces % kThreads;
    static constexpr int kUS = (kUPieces + kThreads - 1) / kThreads, kUTail = kUPieces % kThreads;
    static_assert(kRawTail % 64 == 0 && kUTail % 64 == 0, "tails are whole waves");
    static constexpr int kRawWaves = kRawTail ? kRawTail / 64 : 8, kUWaves = kUTail ? kUTail / 64 : 8;   // waves that issue the last instruction
    static_assert(kSmemBytes <= 160 * 1024, "LDS");
    static_assert(kThreads % kCoutT == 0, "one U offset register: instruction j is j * (kThreads / kCoutT) rows further");
};

// Interpolation points 0, +-a, +-b, inf with a = 3/4, b = 3/2 instead of the textbook 0, +-1, +-2: the same operation count, every
// constant exact in binary, and -- measured in fp32 on 565 input channels of unit-scale data -- 2.2x less rms / 4.4x less maximum
// error (the error of F(4,3) is dominated by the fp32 accumulation of large, mostly cancelling Winograd-domain products; these
// points balance the magnitudes of the 36 positions; a scan over (a, b) is recorded in profiles/r03_wino4_notes.md).
constexpr float kA = 0.75f, kB = 1.5f;
constexpr float kA2 = kA * kA, kB2 = kB * kB, kA2B2 = kA2 * kB2, kS2 = kA2 + kB2, kA3 = kA2 * kA, kB3 = kB2 * kB;

// 1-D input transform, rows of Bt = coefficients of prod_{k != j} (x - a_k) (12 operations):
//   t0 = a2b2 d0 - (a2 + b2) d2 + d4                      t5 = a2b2 d1 - (a2 + b2) d3 + d5
//   t1, t2 = (d4 - b2 d2) +- a (d3 - b2 d1)               t3, t4 = (d4 - a2 d2) +- b (d3 - a2 d1)
__device__ __forceinline__ void bt6(const float (&d)[6], float (&t)[6]) {
#if defined(PWC_W4_EXP) && (PWC_W4_EXP & 8)
    for (int i = 0; i < 6; ++i) t[i] = d[i];
    return;
#endif
    t[0] = __builtin_fmaf(kA2B2, d[0], __builtin_fmaf(-kS2, d[2], d[4]));
    const float p = __builtin_fmaf(-kB2, d[2], d[4]), q = __builtin_fmaf(-kB2, d[1], d[3]);
    t[1] = __builtin_fmaf(kA, q, p);
    t[2] = __builtin_fmaf(-kA, q, p);
    const float r = __builtin_fmaf(-kA2, d[2], d[4]), s = __builtin_fmaf(-kA2, d[1], d[3]);
    t[3] = __builtin_fmaf(kB, s, r);
    t[4] = __builtin_fmaf(-kB, s, r);
    t[5] = __builtin_fmaf(kA2B2, d[1], __builtin_fmaf(-kS2, d[3], d[5]));
}

// The optimiser sinks pure arithmetic towards its first use: without this pin the six row transforms of an iteration (whose results
// are consumed one iteration later) all ended up in ONE MFMA shadow of the next iteration -- 72 VALU operations in a row and 36
// patch values kept alive across the barrier (spills).  An empty volatile asm that "modifies" the values keeps them where they are.
__device__ __forceinline__ void pin6(float (&t)[6]) {
    asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]));
}

// 1-D output transform, At[i][j] = a_j^i:  o0 = m0 + (m1 + m2) + (m3 + m4);  o1 = a (m1 - m2) + b (m3 - m4);
//   o2 = a2 (m1 + m2) + b2 (m3 + m4);  o3 = a3 (m1 - m2) + b3 (m3 - m4) + m5
__device__ __forceinline__ void at6(const float (&m)[6], float (&o)[4]) {
    const float s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
    o[0] = (m[0] + s1) + s2;
    o[1] = __builtin_fmaf(kB, d2, kA * d1);
    o[2] = __builtin_fmaf(kB2, s2, kA2 * s1);
    o[3] = __builtin_fmaf(kB3, d2, kA3 * d1) + m[5];
}

// U[chunk][g = p / 4][k][co][p % 4] <- (G g Gt)[i][j] of w[co][cin = 4 chunk + k], position p = 6 j + i (column-major: the kernel
// consumes one column j of the 6x6 position grid per phase); computed in double, stored as float; zero outside Cin / Cout
__global__ void __launch_bounds__(256)
wino4_pack_kernel(const float *__restrict__ w, float *__restrict__ up, int Cin, int Cout, int CoutP, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int pl = (int)(idx & 3);
    int64_t t = idx >> 2;
    const int co = (int)(t % CoutP);
    t /= CoutP;
    const int k = (int)(t & 3);
    t >>= 2;
    const int g = (int)(t % 9);
    const int chunk = (int)(t / 9);
    const int p = 4 * g + pl, i = p % 6, j = p / 6;
    const int cin = chunk * kCK + k;
    double v = 0.0;
    if (co < Cout && cin < Cin) {
        // G[j] = (1, a_j, a_j^2) / prod_{k != j} (a_j - a_k) for the finite points 0, +-a, +-b; (0, 0, 1) for the point at infinity
        const double a = kA, bb = kB, na = 2 * a * a * (a * a - bb * bb), nb = 2 * bb * bb * (bb * bb - a * a);
        const double G[6][3] = {{1.0 / (a * a * bb * bb), 0.0, 0.0}, {1.0 / na, a / na, a * a / na}, {1.0 / na, -a / na, a * a / na},
                                {1.0 / nb, bb / nb, bb * bb / nb}, {1.0 / nb, -bb / nb, bb * bb / nb}, {0.0, 0.0, 1.0}};
        const float *gw = w + ((int64_t)co * Cin + cin) * 9;
        for (int a = 0; a < 3; ++a) {
            double r = 0.0;
            for (int b = 0; b < 3; ++b) r += (double)gw[a * 3 + b] * G[j][b];
            v += G[i][a] * r;
        }
    }
    up[idx] = (float)v;
}

#define PWC_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")

// Split of a launch's LAST, partial round of workgroups along the input channels (see launch_wino4).  ksplit <= 1: an ordinary launch
// over the given tile positions of every image.
struct TailSplit {
    int per_image;    // tiles of ONE image that belong to this launch: positions pos0 .. pos0 + per_image - 1 of its tiles_x * tiles_y tiles --
    int pos0;         // the same positions of every image, so that an item's result does not depend on its slot in the batch
    int ksplit;       // slices per tile (1: none)
    int cps;          // 4-channel chunks per slice
    float *ws;        // [ksplit][tiles of the launch][Cout][tile rows][tile columns]
};

// ---- the kernel: pair-split positions ------------------------------------------------------------------------------------------------
// The first version gave every wave 16 couts x 16 tiles x all 36 positions (no exchange at all), so the FOUR waves that share a tile
// group each made the same 36 B operands: 144 VALU operations per 36 MFMAs of 32 cycles, the largest single cost (20 % of its time;
// ablations in profiles/r03_wino4_notes.md; dc_conv1 1.78 ms).  Here a wave owns 32 couts (two 16-cout blocks) x 16 tiles x 18
// positions -- the three columns j = 3 hj .. 3 hj + 2 of the 6x6 position grid, all rows -- so one B operand feeds TWO MFMAs and the wave
// only transforms what its three columns need: 6 operations per patch row instead of 12, three column transforms instead of six =
// 72 VALU per 36 MFMAs, and the row-transformed patch is 18 registers instead of 36 (dc_conv1 1.61 ms).  The two waves of a pair
// (hj = 0, 1) hold complementary halves of sum_j T[.][j] At[.][j]: at the end each sends the partial 4x4 outputs of the cout block it
// does NOT finish through LDS (64 floats per lane) and finishes the other.
template <int HJ>
__device__ __forceinline__ void bt3_row(const float (&d)[6], float (&t)[3]) {      // outputs 3 HJ .. 3 HJ + 2 of bt6: 6 operations
#if defined(PWC_W4_EXP) && (PWC_W4_EXP & 8)
    t[0] = d[0]; t[1] = d[1]; t[2] = d[2];
    return;
#endif
    if constexpr (HJ == 0) {
        t[0] = __builtin_fmaf(kA2B2, d[0], __builtin_fmaf(-kS2, d[2], d[4]));
        const float p = __builtin_fmaf(-kB2, d[2], d[4]), q = __builtin_fmaf(-kB2, d[1], d[3]);
        t[1] = __builtin_fmaf(kA, q, p);
        t[2] = __builtin_fmaf(-kA, q, p);
    } else {
        const float r = __builtin_fmaf(-kA2, d[2], d[4]), s = __builtin_fmaf(-kA2, d[1], d[3]);
        t[0] = __builtin_fmaf(kB, s, r);
        t[1] = __builtin_fmaf(-kB, s, r);
        t[2] = __builtin_fmaf(kA2B2, d[1], __builtin_fmaf(-kS2, d[3], d[5]));
    }
    asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]));
}

template <int CB, int TG, int GW, int HJ>
__device__ __forceinline__ void wino4p_body(const float *__restrict__ x, const float *__restrict__ up, const float *__restrict__ bias,
                                            float *__restrict__ y, int Cin_all, int H, int W, int Cout, int CoutP, int tiles_x, int tiles_y,
                                            int64_t bsx, int64_t bsy, float slope, int do_leaky, int co0, int nblk, int ngroups, int split2,
                                            const TailSplit ts) {
    using G = Geo4<CB, TG, GW>;
    constexpr int kRowP = G::kRowP, kRawW = G::kRawW;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // 3 x [raw | U]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int n = lane & 15;                  // tile of the group (B / D column)
    const int kq = lane >> 4;                 // channel of the chunk (A / B k index); D rows 4 kq .. 4 kq + 3
    const int pr = wave >> 1;                           // pair = (32-cout group, tile group); the position half hj = wave & 1 is the template parameter HJ
    const int cg2 = pr % (CB / 2), tgi = pr / (CB / 2);

    // One-dimensional grid of nblk tiles x ngroups cout groups.  Workgroups i, i+8, ... share an XCD: each XCD gets a contiguous run of
    // tiles (halo re-reads hit its L2) and runs the cout groups of one tile BACK TO BACK, so that the second group finds the input
    // tile in that L2 instead of fetching it again (PMC: 2.17x -> 1.4x the algorithmic bytes on dc_conv1).
    // Tail launch (ts.ksplit > 1, see launch_wino4): the tiles of the last, partial round of workgroups, each cut into ksplit slices
    // of the input channels; a slice leaves its raw partial outputs in the workspace, wino4_tail_reduce_kernel finishes them.
    int bid, grp, kz = 0;
    {
        int id = (int)blockIdx.x;
        if (ts.ksplit > 1) {
            kz = id / (nblk * ngroups);                   // slice-major: the slices of a tile land on the same XCD
            id -= kz * (nblk * ngroups);
        }
        if ((nblk & 7) == 0) {
            const int slot = id >> 3;
            grp = slot % ngroups;
            bid = (id & 7) * (nblk >> 3) + slot / ngroups;
        } else {
            grp = id / nblk;
            bid = id % nblk;
        }
    }
    const int tail_tile = bid;                                  // index inside this launch (the workspace is laid out by it)
    const int b = bid / ts.per_image;
    const int pos = ts.pos0 + bid % ts.per_image;
    const int tx = pos % tiles_x;
    const int ty = pos / tiles_x;
    const int cb0 = co0 + grp * G::kCoutT;
    const int ox0 = tx * G::kGW;
    const int oy0 = ty * (G::kGH * TG);
    const int plane = H * W;

    // ---- per-lane LDS-DMA source offsets -------------------------------------------------------------
    // waves are numbered so that the raw and the U tails are issued by the first kRawWaves / kUWaves waves
    unsigned raw_off[G::kRS];
#pragma unroll
    for (int j = 0; j < G::kRS; ++j) {
        const int i = j * kThreads + tid;                 // piece (channel c, row, q) of the padded [4][kPP] image
        const int c = i / G::kPP;
        const int rem = i % G::kPP;
        const int row = rem / kRowP, q = rem % kRowP;
        const int iy = oy0 - 1 + row;
        const int ix = ox0 - 4 + 4 * q;                   // W % 4 == 0: a piece is all-in or all-out
        const bool ok = (i < G::kRawPieces) && (row < G::kRows) && (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);
        raw_off[j] = ok ? (unsigned)(c * plane + iy * W + ix) * 4u : kOOB;
    }
    // U piece p = j * kThreads + tid = (row r = p / CoutT of the [36][CoutT] image, cout c): instruction j reads rows kThreads / CoutT
    // further down -> one per-lane offset + a wave-uniform byte offset per instruction
    const unsigned u_off = (unsigned)((tid / G::kCoutT) * CoutP + tid % G::kCoutT) * 16u;
    const unsigned u_step = (unsigned)(kThreads / G::kCoutT) * (unsigned)CoutP * 16u;

    const int64_t uchunk = (int64_t)36 * kCK * CoutP;                 // floats per chunk of the packed image
    // this workgroup's range of 4-channel chunks: everything, or slice kz of the tail launch
    const int k_lo = kz * ts.cps, k_hi = ts.ksplit > 1 ? min((Cin_all + kCK - 1) / kCK, k_lo + ts.cps) : (Cin_all + kCK - 1) / kCK;
    const int Cin = min(Cin_all, k_hi * kCK) - k_lo * kCK;
    const int nchunks = k_hi - k_lo;
    const float *xb = x + (int64_t)b * bsx + (int64_t)k_lo * kCK * plane;
    const float *ug = up + (int64_t)cb0 * 4 + (int64_t)k_lo * uchunk;
    const int ubytes = (int)(uchunk - (int64_t)cb0 * 4) * 4;
    const unsigned lds0 = pwc::lds_addr(smem);

    auto issue_raw = [&](int chunk) {                                 // raw(chunk) -> slot chunk % 3, one float to the right
        // a chunk past the end gets a zero-sized descriptor: every lane fails the range check, nothing is fetched, zeros land in a slot
        // nobody reads -- so EVERY iteration issues a whole group and the counted waits never change
        const int c0 = (PWC_W4_EXP & 2) ? 0 : chunk * kCK;
        const pwc::v4i32 rs = pwc::make_rsrc(xb + (int64_t)min(c0, Cin - 1) * plane, max(0, min(kCK, Cin - c0)) * plane * 4);
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (chunk % 3) * G::kSlot * 4 + 4 + wave * 1024);
#pragma unroll
        for (int j = 0; j < G::kRS; ++j)
            if (j < G::kRS - 1 || wave < G::kRawWaves) pwc::dma_b128(rs, base + j * kThreads * 16, raw_off[j]);
    };
    auto issue_u = [&](int chunk) {                                   // U(chunk) -> slot chunk % 3
        const pwc::v4i32 rs = pwc::make_rsrc(ug + ((PWC_W4_EXP & 1) ? 0 : (int64_t)min(chunk, nchunks - 1) * uchunk), chunk < nchunks ? ubytes : 0);
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + ((chunk % 3) * G::kSlot + G::kRawFloats) * 4 + wave * 1024);
#pragma unroll
        for (int j = 0; j < G::kUS; ++j)
            if (j < G::kUS - 1 || wave < G::kUWaves) pwc::dma_b128_so(rs, base + j * kThreads * 16, u_off, j * u_step);
    };
    // the same, one instruction at a time (PWC_W4_DMASPREAD: one LDS-DMA per MFMA shadow instead of all seven in two)
    pwc::v4i32 rs_r, rs_u;
    unsigned base_r = 0, base_u = 0;
    auto setup_dma = [&](int k) {
        const int c0 = (k + 3) * kCK;
        rs_r = pwc::make_rsrc(xb + (int64_t)min(c0, Cin - 1) * plane, max(0, min(kCK, Cin - c0)) * plane * 4);
        base_r = __builtin_amdgcn_readfirstlane(lds0 + ((k + 3) % 3) * G::kSlot * 4 + 4 + wave * 1024);
        rs_u = pwc::make_rsrc(ug + (int64_t)min(k + 2, nchunks - 1) * uchunk, k + 2 < nchunks ? ubytes : 0);
        base_u = __builtin_amdgcn_readfirstlane(lds0 + (((k + 2) % 3) * G::kSlot + G::kRawFloats) * 4 + wave * 1024);
    };
    auto dma_piece = [&](int t) {                        // t = 0 .. kRS + kUS - 1 in issue order (raw first)
        if (t < G::kRS) {
            if (t < G::kRS - 1 || wave < G::kRawWaves) pwc::dma_b128(rs_r, base_r + t * kThreads * 16, raw_off[t]);
        } else {
            const int j = t - G::kRS;
            if (j < G::kUS - 1 || wave < G::kUWaves) pwc::dma_b128_so(rs_u, base_u + j * kThreads * 16, u_off, j * u_step);
        }
    };
    // VMEM operations of one group {raw(m + 1), U(m)} issued by THIS wave (wave-uniform; the counted waits need immediates)
    const int n_grp = G::kRS - (wave < G::kRawWaves ? 0 : 1) + G::kUS - (wave < G::kUWaves ? 0 : 1);
    constexpr int kN0 = G::kRS + G::kUS;
    auto wait_groups1 = [&]() {                          // at most one group outstanding
        if (n_grp == kN0) PWC_WAIT_VMCNT(kN0); else if (n_grp == kN0 - 1) PWC_WAIT_VMCNT(kN0 - 1); else PWC_WAIT_VMCNT(kN0 - 2);
    };
    auto wait_groups2 = [&]() {                          // at most two
        if (n_grp == kN0) PWC_WAIT_VMCNT(2 * kN0); else if (n_grp == kN0 - 1) PWC_WAIT_VMCNT(2 * kN0 - 2); else PWC_WAIT_VMCNT(2 * kN0 - 4);
    };


    // this lane's tile inside its group: tile row n / kTC, tile column n % kTC; its patch inside a raw slot: channel kq, rows
    // GH tgi + 4 tr .. + 5, columns 4 tc .. 4 tc + 5 (+ 4: the row lands one float to the right of a 16-byte boundary)
    const int poff = kq * G::kPlane + (G::kGH * tgi + 4 * (n / G::kTC)) * kRawW + 4 + 4 * (n % G::kTC);
    // this lane's U column inside a slot's U image: [g][k = kq][cout = 32 cg2 + 16 cbl + n][4]
    const int uoff = G::kRawFloats + (kq * G::kCoutT + 32 * cg2 + n) * 4;
    constexpr int kUG = kCK * G::kCoutT * 4;              // floats per position group g
    typedef const __attribute__((address_space(3))) f32x4 *lds4_t;
    auto lds_read4 = [](unsigned base_bytes, int off_bytes) { return *reinterpret_cast<lds4_t>((uintptr_t)(base_bytes + off_bytes)); };
    auto load_row = [&](unsigned base_bytes, int a, float (&d)[6]) {
        const f32x4 v = lds_read4(base_bytes, a * kRawW * 4);
        const f32x4 u = lds_read4(base_bytes, a * kRawW * 4 + 16);
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3]; d[4] = u[0]; d[5] = u[1];
    };
    const unsigned poff_b = lds0 + (unsigned)poff * 4u, uoff_b = lds0 + (unsigned)uoff * 4u;

    f32x4 acc[2][18];                         // [cout block][local position 6 jj + i]
    float wA[6][3], wB[6][3];                 // row-transformed patch, this wave's three columns; ping-pong over chunks
    float v0[6], v1[6];                       // V columns, alternating over the SEQUENCE of columns (three per chunk: the parity flips per chunk)
    f32x4 uc[2][2];                           // U groups of column jj = 2 (both cout blocks), carried into the next iteration's first phase
#pragma unroll
    for (int i = 0; i < 6; ++i) v0[i] = v1[i] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) uc[c][0] = uc[c][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // U groups of local column jj: global positions 6 (3 HJ + jj) .. + 5 lie in exactly two groups of four
    auto g0_of = [](int jj) { return (6 * (3 * HJ + jj)) >> 2; };
#define PWC_W4P_MFMA(CBL, Q, UG, VB) acc[CBL][Q] = __builtin_amdgcn_mfma_f32_16x16x4f32((UG), (VB), acc[CBL][Q], 0, 0, 0)

    // One iteration = three phases of twelve MFMAs (6 rows x 2 cout blocks), rotated by one phase: column 2 of chunk k - 1 first (its
    // operands are in registers behind the barrier), then columns 0, 1 of chunk k.  Phase ph also makes V column ph of this chunk and
    // the row transforms of patch rows 2 ph, 2 ph + 1 of the next chunk (24 VALU behind one MFMA), and reads the U groups of column ph.
    // PAR = parity of the chunk: column s of the running sequence of columns uses V buffer s & 1.
    // All seven pieces in two MFMA shadows (as the round-2 kernels do): dc_conv1 1611 us; one piece behind every MFMA 1522, every second 1495,
    // every fourth 1491 us
    // (tools/wino4_exp.sh; the issuing wave stalls 60-185 cycles per piece, and in one long stall the partner wave alone cannot
    // keep the pipe busy).
#ifndef PWC_W4_DMASTRIDE
#define PWC_W4_DMASTRIDE 4
#endif
    constexpr int kDmaStride = PWC_W4_DMASTRIDE;
    static_assert((G::kRS + G::kUS - 1) * kDmaStride + 1 < 36, "the pieces of a group fit one iteration");
    auto iteration = [&](int k, float (&wc)[6][3], float (&wn)[6][3], auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        if (!(PWC_W4_EXP & 16)) {
            wait_groups1();
            __syncthreads();
        }
        unsigned uslot = uoff_b + (unsigned)__builtin_amdgcn_readfirstlane((k % 3) * G::kSlot * 4);
        unsigned rnext = poff_b + (unsigned)__builtin_amdgcn_readfirstlane(((k + 1) % 3) * G::kSlot * 4);
        asm volatile("" : "+v"(uslot), "+v"(rnext));
        f32x4 un[2][2];                                         // U groups of the column the NEXT phase multiplies
        float da[6], db[6];
#pragma unroll
        for (int ph = 0; ph < 3; ++ph) {
            // this phase multiplies: ph 0 -> column 2 of chunk k - 1 (U in uc), ph 1 -> column 0, ph 2 -> column 1 (U in un of the previous phase)
            const int jj = (ph + 2) % 3;
            // V buffer parities along the running sequence of columns (chunk k's columns are elements 3k, 3k+1, 3k+2; 3k = k mod 2)
            const bool use_v1 = ph == 0 ? ((PAR + 1 + 2) & 1) : ((PAR + ph - 1) & 1);
            f32x4 (&ucur)[2][2] = uc;                           // loaded during the previous phase
            const int p0 = 6 * (3 * HJ + jj), gb = p0 >> 2;
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                const int cbl = m / 6, i = m % 6;               // cout block major: its two U groups die after six MFMAs
                const int p = p0 + i;
                if (use_v1) PWC_W4P_MFMA(cbl, 6 * jj + i, ucur[cbl][(p >> 2) - gb][p & 3], v1[i]);
                else        PWC_W4P_MFMA(cbl, 6 * jj + i, ucur[cbl][(p >> 2) - gb][p & 3], v0[i]);
                // ---- in the shadow of this MFMA ----------------------------------------------------------------------
                if (m == 0 && !(PWC_W4_EXP & 32)) { load_row(rnext, 2 * ph, da); load_row(rnext, 2 * ph + 1, db); }
                if (!(PWC_W4_EXP & 4)) {                        // LDS-DMA of {raw(k+3), U(k+2)}: ONE piece per kDmaStride MFMAs
                    if (ph == 0 && m == 0) setup_dma(k);
                    const int t = 12 * ph + m - 1;
                    if (t >= 0 && t % kDmaStride == 0 && t / kDmaStride < G::kRS + G::kUS) dma_piece(t / kDmaStride);
                }
                if (m == 4) {                                   // V column ph of THIS chunk (consumed by the next phase) + two row transforms
                    const float col[6] = {wc[0][ph], wc[1][ph], wc[2][ph], wc[3][ph], wc[4][ph], wc[5][ph]};
                    const bool dst_v1 = (PAR + ph) & 1;
                    if (dst_v1) { bt6(col, v1); pin6(v1); } else { bt6(col, v0); pin6(v0); }
                }
                if (m == 8) { bt3_row<HJ>(da, wn[2 * ph]); bt3_row<HJ>(db, wn[2 * ph + 1]); }
                // U groups of column ph of this chunk, multiplied by the next phase (phase 2's: by the next iteration's phase 0)
                if (m == 6 || m == 10) {
                    const int cb_ = m == 6 ? 0 : 1;
                    const int gn = g0_of(ph);
                    un[cb_][0] = lds_read4(uslot, gn * kUG * 4 + cb_ * 16 * 16);
                    un[cb_][1] = lds_read4(uslot, (gn + 1) * kUG * 4 + cb_ * 16 * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) { uc[c][0] = un[c][0]; uc[c][1] = un[c][1]; }
        }
    };

    // ---- prologue: raw(0) alone, then the groups G(0) = {raw(1), U(0)}, G(1) = {raw(2), U(1)}
    issue_raw(0);
    issue_raw(1);
    issue_u(0);
    issue_raw(2);
    issue_u(1);
    wait_groups2();
    __syncthreads();
    {
        float d[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            load_row(poff_b, a, d);
            bt3_row<HJ>(d, wA[a]);
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 18; ++q) acc[c][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);

    for (int k = 0; k < nchunks; k += 2) {
        iteration(k, wA, wB, std::integral_constant<int, 0>{});
        if (k + 1 < nchunks) iteration(k + 1, wB, wA, std::integral_constant<int, 1>{});
    }
    // column 2 of the last chunk: its V is in the buffer of sequence element 3 (nchunks - 1) + 2, its U groups in uc
    {
        const int p0 = 6 * (3 * HJ + 2), gb = p0 >> 2;
        const bool use_v1 = (nchunks - 1) & 1;                  // sequence element 3 (nchunks - 1) + 2
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            const int cbl = m / 6, i = m % 6, p = p0 + i;
            if (use_v1) PWC_W4P_MFMA(cbl, 12 + i, uc[cbl][(p >> 2) - gb][p & 3], v1[i]);
            else        PWC_W4P_MFMA(cbl, 12 + i, uc[cbl][(p >> 2) - gb][p & 3], v0[i]);
        }
    }
#undef PWC_W4P_MFMA

    // ---- output transform: partial sums over this wave's three columns for both cout blocks; the block 1 - HJ half goes to the partner
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    __syncthreads();                                            // every wave is done with the rings
    float *xsend = smem + (wave * 64) * 64 + lane_e;            // [wave][64 values][64 lanes]
    const float *xrecv = smem + ((wave ^ 1) * 64) * 64 + lane_e;
    float keep[4][4][4];                                        // [cout c][row pp][col q] partial outputs of the block this wave finishes
#pragma unroll
    for (int cbl = 0; cbl < 2; ++cbl) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float t[4][3];
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const float m[6] = {acc[cbl][6 * jj + 0][c], acc[cbl][6 * jj + 1][c], acc[cbl][6 * jj + 2][c],
                                    acc[cbl][6 * jj + 3][c], acc[cbl][6 * jj + 4][c], acc[cbl][6 * jj + 5][c]};
                float o[4];
                at6(m, o);
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) t[pp][jj] = o[pp];
            }
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                float o[4];
                if constexpr (HJ == 0) {                        // columns 0, 1, 2 = points 0, +a, -a
                    const float s = t[pp][1] + t[pp][2], d = t[pp][1] - t[pp][2];
                    o[0] = t[pp][0] + s; o[1] = kA * d; o[2] = kA2 * s; o[3] = kA3 * d;
                } else {                                        // columns 3, 4, 5 = points +b, -b, infinity
                    const float s = t[pp][0] + t[pp][1], d = t[pp][0] - t[pp][1];
                    o[0] = s; o[1] = kB * d; o[2] = kB2 * s; o[3] = __builtin_fmaf(kB3, d, t[pp][2]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (cbl == HJ) keep[c][pp][q] = o[q];
                    else xsend[((c * 4 + pp) * 4 + q) * 64] = o[q];
                }
            }
        }
    }
    __syncthreads();
    const int oyl = oy0 + G::kGH * tgi + 4 * ((lane_e & 15) / G::kTC), ox = ox0 + 4 * ((lane_e & 15) % G::kTC);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int co = cb0 + 32 * cg2 + 16 * HJ + 4 * (lane_e >> 4) + c;
        const float bv = ts.ksplit > 1 ? 0.f : bias[min(co, Cout - 1)];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o[q] = keep[c][pp][q] + xrecv[((c * 4 + pp) * 4 + q) * 64] + bv;
                if (do_leaky && ts.ksplit <= 1) o[q] = leaky(o[q], slope);
            }
            const int oy = oyl + pp;
            if (ts.ksplit > 1) {
                // raw partial sums of slice kz: workspace [slice][tile of the tail][cout of the layer][tile rows][tile columns]
                if (co < Cout)
                    *reinterpret_cast<f32x4 *>(ts.ws + ((((int64_t)kz * nblk + tail_tile) * Cout + co) * (G::kGH * TG) + (oy - oy0)) * G::kGW + (ox - ox0)) =
                        (f32x4){o[0], o[1], o[2], o[3]};
            } else if (co < Cout && oy < H && ox < W) {
                if (!split2) {
                    *reinterpret_cast<f32x4 *>(y + (int64_t)b * bsy + (int64_t)co * plane + (int64_t)oy * W + ox) = (f32x4){o[0], o[1], o[2], o[3]};
                } else {
                    float *d0 = y + ((int64_t)b * 4 + 2 * (oy & 1)) * bsy + (int64_t)co * (plane >> 2) + (int64_t)(oy >> 1) * (W >> 1) + (ox >> 1);
                    *reinterpret_cast<f32x2 *>(d0) = (f32x2){o[0], o[2]};
                    *reinterpret_cast<f32x2 *>(d0 + bsy) = (f32x2){o[1], o[3]};
                }
            }
        }
    }
}

template <int CB, int TG, int GW>
__global__ void __launch_bounds__(kThreads, 1)
conv3x3_wino4p_kernel(const float *__restrict__ x, const float *__restrict__ up, const float *__restrict__ bias,
                      float *__restrict__ y, int Cin, int H, int W, int Cout, int CoutP, int tiles_x, int tiles_y,
                      int64_t bsx, int64_t bsy, float slope, int do_leaky, int co0, int nblk, int ngroups, int split2, const TailSplit ts) {
    // the position half is wave-uniform: two specialisations of the body, every index inside is a compile-time constant
    if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) & 1)
        wino4p_body<CB, TG, GW, 1>(x, up, bias, y, Cin, H, W, Cout, CoutP, tiles_x, tiles_y, bsx, bsy, slope, do_leaky, co0, nblk, ngroups, split2, ts);
    else
        wino4p_body<CB, TG, GW, 0>(x, up, bias, y, Cin, H, W, Cout, CoutP, tiles_x, tiles_y, bsx, bsy, slope, do_leaky, co0, nblk, ngroups, split2, ts);
}

// Finishes the tiles of a tail launch: y = act(bias + sum over slices, in slice order (deterministic)) for couts co0 .. co0 + ncout - 1
// of the ntail tiles of the tail launch; one thread per four pixels of a tile row.
__global__ void __launch_bounds__(256)
wino4_tail_reduce_kernel(const float *__restrict__ ws, const float *__restrict__ bias, float *__restrict__ y, int ksplit, int per_image, int pos0,
                         int ntail, int tiles_x, int th, int tw, int H, int W, int Cout, int co0, int ncout, int64_t bsy, float slope,
                         int do_leaky, int split2, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int q4 = tw >> 2;
    const int xq = (int)(idx % q4);
    int64_t t = idx / q4;
    const int r = (int)(t % th);
    t /= th;
    const int co = co0 + (int)(t % ncout);
    const int tt = (int)(t / ncout);
    const int b = tt / per_image, pos = pos0 + tt % per_image;
    const int tx = pos % tiles_x, ty = pos / tiles_x;
    const int oy = ty * th + r, ox = tx * tw + 4 * xq;
    if (co >= Cout || oy >= H || ox >= W) return;
    const int64_t slice = (int64_t)ntail * Cout * th * tw;
    const float *p = ws + (((int64_t)tt * Cout + co) * th + r) * tw + 4 * xq;
    f32x4 v = *reinterpret_cast<const f32x4 *>(p);
    for (int z = 1; z < ksplit; ++z) v += *reinterpret_cast<const f32x4 *>(p + z * slice);
    const float bv = bias[co];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[i] += bv;
        if (do_leaky) v[i] = leaky(v[i], slope);
    }
    if (!split2) {
        *reinterpret_cast<f32x4 *>(y + (int64_t)b * bsy + ((int64_t)co * H + oy) * W + ox) = v;
    } else {            // PWC_CONV_SPLIT2: image b as its four pixel lattices (the main kernel's epilogue does the same)
        float *d0 = y + ((int64_t)b * 4 + 2 * (oy & 1)) * bsy + (int64_t)co * (((int64_t)H * W) >> 2) + (int64_t)(oy >> 1) * (W >> 1) + (ox >> 1);
        *reinterpret_cast<f32x2 *>(d0) = (f32x2){v[0], v[2]};
        *reinterpret_cast<f32x2 *>(d0 + bsy) = (f32x2){v[1], v[3]};
    }
}

// Inverse of L nested PWC_CONV_SPLIT2 stores (see pwc_hip.h): one thread per four output pixels of a row; the four come from four
// different lattice images (or two, alternating, for L = 1), the store is 16 bytes.
__global__ void __launch_bounds__(256)
lattice_unsplit_kernel(const float *__restrict__ x, float *__restrict__ y, int C, int h, int w, int L, int64_t bsy, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int W = w << L, H = h << L, W4 = W >> 2;
    const int xq = (int)(idx % W4);
    int64_t t = idx / W4;
    const int yy = (int)(t % H);
    t /= H;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int xx = 4 * xq + q;
        int n = b;
        for (int l = 0; l < L; ++l) n = n * 4 + 2 * ((yy >> l) & 1) + ((xx >> l) & 1);
        v[q] = x[(((int64_t)n * C + c) * h + (yy >> L)) * w + (xx >> L)];
    }
    *reinterpret_cast<f32x4 *>(y + (int64_t)b * bsy + ((int64_t)c * H + yy) * W + 4 * xq) = (f32x4){v[0], v[1], v[2], v[3]};
}

inline int cout_padded4(int Cout) { return (Cout + 31) / 32 * 32; }

// One workgroup per CU at a time, so a launch of n workgroups takes ceil(n / 256) rounds and its last round may leave most of the
// chip idle: 896 workgroups (the 64-cout layers of level 2) = 3.5 rounds cost 4, 448 = 1.75 cost 2 -- 12.5 % of those layers.  When the
// last round is partial (and at least one full round precedes it) its tiles go into a second launch that cuts every tile into
// two slices of the input channels, so that the slices fill the chip (128 tiles x 2 = 256 half-length workgroups); a small third
// kernel adds the slices in a fixed order.  Option "w4_tailsplit" = 0 (pwc_set_option / PWC_W4_TAILSPLIT) switches it off;
// "w4_smallsplit" = 0 the whole-launch form for launches smaller than the chip.
constexpr int kCUsDefault = 256;
// CUs of the current device (cached per device; 256 on MI355X in SPX mode -- a partitioned device reports fewer and the plans follow)
inline int device_cus() {
    static std::atomic<int> cached[pwc::kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= pwc::kMaxDevices) return kCUsDefault;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = kCUsDefault;
    cached[dev].store(n, std::memory_order_relaxed);
    return n;
}
struct TailPlan { int main_tiles, ksplit, cps; int64_t ws_bytes; };

// A workgroup costs ~10 us outside its K loop = the time of ~3.3 four-channel chunks (dc_conv1: 3 us per chunk and round)
constexpr double kFixedChunks = 3.3;
// SMALL launches (batches 1-4: script_pwc.py and inference_kitti.py run one pair): fewer workgroups than CUs.  Every tile is cut into
// ksplit slices of the input channels so that tiles x cout groups x slices cover the chip once; ksplit minimises
// rounds x (chunks per slice + fixed cost), at least 6 chunks per slice, at most 8 slices.
inline int wino4_small_ksplit(int64_t nwg, int nchunks, int cus) {
    int best = 1;
    double best_cost = (double)((nwg + cus - 1) / cus) * (nchunks + kFixedChunks);
    for (int k = 2; k <= 8 && nchunks / k >= 6; ++k) {
        const double cost = (double)((nwg * k + cus - 1) / cus) * ((nchunks + k - 1) / k + kFixedChunks) + 0.5 * k;      // + the reduce kernel's reads
        if (cost < 0.9 * best_cost) { best = k; best_cost = cost; }
    }
    return best;
}

// The tail is made of the same (last) tile positions of EVERY image, so the result of an item does not depend on its slot in the batch.
inline TailPlan wino4_tail_plan(int B, int64_t nblk, int ngroups, int nchunks, int Cout, int tile_px, int split2) {
    TailPlan p{(int)nblk, 1, nchunks, 0};
    const int cus = device_cus();
    const int64_t nwg = nblk * ngroups, full = nwg / cus, rem = nwg - full * cus;
    if (full < 1) {
        // whole-launch split (main_tiles = 0): every tile of every image alike, so an item's result does not depend on its slot either
        if (!pwc::option(pwc::OPT_W4_SMALLSPLIT)) return p;
        const int k = wino4_small_ksplit(nwg, nchunks, cus);
        if (k == 1) return p;
        p.main_tiles = 0;
        p.ksplit = k;
        p.cps = (nchunks + k - 1) / k;
        p.ws_bytes = (int64_t)k * nblk * Cout * tile_px * (int64_t)sizeof(float);
        return p;
    }
    if (!pwc::option(pwc::OPT_W4_TAILSPLIT) || split2 || rem == 0 || (cus % ngroups) != 0) return p;
    const int64_t main_tiles = full * cus / ngroups, tail = nblk - main_tiles;
    if ((main_tiles & 7) || (tail & 7) || (tail % B)) return p;  // keep the XCD-aware tile order of both launches; whole positions
    // Two slices, and only when they fit ONE half-length round (rem <= half the CUs): a slice costs ~10 us outside its K loop, so four
    // quarter-length slices per tile measured no gain (448 workgroups = 1.75 rounds: 297 us unsplit, 314 us as 256 + 768 quarter slices),
    // while 896 = 3 rounds + 256 half slices gains 7-10 % (conv2_3 728 -> 668 us, the 64-cout launch of conv2_2 580 -> 542 us).
    const int best = (nchunks / 2 >= 12 && rem * 2 <= cus) ? 2 : 1;
    if (best == 1) return p;
    p.main_tiles = (int)main_tiles;
    p.ksplit = best;
    p.cps = (nchunks + best - 1) / best;
    p.ws_bytes = (int64_t)best * tail * Cout * tile_px * (int64_t)sizeof(float);
    return p;
}

template <int CB, int TG, int GW>
int launch_wino4(const float *x, const float *up, const float *bias, float *y, int B, int Cin, int H, int W, int Cout,
                 int64_t bsx, int64_t bsy, float slope, int do_leaky, hipStream_t st, int co0, int ngroups, int split2,
                 void *workspace, int64_t workspace_bytes) {
    using G = Geo4<CB, TG, GW>;
    constexpr int kSmemP = G::kSmemBytes > 8 * 64 * 64 * 4 ? G::kSmemBytes : 8 * 64 * 64 * 4;       // rings, or the 128 KiB of the final exchange
    static pwc::LdsAttrOnce once;
    if (const int rc = pwc::ensure_lds_attr(once, reinterpret_cast<const void *>(&conv3x3_wino4p_kernel<CB, TG, GW>), kSmemP, "conv3x3_wino4p_kernel"))
        return rc;
    const int CoutP = cout_padded4(Cout);
    constexpr int kTH = G::kGH * TG;
    const int tiles_x = (W + GW - 1) / GW, tiles_y = (H + kTH - 1) / kTH;
    const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
    if (nblk * ngroups > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino4_fwd: grid too large");
    TailPlan tp = wino4_tail_plan(B, nblk, ngroups, (Cin + kCK - 1) / kCK, Cout, kTH * GW, split2);
    if (tp.ksplit > 1 && (!workspace || workspace_bytes < tp.ws_bytes || (reinterpret_cast<uintptr_t>(workspace) & 15u))) tp = TailPlan{(int)nblk, 1, 0, 0};
    if (tp.main_tiles > 0)
        hipLaunchKernelGGL((conv3x3_wino4p_kernel<CB, TG, GW>), dim3((unsigned)(tp.main_tiles * ngroups)), dim3(kThreads), kSmemP, st,
                           x, up, bias, y, Cin, H, W, Cout, CoutP, tiles_x, tiles_y, bsx, bsy, slope, do_leaky, co0, tp.main_tiles, ngroups, split2,
                           TailSplit{tp.main_tiles / B, 0, 1, 0, nullptr});
    if (tp.ksplit > 1) {
        const int ntail = (int)nblk - tp.main_tiles;
        float *ws = static_cast<float *>(workspace);
        hipLaunchKernelGGL((conv3x3_wino4p_kernel<CB, TG, GW>), dim3((unsigned)(ntail * ngroups * tp.ksplit)), dim3(kThreads), kSmemP, st,
                           x, up, bias, y, Cin, H, W, Cout, CoutP, tiles_x, tiles_y, bsx, bsy, slope, do_leaky, co0, ntail, ngroups, 0,
                           TailSplit{ntail / B, tp.main_tiles / B, tp.ksplit, tp.cps, ws});
        const int ncout = min(ngroups * G::kCoutT, Cout - co0);
        const int64_t total = (int64_t)ntail * ncout * kTH * (GW / 4);
        hipLaunchKernelGGL(wino4_tail_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                           ws, bias, y, tp.ksplit, ntail / B, tp.main_tiles / B, ntail, tiles_x, kTH, GW, H, W, Cout, co0, ncout, bsy, slope, do_leaky, split2, total);
    }
    pwc::note_kernel("conv3x3_wino4p_kernel", CB, TG, GW, tp.ksplit, 1, 0);
    return pwc::check_launch("conv3x3_wino4p_kernel");
}

// Tile-group width for a map of W columns: 64 unless that leaves the groups under 85 % full and 32 does better (the 14x32 lattice images of
// dc_conv4: 0.5 vs 1.0)
inline int wino4_gw(int W) {
    const double f64 = (double)W / ((W + 63) / 64 * 64), f32 = (double)W / ((W + 31) / 32 * 32);
    return (f64 < 0.85 && f32 > f64) ? 32 : 64;
}
// rows of a map covered by the launch with `th`-row workgroups, as a fraction
inline double wino4_row_fill(int H, int th) { return (double)H / ((H + th - 1) / th * th); }

}  // namespace

// Can this layer run here at all?  Dilation 1 (a dilated layer's lattices are not contiguous in memory: no 16-byte pieces),
// W % 4 == 0 (aligned 16-byte row pieces and stores).
static bool wino4_supported(int W, int dilation) { return dilation == 1 && W % 4 == 0; }

// Does F(4x4,3x3) beat F(2x2,3x3) (pwc_conv3x3_wino_fwd) for this layer?  Rule measured at batch 16 (profiles/r03_wino4_layers.txt):
// the tile groups are 8 or 16 rows x 64 columns, so the map must fill them, and the launch must cover the chip several times
// over (a workgroup costs ~10 us outside its K loop).  Option "conv_wino4" = 0 (pwc_set_option, default from PWC_CONV_WINO4) switches the route off (A/B runs).
extern "C" int pwc_conv3x3_wino4_preferred(int B, int Cin, int H, int W, int Cout, int dilation) {
    if (B <= 0 || Cin < 32 || H <= 0 || W <= 0 || Cout < 32 || !wino4_supported(W, dilation)) return 0;
    const int knob = pwc::option(pwc::OPT_CONV_WINO4);
    if (!knob) return 0;
    const int n32 = cout_padded4(Cout) / 32;
    const int gw = wino4_gw(W), gh = gw == 64 ? 4 : 8;                 // a tile group is gh rows x gw columns
    const int tiles_x = (W + gw - 1) / gw;
    const int nchunks = (Cin + kCK - 1) / kCK;
    // every launch the layer splits into must cover the chip: the 64-cout launch (two tile groups per workgroup) and, for an odd number
    // of 32-cout blocks, the 32-cout one (four groups) -- conv3_2 (96 couts @56x128) fails on the latter (128 workgroups, x0.94).  A launch
    // smaller than the chip counts with the input-channel slices wino4_tail_plan cuts it into (small batches), provided a slice keeps
    // a K loop worth its fixed cost.
    auto covers = [&](int64_t nblk, int ngroups) {
        const int64_t nwg = nblk * ngroups;
        if (nwg >= 200) return true;
        if (!pwc::option(pwc::OPT_W4_SMALLSPLIT)) return false;
        const int k = wino4_small_ksplit(nwg, nchunks, device_cus());
        return nwg * k >= pwc::option(pwc::OPT_W4_SMALL_MIN_WGS) && nchunks / k >= 6;
    };
    if (n32 >= 2) {
        const int th = 2 * gh;
        if (wino4_row_fill(H, th) < 0.85 || !covers((int64_t)B * tiles_x * ((H + th - 1) / th), n32 / 2)) return 0;
    }
    if (n32 & 1) {
        const int th = 4 * gh;
        if (wino4_row_fill(H, th) < 0.85 || !covers((int64_t)B * tiles_x * ((H + th - 1) / th), 1)) return 0;
    }
    return (double)W / (tiles_x * gw) >= 0.85;
}

extern "C" int64_t pwc_conv3x3_wino4_packed_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return -1;
    return (int64_t)((Cin + kCK - 1) / kCK) * 36 * kCK * cout_padded4(Cout) * (int64_t)sizeof(float);
}

extern "C" int pwc_conv3x3_wino4_pack(const void *w, void *up, int Cin, int Cout, void *stream) {
    if (!w || !up) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino4_pack: null pointer");
    if (Cin <= 0 || Cout <= 0) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino4_pack: bad shape");
    if (!pwc::aligned16(up)) PWC_FAIL(PWC_EALIGN, "pwc_conv3x3_wino4_pack: packed buffer must be 16-byte aligned");
    const int64_t total = pwc_conv3x3_wino4_packed_bytes(Cin, Cout) / (int64_t)sizeof(float);
    hipLaunchKernelGGL(wino4_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(w), static_cast<float *>(up), Cin, Cout, cout_padded4(Cout), total);
    return pwc::check_launch("wino4_pack_kernel");
}

extern "C" int pwc_conv3x3_wino4_fwd(const void *x, const void *up, const void *bias, void *y, int B, int Cin, int H, int W, int Cout,
                                     int dilation, unsigned flags, float leaky_slope, int64_t x_bstride, int64_t y_bstride,
                                     void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !up || !bias || !y) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino4_fwd: null pointer");
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino4_fwd: bad shape");
    if (!wino4_supported(W, dilation))
        PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv3x3_wino4_fwd: needs dilation 1 and W %% 4 == 0 (got dilation %d, W %d): use pwc_conv3x3_wino_fwd", dilation, W);
    if (!pwc::aligned16(up) || !pwc::aligned16(x) || !pwc::aligned16(y) || (x_bstride % 4) || (y_bstride % 4))
        PWC_FAIL(PWC_EALIGN, "pwc_conv3x3_wino4_fwd: tensors must be 16-byte aligned with batch strides that are multiples of 4");
    const int64_t plane = (int64_t)H * W;
    if (x_bstride < Cin * plane || y_bstride < Cout * plane / ((flags & PWC_CONV_SPLIT2) ? 4 : 1))
        PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino4_fwd: batch stride smaller than the tensor");
    if (plane * kCK * 4 >= 0x7fffffffLL) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv3x3_wino4_fwd: image plane too large for 32-bit DMA offsets");
    const float *xf = static_cast<const float *>(x), *uf = static_cast<const float *>(up), *bf = static_cast<const float *>(bias);
    float *yf = static_cast<float *>(y);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int do_leaky = (flags & PWC_ACT_LEAKY) ? 1 : 0;
    const int split2 = (flags & PWC_CONV_SPLIT2) ? 1 : 0;
    if (split2 && ((H & 1) || (W & 7) || (y_bstride & 1)))
        PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino4_fwd: PWC_CONV_SPLIT2 needs even H, W %% 8 == 0 and an even batch stride");
    const int n32 = cout_padded4(Cout) / 32;
    // 64-cout workgroups (4 cout blocks x 2 tile groups) for as many pairs of 32 as there are, one 32-cout launch (2 x 4) for an odd rest
    if (wino4_gw(W) == 64) {
        if (n32 >= 2)
            if (const int rc = launch_wino4<4, 2, 64>(xf, uf, bf, yf, B, Cin, H, W, Cout, x_bstride, y_bstride, leaky_slope, do_leaky, st, 0, n32 / 2, split2, workspace, workspace_bytes))
                return rc;
        if (n32 & 1)
            return launch_wino4<2, 4, 64>(xf, uf, bf, yf, B, Cin, H, W, Cout, x_bstride, y_bstride, leaky_slope, do_leaky, st, (n32 / 2) * 64, 1, split2, workspace, workspace_bytes);
    } else {
        if (n32 >= 2)
            if (const int rc = launch_wino4<4, 2, 32>(xf, uf, bf, yf, B, Cin, H, W, Cout, x_bstride, y_bstride, leaky_slope, do_leaky, st, 0, n32 / 2, split2, workspace, workspace_bytes))
                return rc;
        if (n32 & 1)
            return launch_wino4<2, 4, 32>(xf, uf, bf, yf, B, Cin, H, W, Cout, x_bstride, y_bstride, leaky_slope, do_leaky, st, (n32 / 2) * 64, 1, split2, workspace, workspace_bytes);
    }
    return PWC_OK;
}

// Scratch for the tail / whole-launch split of launch_wino4 (0: neither applies to this layer's launches)
extern "C" int64_t pwc_conv3x3_wino4_workspace_bytes(int B, int Cin, int H, int W, int Cout) {
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || !wino4_supported(W, 1)) return 0;
    const int n32 = cout_padded4(Cout) / 32, nchunks = (Cin + kCK - 1) / kCK;
    const int gw = wino4_gw(W), gh = gw == 64 ? 4 : 8;
    const int64_t tiles_x = (W + gw - 1) / gw;
    int64_t need = 0;
    if (n32 >= 2) need = max(need, wino4_tail_plan(B, (int64_t)B * tiles_x * ((H + 2 * gh - 1) / (2 * gh)), n32 / 2, nchunks, Cout, 2 * gh * gw, 0).ws_bytes);
    if (n32 & 1) need = max(need, wino4_tail_plan(B, (int64_t)B * tiles_x * ((H + 4 * gh - 1) / (4 * gh)), 1, nchunks, Cout, 4 * gh * gw, 0).ws_bytes);
    return need;
}

extern "C" int pwc_lattice_unsplit_f32(const void *x, void *y, int B, int C, int h, int w, int levels, int64_t y_bstride, void *stream) {
    if (!x || !y || B <= 0 || C <= 0 || h <= 0 || w <= 0 || levels < 1 || levels > 4) PWC_FAIL(PWC_EINVAL, "pwc_lattice_unsplit_f32: bad argument");
    const int64_t W = (int64_t)w << levels, H = (int64_t)h << levels;
    if ((W & 3) || !pwc::aligned16(y) || (y_bstride & 3) || y_bstride < C * H * W)
        PWC_FAIL(PWC_EALIGN, "pwc_lattice_unsplit_f32: the output needs 16-byte aligned rows (W %% 4 == 0) and a batch stride >= C*H*W");
    const int64_t total = (int64_t)B * C * H * (W >> 2);
    if ((total + 255) / 256 > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_lattice_unsplit_f32: grid too large");
    hipLaunchKernelGGL(lattice_unsplit_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(x), static_cast<float *>(y), C, h, w, levels, y_bstride, total);
    return pwc::check_launch("lattice_unsplit_kernel");
}

// timing-experiment mask this translation unit was built with (0 in the product; pwc_experiment_mask)
namespace pwc { int exp_mask_wino4() { return PWC_W4_EXP; } }
