// 3x3 / stride 1 convolution (any dilation, padding = dilation) by Winograd F(2x2,3x3) on the gfx950 matrix cores, fp32 throughout.
// Replaces nn.Conv2d(3x3, padding 1) + LeakyReLU(0.1) (reference models/PWCNet.py:26-33) for the large dense-block and
// context layers: 16 multiplications per 2x2 outputs instead of 36, i.e. 2.25x fewer MFMA passes than the direct implicit
// GEMM of pwc_conv_mfma.h for the same fp32 result (up to rounding: the transforms only add and halve).
//
//     Y(2x2) = At [ sum_cin (G g Gt) (.) (Bt d B) ] A        d = 4x4 input patch, g = 3x3 filter
//
// One GEMM per transform position p = 0..15:  M_p[cout, tile] = sum_cin U_p[cout, cin] * V_p[cin, tile].
//   U = G g Gt is computed once per model (pwc_conv3x3_wino_pack) and laid out the way the kernel's LDS wants it;
//   V = Bt d B is computed in the kernel, in registers, from the raw input tile in LDS, hidden under the MFMAs;
//   v_mfma_f32_32x32x2_f32: A = U_p (32 couts x 2 cin), B = V_p (2 cin x 32 tiles), D = 32 couts x 32 tiles.
// A PAIR of waves owns ONE 32-cout block x ONE group of 32 tiles (4 rows x 32 columns of output): 8 of the 16 positions each
// (8 x 16 = 128 accumulator registers); the output transform At M A swaps one row of M between the two through LDS.
// Workgroup = 8 waves = MT cout blocks x (4/MT) tile groups; Cin is consumed in chunks of 4 channels (two MFMA k-steps):
//     iteration k:  barrier | 16 MFMAs per wave on U(k), V(k), with -- one micro-step in the shadow of each MFMA -- the LDS-DMA
//                   of raw(k+2) and U(k+2), the patch reads of raw(k) and the additions that make V(k)
// LDS: raw [3][4][rows+2][34], U [3][16][2][32*MT][2]  (54-106 KiB, at least the 64 KiB of the epilogue's exchange; one workgroup per CU).
// Zero padding, ragged edges and the ragged last channel chunk come from the buffer range check (0 into LDS).
#include <stdlib.h>

#include <type_traits>

#include "pwc_common.h"

namespace pwc_conv {
int splitk_reduce(const float *partial, const float *bias, const float *residual, float *y, int B, int Cout, int plane, int ksplit,
                  int64_t bsy, int64_t bsr, float slope, int do_leaky, hipStream_t st);      // pwc_conv.hip
}

namespace {

using pwc::leaky;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kCK = 4;               // input channels per chunk
constexpr int kTW = 32;              // output columns of a tile group (16 tiles)
constexpr int kGH = 4;               // output rows of a tile group (2 tile rows)
constexpr int kRawW = kTW + 2;
constexpr unsigned kOOB = 0x80000000u;

template <int MT>
struct Geo {
    static constexpr int kTG = 4 / MT;                               // tile groups per workgroup
    static constexpr int kRows = kGH * kTG + 2;
    static constexpr int kRawPlane = kRows * kRawW;                  // floats per staged channel
    static constexpr int kRawElems = kCK * kRawPlane;
    static constexpr int kRawSlots = (kRawElems + kThreads - 1) / kThreads;
    static constexpr int kRawRegion = kRawSlots * kThreads;
    static constexpr int kCoutT = 32 * MT;
    static constexpr int kUFloats = 16 * 2 * kCoutT * 2;             // [pos][kh][cout][step]
    static constexpr int kUSlots = kUFloats / 4 / kThreads;          // 16-byte pieces per thread: 2*MT
};

// U[chunk][pos][kh][co][step] <- G g Gt of w[co][cin = chunk*4 + 2*kh + step], zero padded
__global__ void __launch_bounds__(256)
wino_pack_kernel(const float *__restrict__ w, float *__restrict__ up, int Cin, int Cout, int CoutP, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int step = (int)(i & 1);
    int64_t t = i >> 1;
    const int co = (int)(t % CoutP);
    t /= CoutP;
    const int kh = (int)(t & 1);
    t >>= 1;
    const int pos = (int)(t & 15);
    const int chunk = (int)(t >> 4);
    const int cin = chunk * kCK + 2 * kh + step;
    float v = 0.f;
    if (co < Cout && cin < Cin) {
        const float *g = w + ((int64_t)co * Cin + cin) * 9;
        const int pi = pos >> 2, pj = pos & 3;
        // rows of G: (1,0,0) (1/2,1/2,1/2) (1/2,-1/2,1/2) (0,0,1)
        const float gi[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float r = 0.f;
#pragma unroll
            for (int b = 0; b < 3; ++b) r += g[a * 3 + b] * gi[pj][b];
            s += gi[pi][a] * r;
        }
        v = s;
    }
    up[i] = v;
}

// ---- EIGHT waves (two per SIMD) ------------------------------------------------------------------------------------------
// With one wave per SIMD every instruction that is not an MFMA competes with the MFMAs for the wave's single in-order issue
// slot (a four-wave form with 256 accumulators per wave was measured in round 2: matrix pipe 75 % busy on 128-cout layers, 55 %
// on 32-cout layers, 3-9 % slower; removed in round 3).  Here the 16 positions of a (cout block, tile group) pair are split between
// two waves (8 positions = 128 accumulator registers each), so the workgroup has 8 waves, two per SIMD, and one wave's LDS-DMA /
// operand work issues while the other wave's MFMAs run.  The price is an exchange at the end: At M A needs all four rows of M, each
// wave of a pair holds two; the waves swap one row each through LDS (M1 one way, M2 the other) and each finishes ONE of the two
// output rows.
constexpr int kThreads8 = 512;

// ---- register transform (round 3) ----------------------------------------------------------------------------------------------
// Round 2's kernel (conv3x3_wino8_kernel, removed) staged V = Bt d B in LDS: every thread turned (channel, tile, row) units of the raw
// tile into V values one chunk ahead (4 ds_read_b64 + 4 ds_write_b32 per unit).  Here V never exists in LDS: the MFMA B operand of
// lane (tile n, k-half kh) is V_p[channels 2kh, 2kh+1][tile n] -- a function of that lane's OWN 4x4 patch of those two channels.
// The lane reads the three patch rows its wave's positions need (12 ds_read_b64 per chunk) and makes its 16 operand values with
// 32 additions in MFMA shadows.  Gone: the 16 ds_write_b32 + 16 ds_read_b64 per (channel, tile) of the V round trip -- LDS-array
// cycles per chunk of a 32-cout workgroup 1280 -> ~640 of the 2048 its MFMAs take (ds_write_b32 costs 4 cycles per wave-instruction),
// the V double buffer (8-32 KiB), and one chunk of pipeline depth (raw(k) is read in iteration k, not transformed one ahead).
// Same arithmetic in the same order: bit-identical outputs, 1.00-1.19x per layer (32-cout layers +9 %, 64-cout +4..7 %, 128-cout +1 %;
// profiles/r03_wino_regtr_ab.txt).  The waves of a workgroup that share a tile group repeat the additions (VALU is idle beside the
// MFMAs); positions are split between the two waves of a pair: wave ph owns rows i = 2ph, 2ph+1 of Bt d B,
//   Bt rows: 0 = d0 - d2, 1 = d1 + d2, 2 = d2 - d1, 3 = d1 - d3   ->   ph=0: A0 = d0, B0 = d2, Z = d1;  ph=1: A0 = d2, B0 = d1, Z = d3
//   first row of the wave  w = A0 - B0,  second row  w = B0 + s1 * Z  (s1 = +1 / -1),  then columns (w0-w2, w1+w2, w2-w1, w1-w3).
template <int MT>
struct GeoR {
    using G = Geo<MT>;
    static constexpr int kRing = 3 * (G::kRawRegion + G::kUFloats) * 4;
    static constexpr int kExchange = 4 * 2 * 32 * 64 * 4;                     // epilogue: [pair][sender][32][64 lanes] floats
    static constexpr int kSmemBytes = kRing > kExchange ? kRing : kExchange;
};

template <int MT>
__global__ void __launch_bounds__(kThreads8, 1)
conv3x3_wino8r_kernel(const float *__restrict__ x, const float *__restrict__ up, const float *__restrict__ bias,
                      float *__restrict__ y, int Cin, int H, int W, int Cout, int CoutP, int tiles_x, int tiles_y,
                      int64_t bsx, int64_t bsy, float slope, int do_leaky, int vec2, int dil, int co0,
                      int cps, float *__restrict__ part, int64_t zstride) {
    using G = Geo<MT>;
    constexpr int TG = G::kTG;
    constexpr int RS = (G::kRawElems + kThreads8 - 1) / kThreads8;        // dword LDS-DMAs per thread and chunk
    constexpr int US = G::kUFloats / 4 / kThreads8;                        // 16-byte LDS-DMAs per thread and chunk (= MT)
    static_assert(RS * kThreads8 <= G::kRawRegion && US >= 1 && 3 + RS + US <= 16, "geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *raw = smem;                                  // [3][kRawRegion]
    float *ubuf = smem + 3 * G::kRawRegion;             // [3][kUFloats]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int col = lane & 31;
    const int kh = lane >> 5;

    int bid = blockIdx.x;
    if ((gridDim.x & 7u) == 0) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);      // XCD-contiguous runs of tiles
    const int sub = bid % (dil * dil);                  // pixel lattice of a dilated layer
    bid /= dil * dil;
    const int ry = sub / dil, rx = sub % dil;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int cb = co0 + (int)blockIdx.y * G::kCoutT;
    const int ox0 = tx * kTW;
    const int oy0 = ty * (kGH * TG);
    const int plane = H * W;

    unsigned raw_off[RS];
#pragma unroll
    for (int j = 0; j < RS; ++j) {
        const int i = j * kThreads8 + tid;
        const int c = i / G::kRawPlane;
        const int rem = i % G::kRawPlane;
        const int iy = ry + dil * (oy0 - 1 + rem / kRawW);
        const int ix = rx + dil * (ox0 - 1 + rem % kRawW);
        const bool ok = (i < G::kRawElems) && (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);
        raw_off[j] = ok ? (unsigned)(c * plane + iy * W + ix) * 4u : kOOB;
    }
    unsigned u_off[US];
#pragma unroll
    for (int j = 0; j < US; ++j) {
        const int p = j * kThreads8 + tid;
        const int row = p / (16 * MT);
        const int q = p % (16 * MT);
        u_off[j] = (unsigned)(row * CoutP * 2 + q * 4) * 4u;
    }

    const float *xb = x + (int64_t)b * bsx;
    // split-K (cps > 0): blockIdx.z owns the chunks [z*cps, (z+1)*cps) and writes its raw partial sums (At M A is linear) to
    // part[z][b][Cout][H][W]; bias / LeakyReLU are applied by the fixed-order reduction (pwc_conv::splitk_reduce)
    const int c_lo = cps ? (int)blockIdx.z * cps : 0;
    const int nchunks = cps ? min((Cin + kCK - 1) / kCK - c_lo, cps) : (Cin + kCK - 1) / kCK;
    const int64_t uchunk = (int64_t)64 * CoutP;
    const float *ug = up + cb * 2;
    const int ubytes = (int)(uchunk - cb * 2) * 4;
    const unsigned lds_raw = pwc::lds_addr(raw) + wave * 256;
    const unsigned lds_u = pwc::lds_addr(ubuf) + wave * 1024;

    pwc::v4i32 rs_raw, rs_u;
    unsigned base_raw = 0, base_u = 0;
    auto setup = [&](int chunk, int slot) {                 // descriptors + LDS bases of group {raw(chunk), U(chunk)} -> ring slot
        const int c0 = (c_lo + chunk) * kCK;
        rs_raw = pwc::make_rsrc(xb + (int64_t)c0 * plane, min(kCK, Cin - c0) * plane * 4);
        base_raw = __builtin_amdgcn_readfirstlane(lds_raw + slot * G::kRawRegion * 4);
        rs_u = pwc::make_rsrc(ug + (int64_t)(c_lo + chunk) * uchunk, ubytes);
        base_u = __builtin_amdgcn_readfirstlane(lds_u + slot * G::kUFloats * 4);
    };
    auto issue_all = [&]() {
#pragma unroll
        for (int j = 0; j < RS; ++j) pwc::dma_b32(rs_raw, base_raw + j * kThreads8 * 4, raw_off[j]);
#pragma unroll
        for (int j = 0; j < US; ++j) pwc::dma_b128(rs_u, base_u + j * kThreads8 * 16, u_off[j]);
    };

    const int pw = wave >> 1, ph = wave & 1;            // pair (cout block, tile group); position half
    const int blk = pw % MT, tgw = pw / MT;
    // this lane's patch: tile n = col -> (tile row col >> 4, tile column col & 15) of tile group tgw, channels 2kh + step
    const int pbase = 2 * kh * G::kRawPlane + (kGH * tgw + 2 * (col >> 4)) * kRawW + 2 * (col & 15);
    const int offA = pbase + (2 * ph) * kRawW, offB = pbase + (2 - ph) * kRawW, offZ = pbase + (1 + 2 * ph) * kRawW;
    const float s1 = ph ? -1.f : 1.f;
    const int ua_off = (ph * 8 * 2 + kh) * G::kCoutT * 2 + (blk * 32 + col) * 2;          // U of position p0 = 8*ph

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[p][j] = 0.f;
    f32x2 a2[2][4], b2[2][4];                           // operand sets: local positions 0..3 -> set 0, 4..7 -> set 1; [.][j] = {step 0, step 1}
#pragma unroll
    for (int i = 0; i < 4; ++i) a2[1][i] = b2[1][i] = (f32x2){0.f, 0.f};
    f32x2 rA[2][2], rB[2][2], rZ[2][2];                 // patch rows: [step][columns 0-1 / 2-3]

    // columns of Bt d B from one transformed row w (4 values) of channel `st`, into operand set `set`
    auto columns = [&](int set, int st, float w0, float w1, float w2, float w3) {
        b2[set][0][st] = w0 - w2;
        b2[set][1][st] = w1 + w2;
        b2[set][2][st] = w2 - w1;
        b2[set][3][st] = w1 - w3;
    };

    auto iteration = [&](int k, int r0, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        // ring slots: chunk k in r0 (= k % 3), k+1 in flight -> r1, k+2 issued now -> r2
        const int r1 = (r0 == 2) ? 0 : r0 + 1, r2 = (r1 == 2) ? 0 : r1 + 1;
        const bool do_dma = FULL || (k + 2 < nchunks);
        if (FULL || k + 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RS + US) : "memory");      // group k landed, k+1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float *ua = ubuf + r0 * G::kUFloats + ua_off;
        const float *rw = raw + r0 * G::kRawRegion;
        constexpr int kSetup = 2, kDma0 = 3;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int q = (s >> 3) - 1, i = s & 7;              // q = -1: local positions 4..7 of the previous chunk (set 1)
            const int pq = (q < 0) ? 1 : 0;
            acc[4 * pq + (i & 3)] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[pq][i & 3][i >> 2], b2[pq][i & 3][i >> 2], acc[4 * pq + (i & 3)], 0, 0, 0);
            if (s < 4) a2[0][s] = *reinterpret_cast<const f32x2 *>(ua + s * (2 * G::kCoutT * 2));            // U of local positions 0..3
            if (s >= 8 && s < 12) a2[1][s - 8] = *reinterpret_cast<const f32x2 *>(ua + (s - 4) * (2 * G::kCoutT * 2));   // ... 4..7
            if (s == 0) {
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    rA[st][0] = *reinterpret_cast<const f32x2 *>(rw + offA + st * G::kRawPlane);
                    rA[st][1] = *reinterpret_cast<const f32x2 *>(rw + offA + st * G::kRawPlane + 2);
                }
            }
            if (s == 1) {
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    rB[st][0] = *reinterpret_cast<const f32x2 *>(rw + offB + st * G::kRawPlane);
                    rB[st][1] = *reinterpret_cast<const f32x2 *>(rw + offB + st * G::kRawPlane + 2);
                }
            }
            if (s == 5 || s == 6) {                             // first row of this wave (local positions 0..3): w = A0 - B0
                const int st = s - 5;
                columns(0, st, rA[st][0][0] - rB[st][0][0], rA[st][0][1] - rB[st][0][1], rA[st][1][0] - rB[st][1][0], rA[st][1][1] - rB[st][1][1]);
            }
            if (s == 7) {
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    rZ[st][0] = *reinterpret_cast<const f32x2 *>(rw + offZ + st * G::kRawPlane);
                    rZ[st][1] = *reinterpret_cast<const f32x2 *>(rw + offZ + st * G::kRawPlane + 2);
                }
            }
            if (s == 12 || s == 13) {                           // second row (local positions 4..7): w = B0 + s1 * Z
                const int st = s - 12;
                columns(1, st, __builtin_fmaf(s1, rZ[st][0][0], rB[st][0][0]), __builtin_fmaf(s1, rZ[st][0][1], rB[st][0][1]),
                        __builtin_fmaf(s1, rZ[st][1][0], rB[st][1][0]), __builtin_fmaf(s1, rZ[st][1][1], rB[st][1][1]));
            }
            if (s == kSetup && do_dma) setup(k + 2, r2);
            if (s >= kDma0 && s < kDma0 + RS) {
                if (do_dma) pwc::dma_b32(rs_raw, base_raw + (s - kDma0) * kThreads8 * 4, raw_off[s - kDma0]);
            } else if (s >= kDma0 + RS && s < kDma0 + RS + US) {
                if (do_dma) pwc::dma_b128(rs_u, base_u + (s - kDma0 - RS) * kThreads8 * 16, u_off[s - kDma0 - RS]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // prologue: the two groups the loop expects in flight, {raw(0), U(0)} and {raw(1), U(1)}
    setup(0, 0);
    issue_all();
    if (nchunks > 1) {
        setup(1, 1);
        issue_all();
    }
#ifndef PWC_WINO_NO_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);      // the later-dispatched wave of each SIMD loses every issue arbitration otherwise
#endif
    int r0 = 0;
    int k = 0;
    for (; k + 2 < nchunks; ++k) {
        iteration(k, r0, std::true_type{});
        r0 = (r0 == 2) ? 0 : r0 + 1;
    }
    for (; k < nchunks; ++k) {
        iteration(k, r0, std::false_type{});
        r0 = (r0 == 2) ? 0 : r0 + 1;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)               // local positions 4..7 of the last chunk
        acc[4 + (i & 3)] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[1][i & 3][i >> 2], b2[1][i & 3][i >> 2], acc[4 + (i & 3)], 0, 0, 0);

    // ---- output transform: wave ph=0 holds M rows 0,1 (acc[0..3], acc[4..7]), ph=1 rows 2,3.  Row 0 of Y needs M0 + M1 + M2, row 1
    // needs M1 - M2 - M3: ph=0 sends M1, ph=1 sends M2, 8 cout rows (32 floats per lane) at a time through LDS.
    const int oy = ry + dil * (oy0 + kGH * tgw + 2 * (col >> 4) + ph);
    const int ox = rx + dil * (ox0 + 2 * (col & 15));
    const bool inside = (oy < H) && (ox < W);
    const int64_t obase = (int64_t)b * bsy + (int64_t)oy * W + ox;
    float bvs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bvs[j] = bias[min(cb + blk * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh, Cout - 1)];
    float *xsend = smem + ((pw * 2 + ph) * 32) * 64 + lane;
    const float *xrecv = smem + ((pw * 2 + (ph ^ 1)) * 32) * 64 + lane;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 8; ++jj)
#pragma unroll
            for (int c = 0; c < 4; ++c) xsend[(jj * 4 + c) * 64] = ph ? acc[c][8 * r + jj] : acc[4 + c][8 * r + jj];
        __syncthreads();
        float ya[8], yb[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = 8 * r + jj;
            float tt[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float other = xrecv[(jj * 4 + c) * 64];
                tt[c] = ph ? (other - acc[c][j] - acc[4 + c][j]) : (acc[c][j] + acc[4 + c][j] + other);
            }
            const float bv = part ? 0.f : bvs[j];
            ya[jj] = tt[0] + tt[1] + tt[2] + bv;
            yb[jj] = tt[1] - tt[2] - tt[3] + bv;
            if (do_leaky && !part) { ya[jj] = leaky(ya[jj], slope); yb[jj] = leaky(yb[jj], slope); }
        }
        if (part) {
            const int64_t pb = (int64_t)blockIdx.z * zstride + (int64_t)b * Cout * plane + (int64_t)oy * W + ox;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = 8 * r + jj;
                const int co = cb + blk * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh;
                if (inside && co < Cout) {
                    float *o = part + pb + (int64_t)co * plane;
                    o[0] = ya[jj];
                    if (ox + dil < W) o[dil] = yb[jj];
                }
            }
        } else if (vec2 == 2) {
            // 16-byte stores: the lanes of two neighbouring tiles (2 + 2 pixels of one row) swap halves -- the even lane ends up
            // with the four pixels of cout j, the odd lane with those of cout j + 1 -- so a lane issues 8 stores instead of 16
            const int odd = lane & 1;
#pragma unroll
            for (int jj = 0; jj < 8; jj += 2) {
                const int j = 8 * r + jj;
                const float s0 = odd ? ya[jj] : ya[jj + 1], sB = odd ? yb[jj] : yb[jj + 1];
                const float q0 = __shfl_xor(s0, 1), q1 = __shfl_xor(sB, 1);
                const int co = cb + blk * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh + odd;
                const f32x4 v = odd ? (f32x4){q0, q1, ya[jj + 1], yb[jj + 1]} : (f32x4){ya[jj], yb[jj], q0, q1};
                if (inside && co < Cout) *reinterpret_cast<f32x4 *>(y + obase - 2 * odd + (int64_t)co * plane) = v;
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = 8 * r + jj;
                const int co = cb + blk * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh;
                if (inside && co < Cout) {
                    float *o = y + obase + (int64_t)co * plane;
                    if (vec2) {
                        *reinterpret_cast<f32x2 *>(o) = (f32x2){ya[jj], yb[jj]};
                    } else {
                        o[0] = ya[jj];
                        if (ox + dil < W) o[dil] = yb[jj];
                    }
                }
            }
        }
    }
}

inline int cout_padded(int Cout) { return (Cout + 31) / 32 * 32; }

template <int MT>
int launch_wino(const float *x, const float *up, const float *bias, float *y, int B, int Cin, int H, int W, int Cout, int dil,
                int64_t bsx, int64_t bsy, float slope, int do_leaky, hipStream_t st, int co0, int ngroups,
                int ksplit = 1, int cps = 0, float *part = nullptr) {
    using G = Geo<MT>;
    const int CoutP = cout_padded(Cout);
    const int Hs = (H + dil - 1) / dil, Ws = (W + dil - 1) / dil;            // the largest of the D*D lattices
    const int tiles_x = (Ws + kTW - 1) / kTW, tiles_y = (Hs + kGH * G::kTG - 1) / (kGH * G::kTG);
    const int64_t nblk = (int64_t)B * tiles_x * tiles_y * dil * dil;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino_fwd: grid too large");
    int vec2 = dil == 1 && (W % 2 == 0) && (bsy % 2 == 0) && !(reinterpret_cast<uintptr_t>(y) & 7u);
    if (vec2 && (W % 4 == 0) && (bsy % 4 == 0) && !(reinterpret_cast<uintptr_t>(y) & 15u)) vec2 = 2;      // 16-byte stores (8-wave kernel)
    static pwc::LdsAttrOnce once8r;
    if (const int rc = pwc::ensure_lds_attr(once8r, reinterpret_cast<const void *>(&conv3x3_wino8r_kernel<MT>), GeoR<MT>::kSmemBytes,
                                            "conv3x3_wino8r_kernel"))
        return rc;
    hipLaunchKernelGGL(conv3x3_wino8r_kernel<MT>, dim3((unsigned)nblk, (unsigned)ngroups, (unsigned)ksplit), dim3(kThreads8),
                       GeoR<MT>::kSmemBytes, st, x, up, bias, y, Cin, H, W, Cout, CoutP, tiles_x, tiles_y, bsx, bsy, slope, do_leaky,
                       vec2, dil, co0, cps, part, (int64_t)B * Cout * H * W);
    pwc::note_kernel("conv3x3_wino8r_kernel", MT, G::kTG, 1, dil, 1, 0);
    return pwc::check_launch("conv3x3_wino8r_kernel");
}

// Launch shape of a layer run with ONE cout-tile width (the widest that divides CoutP), and its split-K plan: a launch that leaves
// most CUs idle (levels 5-4: 64-128 workgroups) is cut along Cin into ksplit slices of cps chunks (>= 8 chunks each: a tile costs
// ~9 us outside its K loop), partial sums go to the caller's workspace, pwc_conv::splitk_reduce adds them in fixed order.
struct WinoPlan { int mt, ksplit, cps; int64_t nwg; double fill; };
WinoPlan wino_plan(int B, int Cin, int H, int W, int Cout, int dilation) {
    const int CoutP = cout_padded(Cout);
    const int mt = (CoutP % 128 == 0) ? 4 : (CoutP % 64 == 0) ? 2 : 1;
    const int gh = kGH * (4 / mt);
    const int Hs = (H + dilation - 1) / dilation, Ws = (W + dilation - 1) / dilation;
    const int tiles_x = (Ws + kTW - 1) / kTW, tiles_y = (Hs + gh - 1) / gh;
    WinoPlan p;
    p.mt = mt;
    p.nwg = (int64_t)B * tiles_x * tiles_y * dilation * dilation * (CoutP / (32 * mt));
    p.fill = (double)Hs * Ws / ((double)tiles_y * gh * tiles_x * kTW);
    const int nchunks = (Cin + kCK - 1) / kCK;
    p.ksplit = 1;
    p.cps = 0;
    static const int knob = [] { const char *e = getenv("PWC_WINO_SPLIT"); return (e && *e) ? atoi(e) : -1; }();
    // measured (tools/bench_wino.py layers): with 64 workgroups (level 5 at batch 16) the split form only ties the direct kernel's own
    // split-K; from ~100 (conv4_3: 128) it wins 1.35x
    // ... and only with a long K: conv5aa/b (128 -> 128 @14x32, 32 chunks) in three slices lost to the direct kernel (52 vs 48 us)
    if (knob != 0 && p.nwg < 160 && p.nwg >= 96 && (nchunks >= 64 || knob > 0)) {
        int ks = (int)((320 + p.nwg - 1) / p.nwg);
        if (knob > 0) ks = knob;
        ks = ks < nchunks / 8 ? ks : nchunks / 8;
        if (ks >= 2) {
            p.cps = (nchunks + ks - 1) / ks;
            p.ksplit = (nchunks + p.cps - 1) / p.cps;
        }
    }
    return p;
}

}  // namespace

// Does the Winograd route beat pwc_conv2d_fwd for this layer?  Measured rule (tools/bench_wino.py layers, batch 16): it needs
// enough workgroups to cover the 256 CUs (one workgroup per CU, 8 waves; split-K counts), at least one full 32-row cout block,
// and -- with dilation D, where it runs on the D*D pixel lattices of ceil(H/D) x ceil(W/D) -- tiles that are mostly inside the lattice.
extern "C" int pwc_conv3x3_wino_preferred(int B, int Cin, int H, int W, int Cout, int dilation) {
    if (B <= 0 || Cin < 16 || H <= 0 || W <= 0 || Cout < 32 || dilation < 1 || dilation > 8) return 0;
    const WinoPlan p = wino_plan(B, Cin, H, W, Cout, dilation);
    return p.nwg * p.ksplit >= 160 && p.fill >= 0.7;
}

/* bytes of workspace the split-K form of this layer wants (0: it does not split) */
extern "C" int64_t pwc_conv3x3_wino_workspace_bytes(int B, int Cin, int H, int W, int Cout, int dilation) {
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || dilation < 1) return -1;
    const WinoPlan p = wino_plan(B, Cin, H, W, Cout, dilation);
    return p.ksplit > 1 ? (int64_t)p.ksplit * B * Cout * H * W * (int64_t)sizeof(float) : 0;
}

extern "C" int64_t pwc_conv3x3_wino_packed_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return -1;
    return (int64_t)((Cin + kCK - 1) / kCK) * 64 * cout_padded(Cout) * (int64_t)sizeof(float);
}

extern "C" int pwc_conv3x3_wino_pack(const void *w, void *up, int Cin, int Cout, void *stream) {
    if (!w || !up) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino_pack: null pointer");
    if (Cin <= 0 || Cout <= 0) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino_pack: bad shape");
    if (!pwc::aligned16(up)) PWC_FAIL(PWC_EALIGN, "pwc_conv3x3_wino_pack: packed buffer must be 16-byte aligned");
    const int64_t total = pwc_conv3x3_wino_packed_bytes(Cin, Cout) / (int64_t)sizeof(float);
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(w), static_cast<float *>(up), Cin, Cout, cout_padded(Cout), total);
    return pwc::check_launch("wino_pack_kernel");
}

extern "C" int pwc_conv3x3_wino_fwd(const void *x, const void *up, const void *bias, void *y, int B, int Cin, int H, int W, int Cout,
                                    int dilation, unsigned flags, float leaky_slope, int64_t x_bstride, int64_t y_bstride,
                                    void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !up || !bias || !y) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino_fwd: null pointer");
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino_fwd: bad shape");
    if (dilation < 1 || dilation > 64) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino_fwd: dilation %d", dilation);
    if (!pwc::aligned16(up)) PWC_FAIL(PWC_EALIGN, "pwc_conv3x3_wino_fwd: packed filters must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 3u)
        PWC_FAIL(PWC_EALIGN, "pwc_conv3x3_wino_fwd: tensors must be 4-byte aligned");
    const int64_t plane = (int64_t)H * W;
    if (x_bstride < Cin * plane || y_bstride < Cout * plane) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_wino_fwd: batch stride smaller than the tensor");
    if (plane * kCK * 4 >= 0x7fffffffLL) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv3x3_wino_fwd: image plane too large for 32-bit DMA offsets");
    const float *xf = static_cast<const float *>(x), *uf = static_cast<const float *>(up), *bf = static_cast<const float *>(bias);
    float *yf = static_cast<float *>(y);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int do_leaky = (flags & PWC_ACT_LEAKY) ? 1 : 0;
    const int CoutP = cout_padded(Cout);
    // cout blocks of 32: as many 128-wide workgroups as fit, then a 64-wide and a 32-wide launch for the rest (96 = 64 + 32: the
    // wider the cout tile, the fewer tile groups share a workgroup and the smaller the transform's share of the issue slots)
    static const int force_mt = [] { const char *e = getenv("PWC_WINO_MT"); return (e && *e) ? atoi(e) : 0; }();    // experiments
    const int nblk32 = CoutP / 32;
    // split-K for launches that would leave most CUs idle (needs the caller's workspace; without it the layer runs unsplit)
    const WinoPlan sp = wino_plan(B, Cin, H, W, Cout, dilation);
    if (sp.ksplit > 1 && workspace && !(reinterpret_cast<uintptr_t>(workspace) & 3u) &&
        workspace_bytes >= (int64_t)sp.ksplit * B * Cout * plane * (int64_t)sizeof(float)) {
        float *part = static_cast<float *>(workspace);
        int rc;
        if (sp.mt == 4) rc = launch_wino<4>(xf, uf, bf, yf, B, Cin, H, W, Cout, dilation, x_bstride, y_bstride, leaky_slope, do_leaky, st, 0, nblk32 / 4, sp.ksplit, sp.cps, part);
        else if (sp.mt == 2) rc = launch_wino<2>(xf, uf, bf, yf, B, Cin, H, W, Cout, dilation, x_bstride, y_bstride, leaky_slope, do_leaky, st, 0, nblk32 / 2, sp.ksplit, sp.cps, part);
        else rc = launch_wino<1>(xf, uf, bf, yf, B, Cin, H, W, Cout, dilation, x_bstride, y_bstride, leaky_slope, do_leaky, st, 0, nblk32, sp.ksplit, sp.cps, part);
        if (rc) return rc;
        return pwc_conv::splitk_reduce(part, bf, nullptr, yf, B, Cout, (int)plane, sp.ksplit, y_bstride, 0, leaky_slope, do_leaky, st);
    }
    int co0 = 0;
    if (force_mt == 1 || force_mt == 2) {
        if (force_mt == 2 && nblk32 % 2 == 0)
            return launch_wino<2>(xf, uf, bf, yf, B, Cin, H, W, Cout, dilation, x_bstride, y_bstride, leaky_slope, do_leaky, st, 0, nblk32 / 2);
        return launch_wino<1>(xf, uf, bf, yf, B, Cin, H, W, Cout, dilation, x_bstride, y_bstride, leaky_slope, do_leaky, st, 0, nblk32);
    }
    // ... when the narrowest launch still covers the chip; otherwise one launch of the widest tile that divides CoutP
    const int Hs = (H + dilation - 1) / dilation, Ws = (W + dilation - 1) / dilation;
    const int64_t wg32 = (int64_t)B * ((Ws + kTW - 1) / kTW) * ((Hs + 4 * kGH - 1) / (4 * kGH)) * dilation * dilation;   // 32-cout launch
    const int64_t wg64 = (int64_t)B * ((Ws + kTW - 1) / kTW) * ((Hs + 2 * kGH - 1) / (2 * kGH)) * dilation * dilation;
    const bool ragged = (nblk32 % 4) != 0 && nblk32 != 1 && nblk32 != 2;
    if (ragged && ((nblk32 % 2) ? wg32 : wg64) < 256) {
        if (nblk32 % 2 == 0)
            return launch_wino<2>(xf, uf, bf, yf, B, Cin, H, W, Cout, dilation, x_bstride, y_bstride, leaky_slope, do_leaky, st, 0, nblk32 / 2);
        return launch_wino<1>(xf, uf, bf, yf, B, Cin, H, W, Cout, dilation, x_bstride, y_bstride, leaky_slope, do_leaky, st, 0, nblk32);
    }
    if (nblk32 >= 4) {
        if (const int rc = launch_wino<4>(xf, uf, bf, yf, B, Cin, H, W, Cout, dilation, x_bstride, y_bstride, leaky_slope, do_leaky, st, 0, nblk32 / 4)) return rc;
        co0 = nblk32 / 4 * 128;
    }
    if ((nblk32 % 4) >= 2) {
        if (const int rc = launch_wino<2>(xf, uf, bf, yf, B, Cin, H, W, Cout, dilation, x_bstride, y_bstride, leaky_slope, do_leaky, st, co0, 1)) return rc;
        co0 += 64;
    }
    if (nblk32 % 2)
        return launch_wino<1>(xf, uf, bf, yf, B, Cin, H, W, Cout, dilation, x_bstride, y_bstride, leaky_slope, do_leaky, st, co0, 1);
    return PWC_OK;
}
