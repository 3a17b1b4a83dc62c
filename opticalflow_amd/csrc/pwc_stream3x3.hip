// Streaming 3x3-window kernel for the 2-channel layers that read a whole dense-block arena:
//   HEAD   : predict_flowL = Conv2d(Cin, 2, 3x3, pad 1)                (reference models/PWCNet.py:32-33)
//   UPFEAT : upfeatL / deconvL = ConvTranspose2d(Cin, 2, k4, s2, p1)   (PWCNet.py:35-36)
// Both consume the SAME 3x3 input window per pixel, produce only 2 channels and are bound by streaming
// [B,Cin,H,W] from HBM once (1 GB for predict_flow2 at batch 16), so they share one structure and can run
// FUSED in one pass over the arena (mode HEAD|UPFEAT: predict_flowL + upfeatL, PWCNet.py:207+209 etc.).
//
//   * workgroup = 256 threads = an 8-row x 128-col tile of input pixels; thread = 4 consecutive pixels;
//   * the tile (+1 halo; columns start 4 left of the tile so every 16-byte piece is aligned) streams per
//     4-channel chunk through a 4-deep LDS ring filled by buffer_load_dwordx4 ... lds issued from inline asm
//     (pwc_common.h): three chunks in flight, counted vmcnt, one barrier per chunk; zero padding and the ragged
//     last chunk come from the buffer range check;
//   * the chunk's filter taps travel in the SAME ring slot (the last two 1-KiB strips of the slot, which hold no
//     tile data, carry 80 + 128 floats of weights), and are read back as LDS broadcasts: with one wave per SIMD,
//     per-channel scalar loads from global were fully exposed (3x slower).  The tap DMA is dword-granular with
//     per-lane source offsets that INTERLEAVE the two output channels ([tap][co]), so that
//   * per channel a thread reads its 3 x 6 window (ds_read_b128 + 2 ds_read_b32 per row) and does 36 (HEAD)
//     and/or 64 (UPFEAT) v_pk_fma_f32 on (co0, co1) accumulator pairs: input broadcast x tap pair.  Plain
//     v_fma_f32 runs at half that rate on gfx950 and these kernels are VALU-bound once the stream is hidden;
//   * epilogue: bias, optional LeakyReLU / residual (HEAD), 16-byte stores.
// Needs W % 4 == 0, W >= 64 (PWC_STREAM_MINW) and 16-byte aligned tensors; the dispatchers in pwc_conv.hip / pwc_deconv.hip
// use other kernels otherwise.
#include "pwc_common.h"
#include <cstdlib>

#ifdef PWC_STREAM_NT_LOAD          // experiment: stream the arena with the non-temporal policy
#define PWC_STREAM_DMA pwc::dma_b128_nt
#else
#define PWC_STREAM_DMA pwc::dma_b128
#endif

#ifndef PWC_STREAM_EXP
#define PWC_STREAM_EXP 0           // timing experiments (results invalid): 1 = consumers skip the arithmetic, 2 = every fetch reads chunk 0, 4 = no LDS-DMA inside the loop
#endif

namespace {

using pwc::leaky;

constexpr int kCK = 4;                  // channels per chunk
#ifndef PWC_STREAM_RING
#define PWC_STREAM_RING 3
#endif
constexpr int kRing = PWC_STREAM_RING;  // 3 slots of 24 KiB (8-row tiles) -> two workgroups per CU; the counted vmcnt has 6 bits, so
                                        // an 8-row tile (26 VMEM instructions per chunk) cannot keep more than two chunks in flight
constexpr int kTW = 128;
constexpr int kPitch = kTW + 8;         // floats: cols x0-4 .. x0+131
constexpr int kQuads = kPitch / 4;      // 34 pieces per row
constexpr int kHeadWRow = 20;           // global: {co0: 9 taps, 0, co1: 9 taps, 0} per channel; LDS: [tap][co] + 2 zeros
constexpr int kUpWRow = 32;             // global: nn layout [ci][co][4][4]; LDS: [ky*4+kx][co]
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0x80000000u;
static_assert(kRing >= 3 && kRing <= 5, "ring depth");

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// Tile geometry.  TH rows x 128 columns per workgroup; KS consumer groups of TH*32 threads share the tile, group k
// accumulating channels k*(4/KS).. of every chunk (partial sums meet in LDS after the last chunk).  <8,1> is the
// streaming shape for large maps; <4,4> quarters the per-wave fma chain and doubles the workgroup count for maps that
// give fewer than 256 tiles (level 3 at batch 16: 112 -> 224 workgroups, 2 waves per SIMD instead of 1).
template <int TH, int KS>
struct Cfg {
    static constexpr int kTH = TH, kKS = KS;
    static constexpr int kPix = TH * 32;                     // threads per group; thread = 4 consecutive pixels
    static constexpr int kThreads = kPix * KS;               // consumer threads; one more wave runs the DMA ring
    static constexpr int kLoaderWave = kThreads / 64;
    static constexpr int kBlockThreads = kThreads + 64;
    static constexpr int kRows = TH + 2;
    static constexpr int kPieces = kCK * kRows * kQuads;     // 16-byte pieces of one chunk's tile (1360 for TH = 8)
    static constexpr int kTileInstr = (kPieces + 63) / 64;   // 1-KiB strips (= DMA instructions) carrying the tile (22)
    static constexpr int kInstr = kTileInstr + 2;            // + one strip of head taps + one of upfeat taps
    static constexpr int kVmem = kInstr + 2;                 // VMEM instructions per chunk: tile b128s + 4 x b32 (taps)
    static constexpr int kBuf = kInstr * 256;                // floats per ring slot (24 KiB for TH = 8)
    static constexpr int kHeadWOff = kTileInstr * 256;       // floats
    static constexpr int kUpWOff = (kTileInstr + 1) * 256;
    static constexpr int kCPerGroup = kCK / KS;
    static_assert(kPix % 64 == 0 && kCK % KS == 0, "groups are whole waves and split the chunk evenly");
    static_assert(kPieces * 4 <= kHeadWOff, "weights strip overlaps the tile");
};

constexpr int MODE_HEAD = 1, MODE_UPFEAT = 2;

// Loader wave: start the LDS-DMA of one chunk.  Instructions 0..21 carry the tile (pieces 64*i + lane); then four
// dword DMAs carry the taps into strips 22 (head, 80 floats) and 23 (upfeat, 128 floats), re-ordered to [tap][co]
// by their per-lane source offsets (woff).  A mode that is off still issues its two (all out-of-range) DMAs so that
// the counted vmcnt is the same for every instantiation.
template <int MODE, typename G>
__device__ __forceinline__ void issue(const float *xb, const float *hw, const float *uw, int chunk, int Cin, int plane,
                                      float *buf, const unsigned (&off)[G::kTileInstr], const unsigned (&woff)[4]) {
    constexpr int kInstr = G::kInstr;
    const int c0 = (PWC_STREAM_EXP & 2) ? 0 : chunk * kCK;
    const int cvalid = min(kCK, Cin - c0);
    const pwc::v4i32 r = pwc::make_rsrc(xb + (int64_t)c0 * plane, cvalid * plane * 4);
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)pwc::lds_addr(buf));
#pragma unroll
    for (int i = 0; i < kInstr - 2; ++i) PWC_STREAM_DMA(r, base + i * 1024, off[i]);
    const pwc::v4i32 rh = (MODE & MODE_HEAD) ? pwc::make_rsrc(hw + (int64_t)c0 * kHeadWRow, cvalid * kHeadWRow * 4) : r;
    pwc::dma_b32(rh, base + (kInstr - 2) * 1024, woff[0]);
    pwc::dma_b32(rh, base + (kInstr - 2) * 1024 + 256, woff[1]);
    const pwc::v4i32 ru = (MODE & MODE_UPFEAT) ? pwc::make_rsrc(uw + (int64_t)c0 * kUpWRow, cvalid * kUpWRow * 4) : r;
    pwc::dma_b32(ru, base + (kInstr - 1) * 1024, woff[2]);
    pwc::dma_b32(ru, base + (kInstr - 1) * 1024 + 256, woff[3]);
}

template <int MODE, int TH, int KS>
__global__ void __launch_bounds__((Cfg<TH, KS>::kBlockThreads))
stream3x3_kernel(const float *__restrict__ x, int Cin, int H, int W, int tiles_x, int tiles_y, int64_t bsx,
                 // HEAD: w packed [Cin][20], y [B,2,H,W]
                 const float *__restrict__ hw, const float *__restrict__ hbias, const float *__restrict__ residual,
                 float *__restrict__ hy, int64_t bshy, int64_t bsr, float slope, int do_leaky,
                 // UPFEAT: w [Cin][2][16], y [B,2,2H,2W]
                 const float *__restrict__ uw, const float *__restrict__ ubias, float *__restrict__ uy, int64_t bsuy,
                 // Cin slices (nslice > 1): grid item v = image v / nslice, channels [s * cslice, min(Cin, (s + 1) * cslice)) with
                 // s = v % nslice; raw partial sums (no bias / activation / residual) go to item v of hy / uy = the caller's workspace
                 int nslice, int cslice) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using G = Cfg<TH, KS>;
    constexpr int kTH = G::kTH, kRows = G::kRows, kPieces = G::kPieces, kTileInstr = G::kTileInstr, kVmem = G::kVmem;
    constexpr int kBuf = G::kBuf, kHeadWOff = G::kHeadWOff, kUpWOff = G::kUpWOff, kPix = G::kPix;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int grp = __builtin_amdgcn_readfirstlane(tid / kPix);     // consumer group (channel slice of every chunk)
    const int t = tid % kPix;
    const int ty = t >> 5;              // row inside the tile
    const int tx = t & 31;              // 0..31 group of 4 pixels
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);   // an XCD takes a contiguous run of tiles
    const int bx = bid % tiles_x;
    bid /= tiles_x;
    const int by = bid % tiles_y;
    const int b = bid / tiles_y;                                    // output item (image, or image x slice)
    const int x0 = bx * kTW;
    const int y0 = by * kTH;
    const int plane = H * W;

    const bool sliced = nslice > 1;
    const int c_lo = sliced ? (b % nslice) * cslice : 0;
    const float *xb = x + (int64_t)(sliced ? b / nslice : b) * bsx + (int64_t)c_lo * plane;
    if (sliced) {
        Cin = min(cslice, Cin - c_lo);
        if (MODE & MODE_HEAD) hw += (int64_t)c_lo * kHeadWRow;
        if (MODE & MODE_UPFEAT) uw += (int64_t)c_lo * kUpWRow;
    }
    const int nchunks = (Cin + kCK - 1) / kCK;

    if (wave == G::kLoaderWave) {
        // ================= producer wave: runs the ring, two chunks ahead of the consumers ==================
#ifndef PWC_STREAM_NO_PRIO
        __builtin_amdgcn_s_setprio(3);     // as in the correlation kernel: the ring's issue slots come first
#endif
        unsigned off[kTileInstr];
#pragma unroll
        for (int i = 0; i < kTileInstr; ++i) {
            const int p = i * 64 + lane;
            const int c = p / (kRows * kQuads);
            const int rem = p % (kRows * kQuads);
            const int r = rem / kQuads;
            const int q = rem % kQuads;
            const int iy = y0 - 1 + r;
            const int ix = x0 - 4 + 4 * q;
            const bool ok = (p < kPieces) && (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);   // W % 4 == 0
            off[i] = ok ? (unsigned)(c * plane + iy * W + ix) * 4u : kOOB;
        }
        // tap fetches: LDS float d of the strip <- global float src(d), interleaving the two output channels
        unsigned woff[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int d = h * 64 + lane;
            {   // head: d = c*20 + tap*2 + co  <-  c*20 + co*10 + tap   (d%20 >= 18: zero pad)
                const int c = d / kHeadWRow, r = d % kHeadWRow;
                const bool ok = (MODE & MODE_HEAD) && (d < kCK * kHeadWRow) && (r < 18);
                woff[h] = ok ? (unsigned)(c * kHeadWRow + (r & 1) * 10 + (r >> 1)) * 4u : kOOB;
            }
            {   // upfeat: d = c*32 + t*2 + co  <-  c*32 + co*16 + t
                const int c = d / kUpWRow, r = d % kUpWRow;
                const bool ok = (MODE & MODE_UPFEAT) && (d < kCK * kUpWRow);
                woff[2 + h] = ok ? (unsigned)(c * kUpWRow + (r & 1) * 16 + (r >> 1)) * 4u : kOOB;
            }
        }
#pragma unroll
        for (int k = 0; k < kRing - 1; ++k)
            if (k < nchunks) issue<MODE, G>(xb, hw, uw, k, Cin, plane, smem + k * kBuf, off, woff);
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            // chunk `chunk` has landed; up to kRing - 2 younger ones stay in flight
            constexpr bool kRingOk = (kRing - 2) * kVmem <= 63;          // vmcnt is a 6-bit counter; launch_cfg refuses the others
            const int younger = min(kRing - 2, nchunks - 1 - chunk);
            if (younger >= 3)      wait_vm<(kRing >= 5 && kRingOk ? 3 : 0) * kVmem>();
            else if (younger == 2) wait_vm<(kRing >= 4 && kRingOk ? 2 : 0) * kVmem>();
            else if (younger == 1) wait_vm<kVmem>();
            else                   wait_vm<0>();
            __builtin_amdgcn_s_barrier();      // consumers may read slot chunk%3; they are done with (chunk-1)%3
            if (chunk + kRing - 1 < nchunks && !(PWC_STREAM_EXP & 4))
                issue<MODE, G>(xb, hw, uw, chunk + kRing - 1, Cin, plane, smem + ((chunk + kRing - 1) % kRing) * kBuf, off, woff);
        }
        if constexpr (KS > 1) {            // the two barriers of the consumers' reduction
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    f32x2 hacc[4];                      // HEAD:   [px] -> (co0, co1)
    f32x2 uacc[2][8];                   // UPFEAT: [out row parity][out col 0..7] -> (co0, co1)   (8 output cols for 4 input px)
#pragma unroll
    for (int p = 0; p < 4; ++p) hacc[p] = f32x2{0.f, 0.f};
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int k = 0; k < 8; ++k) uacc[py][k] = f32x2{0.f, 0.f};

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float *cur = smem + (chunk % kRing) * kBuf;
        // channels past Cin in the last chunk: tile AND taps were range-checked to 0, so they add exactly 0
#pragma unroll
        for (int cc = 0; cc < ((PWC_STREAM_EXP & 1) ? 0 : G::kCPerGroup); ++cc) {
            const int c = grp * G::kCPerGroup + cc;
            // window columns -1 .. 4 of the thread's 4 pixels = staged floats 4 tx + 3 .. 4 tx + 8: three ALIGNED 16-byte reads
            // (conflict-free, 4 LDS cycles each).  Reading exactly the six floats (b32 + b128 + b32) is what the first version
            // asked for; the compiler merged them into three ds_read2_b32 at a 16-byte lane stride = 4-way bank conflicts on
            // the 32-bank path, 96 LDS cycles per row instead of 12 -- the kernel was LDS-bound at 0.35 of HBM (round-3 ablation).
            const float *t = cur + (c * kRows + ty) * kPitch + 4 * tx;
            float v[3][6];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const f32x4 *row = reinterpret_cast<const f32x4 *>(t + r * kPitch);
                f32x4 l = row[0], m = row[1], h = row[2];
                asm volatile("" : "+v"(l), "+v"(h));          // keep the edge reads whole: narrowed to b64 / b32 they conflict again
                v[r][0] = l[3];
                v[r][1] = m[0]; v[r][2] = m[1]; v[r][3] = m[2]; v[r][4] = m[3];
                v[r][5] = h[0];
            }
            if constexpr (MODE & MODE_HEAD) {
                const float4 *wq = reinterpret_cast<const float4 *>(cur + kHeadWOff + c * kHeadWRow);   // LDS broadcast
                const float4 q0 = wq[0], q1 = wq[1], q2 = wq[2], q3 = wq[3];
                const float2 q4 = *reinterpret_cast<const float2 *>(cur + kHeadWOff + c * kHeadWRow + 16);
                const f32x2 wk[9] = {{q0.x, q0.y}, {q0.z, q0.w}, {q1.x, q1.y}, {q1.z, q1.w}, {q2.x, q2.y},
                                     {q2.z, q2.w}, {q3.x, q3.y}, {q3.z, q3.w}, {q4.x, q4.y}};          // [tap] -> (co0, co1)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            const float a = v[ky][p + kx];
                            hacc[p] = __builtin_elementwise_fma(f32x2{a, a}, wk[ky * 3 + kx], hacc[p]);
                        }
            }
            if constexpr (MODE & MODE_UPFEAT) {
                // oy = 2*iy' - 1 + ky: out row 2*iy+py takes py=0: (iy-1,ky=3),(iy,ky=1); py=1: (iy,ky=2),(iy+1,ky=0)
                // window row index a: iy-1+a ; window col index e: ix-1+e, pixel p sits at e = p+1
                const float4 *wq = reinterpret_cast<const float4 *>(cur + kUpWOff + c * kUpWRow);       // LDS broadcast
                f32x2 k[16];                                                                            // [ky*4+kx] -> (co0, co1)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 q = wq[j];
                    k[2 * j] = f32x2{q.x, q.y};
                    k[2 * j + 1] = f32x2{q.z, q.w};
                }
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const int kya = py ? 2 : 3, kyb = py ? 0 : 1;        // window rows a = py, py+1
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
                            const int kxa = px ? 2 : 3, kxb = px ? 0 : 1;    // window cols e = p+px, p+px+1
                            const float a0 = v[py][p + px], a1 = v[py][p + px + 1];
                            const float b0 = v[py + 1][p + px], b1 = v[py + 1][p + px + 1];
                            f32x2 s = uacc[py][2 * p + px];
                            s = __builtin_elementwise_fma(f32x2{a0, a0}, k[kya * 4 + kxa], s);
                            s = __builtin_elementwise_fma(f32x2{a1, a1}, k[kya * 4 + kxb], s);
                            s = __builtin_elementwise_fma(f32x2{b0, b0}, k[kyb * 4 + kxa], s);
                            s = __builtin_elementwise_fma(f32x2{b1, b1}, k[kyb * 4 + kxb], s);
                            uacc[py][2 * p + px] = s;
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // LDS reads done before the ring slot is refilled
    }

    if constexpr (KS > 1) {
        // partial sums of groups 1.. -> LDS (the ring is free once every consumer passed the first barrier) -> group 0
        constexpr int kAcc = ((MODE & MODE_HEAD) ? 8 : 0) + ((MODE & MODE_UPFEAT) ? 32 : 0);
        constexpr int kHeadAcc = (MODE & MODE_HEAD) ? 8 : 0;
        __builtin_amdgcn_s_barrier();
        if (grp > 0) {
            float *red = smem + (grp - 1) * kAcc * kPix + t;
            if constexpr (MODE & MODE_HEAD) {
#pragma unroll
                for (int p = 0; p < 4; ++p) { red[(2 * p) * kPix] = hacc[p][0]; red[(2 * p + 1) * kPix] = hacc[p][1]; }
            }
            if constexpr (MODE & MODE_UPFEAT) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    red[(kHeadAcc + 2 * j) * kPix] = uacc[j >> 3][j & 7][0];
                    red[(kHeadAcc + 2 * j + 1) * kPix] = uacc[j >> 3][j & 7][1];
                }
            }
        }
        __builtin_amdgcn_s_barrier();
        if (grp > 0) return;
#pragma unroll
        for (int k = 1; k < KS; ++k) {
            const float *red = smem + (k - 1) * kAcc * kPix + t;
            if constexpr (MODE & MODE_HEAD) {
#pragma unroll
                for (int p = 0; p < 4; ++p) { hacc[p][0] += red[(2 * p) * kPix]; hacc[p][1] += red[(2 * p + 1) * kPix]; }
            }
            if constexpr (MODE & MODE_UPFEAT) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    uacc[j >> 3][j & 7][0] += red[(kHeadAcc + 2 * j) * kPix];
                    uacc[j >> 3][j & 7][1] += red[(kHeadAcc + 2 * j + 1) * kPix];
                }
            }
        }
    }

    const int oy = y0 + ty;
    const int ox = x0 + 4 * tx;
    if (oy >= H || ox >= W) return;
    if constexpr (MODE & MODE_HEAD) {
#pragma unroll
        for (int co = 0; co < 2; ++co) {
            const int64_t o = (int64_t)co * plane + (int64_t)oy * W + ox;
            const float bv = sliced ? 0.f : hbias[co];
            float4 v = make_float4(hacc[0][co] + bv, hacc[1][co] + bv, hacc[2][co] + bv, hacc[3][co] + bv);
            if (do_leaky) { v.x = leaky(v.x, slope); v.y = leaky(v.y, slope); v.z = leaky(v.z, slope); v.w = leaky(v.w, slope); }
            if (residual) {
                const float4 rr = *reinterpret_cast<const float4 *>(residual + (int64_t)b * bsr + o);
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            *reinterpret_cast<float4 *>(hy + (int64_t)b * bshy + o) = v;
        }
    }
    if constexpr (MODE & MODE_UPFEAT) {
        const int Wo = 2 * W;
#pragma unroll
        for (int co = 0; co < 2; ++co) {
            const float bv = sliced ? 0.f : ubias[co];
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                float *p = uy + (int64_t)b * bsuy + (int64_t)co * 4 * plane + (int64_t)(2 * oy + py) * Wo + 2 * ox;
                *reinterpret_cast<float4 *>(p) = make_float4(uacc[py][0][co] + bv, uacc[py][1][co] + bv,
                                                             uacc[py][2][co] + bv, uacc[py][3][co] + bv);
                *reinterpret_cast<float4 *>(p + 4) = make_float4(uacc[py][4][co] + bv, uacc[py][5][co] + bv,
                                                                 uacc[py][6][co] + bv, uacc[py][7][co] + bv);
            }
        }
    }
}

// Sum of the Cin slices' partial sums in fixed slice order + bias (+ LeakyReLU, + residual for the head): thread = four consecutive
// outputs of one image; the head's 2 x H x W come first, then the deconvolution's 2 x 2H x 2W.
__global__ void __launch_bounds__(256)
stream3x3_slice_reduce_kernel(const float *__restrict__ ph, const float *__restrict__ pu, int nslice, int plane, int head_quads, int quads,
                              const float *__restrict__ hbias, const float *__restrict__ residual, float *__restrict__ hy, int64_t bshy,
                              int64_t bsr, float slope, int do_leaky, const float *__restrict__ ubias, float *__restrict__ uy, int64_t bsuy) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (q >= quads) return;
    const bool head = q < head_quads;
    const int o = 4 * (head ? q : q - head_quads);                   // offset inside the image's [2][H][W] / [2][2H][2W]
    const int64_t per = head ? 2 * (int64_t)plane : 8 * (int64_t)plane;
    const float *p = (head ? ph : pu) + (int64_t)b * nslice * per + o;
    float4 v = *reinterpret_cast<const float4 *>(p);
    for (int k = 1; k < nslice; ++k) {
        const float4 t = *reinterpret_cast<const float4 *>(p + k * per);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const float bv = head ? hbias[o / plane] : ubias[o / (4 * plane)];
    v.x += bv; v.y += bv; v.z += bv; v.w += bv;
    if (head) {
        if (do_leaky) { v.x = leaky(v.x, slope); v.y = leaky(v.y, slope); v.z = leaky(v.z, slope); v.w = leaky(v.w, slope); }
        if (residual) {
            const float4 rr = *reinterpret_cast<const float4 *>(residual + (int64_t)b * bsr + o);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        *reinterpret_cast<float4 *>(hy + (int64_t)b * bshy + o) = v;
    } else {
        *reinterpret_cast<float4 *>(uy + (int64_t)b * bsuy + o) = v;
    }
}

// Cin slices for launches of a few tiles (predict_flow2 of a single pair is 28 8-row tiles, refused by
// stream3x3_ok, and ran 81 + 5 us on the split-K MFMA kernel with 2 of 16 couts used -- 5 % of that forward): a tile's workgroup walks
// Cin chunk by chunk behind one barrier per chunk, so few workgroups x 142 chunks are a latency chain.  The launch is cut along Cin
// into S slices, (image, slice) pairs run as S x as many <TH4,KS4> workgroups, and stream3x3_slice_reduce_kernel adds the partial sums
// in fixed order.  Option "stream_slice_wgs" = workgroups to reach (0 = off).  The head alone: only under stream3x3_ok's 64 tiles
// (profiles/r04_ab_stream_slices.txt: batch 1 +2.3 % in the median, batch 2 +0.7 %; level 2 at batch 4-8, where the one-pass kernel
// already runs: nothing).  Head + upfeat: up to 256 tiles (profiles/r04_microbench_head_slices.txt: level 4 at batch 16 115 -> 84 us).
struct SlicePlan { int nslice, cslice; };
inline SlicePlan slice_plan(int B, int Cin, int H, int W, int max_tiles8 = 64) {
    SlicePlan p{1, Cin};
    const int target = pwc::option(pwc::OPT_STREAM_SLICE_WGS);
    const int64_t nblk8 = (int64_t)B * ((W + kTW - 1) / kTW) * ((H + 7) / 8);
    const int64_t tiles4 = (int64_t)B * ((W + kTW - 1) / kTW) * ((H + 3) / 4);
    if (target <= 0 || nblk8 >= max_tiles8 || tiles4 * 4 > target * 3) return p;   // the one-pass kernel's range / within 3/4 of the target
    int S = (int)((target + tiles4 - 1) / tiles4);
    if (S > 16) S = 16;
    int cs = ((Cin + S - 1) / S + kCK - 1) / kCK * kCK;                           // whole chunks, at least eight of them
    if (cs < 32) cs = 32;
    S = (Cin + cs - 1) / cs;
    if (S > 1) { p.nslice = S; p.cslice = cs; }
    return p;
}
template <int MODE> inline int64_t slice_ws_bytes(const SlicePlan &p, int B, int H, int W) {
    if (p.nslice <= 1) return 0;
    return (int64_t)p.nslice * B * (((MODE & MODE_HEAD) ? 2 : 0) + ((MODE & MODE_UPFEAT) ? 8 : 0)) * H * W * (int64_t)sizeof(float);
}

template <int MODE, int TH, int KS>
int launch_cfg(const float *x, int B, int Cin, int H, int W, int64_t bsx,
               const float *hw, const float *hbias, const float *residual, float *hy, int64_t bshy, int64_t bsr,
               float slope, int do_leaky, const float *uw, const float *ubias, float *uy, int64_t bsuy, hipStream_t st,
               int nslice = 1, int cslice = 0) {
    using G = Cfg<TH, KS>;
    if constexpr ((kRing - 2) * G::kVmem > 63) return PWC_EUNSUPPORTED;      // experiment builds with a deeper ring
    const int tiles_x = (W + kTW - 1) / kTW;
    const int tiles_y = (H + TH - 1) / TH;
    const int64_t nblk = (int64_t)B * nslice * tiles_x * tiles_y;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "stream3x3: grid too large");
    constexpr int kAcc = ((MODE & MODE_HEAD) ? 8 : 0) + ((MODE & MODE_UPFEAT) ? 32 : 0);
    constexpr int ring = kRing * G::kBuf * 4, red = (KS - 1) * kAcc * G::kPix * 4;
    constexpr int smem = ring > red ? ring : red;
    auto kern = stream3x3_kernel<MODE, TH, KS>;
    static pwc::LdsAttrOnce attr;       // per instantiation, tracked per device
    if (const int rc = pwc::ensure_lds_attr(attr, reinterpret_cast<const void *>(kern), smem, "stream3x3")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(G::kBlockThreads), smem, st,
                       x, Cin, H, W, tiles_x, tiles_y, bsx, hw, hbias, residual, hy, bshy, bsr, slope, do_leaky,
                       uw, ubias, uy, bsuy, nslice, cslice);
    return pwc::check_launch("stream3x3_kernel");
}

// PWC_STREAM_CFG=81|42|44 pins the tile shape (experiments); default: <4,4> when 8-row tiles would leave CUs idle.
inline int stream_cfg_override() {
    static const int v = [] { const char *e = getenv("PWC_STREAM_CFG"); return e ? atoi(e) : 0; }();
    return v;
}

template <int MODE>
int launch(const float *x, int B, int Cin, int H, int W, int64_t bsx,
           const float *hw, const float *hbias, const float *residual, float *hy, int64_t bshy, int64_t bsr,
           float slope, int do_leaky, const float *uw, const float *ubias, float *uy, int64_t bsuy, hipStream_t st,
           float *ws = nullptr, int64_t ws_bytes = 0) {
    const int64_t nblk8 = (int64_t)B * ((W + kTW - 1) / kTW) * ((H + 7) / 8);
    // head + upfeat (100 v_pk_fma_f32 per channel and thread: a workgroup keeps its CU's VALUs busy, so fewer workgroups than CUs leave
    // the rest of the chip idle) is sliced up to 256 tiles; the head alone only below the one-pass kernel's range
    const SlicePlan sp = slice_plan(B, Cin, H, W, (MODE & MODE_UPFEAT) ? 256 : 64);
    if (sp.nslice > 1 && ws && pwc::aligned16(ws) && ws_bytes >= slice_ws_bytes<MODE>(sp, B, H, W) && stream_cfg_override() == 0) {
        const int plane = H * W;
        float *ph = ws, *pu = ws + ((MODE & MODE_HEAD) ? (int64_t)sp.nslice * B * 2 * plane : 0);
        if (const int rc = launch_cfg<MODE, 4, 4>(x, B, Cin, H, W, bsx, hw, nullptr, nullptr, ph, 2 * (int64_t)plane, 0, 0.f, 0,
                                                  uw, nullptr, pu, 8 * (int64_t)plane, st, sp.nslice, sp.cslice)) return rc;
        const int head_quads = (MODE & MODE_HEAD) ? plane / 2 : 0, quads = head_quads + ((MODE & MODE_UPFEAT) ? 2 * plane : 0);
        hipLaunchKernelGGL(stream3x3_slice_reduce_kernel, dim3((unsigned)((quads + 255) / 256), (unsigned)B), dim3(256), 0, st,
                           ph, pu, sp.nslice, plane, head_quads, quads, hbias, residual, hy, bshy, bsr, slope, do_leaky, ubias, uy, bsuy);
        return pwc::check_launch("stream3x3_slice_reduce_kernel");
    }
    int cfg = stream_cfg_override();
    if (cfg != 81 && cfg != 42 && cfg != 44 && cfg != 41) cfg = nblk8 < 256 ? 44 : 81;
#define PWC_STREAM_GO(TH, KS) return launch_cfg<MODE, TH, KS>(x, B, Cin, H, W, bsx, hw, hbias, residual, hy, bshy, bsr, slope, do_leaky, uw, ubias, uy, bsuy, st)
    if (cfg == 44) PWC_STREAM_GO(4, 4);
    if (cfg == 42) PWC_STREAM_GO(4, 2);
    if (cfg == 41) PWC_STREAM_GO(4, 1);
    PWC_STREAM_GO(8, 1);
#undef PWC_STREAM_GO
}

inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

namespace pwc_conv {

// true when the streaming kernel applies to this geometry (the callers keep their other kernels otherwise)
int64_t stream3x3_head_workspace_bytes(int B, int Cin, int H, int W) { return slice_ws_bytes<MODE_HEAD>(slice_plan(B, Cin, H, W), B, H, W); }

// the flow head alone with Cin slices: launches under the 64 tiles stream3x3_ok asks for (predict_flow2 of one or two pairs) fill the
// chip through their slices -- needs the caller's workspace
int64_t stream3x3_head_upfeat_workspace_bytes(int B, int Cin, int H, int W) {
    return slice_ws_bytes<MODE_HEAD | MODE_UPFEAT>(slice_plan(B, Cin, H, W, 256), B, H, W);
}

// head + upfeat below stream3x3_ok's 64 tiles: possible with Cin slices and the caller's workspace (the plan decides whether it beats
// the 10-channel convolution there: option "head_sliced_min_tiles", read by the Python engine)
bool stream3x3_head_upfeat_sliced_ok(int B, int Cin, int H, int W, const void *x, int64_t bsx, const void *ws, int64_t ws_bytes) {
    static const int min_w = [] { const char *e = getenv("PWC_STREAM_MINW"); return (e && *e) ? atoi(e) : 64; }();
    const SlicePlan sp = slice_plan(B, Cin, H, W, 256);
    return sp.nslice > 1 && ws && al16(ws) && ws_bytes >= slice_ws_bytes<MODE_HEAD | MODE_UPFEAT>(sp, B, H, W) && (W % 4 == 0) && (W >= min_w) &&
           al16(x) && (bsx % 4 == 0) && ((int64_t)H * W * kCK * 4 < 0x7fffffffLL) && (int64_t)B * ((W + kTW - 1) / kTW) * ((H + 7) / 8) >= 4;
}

bool stream3x3_head_sliced_ok(int B, int Cin, int H, int W, const void *x, int64_t bsx, const void *ws, int64_t ws_bytes) {
    static const int min_w = [] { const char *e = getenv("PWC_STREAM_MINW"); return (e && *e) ? atoi(e) : 64; }();
    const SlicePlan sp = slice_plan(B, Cin, H, W);
    return sp.nslice > 1 && ws && al16(ws) && ws_bytes >= slice_ws_bytes<MODE_HEAD>(sp, B, H, W) && (W % 4 == 0) && (W >= min_w) && al16(x) &&
           (bsx % 4 == 0) && ((int64_t)H * W * kCK * 4 < 0x7fffffffLL) && (int64_t)B * ((W + kTW - 1) / kTW) * ((H + 7) / 8) >= 8;
}

bool stream3x3_ok(int B, int Cin, int H, int W, const void *x, int64_t bsx) {
    static const int min_w = [] { const char *e = getenv("PWC_STREAM_MINW"); return (e && *e) ? atoi(e) : 64; }();    // 64-column maps (level 4) run half-filled 128-column tiles: still ahead of split-K MFMA head + deconv (-45 us)
    return (W % 4 == 0) && (W >= min_w) && al16(x) && (bsx % 4 == 0) && ((int64_t)H * W * kCK * 4 < 0x7fffffffLL) &&
           (int64_t)B * ((W + kTW - 1) / kTW) * ((H + 7) / 8) >= 64;    // fewer: the split-K MFMA head + deconv kernel win (24 measured: -1.6 % at batch 4)
}

// w = packed head taps [Cin][20] (tail of pwc_conv3x3_pack's buffer for Cout == 2)
int stream3x3_head(const float *x, const float *w, const float *bias, const float *residual, float *y,
                   int B, int Cin, int H, int W, int64_t bsx, int64_t bsy, int64_t bsr,
                   float slope, int do_leaky, hipStream_t st, float *ws, int64_t ws_bytes) {
    if (!al16(y) || (bsy % 4) || !al16(w) || (residual && (!al16(residual) || (bsr % 4)))) return PWC_EUNSUPPORTED;
    return launch<MODE_HEAD>(x, B, Cin, H, W, bsx, w, bias, residual, y, bsy, bsr, slope, do_leaky,
                             nullptr, nullptr, nullptr, 0, st, ws, ws_bytes);
}

// w = nn.ConvTranspose2d weight [Cin][2][4][4]
int stream3x3_upfeat(const float *x, const float *w, const float *bias, float *y,
                     int B, int Cin, int H, int W, int64_t bsx, int64_t bsy, hipStream_t st) {
    if (!al16(y) || (bsy % 4) || !al16(w)) return PWC_EUNSUPPORTED;
    return launch<MODE_UPFEAT>(x, B, Cin, H, W, bsx, nullptr, nullptr, nullptr, nullptr, 0, 0, 0.f, 0,
                               w, bias, y, bsy, st);
}

int stream3x3_head_upfeat(const float *x, int B, int Cin, int H, int W, int64_t bsx,
                          const float *hw, const float *hbias, float *hy, int64_t bshy,
                          const float *uw, const float *ubias, float *uy, int64_t bsuy, hipStream_t st, float *ws, int64_t ws_bytes) {
    if (!al16(hy) || (bshy % 4) || !al16(uy) || (bsuy % 4) || !al16(hw) || !al16(uw)) return PWC_EUNSUPPORTED;
    return launch<MODE_HEAD | MODE_UPFEAT>(x, B, Cin, H, W, bsx, hw, hbias, nullptr, hy, bshy, 0, 0.f, 0,
                                           uw, ubias, uy, bsuy, st, ws, ws_bytes);
}

}  // namespace pwc_conv

// timing-experiment mask this translation unit was built with (0 in the product; pwc_experiment_mask)
namespace pwc { int exp_mask_stream3x3() { return PWC_STREAM_EXP; } }
