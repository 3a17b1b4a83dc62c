// Half-precision correlation and warp on channel-blocked ("c8") activations [B][ceil(C/8)][H][W][8] -- the
// HBM-bound companions of the fp16 MFMA convolution (pwc_conv_f16.hip).  fp32 accumulation / coordinates.
//
//   corr81_c8_kernel : PWC-Net's cost volume (pad 4, kernel 1, max displacement 4, strides 1; reference
//                      models/correlation_package/correlation.py:12-40, correlation_cuda_kernel.cu:73-147) + fused
//                      scale and LeakyReLU, written as 11 channel groups (81 channels + 7 zeros) so that it can land
//                      directly in an arena slot.  One thread = one pixel x 81 displacements: channel-innermost data
//                      makes the inner product a chain of v_dot2_f32_f16 on 16-byte LDS reads; the in2 halo tile of a
//                      chunk of 4 channel groups is staged by LDS-DMA (zero padding = range check).
//   warp_c8_kernel   : PWCDCNet.warp (models/PWCNet.py:141-177), same coordinate arithmetic and mask rule as the fp32
//                      kernel (pwc_warp.hip); a tap is one 16-byte gather per channel group.
#include <stdlib.h>

#include "pwc_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned kOOB = 0x80000000u;

// ---- correlation -------------------------------------------------------------------------------------------
constexpr int kCTH = 8, kCTW = 32;                   // pixel tile
constexpr int kCHaloH = kCTH + 8, kCHaloW = kCTW + 8;
constexpr int kCGroups = 4;                           // channel groups (32 channels) staged per step
constexpr int kCPieces = kCGroups * kCHaloH * kCHaloW;           // 2560 16-byte pieces
constexpr int kCThreads = 256;
constexpr int kCInstr = kCPieces / kCThreads;          // 10 LDS-DMA instructions per wave and step
static_assert(kCPieces % kCThreads == 0, "tile pieces must divide evenly");

__device__ __forceinline__ float dot8(h8 a, h8 b, float acc) {
    acc = __builtin_amdgcn_fdot2(h2{a[0], a[1]}, h2{b[0], b[1]}, acc, false);
    acc = __builtin_amdgcn_fdot2(h2{a[2], a[3]}, h2{b[2], b[3]}, acc, false);
    acc = __builtin_amdgcn_fdot2(h2{a[4], a[5]}, h2{b[4], b[5]}, acc, false);
    acc = __builtin_amdgcn_fdot2(h2{a[6], a[7]}, h2{b[6], b[7]}, acc, false);
    return acc;
}

__global__ void __launch_bounds__(kCThreads)
corr81_c8_kernel(const _Float16 *__restrict__ in1, const _Float16 *__restrict__ in2, _Float16 *__restrict__ out,
                 int Cg, int H, int W, int tiles_x, int tiles_y, int64_t bs1, int64_t bs2, int64_t bso,
                 float scale, float slope, int do_leaky) {
    __shared__ __attribute__((aligned(16))) h8 tile[kCPieces];           // [g][row 16][col 40]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx * kCTW, y0 = ty * kCTH;
    const int plane = H * W;
    const int px = tid & 31, py = tid >> 5;
    const int x = x0 + px, y = y0 + py;
    const bool inside = (x < W) && (y < H);

    // per-lane source offsets of this wave's DMA instructions (pieces wave*64 + lane + 256*i)
    unsigned off[kCInstr];
#pragma unroll
    for (int i = 0; i < kCInstr; ++i) {
        const int p = i * kCThreads + wave * 64 + lane;
        const int g = p / (kCHaloH * kCHaloW);
        const int rem = p % (kCHaloH * kCHaloW);
        const int iy = y0 - 4 + rem / kCHaloW;
        const int ix = x0 - 4 + rem % kCHaloW;
        const bool ok = (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);
        off[i] = ok ? (unsigned)(g * plane + iy * W + ix) * 16u : kOOB;
    }

    float acc[81];
#pragma unroll
    for (int d = 0; d < 81; ++d) acc[d] = 0.f;

    const _Float16 *p1 = in1 + (int64_t)b * bs1;
    const _Float16 *p2 = in2 + (int64_t)b * bs2;
    for (int g0 = 0; g0 < Cg; g0 += kCGroups) {
        const int gv = min(kCGroups, Cg - g0);
        const pwc::v4i32 r2 = pwc::make_rsrc(p2 + (int64_t)g0 * plane * 8, gv * plane * 16);
        const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)pwc::lds_addr(tile)) + wave * 1024;
        __syncthreads();                                   // previous step's reads are done
#pragma unroll
        for (int i = 0; i < kCInstr; ++i) pwc::dma_b128(r2, base + i * (kCThreads * 16), off[i]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int g = 0; g < gv; ++g) {
            h8 a = {0, 0, 0, 0, 0, 0, 0, 0};
            if (inside) a = *reinterpret_cast<const h8 *>(p1 + ((int64_t)(g0 + g) * plane + (int64_t)y * W + x) * 8);
            const h8 *t = tile + (g * kCHaloH + py) * kCHaloW + px;
#pragma unroll
            for (int dy = 0; dy < 9; ++dy)
#pragma unroll
                for (int dx = 0; dx < 9; ++dx)
                    acc[dy * 9 + dx] = dot8(a, t[dy * kCHaloW + dx], acc[dy * 9 + dx]);
        }
    }
    if (!inside) return;
    _Float16 *po = out + (int64_t)b * bso + ((int64_t)y * W + x) * 8;
#pragma unroll
    for (int g = 0; g < 11; ++g) {
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = g * 8 + j;
            float v = 0.f;
            if (d < 81) {
                v = acc[d < 81 ? d : 0] * scale;
                if (do_leaky) v = pwc::leaky(v, slope);
            }
            o[j] = pwc::sat_half(v);
        }
        *reinterpret_cast<h8 *>(po + (int64_t)g * plane * 8) = o;
    }
}

// Small maps (pyramid levels 6-4): the tiled kernel above has only B*ceil(H/8)*ceil(W/32) workgroups, each walking all
// channel groups serially (level 6: 16 workgroups x 7 DMA round trips = 52 us).  Here one thread owns ONE output value
// (pixel, displacement) and streams the channel groups straight from L2: 88 threads per pixel, no LDS.
__global__ void __launch_bounds__(256)
corr81_c8_direct_kernel(const _Float16 *__restrict__ in1, const _Float16 *__restrict__ in2, _Float16 *__restrict__ out,
                        int Cg, int H, int W, int64_t total, int64_t bs1, int64_t bs2, int64_t bso,
                        float scale, float slope, int do_leaky) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int d = (int)(i % 88);                       // output channel 0..87 (81..87 = zero padding)
    int64_t t = i / 88;
    const int plane = H * W;
    const int pix = (int)(t % plane);
    const int b = (int)(t / plane);
    const int y = pix / W, x = pix - y * W;
    float acc = 0.f;
    if (d < 81) {
        const int y2 = y + d / 9 - 4, x2 = x + d % 9 - 4;
        if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) {
            const _Float16 *p1 = in1 + (int64_t)b * bs1 + (int64_t)pix * 8;
            const _Float16 *p2 = in2 + (int64_t)b * bs2 + ((int64_t)y2 * W + x2) * 8;
            for (int g = 0; g < Cg; ++g, p1 += (int64_t)plane * 8, p2 += (int64_t)plane * 8)
                acc = dot8(*reinterpret_cast<const h8 *>(p1), *reinterpret_cast<const h8 *>(p2), acc);
        }
        acc *= scale;
        if (do_leaky) acc = pwc::leaky(acc, slope);
    }
    out[(int64_t)b * bso + ((int64_t)(d >> 3) * plane + pix) * 8 + (d & 7)] = pwc::sat_half(acc);
}

// ---- warp ---------------------------------------------------------------------------------------------------
// ENTRY = the whole entry of a decoder level in one pass (pwc_level_entry_c8_f16).  The flow travels in fp32:
// `flow32` is the fp32 c8 output of the level above's head convolution (channels 0,1 of its group), the thread
// applies deconvL (ConvTranspose2d(2,2,k4,s2,p1), PWCNet.py:84,208) to it in fp32 -- 2x2 input pixels x 2 channels per
// output -- and warps with that fp32 up_flow; up_feat arrives as the 4-phase fp32 output of the 3x3 form of upfeatL
// (channel co*4 + py*2 + px).  (up_flow, up_feat) are also written, rounded to half, into channels 0..3 of the
// arena's flow group (they are inputs of the level's convolutions), and the thread copies its pixel of c1 into the arena.
struct LevelEntry {
    const float *flow32, *feat_phases, *dw, *db;       // dw: [ci 2][co 2][ky 4][kx 4] (nn.ConvTranspose2d layout), db: [2]
    const _Float16 *c1;
    _Float16 *fg, *c1_dst;
    int64_t bs_flow, bs_featp, bs_c1, bs_fg, bs_c1dst;
};


// up_flow = deconvL(flow) at output pixel (yy, xx) of image b, in fp32 (ConvTranspose2d(2,2,k4,s2,p1): 2x2 input pixels x 2 channels)
__device__ __forceinline__ void entry_up_flow(const LevelEntry &e, int b, int yy, int xx, int H, int W, float &acc0, float &acc1) {
    const int hh = H >> 1, wh = W >> 1;
    const int py = yy & 1, px = xx & 1;
    // output row 2Y+py takes input rows r0 = Y-1+py (kernel row 3-py) and r0+1 (kernel row 1-py); same along x
    const int r0 = (yy >> 1) - 1 + py, c0 = (xx >> 1) - 1 + px;
    const float *fb = e.flow32 + (int64_t)b * e.bs_flow;
    acc0 = e.db[0];
    acc1 = e.db[1];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int r = r0 + a, ky = 3 - py - 2 * a;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int cc = c0 + c, kx = 3 - px - 2 * c;
            if (r < 0 || r >= hh || cc < 0 || cc >= wh) continue;
            const float2 f = *reinterpret_cast<const float2 *>(fb + ((int64_t)r * wh + cc) * 8);
            acc0 = fmaf(f.x, e.dw[(0 * 2 + 0) * 16 + ky * 4 + kx], acc0);
            acc0 = fmaf(f.y, e.dw[(1 * 2 + 0) * 16 + ky * 4 + kx], acc0);
            acc1 = fmaf(f.x, e.dw[(0 * 2 + 1) * 16 + ky * 4 + kx], acc1);
            acc1 = fmaf(f.y, e.dw[(1 * 2 + 1) * 16 + ky * 4 + kx], acc1);
        }
    }
}

// bilinear taps of PWCDCNet.warp at pixel (xx, yy) displaced by (u, v): element offsets of the four taps inside a channel group's
// plane and their weights (zero outside the image or under the mask threshold).  Same arithmetic as pwc_warp.hip::make_taps
// (PWCNet.py:162-163 + grid_sample's un-normalisation).
struct Taps8 {
    int64_t o00, o01, o10, o11;
    float w00, w01, w10, w11;
};
__device__ __forceinline__ Taps8 make_taps8(float u, float v, int xx, int yy, int H, int W, int align_corners, float thr) {
    const float gx = 2.0f * ((float)xx + u) / (float)max(W - 1, 1) - 1.0f;
    const float gy = 2.0f * ((float)yy + v) / (float)max(H - 1, 1) - 1.0f;
    float ix, iy;
    if (align_corners) {
        ix = (gx + 1.0f) / 2.0f * (float)(W - 1);
        iy = (gy + 1.0f) / 2.0f * (float)(H - 1);
    } else {
        ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
        iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
    }
    ix = fminf(fmaxf(ix, -16.0f), (float)W + 16.0f);
    iy = fminf(fmaxf(iy, -16.0f), (float)H + 16.0f);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float ax1 = ix - fx, ay1 = iy - fy, ax0 = 1.0f - ax1, ay0 = 1.0f - ay1;
    const bool vx0 = (x0 >= 0) && (x0 < W), vx1 = (x0 + 1 >= 0) && (x0 + 1 < W);
    const bool vy0 = (y0 >= 0) && (y0 < H), vy1 = (y0 + 1 >= 0) && (y0 + 1 < H);
    Taps8 t;
    t.w00 = (vx0 && vy0) ? ay0 * ax0 : 0.f;
    t.w01 = (vx1 && vy0) ? ay0 * ax1 : 0.f;
    t.w10 = (vx0 && vy1) ? ay1 * ax0 : 0.f;
    t.w11 = (vx1 && vy1) ? ay1 * ax1 : 0.f;
    const float msum = ((t.w00 + t.w01) + t.w10) + t.w11;
    if (!(msum >= thr)) t.w00 = t.w01 = t.w10 = t.w11 = 0.f;
    const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1);
    const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0 + 1, 0), H - 1);
    t.o00 = ((int64_t)yc0 * W + xc0) * 8;
    t.o01 = ((int64_t)yc0 * W + xc1) * 8;
    t.o10 = ((int64_t)yc1 * W + xc0) * 8;
    t.o11 = ((int64_t)yc1 * W + xc1) * 8;
    return t;
}
__device__ __forceinline__ h8 blend_taps8(const _Float16 *src, const Taps8 &t) {
    const h8 a = *reinterpret_cast<const h8 *>(src + t.o00), bq = *reinterpret_cast<const h8 *>(src + t.o01);
    const h8 c = *reinterpret_cast<const h8 *>(src + t.o10), d = *reinterpret_cast<const h8 *>(src + t.o11);
    h8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        o[j] = (_Float16)((((float)a[j] * t.w00 + (float)bq[j] * t.w01) + (float)c[j] * t.w10) + (float)d[j] * t.w11);
    return o;
}

template <bool ENTRY>
__global__ void __launch_bounds__(256)
warp_c8_kernel(const _Float16 *__restrict__ x, const _Float16 *__restrict__ flo, _Float16 *__restrict__ out,
               int Cg, int H, int W, int64_t npix, int flo_channel, int64_t bsx, int64_t bsf, int64_t bso,
               float flow_scale, int align_corners, float thr, LevelEntry e) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int64_t plane = (int64_t)H * W;
    const int b = (int)(i / plane);
    const int pix = (int)(i - (int64_t)b * plane);
    const int yy = pix / W, xx = pix - yy * W;
    float u, v;
    if constexpr (ENTRY) {
        float acc0, acc1;
        entry_up_flow(e, b, yy, xx, H, W, acc0, acc1);
        const int64_t sp = ((int64_t)(yy >> 1) * (W >> 1) + (xx >> 1)) * 8;
        const int ph = (yy & 1) * 2 + (xx & 1);
        const float *pq = e.feat_phases + (int64_t)b * e.bs_featp + sp;
        h4 o;
        o[0] = pwc::sat_half(acc0); o[1] = pwc::sat_half(acc1); o[2] = pwc::sat_half(pq[ph]); o[3] = pwc::sat_half(pq[4 + ph]);
        *reinterpret_cast<h4 *>(e.fg + (int64_t)b * e.bs_fg + (int64_t)pix * 8) = o;
        u = acc0 * flow_scale;
        v = acc1 * flow_scale;
        const _Float16 *cs = e.c1 + (int64_t)b * e.bs_c1 + (int64_t)pix * 8;
        _Float16 *cd = e.c1_dst + (int64_t)b * e.bs_c1dst + (int64_t)pix * 8;
        // four groups' loads in flight before the first store (the loop as written made one round trip per group)
        int g = 0;
        for (; g + 4 <= Cg; g += 4, cs += 4 * plane * 8, cd += 4 * plane * 8) {
            h8 r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = *reinterpret_cast<const h8 *>(cs + (int64_t)q * plane * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<h8 *>(cd + (int64_t)q * plane * 8) = r[q];
        }
        for (; g < Cg; ++g, cs += plane * 8, cd += plane * 8)
            *reinterpret_cast<h8 *>(cd) = *reinterpret_cast<const h8 *>(cs);
    } else {
        const _Float16 *f = flo + (int64_t)b * bsf + (int64_t)pix * 8 + flo_channel;
        u = (float)f[0] * flow_scale;
        v = (float)f[1] * flow_scale;
    }
    const Taps8 t = make_taps8(u, v, xx, yy, H, W, align_corners, thr);
    const _Float16 *src = x + (int64_t)b * bsx;
    _Float16 *dst = out + (int64_t)b * bso + (int64_t)pix * 8;
    // two groups' eight taps in flight (four: 64 more registers, one workgroup less per CU)
    int g = 0;
    for (; g + 2 <= Cg; g += 2, src += 2 * plane * 8, dst += 2 * plane * 8) {
        const h8 r0 = blend_taps8(src, t), r1 = blend_taps8(src + plane * 8, t);
        *reinterpret_cast<h8 *>(dst) = r0;
        *reinterpret_cast<h8 *>(dst + plane * 8) = r1;
    }
    for (; g < Cg; ++g, src += plane * 8, dst += plane * 8) *reinterpret_cast<h8 *>(dst) = blend_taps8(src, t);
}

// ---- level entry + warp + correlation in ONE kernel (VERDICT r3 next #1c) ----------------------------------------------------------
// corr81_c8_kernel with the LDS-DMA of the in2 halo tile replaced by its production: the 256 threads of a tile own its 16 x 40 halo
// pixels (three slots each, the last one part-filled), compute up_flow there exactly as the entry kernel does (fp32 deconvolution of
// the level above's fp32 flow), derive the bilinear taps ONCE per tile and, per step of four channel groups, gather four 16-byte taps
// per (pixel, group), blend in fp32 and write the half result into the LDS tile -- the warped tensor never exists in memory.  The
// thread's own pixel does the rest of the level's entry on the way: (up_flow, up_feat) into the arena's flow group, and the c1 values
// it loads for the dot products anyway are also stored into the arena.  One launch instead of two, and 58 MB (level 2, batch 16) of
// warped features + 29 MB of c1 re-read less.  Statement-for-statement the arithmetic of warp_c8_kernel<true> + corr81_c8_kernel:
// bit-identical results.
constexpr int kCHaloPx = kCHaloH * kCHaloW;                      // 640
constexpr int kCSlots = (kCHaloPx + kCThreads - 1) / kCThreads;  // 3

__global__ void __launch_bounds__(kCThreads, 3)
level_corr81_c8_kernel(const _Float16 *__restrict__ x2, _Float16 *__restrict__ out, int Cg, int H, int W, int tiles_x, int tiles_y,
                       int64_t bsx, int64_t bso, float flow_scale, int align_corners, float thr, LevelEntry e,
                       float scale, float slope, int do_leaky) {
    __shared__ __attribute__((aligned(16))) h8 tile[kCPieces];           // [g][row 16][col 40]

    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = tx * kCTW, y0 = ty * kCTH;
    const int plane = H * W;
    const int px = tid & 31, py = tid >> 5;
    const int x = x0 + px, y = y0 + py;
    const bool inside = (x < W) && (y < H);

    // the thread's own pixel: flow group of the arena (deconvL / upfeatL of the level above)
    if (inside) {
        float acc0, acc1;
        entry_up_flow(e, b, y, x, H, W, acc0, acc1);
        const int64_t sp = ((int64_t)(y >> 1) * (W >> 1) + (x >> 1)) * 8;
        const int ph = (y & 1) * 2 + (x & 1);
        const float *pq = e.feat_phases + (int64_t)b * e.bs_featp + sp;
        h4 o;
        o[0] = pwc::sat_half(acc0); o[1] = pwc::sat_half(acc1); o[2] = pwc::sat_half(pq[ph]); o[3] = pwc::sat_half(pq[4 + ph]);
        *reinterpret_cast<h4 *>(e.fg + (int64_t)b * e.bs_fg + ((int64_t)y * W + x) * 8) = o;
    }
    // the thread's halo pixels: taps once per tile (a halo pixel outside the image is the correlation's zero padding)
    Taps8 taps[kCSlots];
    bool live[kCSlots];
#pragma unroll
    for (int i = 0; i < kCSlots; ++i) {
        const int hp = tid + i * kCThreads;
        const int hy = y0 - 4 + hp / kCHaloW, hx = x0 - 4 + hp % kCHaloW;
        live[i] = (hp < kCHaloPx) && (hy >= 0) && (hy < H) && (hx >= 0) && (hx < W);
        float u = 0.f, v = 0.f;
        if (live[i]) {
            entry_up_flow(e, b, hy, hx, H, W, u, v);
            u *= flow_scale;
            v *= flow_scale;
        }
        taps[i] = make_taps8(u, v, live[i] ? hx : 0, live[i] ? hy : 0, H, W, align_corners, thr);
    }

    float acc[81];
#pragma unroll
    for (int d = 0; d < 81; ++d) acc[d] = 0.f;

    const _Float16 *p1 = e.c1 + (int64_t)b * e.bs_c1;
    _Float16 *pc = e.c1_dst + (int64_t)b * e.bs_c1dst;
    const _Float16 *p2 = x2 + (int64_t)b * bsx;
    const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int g0 = 0; g0 < Cg; g0 += kCGroups) {
        const int gv = min(kCGroups, Cg - g0);
        __syncthreads();                                   // previous step's reads are done
#pragma unroll
        for (int i = 0; i < kCSlots; ++i) {
            const int hp = tid + i * kCThreads;
            if (hp >= kCHaloPx) continue;
            // UNCONDITIONAL loads (a dead pixel's taps point at element 0, a group past the last one re-reads the last): a load behind
            // a run-time condition makes hipcc branch around it and wait for each one (cdna guide, GEMV item 4c); the selects are
            // v_cndmask.  The 16 taps of a pixel's four groups are in flight together.
#pragma unroll
            for (int gp = 0; gp < kCGroups; gp += 2) {     // two groups' eight taps in flight (four would cost a third workgroup per CU)
                h8 r[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const h8 v = blend_taps8(p2 + (int64_t)(g0 + min(gp + k, gv - 1)) * plane * 8, taps[i]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) r[k][j] = live[i] ? v[j] : (_Float16)0;
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (gp + k < gv) tile[(gp + k) * kCHaloPx + hp] = r[k];
            }
        }
        __syncthreads();
        for (int g = 0; g < gv; ++g) {
            h8 a = zero;
            if (inside) {
                const int64_t o1 = ((int64_t)(g0 + g) * plane + (int64_t)y * W + x) * 8;
                a = *reinterpret_cast<const h8 *>(p1 + o1);
                *reinterpret_cast<h8 *>(pc + o1) = a;                   // c1 into its arena slot
            }
            const h8 *t = tile + (g * kCHaloH + py) * kCHaloW + px;
#pragma unroll
            for (int dy = 0; dy < 9; ++dy)
#pragma unroll
                for (int dx = 0; dx < 9; ++dx)
                    acc[dy * 9 + dx] = dot8(a, t[dy * kCHaloW + dx], acc[dy * 9 + dx]);
        }
    }
    if (!inside) return;
    _Float16 *po = out + (int64_t)b * bso + ((int64_t)y * W + x) * 8;
#pragma unroll
    for (int g = 0; g < 11; ++g) {
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int d = g * 8 + j;
            float v = 0.f;
            if (d < 81) {
                v = acc[d < 81 ? d : 0] * scale;
                if (do_leaky) v = pwc::leaky(v, slope);
            }
            o[j] = pwc::sat_half(v);
        }
        *reinterpret_cast<h8 *>(po + (int64_t)g * plane * 8) = o;
    }
}

// ---- first pyramid layer ----------------------------------------------------------------------------------------
// conv1a (3 -> 16, stride 2, pad 1) + LeakyReLU straight from the float32 NCHW image to c8 halves: with K = 27 the MFMA
// kernel is pure per-workgroup overhead (324 us at batch 16 + 106 us of layout conversion); here one thread computes
// the 16 channels of one output pixel from its 3x3x3 window (432 fma), weights broadcast from LDS.
// waves_per_eu: left alone, the compiler hoists all 108 weight reads above the fma chain (500 registers, ONE wave per
// SIMD, every latency exposed: 127 us per launch at batch 16); capped at PWC_IMAGE_CONV_WAVES waves' worth of registers
// the reads interleave with the fmas and the other waves hide the image loads.
#ifndef PWC_IMAGE_CONV_WAVES
#define PWC_IMAGE_CONV_WAVES 8
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PWC_IMAGE_CONV_WAVES, 8)))
image_conv_s2_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                     _Float16 *__restrict__ y, int H, int W, int Ho, int Wo, int64_t npix, int64_t bsx, int64_t bsy,
                     float slope) {
    __shared__ __attribute__((aligned(16))) float sw[27][16];
    __shared__ float sb[16];
    for (int i = threadIdx.x; i < 27 * 16; i += 256) sw[i / 16][i % 16] = w[(i % 16) * 27 + i / 16];   // w[co][ci][ky][kx]
    if (threadIdx.x < 16) sb[threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int64_t oplane = (int64_t)Ho * Wo;
    const int b = (int)(i / oplane);
    const int pix = (int)(i - (int64_t)b * oplane);
    const int oy = pix / Wo, ox = pix - oy * Wo;
    f32x2 acc[8];                                    // (co, co+1) pairs: every fma below is a v_pk_fma_f32
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = f32x2{sb[2 * q], sb[2 * q + 1]};
    const float *xb = x + (int64_t)b * bsx;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                // unconditional load from the clamped address + select: no divergent branch per tap
                const bool ok = (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);
                const float ld = xb[((int64_t)ci * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)];
                const float v = ok ? ld : 0.f;
                const float4 *wr = reinterpret_cast<const float4 *>(sw[(ci * 3 + ky) * 3 + kx]);      // 4 broadcast b128 reads
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 wq = wr[q];
                    acc[2 * q] = __builtin_elementwise_fma(f32x2{v, v}, f32x2{wq.x, wq.y}, acc[2 * q]);
                    acc[2 * q + 1] = __builtin_elementwise_fma(f32x2{v, v}, f32x2{wq.z, wq.w}, acc[2 * q + 1]);
                }
            }
        }
    _Float16 *yo = y + (int64_t)b * bsy + (int64_t)pix * 8;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = pwc::sat_half(pwc::leaky(acc[g * 4 + j / 2][j & 1], slope));
        *reinterpret_cast<h8 *>(yo + (int64_t)g * oplane * 8) = o;
    }
}

}  // namespace

extern "C" int pwc_image_conv_s2_c8_f16(const void *x, const void *w, const void *bias, void *y, int B, int H, int W,
                                        float leaky_slope, int64_t x_bstride, int64_t y_bstride, void *stream) {
    if (!x || !w || !bias || !y || B <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_image_conv_s2_c8_f16: bad argument");
    if (!pwc::aligned16(y) || (y_bstride % 8)) PWC_FAIL(PWC_EALIGN, "pwc_image_conv_s2_c8_f16: output must be 16-byte aligned");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int64_t npix = (int64_t)B * Ho * Wo;
    const int64_t nblk = (npix + 255) / 256;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_image_conv_s2_c8_f16: grid too large");
    hipLaunchKernelGGL(image_conv_s2_kernel, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(x), static_cast<const float *>(w), static_cast<const float *>(bias),
                       static_cast<_Float16 *>(y), H, W, Ho, Wo, npix, x_bstride, y_bstride, leaky_slope);
    return pwc::check_launch("image_conv_s2_kernel");
}

extern "C" int pwc_corr81_c8_f16(const void *in1, const void *in2, void *out, int B, int C, int H, int W,
                                 float corr_multiply, unsigned flags, float leaky_slope,
                                 int64_t in1_bstride, int64_t in2_bstride, int64_t out_bstride, void *stream) {
    if (!in1 || !in2 || !out) PWC_FAIL(PWC_EINVAL, "pwc_corr81_c8_f16: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_corr81_c8_f16: bad shape");
    if (!pwc::aligned16(in1) || !pwc::aligned16(in2) || !pwc::aligned16(out) || (in1_bstride % 8) || (in2_bstride % 8) || (out_bstride % 8))
        PWC_FAIL(PWC_EALIGN, "pwc_corr81_c8_f16: tensors must be 16-byte aligned with batch strides that are multiples of 8");
    const int64_t plane = (int64_t)H * W;
    const int cg = (C + 7) / 8;
    if (in1_bstride < cg * plane * 8 || in2_bstride < cg * plane * 8 || out_bstride < 11 * plane * 8)
        PWC_FAIL(PWC_EINVAL, "pwc_corr81_c8_f16: batch stride smaller than the tensor");
    if (plane * 16 * kCGroups >= 0x7fffffffLL) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_corr81_c8_f16: image plane too large for 32-bit DMA offsets");
    const int tiles_x = (W + kCTW - 1) / kCTW, tiles_y = (H + kCTH - 1) / kCTH;
    const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_corr81_c8_f16: grid too large");
    const float scale = (flags & PWC_CORR_NORMALIZE) ? 1.0f / (float)C : corr_multiply;
    // few tiles: one thread per output value instead (PWC_CORR16_DIRECT_BELOW overrides the tile-count threshold)
    static const int direct_below = [] { const char *e = getenv("PWC_CORR16_DIRECT_BELOW"); return (e && *e) ? atoi(e) : 100; }();   // level 4 at batch 16 (128 tiles): tiled 25 us vs direct 38 us; level 5 (32 tiles): direct 15 vs 30
    if (nblk < direct_below) {
        const int64_t total = (int64_t)B * plane * 88;
        hipLaunchKernelGGL(corr81_c8_direct_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const _Float16 *>(in1), static_cast<const _Float16 *>(in2), static_cast<_Float16 *>(out),
                           cg, H, W, total, in1_bstride, in2_bstride, out_bstride, scale, leaky_slope,
                           (flags & PWC_ACT_LEAKY) ? 1 : 0);
        return pwc::check_launch("corr81_c8_direct_kernel");
    }
    hipLaunchKernelGGL(corr81_c8_kernel, dim3((unsigned)nblk), dim3(kCThreads), 0, static_cast<hipStream_t>(stream),
                       static_cast<const _Float16 *>(in1), static_cast<const _Float16 *>(in2), static_cast<_Float16 *>(out),
                       cg, H, W, tiles_x, tiles_y, in1_bstride, in2_bstride, out_bstride, scale, leaky_slope,
                       (flags & PWC_ACT_LEAKY) ? 1 : 0);
    return pwc::check_launch("corr81_c8_kernel");
}

extern "C" int pwc_warp_c8_f16(const void *x, const void *flo, void *out, int B, int C, int H, int W, int flo_channel,
                               float flow_scale, int align_corners, float mask_threshold,
                               int64_t x_bstride, int64_t flo_bstride, int64_t out_bstride, void *stream) {
    if (!x || !flo || !out) PWC_FAIL(PWC_EINVAL, "pwc_warp_c8_f16: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_warp_c8_f16: bad shape");
    if (flo_channel < 0 || flo_channel > 6) PWC_FAIL(PWC_EINVAL, "pwc_warp_c8_f16: (u,v) must be channels k,k+1 of ONE group (k=%d)", flo_channel);
    if (!pwc::aligned16(x) || !pwc::aligned16(out) || (x_bstride % 8) || (out_bstride % 8))
        PWC_FAIL(PWC_EALIGN, "pwc_warp_c8_f16: tensors must be 16-byte aligned with batch strides that are multiples of 8");
    if (x == out) PWC_FAIL(PWC_EINVAL, "pwc_warp_c8_f16: in-place warp is not defined");
    const int64_t npix = (int64_t)B * H * W;
    const int64_t nblk = (npix + 255) / 256;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_warp_c8_f16: grid too large");
    hipLaunchKernelGGL(warp_c8_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const _Float16 *>(x), static_cast<const _Float16 *>(flo), static_cast<_Float16 *>(out),
                       (C + 7) / 8, H, W, npix, flo_channel, x_bstride, flo_bstride, out_bstride,
                       flow_scale, align_corners, mask_threshold, LevelEntry{});
    return pwc::check_launch("warp_c8_kernel");
}

extern "C" int pwc_level_entry_c8_f16(const void *c1, const void *c2, const void *flow32, const void *feat_phases,
                                      const void *deconv_w, const void *deconv_b,
                                      void *c1_dst, void *flow_group, void *warped, int B, int C, int H, int W,
                                      float flow_scale, int align_corners, float mask_threshold,
                                      int64_t c1_bstride, int64_t c2_bstride, int64_t flow32_bstride,
                                      int64_t feat_phases_bstride, int64_t c1_dst_bstride, int64_t flow_group_bstride,
                                      int64_t warped_bstride, void *stream) {
    if (!c1 || !c2 || !flow32 || !feat_phases || !deconv_w || !deconv_b || !c1_dst || !flow_group || !warped)
        PWC_FAIL(PWC_EINVAL, "pwc_level_entry_c8_f16: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1))
        PWC_FAIL(PWC_EINVAL, "pwc_level_entry_c8_f16: H and W must be positive and even (twice the level above), got %dx%d", H, W);
    const void *ptrs[7] = {c1, c2, flow32, feat_phases, c1_dst, flow_group, warped};
    const int64_t strides[7] = {c1_bstride, c2_bstride, flow32_bstride, feat_phases_bstride, c1_dst_bstride,
                                flow_group_bstride, warped_bstride};
    for (int k = 0; k < 7; ++k)
        if (!pwc::aligned16(ptrs[k]) || (strides[k] % 8))
            PWC_FAIL(PWC_EALIGN, "pwc_level_entry_c8_f16: tensors must be 16-byte aligned with batch strides that are multiples of 8");
    if (c2 == warped || c1 == c1_dst) PWC_FAIL(PWC_EINVAL, "pwc_level_entry_c8_f16: in-place operands");
    const int64_t npix = (int64_t)B * H * W;
    const int64_t nblk = (npix + 255) / 256;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_level_entry_c8_f16: grid too large");
    LevelEntry e{static_cast<const float *>(flow32), static_cast<const float *>(feat_phases),
                 static_cast<const float *>(deconv_w), static_cast<const float *>(deconv_b),
                 static_cast<const _Float16 *>(c1), static_cast<_Float16 *>(flow_group), static_cast<_Float16 *>(c1_dst),
                 flow32_bstride, feat_phases_bstride, c1_bstride, flow_group_bstride, c1_dst_bstride};
    hipLaunchKernelGGL(warp_c8_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const _Float16 *>(c2), static_cast<const _Float16 *>(nullptr), static_cast<_Float16 *>(warped),
                       (C + 7) / 8, H, W, npix, 0, c2_bstride, (int64_t)0, warped_bstride,
                       flow_scale, align_corners, mask_threshold, e);
    return pwc::check_launch("warp_c8_kernel<entry>");
}

/* Entry of a decoder level below the coarsest AND its cost volume in one launch (PWCNet.py:208-214: up_flow = deconvL(flow),
 * up_feat = upfeatL(x), corr = LeakyReLU(corr(c1, warp(c2, up_flow * s)))): pwc_level_entry_c8_f16 followed by pwc_corr81_c8_f16
 * with the warped features kept in LDS.  Bit-identical to the two calls. */
extern "C" int pwc_level_corr81_c8_f16(const void *c1, const void *c2, const void *flow32, const void *feat_phases,
                                       const void *deconv_w, const void *deconv_b,
                                       void *c1_dst, void *flow_group, void *corr_out, int B, int C, int H, int W,
                                       float flow_scale, int align_corners, float mask_threshold,
                                       float corr_multiply, unsigned flags, float leaky_slope,
                                       int64_t c1_bstride, int64_t c2_bstride, int64_t flow32_bstride,
                                       int64_t feat_phases_bstride, int64_t c1_dst_bstride, int64_t flow_group_bstride,
                                       int64_t corr_bstride, void *stream) {
    if (!c1 || !c2 || !flow32 || !feat_phases || !deconv_w || !deconv_b || !c1_dst || !flow_group || !corr_out)
        PWC_FAIL(PWC_EINVAL, "pwc_level_corr81_c8_f16: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1))
        PWC_FAIL(PWC_EINVAL, "pwc_level_corr81_c8_f16: H and W must be positive and even (twice the level above), got %dx%d", H, W);
    const void *ptrs[7] = {c1, c2, flow32, feat_phases, c1_dst, flow_group, corr_out};
    const int64_t strides[7] = {c1_bstride, c2_bstride, flow32_bstride, feat_phases_bstride, c1_dst_bstride,
                                flow_group_bstride, corr_bstride};
    for (int k = 0; k < 7; ++k)
        if (!pwc::aligned16(ptrs[k]) || (strides[k] % 8))
            PWC_FAIL(PWC_EALIGN, "pwc_level_corr81_c8_f16: tensors must be 16-byte aligned with batch strides that are multiples of 8");
    if (c1 == c1_dst) PWC_FAIL(PWC_EINVAL, "pwc_level_corr81_c8_f16: in-place operands");
    const int cg = (C + 7) / 8;
    const int64_t plane = (int64_t)H * W;
    if (c1_bstride < cg * plane * 8 || c2_bstride < cg * plane * 8 || c1_dst_bstride < cg * plane * 8 || corr_bstride < 11 * plane * 8 ||
        flow_group_bstride < plane * 8)
        PWC_FAIL(PWC_EINVAL, "pwc_level_corr81_c8_f16: batch stride smaller than the tensor");
    const int tiles_x = (W + kCTW - 1) / kCTW, tiles_y = (H + kCTH - 1) / kCTH;
    const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_level_corr81_c8_f16: grid too large");
    const float scale = (flags & PWC_CORR_NORMALIZE) ? 1.0f / (float)C : corr_multiply;
    LevelEntry e{static_cast<const float *>(flow32), static_cast<const float *>(feat_phases),
                 static_cast<const float *>(deconv_w), static_cast<const float *>(deconv_b),
                 static_cast<const _Float16 *>(c1), static_cast<_Float16 *>(flow_group), static_cast<_Float16 *>(c1_dst),
                 flow32_bstride, feat_phases_bstride, c1_bstride, flow_group_bstride, c1_dst_bstride};
    hipLaunchKernelGGL(level_corr81_c8_kernel, dim3((unsigned)nblk), dim3(kCThreads), 0, static_cast<hipStream_t>(stream),
                       static_cast<const _Float16 *>(c2), static_cast<_Float16 *>(corr_out), cg, H, W, tiles_x, tiles_y,
                       c2_bstride, corr_bstride, flow_scale, align_corners, mask_threshold, e,
                       scale, leaky_slope, (flags & PWC_ACT_LEAKY) ? 1 : 0);
    return pwc::check_launch("level_corr81_c8_kernel");
}
