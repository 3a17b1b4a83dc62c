// Sample coordinates, bilinear taps and validity mask of PWCDCNet.warp (reference models/PWCNet.py:141-177) -- shared by the
// stand-alone warp kernels (pwc_warp.hip) and the correlation kernel that warps its second operand on the fly (pwc_corr.hip).
#pragma once
#include "pwc_common.h"

namespace pwc_warp {

struct Taps {
    int o00, o01, o10, o11;     // element offsets inside a plane (clamped, always valid)
    float w00, w01, w10, w11;   // bilinear weight * in-bounds * mask
    int x0;                     // unclamped column of the left taps (pair loads, see make_pair_taps)
    int r0, r1;                 // rows of the top / bottom taps (clamped): o00 = r0 * W + ..., o10 = r1 * W + ...
};

__device__ __forceinline__ Taps make_taps(float px, float py, int H, int W, int align_corners, float thr) {
    const float gx = 2.0f * px / (float)max(W - 1, 1) - 1.0f;
    const float gy = 2.0f * py / (float)max(H - 1, 1) - 1.0f;
    float ix, iy;
    if (align_corners) {
        ix = (gx + 1.0f) / 2.0f * (float)(W - 1);
        iy = (gy + 1.0f) / 2.0f * (float)(H - 1);
    } else {
        ix = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
        iy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
    }
    // keep the integer conversion defined for wild flows; anything this far out has no valid tap
    ix = fminf(fmaxf(ix, -16.0f), (float)W + 16.0f);
    iy = fminf(fmaxf(iy, -16.0f), (float)H + 16.0f);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float ax1 = ix - fx, ay1 = iy - fy;
    const float ax0 = 1.0f - ax1, ay0 = 1.0f - ay1;
    const bool vx0 = (x0 >= 0) && (x0 < W), vx1 = (x0 + 1 >= 0) && (x0 + 1 < W);
    const bool vy0 = (y0 >= 0) && (y0 < H), vy1 = (y0 + 1 >= 0) && (y0 + 1 < H);
    Taps t;
    t.w00 = (vx0 && vy0) ? ay0 * ax0 : 0.0f;
    t.w01 = (vx1 && vy0) ? ay0 * ax1 : 0.0f;
    t.w10 = (vx0 && vy1) ? ay1 * ax0 : 0.0f;
    t.w11 = (vx1 && vy1) ? ay1 * ax1 : 0.0f;
    // grid_sample(ones) accumulates nw, ne, sw, se in this order (same order as the value sum)
    const float msum = ((t.w00 + t.w01) + t.w10) + t.w11;
    if (!(msum >= thr)) { t.w00 = t.w01 = t.w10 = t.w11 = 0.0f; }
    const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1);
    const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0 + 1, 0), H - 1);
    t.o00 = yc0 * W + xc0;
    t.o01 = yc0 * W + xc1;
    t.o10 = yc1 * W + xc0;
    t.o11 = yc1 * W + xc1;
    t.x0 = x0;
    t.r0 = yc0;
    t.r1 = yc1;
    return t;
}

// The same taps as TWO 8-byte row pairs (W >= 2): the pair of a row starts at column xb = clamp(x0, 0, W-2), and the two bilinear
// weights of the row are re-assigned to the pair's elements (at the image border one tap is invalid -- weight 0 -- and the other
// may sit on either element).  sum = ((a_top*wa + b_top*wb) + a_bot*wc) + b_bot*wd reproduces tap4() bit for bit: only the
// position of an exact zero term changes.
struct PairTaps {
    int otop, obot;             // element offsets of the pairs inside a plane
    int rtop, rbot, xb;         // the same as rows and first column (no integer division downstream: ~35 instructions each on gfx950)
    float wa, wb, wc, wd;
};

__device__ __forceinline__ PairTaps make_pair_taps(float px, float py, int H, int W, int align_corners, float thr) {
    const Taps t = make_taps(px, py, H, W, align_corners, thr);
    const int xb = min(max(t.x0, 0), W - 2);
    PairTaps p;
    p.rtop = t.r0;
    p.rbot = t.r1;
    p.xb = xb;
    p.otop = t.r0 * W + xb;
    p.obot = t.r1 * W + xb;
    const bool same = (t.x0 == xb), left_edge = (t.x0 + 1 == xb), right_edge = (t.x0 == xb + 1);
    p.wa = same ? t.w00 : (left_edge ? t.w01 : 0.0f);
    p.wb = same ? t.w01 : (right_edge ? t.w00 : 0.0f);
    p.wc = same ? t.w10 : (left_edge ? t.w11 : 0.0f);
    p.wd = same ? t.w11 : (right_edge ? t.w10 : 0.0f);
    return p;
}

// nw*w00 + ne*w01 + sw*w10 + se*w11 in grid_sample's association ((nw + ne) + sw) + se, as one multiply and three fused
// multiply-adds (four VALU instructions instead of seven; one rounding per term instead of two).  Every warp in this library
// -- stand-alone, fp16, fused into the correlation -- blends through this function, so they agree bit for bit.
__device__ __forceinline__ float blend4(float v00, float v01, float v10, float v11, float w00, float w01, float w10, float w11) {
    return fmaf(v11, w11, fmaf(v10, w10, fmaf(v01, w01, v00 * w00)));
}

__device__ __forceinline__ float tap4(const Taps &t, float v00, float v01, float v10, float v11) {
    return blend4(v00, v01, v10, v11, t.w00, t.w01, t.w10, t.w11);
}


}  // namespace pwc_warp
