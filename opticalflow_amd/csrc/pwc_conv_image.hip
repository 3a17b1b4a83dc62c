// conv1a of the feature pyramid -- nn.Conv2d(3, 16, 3x3, stride 2, padding 1) + LeakyReLU on the image (reference models/PWCNet.py:52,184-187)
// -- as its own fp32 kernel (round 3).  Through the generic implicit-GEMM kernel the layer was bound by the issue rate of its dword
// LDS-DMA (70 instructions of 256 B per workgroup for K = 27): 2 x 92 us for the two images of a 16-pair batch against ~80 us of HBM
// time (176 MB in, 235 MB out).  Here a workgroup stages its 17 x 132 x 3 input patch with 16-byte loads, and
//   v_mfma_f32_16x16x4_f32:  D[16 couts][16 pixels] += A[cout][k] * B[k][pixel],   k = (ci, ky, kx), 27 padded to 28 = 7 steps
// with the whole filter bank in SEVEN registers per lane (A operand: lane = (cout l & 15, k = 4 s + (l >> 4))) and the B operand
// gathered from the patch (lane = (pixel l & 15, the same k)): 7 ds_read_b32 + 7 MFMAs per 16 pixels x 16 couts.
// Tile = 8 output rows x 64 output columns; wave w owns rows 2w, 2w + 1 as eight groups of 16 pixels.
// Needs Cin = 3, Cout = 16, even H and W, W % 4 == 0, 16-byte aligned x with a batch stride that is a multiple of 4.
#include "pwc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTH = 8, kTW = 64;                   // output tile
constexpr int kInH = 2 * kTH + 1;                  // 17 input rows: 2 oy0 - 1 .. 2 oy0 + 15
constexpr int kInW = 2 * kTW + 4;                  // 132 staged columns: 2 ox0 - 4 .. 2 ox0 + 127 (16-byte pieces; the first needed is 2 ox0 - 1)
constexpr int kQ = kInW / 4;                       // 33 pieces per row
constexpr int kPieces = 3 * kInH * kQ;             // 1683
constexpr int kSlots = (kPieces + 255) / 256;      // 7 per thread

__global__ void __launch_bounds__(256)
image_conv_s2_f32_kernel(const float *__restrict__ x, const float *__restrict__ wp, const float *__restrict__ bias, float *__restrict__ y,
                         int H, int W, int CoutP, int tiles_x, int tiles_y, int64_t bsx, int64_t bsy, float slope, int do_leaky) {
    __shared__ __attribute__((aligned(16))) float tile[3 * kInH * kInW];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int n = lane & 15, kq = lane >> 4;
    int bid = blockIdx.x;
    if ((gridDim.x & 7u) == 0) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);       // an XCD takes a contiguous run of tiles
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int ox0 = tx * kTW, oy0 = ty * kTH;
    const int Ho = H >> 1, Wo = W >> 1;
    const float *xb = x + (int64_t)b * bsx;

    // ---- the patch: 16-byte pieces, zero outside the image (padding 1: row -1 / column -1; W % 4 == 0: a piece is all-in or all-out)
#pragma unroll
    for (int j = 0; j < kSlots; ++j) {
        const int i = j * 256 + tid;
        if (i < kPieces) {
            const int ci = i / (kInH * kQ), rem = i % (kInH * kQ);
            const int r = rem / kQ, q = rem % kQ;
            const int iy = 2 * oy0 - 1 + r, ix = 2 * ox0 - 4 + 4 * q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4 *>(xb + ((int64_t)ci * H + iy) * W + ix);
            *reinterpret_cast<f32x4 *>(tile + (ci * kInH + r) * kInW + 4 * q) = v;
        }
    }
    // ---- filters: lane (cout n, k = 4 s + kq) from the packed image [cin][tap][CoutP] of pwc_conv3x3_pack; k = 27 is the zero pad
    float a[7];
    int koff[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int k = 4 * s + kq;
        a[s] = k < 27 ? wp[(int64_t)k * CoutP + n] : 0.f;                         // k = ci * 9 + tap
        const int ci = k / 9, tap = k % 9;
        koff[s] = k < 27 ? (ci * kInH + tap / 3) * kInW + tap % 3 + 3 : 0;        // + 3: staged column 0 is image column 2 ox0 - 4
    }
    __syncthreads();

    f32x4 acc[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int row = 2 * wave + (g >> 2), col = 16 * (g & 3) + n;
        const float *p = tile + (2 * row) * kInW + 2 * col;
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            // K is padded from 27 to 28: the pad step multiplies an explicit zero, not a real image value by a zero filter -- 0 x Inf would
            // put a NaN into an output whose 3x3 window does not contain the non-finite pixel
            const float bval = (4 * s + kq < 27) ? p[koff[s]] : 0.f;
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bval, c, 0, 0, 0);
        }
        acc[g] = c;
    }
    // ---- bias, LeakyReLU, stores: D rows = couts 4 kq .. 4 kq + 3, column = pixel n (64 contiguous bytes per cout and group)
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[4 * kq + r];
    const int64_t oplane = (int64_t)Ho * Wo;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int oy = oy0 + 2 * wave + (g >> 2), ox = ox0 + 16 * (g & 3) + n;
        if (oy >= Ho || ox >= Wo) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[g][r] + bv[r];
            if (do_leaky) v = pwc::leaky(v, slope);
            y[(int64_t)b * bsy + (int64_t)(4 * kq + r) * oplane + (int64_t)oy * Wo + ox] = v;
        }
    }
}

}  // namespace

namespace pwc_conv {

// conv1a on the image (Cin 3 -> Cout 16, stride 2): PWC_OK, or PWC_EUNSUPPORTED when the geometry is not this kernel's (the caller
// then takes the generic kernel).  wp = pwc_conv3x3_pack's image ([cin padded to 8][tap][CoutP]).
int image_conv_s2(const float *x, const float *wp, const float *bias, float *y, int B, int Cin, int H, int W, int Cout, int CoutP,
                  int64_t bsx, int64_t bsy, float slope, int do_leaky, hipStream_t st) {
    static const bool on = [] { const char *e = getenv("PWC_CONV_IMAGE"); return !(e && e[0] == '0'); }();
    if (!on || Cin != 3 || Cout != 16 || (H & 1) || (W & 3) || !pwc::aligned16(x) || (bsx & 3)) return PWC_EUNSUPPORTED;
    const int Ho = H / 2, Wo = W / 2;
    const int tiles_x = (Wo + kTW - 1) / kTW, tiles_y = (Ho + kTH - 1) / kTH;
    const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
    if (nblk > 0x7fffffffLL) return PWC_EUNSUPPORTED;
    hipLaunchKernelGGL(image_conv_s2_f32_kernel, dim3((unsigned)nblk), dim3(256), 0, st, x, wp, bias, y, H, W, CoutP, tiles_x, tiles_y,
                       bsx, bsy, slope, do_leaky);
    pwc::note_kernel("image_conv_s2_f32_kernel", 1, 2, 2, 1, 0, 0);
    return pwc::check_launch("image_conv_s2_f32_kernel");
}

}  // namespace pwc_conv
