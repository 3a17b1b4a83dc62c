// 3x3 convolution with a tiny Cout (the 2-channel flow heads predict_flow{6..2} and dc_conv7,
// reference models/PWCNet.py:32-33, used :207,221,235,251,265,268).
//
// With Cout = 2 an MFMA tile would be 94 % padding, and the op is a stream over the whole [B,Cin,H,W]
// input (1 GB for predict_flow2 at batch 16) with 18 fma per loaded window -- HBM-bound, GEMV-like.  So:
//   * workgroup = 8 waves that all cover the SAME 4-row x 16-col pixel tile (one pixel per lane) and SPLIT
//     the reduction over Cin (wave w takes channels w, w+8, ...): a level-6 head (112 pixels, 529
//     channels) still spreads over 8 x (B*7) waves instead of being one long serial loop;
//   * no LDS staging and no barrier in the loop: each lane loads its 3x3 window straight from global
//     (clamped offsets computed once; neighbouring lanes share cache lines), 4 channels unrolled so ~36
//     loads per lane are in flight; weights are wave-uniform scalar loads;
//   * the 8 partial sums meet in LDS and are added in wave order (deterministic), then bias / optional
//     LeakyReLU / optional residual (flow2 + dc_conv7(..), PWCNet.py:268).
// Any H, W and alignment (dword accesses only).
#include "pwc_common.h"

namespace {

using pwc::leaky;

constexpr int kHWaves = 8;
constexpr int kHThreads = 64 * kHWaves;
constexpr int kHTH = 4;                 // tile rows
constexpr int kHTW = 16;                // tile cols

template <int CO>
__global__ void __launch_bounds__(kHThreads)
conv3x3_head_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                    const float *__restrict__ residual, float *__restrict__ y,
                    int Cin, int H, int W, int tiles_x, int tiles_y,
                    int64_t bsx, int64_t bsy, int64_t bsr, float slope, int do_leaky) {
    __shared__ float red[kHWaves][CO][64];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    int bid = blockIdx.x;
    const int bx = bid % tiles_x;
    bid /= tiles_x;
    const int by = bid % tiles_y;
    const int b = bid / tiles_y;
    const int oy = by * kHTH + (lane >> 4);
    const int ox = bx * kHTW + (lane & 15);
    const int64_t plane = (int64_t)H * W;
    const float *xb = x + (int64_t)b * bsx;

    int off[3][3];
    float msk[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int yy = oy - 1 + a, xx = ox - 1 + c;
            const bool ok = (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W);
            off[a][c] = min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1);
            msk[a][c] = ok ? 1.f : 0.f;
        }

    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = 0.f;

#pragma unroll 4
    for (int ci = wave; ci < Cin; ci += kHWaves) {
        const float *xp = xb + (int64_t)ci * plane;
        float v[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) v[a][c] = xp[off[a][c]] * msk[a][c];
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const float *wk = w + (int64_t)ci * 20 + co * 10;               // wave-uniform; tail layout [ci][co][10]
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[co] = fmaf(v[a][c], wk[a * 3 + c], acc[co]);
        }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co) red[wave][co][lane] = acc[co];
    __syncthreads();
    if (wave < CO) {
        float s = bias[wave];
#pragma unroll
        for (int k = 0; k < kHWaves; ++k) s += red[k][wave][lane];
        if (oy < H && ox < W) {
            if (do_leaky) s = leaky(s, slope);
            const int64_t o = (int64_t)wave * plane + (int64_t)oy * W + ox;
            if (residual) s += residual[(int64_t)b * bsr + o];
            y[(int64_t)b * bsy + o] = s;
        }
    }
}

}  // namespace

namespace pwc_conv {

// Returns PWC_EUNSUPPORTED when the head kernel does not apply (caller falls back to the MFMA kernel).
int run_head(const float *x, const float *w_raw, const float *bias, const float *residual, float *y,
             int B, int Cin, int H, int W, int Cout, int64_t bsx, int64_t bsy, int64_t bsr,
             float slope, int do_leaky, hipStream_t st) {
    if (Cout != 2) return PWC_EUNSUPPORTED;
    const int tiles_x = (W + kHTW - 1) / kHTW;
    const int tiles_y = (H + kHTH - 1) / kHTH;
    const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: grid too large");
    hipLaunchKernelGGL((conv3x3_head_kernel<2>), dim3((unsigned)nblk), dim3(kHThreads), 0, st,
                       x, w_raw, bias, residual, y, Cin, H, W, tiles_x, tiles_y, bsx, bsy, bsr, slope, do_leaky);
    return pwc::check_launch("conv3x3_head_kernel");
}

}  // namespace pwc_conv
