// Convolution entry points of libpwc_hip.so: filter packing, the 3x3 MFMA implicit GEMM dispatch
// (kernel template in pwc_conv_mfma.h, instantiated per (stride, dilation) in pwc_conv_s*d*.hip) and the
// 4x4/stride-2 transposed convolution.
//
// Replaces nn.Conv2d(3x3)+LeakyReLU(0.1) (reference models/PWCNet.py:26-33) and
// nn.ConvTranspose2d(k4,s2,p1) (PWCNet.py:35-36) for every layer PWCDCNet.forward runs (PWCNet.py:184-268).
#include "pwc_conv_mfma.h"

namespace pwc_conv {
int run_s1d1(const ConvArgs &a);
int run_s1d2(const ConvArgs &a);
int run_s1d4(const ConvArgs &a);
int run_s1d8(const ConvArgs &a);
int run_s1d16(const ConvArgs &a);
int run_s2d1(const ConvArgs &a);
}  // namespace pwc_conv

namespace {

using pwc::from_f32;
using pwc::to_f32;
using pwc_conv::kCK;

inline int cout_padded(int Cout) { return (Cout + 31) / 32 * 32; }
inline int conv_chunks(int Cin) { return (Cin + kCK - 1) / kCK; }

// wp[chunk][c][tap][co] <- w[co][chunk*8 + c][tap], zero padded to CoutP columns / 8-channel chunks
__global__ void __launch_bounds__(256)
pack3x3_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cin, int Cout, int CoutP, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % CoutP);
    int64_t t = i / CoutP;
    const int tap = (int)(t % 9);
    t /= 9;
    const int cin = (int)t;                    // chunk*8 + c
    float v = 0.f;
    if (co < Cout && cin < Cin) v = w[((int64_t)co * Cin + cin) * 9 + tap];
    wp[i] = v;
}

// ConvTranspose2d(k=4, s=2, p=1), tiny Cout (2 in PWC-Net): one thread per INPUT pixel produces the
// 2x2 output quad it anchors; weights are wave-uniform.
template <typename T, int CO>
__global__ void __launch_bounds__(256)
deconv4x4s2_kernel(const T *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                   T *__restrict__ y, int Cin, int H, int W, int64_t bsx, int64_t bsy, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ix = (int)(idx % W);
    int64_t t = idx / W;
    const int iy = (int)(t % H);
    const int b = (int)(t / H);
    const int64_t plane = (int64_t)H * W;
    const T *xb = x + (int64_t)b * bsx;

    float acc[CO][2][2];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const float bv = bias[co];
        acc[co][0][0] = acc[co][0][1] = acc[co][1][0] = acc[co][1][1] = bv;
    }
    // neighbourhood offsets (clamped) and validity
    int yo[3], xo[3];
    bool yv[3], xv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int yy = iy - 1 + k, xx = ix - 1 + k;
        yv[k] = (yy >= 0 && yy < H);
        xv[k] = (xx >= 0 && xx < W);
        yo[k] = min(max(yy, 0), H - 1) * W;
        xo[k] = min(max(xx, 0), W - 1);
    }
    for (int ci = 0; ci < Cin; ++ci) {
        const T *xp = xb + (int64_t)ci * plane;
        float v[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) v[a][c] = (yv[a] && xv[c]) ? to_f32<T>(xp[yo[a] + xo[c]]) : 0.f;
        const float *wc = w + (int64_t)ci * CO * 16;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const float *k = wc + co * 16;   // k[ky*4+kx]
            // oy = 2*iy' - 1 + ky.  Output row 2*iy+py takes:  py=0: (iy-1, ky=3), (iy, ky=1);  py=1: (iy, ky=2), (iy+1, ky=0)
#pragma unroll
            for (int py = 0; py < 2; ++py) {
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const int ra = py, rb = py + 1;                // rows in v[]: (iy-1+ra), (iy-1+rb)
                    const int kya = (py == 0) ? 3 : 2, kyb = (py == 0) ? 1 : 0;
                    const int ca = px, cb = px + 1;
                    const int kxa = (px == 0) ? 3 : 2, kxb = (px == 0) ? 1 : 0;
                    float s = acc[co][py][px];
                    s = fmaf(v[ra][ca], k[kya * 4 + kxa], s);
                    s = fmaf(v[ra][cb], k[kya * 4 + kxb], s);
                    s = fmaf(v[rb][ca], k[kyb * 4 + kxa], s);
                    s = fmaf(v[rb][cb], k[kyb * 4 + kxb], s);
                    acc[co][py][px] = s;
                }
            }
        }
    }
    const int Wo = 2 * W;
    const int64_t oplane = (int64_t)4 * plane;
    T *yb = y + (int64_t)b * bsy + (int64_t)(2 * iy) * Wo + 2 * ix;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        T *p = yb + (int64_t)co * oplane;
        p[0] = from_f32<T>(acc[co][0][0]);
        p[1] = from_f32<T>(acc[co][0][1]);
        p[Wo] = from_f32<T>(acc[co][1][0]);
        p[Wo + 1] = from_f32<T>(acc[co][1][1]);
    }
}

}  // namespace

extern "C" int64_t pwc_conv3x3_packed_bytes(int Cin, int Cout, int dtype) {
    if (Cin <= 0 || Cout <= 0 || dtype != PWC_F32) return -1;
    return (int64_t)conv_chunks(Cin) * kCK * 9 * cout_padded(Cout) * (int64_t)sizeof(float);
}

extern "C" int pwc_conv3x3_pack(const void *w, void *wp, int Cin, int Cout, int dtype, void *stream) {
    if (!w || !wp) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_pack: null pointer");
    if (Cin <= 0 || Cout <= 0) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_pack: bad shape");
    if (dtype != PWC_F32) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv3x3_pack: weights must be f32");
    if (!pwc::aligned16(wp)) PWC_FAIL(PWC_EALIGN, "pwc_conv3x3_pack: packed buffer must be 16-byte aligned");
    const int64_t total = pwc_conv3x3_packed_bytes(Cin, Cout, dtype) / (int64_t)sizeof(float);
    const int64_t nblk = (total + 255) / 256;
    hipLaunchKernelGGL(pack3x3_kernel, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(w), static_cast<float *>(wp), Cin, Cout, cout_padded(Cout), total);
    return pwc::check_launch("pack3x3_kernel");
}

extern "C" int pwc_conv2d_fwd(const void *x, const void *wp, const void *bias, const void *residual, void *y,
                              int B, int Cin, int H, int W, int Cout,
                              int stride, int dilation, int dtype, unsigned flags, float leaky_slope,
                              int64_t x_bstride, int64_t y_bstride, int64_t res_bstride, void *stream) {
    if (!x || !wp || !bias || !y) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: null pointer");
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: bad shape");
    if (dtype != PWC_F32) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv2d_fwd: dtype %d", dtype);
    if (!pwc::aligned16(wp)) PWC_FAIL(PWC_EALIGN, "pwc_conv2d_fwd: packed weights must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 3u)
        PWC_FAIL(PWC_EALIGN, "pwc_conv2d_fwd: tensors must be 4-byte aligned");
    if ((flags & PWC_CONV_RESIDUAL) && !residual) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: residual flag without pointer");
    const int64_t plane = (int64_t)H * W;
    if (x_bstride < Cin * plane) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: x batch stride < Cin*H*W");
    if (plane * pwc_conv::kCK * 4 >= 0x7fffffffLL) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv2d_fwd: image plane too large for 32-bit DMA offsets");
    pwc_conv::ConvArgs a;
    a.x = static_cast<const float *>(x);
    a.wp = static_cast<const float *>(wp);
    a.bias = static_cast<const float *>(bias);
    a.residual = (flags & PWC_CONV_RESIDUAL) ? static_cast<const float *>(residual) : nullptr;
    a.y = static_cast<float *>(y);
    a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.CoutP = cout_padded(Cout);
    a.Ho = (H - 1) / stride + 1;
    a.Wo = (W - 1) / stride + 1;
    a.bsx = x_bstride; a.bsy = y_bstride; a.bsr = res_bstride;
    a.slope = leaky_slope;
    a.do_leaky = (flags & PWC_ACT_LEAKY) ? 1 : 0;
    a.stream = static_cast<hipStream_t>(stream);
    if (stride == 1) {
        switch (dilation) {
            case 1: return pwc_conv::run_s1d1(a);
            case 2: return pwc_conv::run_s1d2(a);
            case 4: return pwc_conv::run_s1d4(a);
            case 8: return pwc_conv::run_s1d8(a);
            case 16: return pwc_conv::run_s1d16(a);
        }
    } else if (stride == 2 && dilation == 1) {
        return pwc_conv::run_s2d1(a);
    }
    PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv2d_fwd: stride %d dilation %d has no kernel (stride 1: dilation 1,2,4,8,16; stride 2: dilation 1)",
             stride, dilation);
}

extern "C" int pwc_deconv4x4s2_fwd(const void *x, const void *w, const void *bias, void *y,
                                   int B, int Cin, int H, int W, int Cout, int dtype,
                                   int64_t x_bstride, int64_t y_bstride, void *stream) {
    if (!x || !w || !bias || !y) PWC_FAIL(PWC_EINVAL, "pwc_deconv4x4s2_fwd: null pointer");
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_deconv4x4s2_fwd: bad shape");
    if (Cout != 2) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_deconv4x4s2_fwd: Cout=%d (PWC-Net only has 2-channel deconvs)", Cout);
    if (dtype != PWC_F32) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_deconv4x4s2_fwd: dtype %d", dtype);
    const int64_t total = (int64_t)B * H * W;
    const int64_t nblk = (total + 255) / 256;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_deconv4x4s2_fwd: grid too large");
    hipLaunchKernelGGL((deconv4x4s2_kernel<float, 2>), dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(x), static_cast<const float *>(w), static_cast<const float *>(bias),
                       static_cast<float *>(y), Cin, H, W, x_bstride, y_bstride, total);
    return pwc::check_launch("deconv4x4s2_kernel");
}
