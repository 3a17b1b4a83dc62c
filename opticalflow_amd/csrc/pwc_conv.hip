// 3x3 convolution stack of PWC-Net as an im2col-free implicit GEMM on the gfx950 matrix cores.
//
// Replaces nn.Conv2d(3x3)+LeakyReLU(0.1) (reference models/PWCNet.py:26-33) and
// nn.ConvTranspose2d(k4,s2,p1) (PWCNet.py:35-36) for every layer PWCDCNet.forward runs
// (PWCNet.py:184-268).
//
// GEMM view (fp32 in, fp32 accumulate, exact: v_mfma_f32_32x32x2_f32 is an fp32 fma chain):
//     Y[cout, pixel] = sum_{cin, ky, kx} Wt[cout, (cin,ky,kx)] * X[cin, y*s + ky*d - d, x*s + kx*d - d]
//   A operand = weights  (M = cout, lane l holds A[cout = l&31][k = l>>5])
//   B operand = input    (N = pixel, lane l holds B[k = l>>5][pixel = l&31])
//   D         = 32 cout x 32 pixel, column (pixel) on the lane -> a store instruction writes 128-byte
//               row segments of the NCHW output.
// The two k values of one MFMA are the channel pair (2cp, 2cp+1) at one filter tap, so both operands
// are plain ds_read_b32 of 32 consecutive dwords per half-wave (conflict-free) from
//   input  tile  [cin][rows + halo][cols + halo]      (im2col-free: the tap is an address offset)
//   weight chunk [cin][tap][cout]                      (pre-packed once on the device, see pack kernel)
// Workgroup = 4 waves = 8 rows x 32 cols of output pixels x up to 128 couts; wave w owns rows 2w,2w+1
// (2 N-tiles) x MT M-tiles -> MT*2 accumulator tiles of 16 VGPRs.  Cin is consumed in chunks of 8.
// Operands may be channel slices of a wider arena: only the batch stride is free.
#include "pwc_common.h"

namespace {

using pwc::from_f32;
using pwc::leaky;
using pwc::to_f32;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kCK = 8;          // input channels per LDS chunk
constexpr int kTileH = 8;       // output rows per workgroup
constexpr int kTileW = 32;      // output cols per workgroup (= MFMA N)
constexpr int kConvThreads = 256;

template <int MT, int S, int D>
struct ConvGeom {
    static constexpr bool kRowSep = (D >= 16);                        // stage the 3 ky row sets separately
    static constexpr int kInW = (kTileW - 1) * S + 2 * D + 1;
    static constexpr int kPitch = kInW + ((kInW % 2) ? 0 : 1);        // odd pitch
    static constexpr int kInH = kRowSep ? 3 * kTileH : (kTileH - 1) * S + 2 * D + 1;
    static constexpr int kCH = kInH * kPitch;                         // floats per staged channel
    static constexpr int kKyStride = kRowSep ? kTileH * kPitch : D * kPitch;
    static constexpr int kCoutT = 32 * MT;
    static constexpr int kWChunk = kCK * 9 * kCoutT;                  // floats per weight chunk
    static constexpr int kInChunk = kCK * kCH;
    static constexpr int kSmemFloats = kInChunk + kWChunk;
};

// groups of output channels handled by one workgroup column
inline int conv_mt(int Cout) {
    const int tiles = (Cout + 31) / 32;
    const int groups = (tiles + 3) / 4;
    return (tiles + groups - 1) / groups;
}
inline int conv_groups(int Cout) {
    const int tiles = (Cout + 31) / 32;
    return (tiles + 3) / 4;
}
inline int conv_chunks(int Cin) { return (Cin + kCK - 1) / kCK; }

// wp[g][chunk][c][tap][co] <- w[g*32*MT + co][chunk*8 + c][tap], zero padded
__global__ void __launch_bounds__(256)
pack3x3_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cin, int Cout, int MT, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int coutT = 32 * MT;
    const int co = (int)(i % coutT);
    int64_t t = i / coutT;
    const int tap = (int)(t % 9);
    t /= 9;
    const int c = (int)(t % kCK);
    t /= kCK;
    const int nchunks = (Cin + kCK - 1) / kCK;
    const int chunk = (int)(t % nchunks);
    const int g = (int)(t / nchunks);
    const int cout = g * coutT + co;
    const int cin = chunk * kCK + c;
    float v = 0.f;
    if (cout < Cout && cin < Cin) v = w[((int64_t)cout * Cin + cin) * 9 + tap];
    wp[i] = v;
}

template <typename T, int MT, int S, int D>
__global__ void __launch_bounds__(kConvThreads)
conv3x3_mfma_kernel(const T *__restrict__ x, const float *__restrict__ wp, const float *__restrict__ bias,
                    const T *__restrict__ residual, T *__restrict__ y,
                    int Cin, int H, int W, int Cout, int Ho, int Wo, int tiles_x, int tiles_y,
                    int64_t bsx, int64_t bsy, int64_t bsr, float slope, int do_leaky) {
    using G = ConvGeom<MT, S, D>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *s_in = smem;
    float *s_w = smem + G::kInChunk;

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int col = lane & 31;
    const int kh = lane >> 5;

    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int g = blockIdx.y;
    const int ox0 = tx * kTileW;
    const int oy0 = ty * kTileH;

    // accumulators start at the bias: D row (cout) of register j is (j&3) + 8*(j>>2) + 4*kh
    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int co = g * G::kCoutT + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh;
            const float bv = (co < Cout) ? bias[co] : 0.f;
            acc[mt][0][j] = bv;
            acc[mt][1][j] = bv;
        }
    }

    const int64_t plane = (int64_t)H * W;
    const T *xb = x + (int64_t)b * bsx;
    const int nchunks = (Cin + kCK - 1) / kCK;
    const float *wg = wp + (int64_t)g * nchunks * G::kWChunk;

    const float *rd_in = s_in + kh * G::kCH + (2 * wave) * S * G::kPitch + col * S;
    const float *rd_w = s_w + kh * 9 * G::kCoutT + col;

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int c0 = chunk * kCK;
        if (chunk) __syncthreads();
        // ---- stage the input tile (zero padded) ------------------------------------------------
        for (int i = tid; i < kCK * G::kInH * G::kInW; i += kConvThreads) {
            const int c = i / (G::kInH * G::kInW);
            const int rem = i % (G::kInH * G::kInW);
            const int r = rem / G::kInW;
            const int xx = rem % G::kInW;
            int iy;
            if constexpr (G::kRowSep) {
                iy = (oy0 + (r % kTileH)) - D + (r / kTileH) * D;
            } else {
                iy = oy0 * S - D + r;
            }
            const int ix = ox0 * S - D + xx;
            float v = 0.f;
            if ((c0 + c) < Cin && iy >= 0 && iy < H && ix >= 0 && ix < W)
                v = to_f32<T>(xb[(int64_t)(c0 + c) * plane + (int64_t)iy * W + ix]);
            s_in[c * G::kCH + r * G::kPitch + xx] = v;
        }
        // ---- stage the weight chunk (contiguous in the packed layout) ---------------------------
        {
            const float4 *src = reinterpret_cast<const float4 *>(wg + (int64_t)chunk * G::kWChunk);
            float4 *dst = reinterpret_cast<float4 *>(s_w);
            for (int i = tid; i < G::kWChunk / 4; i += kConvThreads) dst[i] = src[i];
        }
        __syncthreads();
        // ---- 9 taps x 4 channel pairs of MFMA k-steps --------------------------------------------
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int cp = 0; cp < kCK / 2; ++cp) {
                float a[MT], bv[2];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a[mt] = rd_w[(cp * 2 * 9 + tap) * G::kCoutT + mt * 32];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    bv[nt] = rd_in[cp * 2 * G::kCH + ky * G::kKyStride + kx * D + nt * S * G::kPitch];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], bv[nt], acc[mt][nt], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: LeakyReLU / residual, 128-byte row-segment stores ----------------------------
    const int ox = ox0 + col;
    const int64_t oplane = (int64_t)Ho * Wo;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int oy = oy0 + 2 * wave + nt;
        if (oy >= Ho || ox >= Wo) continue;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = g * G::kCoutT + mt * 32 + (j & 3) + 8 * (j >> 2) + 4 * kh;
                if (co >= Cout) continue;
                float v = acc[mt][nt][j];
                if (do_leaky) v = leaky(v, slope);
                const int64_t off = (int64_t)co * oplane + (int64_t)oy * Wo + ox;
                if (residual) v += to_f32<T>(residual[(int64_t)b * bsr + off]);
                y[(int64_t)b * bsy + off] = from_f32<T>(v);
            }
        }
    }
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
            // output row 2*iy + py takes input rows (iy + py - 1 + q) with ky = 3 - py - 2q ... written out:
            //   py=0: (iy-1, ky=3), (iy, ky=1)      py=1: (iy, ky=2), (iy+1, ky=0)
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

template <typename T, int MT, int S, int D>
int launch_conv(const void *x, const void *wp, const void *bias, const void *residual, void *y,
                int B, int Cin, int H, int W, int Cout, int Ho, int Wo, unsigned flags, float slope,
                int64_t bsx, int64_t bsy, int64_t bsr, hipStream_t st) {
    using G = ConvGeom<MT, S, D>;
    const int tiles_x = (Wo + kTileW - 1) / kTileW;
    const int tiles_y = (Ho + kTileH - 1) / kTileH;
    const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: grid too large");
    const size_t smem = (size_t)G::kSmemFloats * sizeof(float);
    auto kern = conv3x3_mfma_kernel<T, MT, S, D>;
    static bool attr_set = false;   // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) PWC_FAIL((int)e, "pwc_conv2d_fwd: hipFuncSetAttribute(%zu B LDS): %s", smem, hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)conv_groups(Cout)), dim3(kConvThreads), smem, st,
                       static_cast<const T *>(x), static_cast<const float *>(wp), static_cast<const float *>(bias),
                       static_cast<const T *>(residual), static_cast<T *>(y),
                       Cin, H, W, Cout, Ho, Wo, tiles_x, tiles_y, bsx, bsy, bsr, slope,
                       (flags & PWC_ACT_LEAKY) ? 1 : 0);
    return pwc::check_launch("conv3x3_mfma_kernel");
}

template <typename T, int S, int D>
int dispatch_mt(int mt, const void *x, const void *wp, const void *bias, const void *residual, void *y,
                int B, int Cin, int H, int W, int Cout, int Ho, int Wo, unsigned flags, float slope,
                int64_t bsx, int64_t bsy, int64_t bsr, hipStream_t st) {
    switch (mt) {
        case 1: return launch_conv<T, 1, S, D>(x, wp, bias, residual, y, B, Cin, H, W, Cout, Ho, Wo, flags, slope, bsx, bsy, bsr, st);
        case 2: return launch_conv<T, 2, S, D>(x, wp, bias, residual, y, B, Cin, H, W, Cout, Ho, Wo, flags, slope, bsx, bsy, bsr, st);
        case 3: return launch_conv<T, 3, S, D>(x, wp, bias, residual, y, B, Cin, H, W, Cout, Ho, Wo, flags, slope, bsx, bsy, bsr, st);
        case 4: return launch_conv<T, 4, S, D>(x, wp, bias, residual, y, B, Cin, H, W, Cout, Ho, Wo, flags, slope, bsx, bsy, bsr, st);
    }
    PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: internal MT=%d", mt);
}

template <typename T>
int dispatch_conv(const void *x, const void *wp, const void *bias, const void *residual, void *y,
                  int B, int Cin, int H, int W, int Cout, int stride, int dil, unsigned flags, float slope,
                  int64_t bsx, int64_t bsy, int64_t bsr, hipStream_t st) {
    const int Ho = (H - 1) / stride + 1;
    const int Wo = (W - 1) / stride + 1;
    const int mt = conv_mt(Cout);
#define PWC_CONV_CASE(S_, D_) \
    if (stride == S_ && dil == D_) return dispatch_mt<T, S_, D_>(mt, x, wp, bias, residual, y, B, Cin, H, W, Cout, Ho, Wo, flags, slope, bsx, bsy, bsr, st);
    PWC_CONV_CASE(1, 1)
    PWC_CONV_CASE(1, 2)
    PWC_CONV_CASE(1, 4)
    PWC_CONV_CASE(1, 8)
    PWC_CONV_CASE(1, 16)
    PWC_CONV_CASE(2, 1)
#undef PWC_CONV_CASE
    PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv2d_fwd: stride %d dilation %d has no kernel (stride 1: dilation 1,2,4,8,16; stride 2: dilation 1)",
             stride, dil);
}

}  // namespace

extern "C" int64_t pwc_conv3x3_packed_bytes(int Cin, int Cout, int dtype) {
    if (Cin <= 0 || Cout <= 0 || dtype != PWC_F32) return -1;
    return (int64_t)conv_groups(Cout) * conv_chunks(Cin) * kCK * 9 * 32 * conv_mt(Cout) * (int64_t)sizeof(float);
}

extern "C" int pwc_conv3x3_pack(const void *w, void *wp, int Cin, int Cout, int dtype, void *stream) {
    if (!w || !wp) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_pack: null pointer");
    if (Cin <= 0 || Cout <= 0) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_pack: bad shape");
    if (dtype != PWC_F32) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv3x3_pack: weights must be f32");
    if (!pwc::aligned16(wp)) PWC_FAIL(PWC_EALIGN, "pwc_conv3x3_pack: packed buffer must be 16-byte aligned");
    const int64_t total = pwc_conv3x3_packed_bytes(Cin, Cout, dtype) / (int64_t)sizeof(float);
    const int64_t nblk = (total + 255) / 256;
    hipLaunchKernelGGL(pack3x3_kernel, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(w), static_cast<float *>(wp), Cin, Cout, conv_mt(Cout), total);
    return pwc::check_launch("pack3x3_kernel");
}

extern "C" int pwc_conv2d_fwd(const void *x, const void *wp, const void *bias, const void *residual, void *y,
                              int B, int Cin, int H, int W, int Cout,
                              int stride, int dilation, int dtype, unsigned flags, float leaky_slope,
                              int64_t x_bstride, int64_t y_bstride, int64_t res_bstride, void *stream) {
    if (!x || !wp || !bias || !y) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: null pointer");
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: bad shape");
    if (!pwc::aligned16(wp)) PWC_FAIL(PWC_EALIGN, "pwc_conv2d_fwd: packed weights must be 16-byte aligned");
    if ((flags & PWC_CONV_RESIDUAL) && !residual) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: residual flag without pointer");
    const void *res = (flags & PWC_CONV_RESIDUAL) ? residual : nullptr;
    const int64_t plane = (int64_t)H * W;
    if (x_bstride < Cin * plane) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: x batch stride < Cin*H*W");
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case PWC_F32:
            return dispatch_conv<float>(x, wp, bias, res, y, B, Cin, H, W, Cout, stride, dilation, flags, leaky_slope,
                                        x_bstride, y_bstride, res_bstride, st);
        default:
            PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv2d_fwd: dtype %d", dtype);
    }
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
