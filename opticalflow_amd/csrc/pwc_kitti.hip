// Image-space pre / post processing of the KITTI evaluation loop as two kernels (reference inference_kitti.py:53-91, 175-178, 208-224;
// host mirror opticalflow_amd/kitti.py).  Round 3: the same steps as ~30 PyTorch elementwise / pad / cat / interpolate launches inside
// the captured graph were 14.6 % of the fp16 stream's GPU time (profiles/r03_kitti_share_before.txt; after: r03_kitti_share_fp16.txt).
//
//   ingest : uint8 pairs [n][2][H][W][3] (HWC, RGB) -> float [n][6][Hp][Wp], Hp / Wp = H / W rounded up to multiples of 64:
//            ToTensor (/ 255) + ImageNet normalisation ((v - mean) / std per channel, inference_kitti.py:175-178), the two images
//            concatenated along the channels (:208-210), replicate padding at the bottom / right (:53-63).  Same arithmetic and
//            order as the host mirror's torch expression on the device ((float(u) * (1 / 255) - mean) / std, IEEE division, no contraction).
//   upflow : the network's quarter-resolution flow [n][2][Hq][Wq] -> [n][2][h][w]: crop to (hc, wc) (the reference removes the FULL
//            resolution pad amounts from the quarter-resolution flow, :66-71,220), bilinear resize with align_corners = True
//            (F.interpolate's arithmetic: source = dst * (in - 1) / (out - 1), weights 1 - l and l), u * (w / wc), v * (h / hc)
//            (:73-91).
#include <stdint.h>

#include "pwc_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Norm3 { float mean[3], std[3]; };
// `t / 255.0` of the host mirror: PyTorch's device kernel divides a tensor by a Python scalar as a multiplication by the scalar's float
// reciprocal, and the captured pipeline must stay bit-identical to the eager one (tests/test_kitti.py)
constexpr float kInv255 = 1.0f / 255.0f;

// thread = four consecutive output columns of one (item, image, row): 12 source bytes (fewer at the right edge) -> three 16-byte stores
__global__ void __launch_bounds__(256)
kitti_ingest_kernel(const uint8_t *__restrict__ src, float *__restrict__ dst, int H, int W, int Hp, int Wp, int64_t bsd, Norm3 nm, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int wq = Wp >> 2;
    const int xq = (int)(idx % wq);
    int64_t t = idx / wq;
    const int y = (int)(t % Hp);
    t /= Hp;
    const int im = (int)(t & 1);
    const int64_t b = t >> 1;
    const uint8_t *row = src + (((b * 2 + im) * H + min(y, H - 1)) * (int64_t)W) * 3;
    float v[3][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint8_t *px = row + (int64_t)min(4 * xq + q, W - 1) * 3;          // replicate padding = clamped source coordinate
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c][q] = ((float)px[c] * kInv255 - nm.mean[c]) / nm.std[c];
    }
    float *o = dst + b * bsd + ((int64_t)(3 * im) * Hp + y) * Wp + 4 * xq;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        *reinterpret_cast<f32x4 *>(o + (int64_t)c * Hp * Wp) = (f32x4){v[c][0], v[c][1], v[c][2], v[c][3]};
}

// thread = one output pixel, both channels
__global__ void __launch_bounds__(256)
flow_upsample_kernel(const float *__restrict__ q, float *__restrict__ out, int Hq, int Wq, int hc, int wc, int h, int w, int64_t bsq,
                     float rh, float rw, float su, float sv, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % w);
    int64_t t = idx / w;
    const int y = (int)(t % h);
    const int64_t b = t / h;
    const float fy = rh * (float)y, fx = rw * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < hc - 1 ? 1 : 0), x1 = x0 + (x0 < wc - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, my = 1.0f - ly, mx = 1.0f - lx;
    const float *p = q + b * bsq;
    const int64_t plane = (int64_t)Hq * Wq;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float *pc = p + c * plane;
        const float v = my * (mx * pc[(int64_t)y0 * Wq + x0] + lx * pc[(int64_t)y0 * Wq + x1]) +
                        ly * (mx * pc[(int64_t)y1 * Wq + x0] + lx * pc[(int64_t)y1 * Wq + x1]);
        out[((b * 2 + c) * h + y) * (int64_t)w + x] = v * (c == 0 ? su : sv);
    }
}

}  // namespace

extern "C" int pwc_kitti_ingest_u8(const void *pairs_u8, void *x, int n, int H, int W, const float *mean3, const float *std3,
                                   int64_t x_bstride, void *stream) {
    if (!pairs_u8 || !x || !mean3 || !std3) PWC_FAIL(PWC_EINVAL, "pwc_kitti_ingest_u8: null pointer");
    if (n <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_kitti_ingest_u8: bad shape");
    const int Hp = (H + 63) / 64 * 64, Wp = (W + 63) / 64 * 64;
    if (!pwc::aligned16(x) || (x_bstride & 3) || x_bstride < (int64_t)6 * Hp * Wp)
        PWC_FAIL(PWC_EALIGN, "pwc_kitti_ingest_u8: x must be 16-byte aligned with a batch stride >= 6*Hp*Wp that is a multiple of 4");
    Norm3 nm;
    for (int c = 0; c < 3; ++c) {
        nm.mean[c] = mean3[c];
        nm.std[c] = std3[c];
        if (!(nm.std[c] != 0.f)) PWC_FAIL(PWC_EINVAL, "pwc_kitti_ingest_u8: std must be non-zero");
    }
    const int64_t total = (int64_t)n * 2 * Hp * (Wp >> 2);
    if ((total + 255) / 256 > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_kitti_ingest_u8: grid too large");
    hipLaunchKernelGGL(kitti_ingest_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint8_t *>(pairs_u8), static_cast<float *>(x), H, W, Hp, Wp, x_bstride, nm, total);
    return pwc::check_launch("kitti_ingest_kernel");
}

extern "C" int pwc_flow_upsample_f32(const void *flow_q, void *out, int n, int Hq, int Wq, int crop_h, int crop_w, int out_h, int out_w,
                                     int64_t q_bstride, void *stream) {
    if (!flow_q || !out) PWC_FAIL(PWC_EINVAL, "pwc_flow_upsample_f32: null pointer");
    if (n <= 0 || Hq <= 0 || Wq <= 0 || crop_h <= 0 || crop_w <= 0 || crop_h > Hq || crop_w > Wq || out_h <= 0 || out_w <= 0)
        PWC_FAIL(PWC_EINVAL, "pwc_flow_upsample_f32: bad shape");
    if (q_bstride < (int64_t)2 * Hq * Wq) PWC_FAIL(PWC_EINVAL, "pwc_flow_upsample_f32: batch stride smaller than the tensor");
    // F.interpolate(align_corners = True): scale = (in - 1) / (out - 1) in float (0 for a one-pixel output)
    const float rh = out_h > 1 ? (float)(crop_h - 1) / (float)(out_h - 1) : 0.f;
    const float rw = out_w > 1 ? (float)(crop_w - 1) / (float)(out_w - 1) : 0.f;
    const float su = (float)((double)out_w / (double)crop_w), sv = (float)((double)out_h / (double)crop_h);
    const int64_t total = (int64_t)n * out_h * out_w;
    if ((total + 255) / 256 > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_flow_upsample_f32: grid too large");
    hipLaunchKernelGGL(flow_upsample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(flow_q), static_cast<float *>(out), Hq, Wq, crop_h, crop_w, out_h, out_w, q_bstride,
                       rh, rw, su, sv, total);
    return pwc::check_launch("flow_upsample_kernel");
}
