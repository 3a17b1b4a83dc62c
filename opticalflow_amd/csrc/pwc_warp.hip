// Fused bilinear backward-warp + validity mask for gfx950.
//
// Replaces PWCDCNet.warp (reference models/PWCNet.py:141-177), which builds the mesh on the host,
// copies it and a ones tensor to the device and calls grid_sample twice.  Here one kernel computes
// the source coordinate of a pixel once, derives the four taps, their weights and the mask from
// them, and reuses all of it across the C channels.
//
// Coordinates (fp32 always, also for fp16 tensors):
//   g  = 2*(x + s*u)/max(W-1,1) - 1                              (PWCNet.py:162)
//   ix = ((g+1)*W - 1)/2        align_corners = 0  (what torch>=1.3 executes for PWCNet.py:166)
//   ix = (g+1)/2*(W-1)          align_corners = 1
// evaluated in exactly this operation order so that pixels whose mask sum sits on the 0.9999
// threshold (PWCNet.py:174) fall on the same side as in the reference.
//
// Work split: one thread = 4 consecutive pixels of one row for a block of CPT channels; lanes of a
// wave cover 256 consecutive pixels, so the stores are 16 B/lane contiguous and the tap gathers of a
// smooth flow field stay within a few cache lines per wave.  HBM-bound: algorithmic bytes
// (2*C + 2)*H*W*sizeof(T) per image.
#include "pwc_common.h"
#include "pwc_warp_taps.h"

namespace {

using pwc::from_f32;
using pwc::to_f32;

constexpr int kWarpThreads = 256;
constexpr int kCPT = 8;  // channels per thread-iteration block (gridDim.y splits C)

using pwc_warp::Taps;
using pwc_warp::make_taps;
using pwc_warp::tap4;

template <typename T>
__global__ void __launch_bounds__(kWarpThreads)
warp_kernel(const T *__restrict__ x, const T *__restrict__ flo, T *__restrict__ out,
            int C, int H, int W, int quads_per_row, int64_t nquads,
            int64_t bsx, int64_t bsf, int64_t bso,
            float flow_scale, int align_corners, float thr, int vec) {
    const int64_t qi = (int64_t)blockIdx.x * kWarpThreads + threadIdx.x;
    if (qi >= nquads) return;
    const int qx = (int)(qi % quads_per_row);
    int64_t t = qi / quads_per_row;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    const int x0 = qx * 4;
    const int64_t plane = (int64_t)H * W;
    const T *fu = flo + (int64_t)b * bsf + (int64_t)y * W + x0;
    const T *fv = fu + plane;
    const int npx = min(4, W - x0);

    Taps tp[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int pp = (p < npx) ? p : 0;
        const float u = to_f32<T>(fu[pp]) * flow_scale;
        const float v = to_f32<T>(fv[pp]) * flow_scale;
        tp[p] = make_taps((float)(x0 + pp) + u, (float)y + v, H, W, align_corners, thr);
    }

    const int c_begin = blockIdx.y * kCPT;
    const int c_end = min(C, c_begin + kCPT);
    const T *src = x + (int64_t)b * bsx + (int64_t)c_begin * plane;
    T *dst = out + (int64_t)b * bso + (int64_t)c_begin * plane + (int64_t)y * W + x0;
    for (int c = c_begin; c < c_end; ++c, src += plane, dst += plane) {
        float r[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const Taps &q = tp[p];
            r[p] = tap4(q, to_f32<T>(src[q.o00]), to_f32<T>(src[q.o01]), to_f32<T>(src[q.o10]), to_f32<T>(src[q.o11]));
        }
        if (vec) {
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<float4 *>(dst) = make_float4(r[0], r[1], r[2], r[3]);
            } else {
                __half2 lo = __floats2half2_rn(r[0], r[1]);
                __half2 hi = __floats2half2_rn(r[2], r[3]);
                uint2 raw;
                raw.x = *reinterpret_cast<unsigned *>(&lo);
                raw.y = *reinterpret_cast<unsigned *>(&hi);
                *reinterpret_cast<uint2 *>(dst) = raw;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (p < npx) dst[p] = from_f32<T>(r[p]);
        }
    }
}

template <typename T>
int launch_warp(const void *x, const void *flo, void *out, int B, int C, int H, int W,
                float flow_scale, int align_corners, float thr,
                int64_t bsx, int64_t bsf, int64_t bso, hipStream_t st) {
    const int quads_per_row = (W + 3) / 4;
    const int64_t nquads = (int64_t)B * H * quads_per_row;
    const int64_t nblk = (nquads + kWarpThreads - 1) / kWarpThreads;
    const int cblk = (C + kCPT - 1) / kCPT;
    if (nblk > 0x7fffffffLL || cblk > 65535) PWC_FAIL(PWC_EINVAL, "pwc_warp_fwd: grid too large");
    const uintptr_t amask = (uintptr_t)(4 * sizeof(T) - 1);
    const int vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & amask) == 0) && (bso % 4 == 0);
    hipLaunchKernelGGL((warp_kernel<T>), dim3((unsigned)nblk, (unsigned)cblk), dim3(kWarpThreads), 0, st,
                       static_cast<const T *>(x), static_cast<const T *>(flo), static_cast<T *>(out),
                       C, H, W, quads_per_row, nquads, bsx, bsf, bso, flow_scale, align_corners, thr, vec);
    return pwc::check_launch("warp_kernel");
}

// ---- backward --------------------------------------------------------------------------------------------
// d out / d x  : scatter of mask * bilinear weight * grad_out to the four taps.  Which output pixels sample a given source
//                pixel depends on the flow, so a gather form would need a search window as large as the largest flow; the
//                scatter is made DETERMINISTIC instead: with a caller-provided workspace the contributions are accumulated
//                as 64-bit fixed-point integers (integer addition is associative, so the atomics' arrival order does not
//                matter), scaled so that the largest |grad_out| sits 40 bits above the unit in the last place (fp32 has
//                24), and converted to float once at the end.  Without a workspace: float atomicAdd, like torch's
//                grid_sample backward (summation order not fixed);
// d out / d flo: mask * sum_c grad_out * (d sample / d ix, d sample / d iy) * d(ix,iy)/d(u,v), where
//                d ix / d u = flow_scale * W / max(W-1,1)        (align_corners = 0)
//                           = flow_scale * (W-1) / max(W-1,1)    (align_corners = 1).
// The mask is a constant: the reference thresholds it in place (PWCNet.py:174-175), which cuts its graph.
// largest |grad_out| as float bits (order-independent: max), into *out (zeroed before).  The bit pattern of |v| orders like the
// value for finite numbers and puts +inf (0x7f800000) and every NaN (> 0x7f800000) above them, so a non-finite gradient leaves
// *out >= 0x7f800000: the fixed-point path cannot represent it and the kernels below switch to float atomics for that call
// (Inf / NaN then propagate into grad_x exactly as in the float path and in torch's grid_sample backward).
constexpr unsigned kNonFiniteBits = 0x7f800000u;
__global__ void __launch_bounds__(256)
absmax_kernel(const float *__restrict__ v, int64_t n, unsigned *out) {
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        m = max(m, __float_as_uint(fabsf(v[i])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// fixed-point scale for a tensor whose largest magnitude has float bits `mbits`: 2^(40 - exponent)
__device__ __forceinline__ float fixed_scale(unsigned mbits) {
    const int e = (int)(mbits >> 23) - 127;                  // floor(log2(max)); -127 for zero / subnormal
    return __uint_as_float((unsigned)(min(max(40 - e, -126), 127) + 127) << 23);
}

__global__ void __launch_bounds__(256)
fixed_to_float_kernel(const long long *__restrict__ acc, const unsigned *__restrict__ mbits, float *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || *mbits >= kNonFiniteBits) return;            // non-finite grad_out: warp_bwd_kernel already wrote grad_x (float atomics)
    out[i] = (float)((double)acc[i] / (double)fixed_scale(*mbits));
}

template <bool FIXED>
__global__ void __launch_bounds__(kWarpThreads)
warp_bwd_kernel(const float *__restrict__ x, const float *__restrict__ flo, const float *__restrict__ go,
                float *__restrict__ gx, long long *__restrict__ gacc, const unsigned *__restrict__ mbits,
                float *__restrict__ gflo, int C, int H, int W, int64_t npix,
                float flow_scale, int align_corners, float thr) {
    const int64_t i = (int64_t)blockIdx.x * kWarpThreads + threadIdx.x;
    if (i >= npix) return;
    const int64_t plane = (int64_t)H * W;
    const int b = (int)(i / plane);
    const int pix = (int)(i - (int64_t)b * plane);
    const int y = pix / W, xx = pix - y * W;
    const float u = flo[(int64_t)b * 2 * plane + pix] * flow_scale;
    const float v = flo[(int64_t)b * 2 * plane + plane + pix] * flow_scale;
    const float px = (float)xx + u, py = (float)y + v;
    // same coordinate arithmetic as make_taps, but the unmasked factors are needed separately
    const float gxn = 2.0f * px / (float)max(W - 1, 1) - 1.0f;
    const float gyn = 2.0f * py / (float)max(H - 1, 1) - 1.0f;
    float ix, iy;
    if (align_corners) {
        ix = (gxn + 1.0f) / 2.0f * (float)(W - 1);
        iy = (gyn + 1.0f) / 2.0f * (float)(H - 1);
    } else {
        ix = ((gxn + 1.0f) * (float)W - 1.0f) / 2.0f;
        iy = ((gyn + 1.0f) * (float)H - 1.0f) / 2.0f;
    }
    const bool wild = (ix < -16.0f) || (ix > (float)W + 16.0f) || (iy < -16.0f) || (iy > (float)H + 16.0f);
    ix = fminf(fmaxf(ix, -16.0f), (float)W + 16.0f);
    iy = fminf(fmaxf(iy, -16.0f), (float)H + 16.0f);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float ax1 = ix - fx, ay1 = iy - fy, ax0 = 1.0f - ax1, ay0 = 1.0f - ay1;
    const bool vx0 = (x0 >= 0) && (x0 < W), vx1 = (x0 + 1 >= 0) && (x0 + 1 < W);
    const bool vy0 = (y0 >= 0) && (y0 < H), vy1 = (y0 + 1 >= 0) && (y0 + 1 < H);
    const bool v00 = vx0 && vy0, v01 = vx1 && vy0, v10 = vx0 && vy1, v11 = vx1 && vy1;
    const float w00 = v00 ? ay0 * ax0 : 0.f, w01 = v01 ? ay0 * ax1 : 0.f;
    const float w10 = v10 ? ay1 * ax0 : 0.f, w11 = v11 ? ay1 * ax1 : 0.f;
    const float msum = ((w00 + w01) + w10) + w11;
    const bool keep = (msum >= thr) && !wild;
    float *gfu = gflo + (int64_t)b * 2 * plane + pix;
    if (!keep) {
        gfu[0] = 0.f;
        gfu[plane] = 0.f;
        return;
    }
    const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1);
    const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0 + 1, 0), H - 1);
    const int o00 = yc0 * W + xc0, o01 = yc0 * W + xc1, o10 = yc1 * W + xc0, o11 = yc1 * W + xc1;
    const float *xb = x + (int64_t)b * C * plane;
    const int64_t gbase = (int64_t)b * C * plane;
    const float *gob = go + (int64_t)b * C * plane + pix;
    float fscale = 0.f;
    bool fixed = FIXED;
    if constexpr (FIXED) {
        fixed = *mbits < kNonFiniteBits;                        // uniform over the grid
        fscale = fixed_scale(*mbits);
    }
    float dix = 0.f, diy = 0.f;
    for (int c = 0; c < C; ++c, xb += plane, gob += plane) {
        const float g = gob[0];
        const float s00 = v00 ? xb[o00] : 0.f, s01 = v01 ? xb[o01] : 0.f;
        const float s10 = v10 ? xb[o10] : 0.f, s11 = v11 ? xb[o11] : 0.f;
        dix += g * (ay0 * (s01 - s00) + ay1 * (s11 - s10));
        diy += g * (ax0 * (s10 - s00) + ax1 * (s11 - s01));
        const int64_t cb = gbase + (int64_t)c * plane;
        if (fixed) {
            // round(g * w * 2^k) as int64: deterministic per contribution, associative in the sum
            unsigned long long *a = reinterpret_cast<unsigned long long *>(gacc) + cb;
            if (w00 != 0.f) atomicAdd(a + o00, (unsigned long long)__double2ll_rn((double)(g * w00) * (double)fscale));
            if (w01 != 0.f) atomicAdd(a + o01, (unsigned long long)__double2ll_rn((double)(g * w01) * (double)fscale));
            if (w10 != 0.f) atomicAdd(a + o10, (unsigned long long)__double2ll_rn((double)(g * w10) * (double)fscale));
            if (w11 != 0.f) atomicAdd(a + o11, (unsigned long long)__double2ll_rn((double)(g * w11) * (double)fscale));
        } else {
            float *gxb = gx + cb;
            if (w00 != 0.f) atomicAdd(gxb + o00, g * w00);
            if (w01 != 0.f) atomicAdd(gxb + o01, g * w01);
            if (w10 != 0.f) atomicAdd(gxb + o10, g * w10);
            if (w11 != 0.f) atomicAdd(gxb + o11, g * w11);
        }
    }
    const float sx = align_corners ? (float)(W - 1) / (float)max(W - 1, 1) : (float)W / (float)max(W - 1, 1);
    const float sy = align_corners ? (float)(H - 1) / (float)max(H - 1, 1) : (float)H / (float)max(H - 1, 1);
    gfu[0] = dix * flow_scale * sx;
    gfu[plane] = diy * flow_scale * sy;
}

}  // namespace

extern "C" int64_t pwc_warp_bwd_workspace_bytes(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return -1;
    return (int64_t)B * C * H * W * 8 + 16;                     // int64 accumulators + the |grad_out| maximum
}

extern "C" int pwc_warp_bwd(const void *x, const void *flo, const void *grad_out, void *grad_x, void *grad_flo,
                            int B, int C, int H, int W,
                            float flow_scale, int align_corners, float mask_threshold, int dtype,
                            void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !flo || !grad_out || !grad_x || !grad_flo) PWC_FAIL(PWC_EINVAL, "pwc_warp_bwd: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_warp_bwd: bad shape %dx%dx%dx%d", B, C, H, W);
    if (dtype != PWC_F32) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_warp_bwd: dtype %d (f32 only)", dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t npix = (int64_t)B * H * W;
    const int64_t nel = npix * C;
    const int64_t nblk = (npix + kWarpThreads - 1) / kWarpThreads;
    if (nblk > 0x7fffffffLL || (nel + 255) / 256 > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_warp_bwd: grid too large");
    const float *xf = static_cast<const float *>(x), *ff = static_cast<const float *>(flo), *gf = static_cast<const float *>(grad_out);
    if (workspace) {
        // deterministic path: 64-bit fixed-point accumulation in the caller's workspace
        if (workspace_bytes < pwc_warp_bwd_workspace_bytes(B, C, H, W) || (reinterpret_cast<uintptr_t>(workspace) & 7u))
            PWC_FAIL(PWC_EINVAL, "pwc_warp_bwd: workspace needs %lld bytes, 8-byte aligned", (long long)pwc_warp_bwd_workspace_bytes(B, C, H, W));
        long long *acc = static_cast<long long *>(workspace);
        unsigned *mbits = reinterpret_cast<unsigned *>(acc + nel);
        hipError_t e = hipMemsetAsync(workspace, 0, (size_t)nel * 8 + 16, st);
        if (e == hipSuccess) e = hipMemsetAsync(grad_x, 0, (size_t)nel * sizeof(float), st);     // target of the non-finite fallback
        if (e != hipSuccess) { pwc::set_error("pwc_warp_bwd: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
        const int mblk = (int)((nel + 255) / 256 < 2048 ? (nel + 255) / 256 : 2048);
        hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)mblk), dim3(256), 0, st, gf, nel, mbits);
        hipLaunchKernelGGL(warp_bwd_kernel<true>, dim3((unsigned)nblk), dim3(kWarpThreads), 0, st, xf, ff, gf,
                           static_cast<float *>(grad_x), acc, mbits, static_cast<float *>(grad_flo), C, H, W, npix,
                           flow_scale, align_corners, mask_threshold);
        hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, acc, mbits,
                           static_cast<float *>(grad_x), nel);
        return pwc::check_launch("warp_bwd_kernel<fixed>");
    }
    hipError_t e = hipMemsetAsync(grad_x, 0, (size_t)nel * sizeof(float), st);
    if (e != hipSuccess) { pwc::set_error("pwc_warp_bwd: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(warp_bwd_kernel<false>, dim3((unsigned)nblk), dim3(kWarpThreads), 0, st, xf, ff, gf,
                       static_cast<float *>(grad_x), static_cast<long long *>(nullptr), static_cast<const unsigned *>(nullptr),
                       static_cast<float *>(grad_flo), C, H, W, npix, flow_scale, align_corners, mask_threshold);
    return pwc::check_launch("warp_bwd_kernel");
}

extern "C" int pwc_warp_fwd(const void *x, const void *flo, void *out,
                            int B, int C, int H, int W,
                            float flow_scale, int align_corners, float mask_threshold, int dtype,
                            int64_t x_bstride, int64_t flo_bstride, int64_t out_bstride,
                            void *stream) {
    if (!x || !flo || !out) PWC_FAIL(PWC_EINVAL, "pwc_warp_fwd: null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_warp_fwd: bad shape %dx%dx%dx%d", B, C, H, W);
    const int64_t plane = (int64_t)H * W;
    if (x_bstride < C * plane || out_bstride < C * plane || flo_bstride < 2 * plane)
        PWC_FAIL(PWC_EINVAL, "pwc_warp_fwd: batch stride smaller than the tensor");
    if (x == out) PWC_FAIL(PWC_EINVAL, "pwc_warp_fwd: in-place warp is not defined");
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case PWC_F32:
            return launch_warp<float>(x, flo, out, B, C, H, W, flow_scale, align_corners, mask_threshold,
                                      x_bstride, flo_bstride, out_bstride, st);
        case PWC_F16:
            return launch_warp<__half>(x, flo, out, B, C, H, W, flow_scale, align_corners, mask_threshold,
                                       x_bstride, flo_bstride, out_bstride, st);
        default:
            PWC_FAIL(PWC_EUNSUPPORTED, "pwc_warp_fwd: dtype %d", dtype);
    }
}
