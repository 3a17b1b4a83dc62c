// ConvTranspose2d(kernel 4, stride 2, padding 1) with 2 output channels: deconv{6..3} (2 -> 2) and
// upfeat{6..3} (529..597 -> 2) of the reference (models/PWCNet.py:35-36, used :208-209,222-223,236-237,252-253).
//
//   y[co, 2*iy + py, 2*ix + px] = bias[co] + sum_ci sum_{2x2 taps} x[ci, iy + dy, ix + dx] * w[ci, co, ky, kx]
//   with oy = 2*iy' - 1 + ky:   py = 0 -> (iy-1, ky=3), (iy, ky=1);   py = 1 -> (iy, ky=2), (iy+1, ky=0)   (same in x)
//
// Input-anchored: a lane owns one input pixel and produces the 2x2 output quad it anchors from its 3x3
// neighbourhood, so every input value is loaded once per lane and reused by 4 outputs x 2 couts.  The
// reduction over Cin is split across the 8 or 16 waves of the workgroup (wave w takes channels w, w+waves, ...), which
// all cover the SAME 8x8 pixel tile; partial sums meet in LDS.  That keeps ~600-channel upfeat layers from
// being one long serial loop per thread (the first version ran 300 us per call regardless of level).
// Weights are wave-uniform (scalar loads).  fp32 only.
// Small levels (a single pair: 2-112 workgroups) take 21-26 us per launch whatever the level: that is VALU ISSUE on the few CUs that
// hold the whole Cin x 64-pixel product (16 waves x ~35 channels x ~90 instructions on one CU), not latency -- round 4 measured
// deeper load pipelines (two / three channels per trip: 24-30 us) and lane-per-weight vector loads + v_readlane (28-33 us), both
// slower; what would help is more CUs per pixel tile (channel slices in separate workgroups + a deterministic combine).
#include "pwc_common.h"

namespace pwc_conv {
bool stream3x3_ok(int B, int Cin, int H, int W, const void *x, int64_t bsx);
int stream3x3_upfeat(const float *x, const float *w, const float *bias, float *y,
                     int B, int Cin, int H, int W, int64_t bsx, int64_t bsy, hipStream_t st);
}  // namespace pwc_conv

namespace {

// waves per workgroup: 16 while the grid cannot fill the chip anyway (batch-1 forward 2.47 -> 2.41 ms), 8 otherwise
constexpr int kDTH = 8;              // 8 x 8 input pixels per workgroup
constexpr int kDTW = 8;

template <int CO, int kDWaves>
__global__ void __launch_bounds__(64 * kDWaves)
deconv4x4s2_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                   float *__restrict__ y, int Cin, int H, int W, int tiles_x, int tiles_y,
                   int64_t bsx, int64_t bsy) {
    __shared__ float red[kDWaves][CO * 4][64];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int iy = ty * kDTH + (lane >> 3);
    const int ix = tx * kDTW + (lane & 7);
    const int64_t plane = (int64_t)H * W;
    const float *xb = x + (int64_t)b * bsx;

    // clamped neighbourhood offsets + validity (out-of-image taps contribute 0)
    int off[3][3];
    float msk[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int yy = iy - 1 + a, xx = ix - 1 + c;
            const bool ok = (yy >= 0) && (yy < H) && (xx >= 0) && (xx < W);
            off[a][c] = min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1);
            msk[a][c] = ok ? 1.f : 0.f;
        }

    float acc[CO][2][2];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co][0][0] = acc[co][0][1] = acc[co][1][0] = acc[co][1][1] = 0.f;

    for (int ci = wave; ci < Cin; ci += kDWaves) {
        const float *xp = xb + (int64_t)ci * plane;
        float v[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) v[a][c] = xp[off[a][c]] * msk[a][c];
        const float *wc = w + (int64_t)ci * CO * 16;          // wave-uniform
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            const float *k = wc + co * 16;                    // k[ky*4 + kx]
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const int kya = py ? 2 : 3, kyb = py ? 0 : 1;       // rows v[py], v[py+1]
                    const int kxa = px ? 2 : 3, kxb = px ? 0 : 1;       // cols v[.][px], v[.][px+1]
                    float s = acc[co][py][px];
                    s = fmaf(v[py][px], k[kya * 4 + kxa], s);
                    s = fmaf(v[py][px + 1], k[kya * 4 + kxb], s);
                    s = fmaf(v[py + 1][px], k[kyb * 4 + kxa], s);
                    s = fmaf(v[py + 1][px + 1], k[kyb * 4 + kxb], s);
                    acc[co][py][px] = s;
                }
        }
    }
#pragma unroll
    for (int co = 0; co < CO; ++co)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave][co * 4 + q][lane] = acc[co][q >> 1][q & 1];
    __syncthreads();
    // wave w finishes output value (co, q) = w of every pixel: fixed summation order -> deterministic
    if (wave < CO * 4) {
        float s = bias[wave >> 2];
#pragma unroll
        for (int k = 0; k < kDWaves; ++k) s += red[k][wave][lane];
        if (iy < H && ix < W) {
            const int co = wave >> 2, py = (wave >> 1) & 1, px = wave & 1;
            y[(int64_t)b * bsy + (int64_t)co * 4 * plane + (int64_t)(2 * iy + py) * (2 * W) + 2 * ix + px] = s;
        }
    }
}

// ---- small levels: predict_flowL and upfeatL as ONE 10-channel 3x3 convolution, finished here -------------------------------------
// deconv4x4s2_kernel above is VALU-issue bound on the few CUs a small map gives it (21-26 us per launch).  The host plan instead runs
// ConvTranspose2d(k4, s2, p1) as what it is -- a 3x3 convolution with four output phases per channel -- together with the flow head
// on the matrix cores (pwc_conv2d_fwd, split-K over the whole chip): `head` = [flow u, v | upfeat phases co*4 + py*2 + px].  This
// kernel is the rest of the level's exit (PWCNet.py:208-209 ...): up_flow = deconvL(flow) (2 -> 2 channels: 2x2 input pixels x 2
// channels per output, fp32) and up_feat = the pixel shuffle of the phases, one thread per output pixel, four channels written.
__global__ void __launch_bounds__(256)
upsample_entry_kernel(const float *__restrict__ head, const float *__restrict__ dw, const float *__restrict__ db,
                      float *__restrict__ out, int h, int w, int64_t npix, int64_t bsh, int64_t bso) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const int H = 2 * h, W = 2 * w;
    const int64_t plane = (int64_t)H * W, ip = (int64_t)h * w;
    const int b = (int)(i / plane);
    const int pix = (int)(i - (int64_t)b * plane);
    const int yy = pix / W, xx = pix - yy * W;
    const int py = yy & 1, px = xx & 1;
    // output row 2Y+py takes input rows r0 = Y-1+py (kernel row 3-py) and r0+1 (kernel row 1-py); same along x
    const int r0 = (yy >> 1) - 1 + py, c0 = (xx >> 1) - 1 + px;
    const float *hb = head + (int64_t)b * bsh;
    float acc0 = db[0], acc1 = db[1];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int r = r0 + a, ky = 3 - py - 2 * a;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int cc = c0 + c, kx = 3 - px - 2 * c;
            if (r < 0 || r >= h || cc < 0 || cc >= w) continue;
            const float fu = hb[(int64_t)r * w + cc], fv = hb[ip + (int64_t)r * w + cc];
            acc0 = fmaf(fu, dw[(0 * 2 + 0) * 16 + ky * 4 + kx], acc0);          // dw[ci][co][ky][kx]
            acc0 = fmaf(fv, dw[(1 * 2 + 0) * 16 + ky * 4 + kx], acc0);
            acc1 = fmaf(fu, dw[(0 * 2 + 1) * 16 + ky * 4 + kx], acc1);
            acc1 = fmaf(fv, dw[(1 * 2 + 1) * 16 + ky * 4 + kx], acc1);
        }
    }
    const int64_t sp = (int64_t)(yy >> 1) * w + (xx >> 1);
    const int ph = py * 2 + px;
    float *ob = out + (int64_t)b * bso + pix;
    ob[0] = acc0;
    ob[plane] = acc1;
    ob[2 * plane] = hb[(2 + ph) * ip + sp];
    ob[3 * plane] = hb[(6 + ph) * ip + sp];
}

}  // namespace

extern "C" int pwc_deconv4x4s2_fwd(const void *x, const void *w, const void *bias, void *y,
                                   int B, int Cin, int H, int W, int Cout, int dtype,
                                   int64_t x_bstride, int64_t y_bstride, void *stream) {
    if (!x || !w || !bias || !y) PWC_FAIL(PWC_EINVAL, "pwc_deconv4x4s2_fwd: null pointer");
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_deconv4x4s2_fwd: bad shape");
    if (Cout != 2) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_deconv4x4s2_fwd: Cout=%d (PWC-Net only has 2-channel deconvs)", Cout);
    if (dtype != PWC_F32) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_deconv4x4s2_fwd: dtype %d", dtype);
    // wide images with many channels: stream the input through the LDS ring (pwc_stream3x3.hip)
    if (Cin >= 16 && pwc_conv::stream3x3_ok(B, Cin, H, W, x, x_bstride)) {
        const int rc = pwc_conv::stream3x3_upfeat(static_cast<const float *>(x), static_cast<const float *>(w),
                                                  static_cast<const float *>(bias), static_cast<float *>(y), B, Cin, H, W,
                                                  x_bstride, y_bstride, static_cast<hipStream_t>(stream));
        if (rc != PWC_EUNSUPPORTED) return rc;
    }
    const int tiles_x = (W + kDTW - 1) / kDTW;
    const int tiles_y = (H + kDTH - 1) / kDTH;
    const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_deconv4x4s2_fwd: grid too large");
    if (nblk <= 256)
        hipLaunchKernelGGL((deconv4x4s2_kernel<2, 16>), dim3((unsigned)nblk), dim3(64 * 16), 0, static_cast<hipStream_t>(stream),
                           static_cast<const float *>(x), static_cast<const float *>(w), static_cast<const float *>(bias),
                           static_cast<float *>(y), Cin, H, W, tiles_x, tiles_y, x_bstride, y_bstride);
    else
        hipLaunchKernelGGL((deconv4x4s2_kernel<2, 8>), dim3((unsigned)nblk), dim3(64 * 8), 0, static_cast<hipStream_t>(stream),
                           static_cast<const float *>(x), static_cast<const float *>(w), static_cast<const float *>(bias),
                           static_cast<float *>(y), Cin, H, W, tiles_x, tiles_y, x_bstride, y_bstride);
    return pwc::check_launch("deconv4x4s2_kernel");
}

/* Exit of a decoder level whose flow head and upfeatL ran as one 10-channel 3x3 convolution (models/PWCNet.py:207-209, 221-223 ...):
 * head [B,10,h,w] = [flow (2) | upfeat phases co*4 + py*2 + px (8)]; out [B,4,2h,2w] = [deconvL(flow) (2) | up_feat (2)], the four
 * channels the next level's arena holds behind c1.  deconv_w [2,2,4,4] / deconv_b [2]: nn.ConvTranspose2d(2,2,4,2,1) of deconvL. */
extern "C" int pwc_upsample_entry_f32(const void *head, const void *deconv_w, const void *deconv_b, void *out, int B, int h, int w,
                                      int64_t head_bstride, int64_t out_bstride, void *stream) {
    if (!head || !deconv_w || !deconv_b || !out) PWC_FAIL(PWC_EINVAL, "pwc_upsample_entry_f32: null pointer");
    if (B <= 0 || h <= 0 || w <= 0) PWC_FAIL(PWC_EINVAL, "pwc_upsample_entry_f32: bad shape");
    if (head_bstride < (int64_t)10 * h * w || out_bstride < (int64_t)16 * h * w)
        PWC_FAIL(PWC_EINVAL, "pwc_upsample_entry_f32: batch stride smaller than the tensor");
    const int64_t npix = (int64_t)B * 4 * h * w;
    if ((npix + 255) / 256 > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_upsample_entry_f32: grid too large");
    hipLaunchKernelGGL(upsample_entry_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(head), static_cast<const float *>(deconv_w), static_cast<const float *>(deconv_b),
                       static_cast<float *>(out), h, w, npix, head_bstride, out_bstride);
    return pwc::check_launch("upsample_entry_kernel");
}
