// Convolution entry points of libpwc_hip.so: filter packing and the dispatch of the 3x3 convolutions to
//   * the MFMA implicit GEMM (kernel template in pwc_conv_mfma.h, instantiated per (stride, dilation) in
//     pwc_conv_s*d*.hip), or
//   * the VALU kernel for the 2-channel flow heads (pwc_conv_head.hip).
// Replaces nn.Conv2d(3x3)+LeakyReLU(0.1) (reference models/PWCNet.py:26-33) for every layer
// PWCDCNet.forward runs (PWCNet.py:184-268).  The transposed convolutions live in pwc_deconv.hip.
#include <algorithm>

#include "pwc_conv_mfma.h"

namespace pwc_conv {
int run_s1d1(const ConvArgs &a);
int run_s1d2(const ConvArgs &a);
int run_s1d4(const ConvArgs &a);
int run_s1d8(const ConvArgs &a);
int run_s1d16(const ConvArgs &a);
int run_s2d1(const ConvArgs &a);
int run_head(const float *x, const float *w_raw, const float *bias, const float *residual, float *y,
             int B, int Cin, int H, int W, int Cout, int64_t bsx, int64_t bsy, int64_t bsr,
             float slope, int do_leaky, hipStream_t st);
int image_conv_s2(const float *x, const float *wp, const float *bias, float *y, int B, int Cin, int H, int W, int Cout, int CoutP,
                  int64_t bsx, int64_t bsy, float slope, int do_leaky, hipStream_t st);      // pwc_conv_image.hip
bool stream3x3_ok(int B, int Cin, int H, int W, const void *x, int64_t bsx);
bool stream3x3_head_sliced_ok(int B, int Cin, int H, int W, const void *x, int64_t bsx, const void *ws, int64_t ws_bytes);
bool stream3x3_head_upfeat_sliced_ok(int B, int Cin, int H, int W, const void *x, int64_t bsx, const void *ws, int64_t ws_bytes);
int64_t stream3x3_head_workspace_bytes(int B, int Cin, int H, int W);
int64_t stream3x3_head_upfeat_workspace_bytes(int B, int Cin, int H, int W);
int stream3x3_head(const float *x, const float *w, const float *bias, const float *residual, float *y,
                   int B, int Cin, int H, int W, int64_t bsx, int64_t bsy, int64_t bsr,
                   float slope, int do_leaky, hipStream_t st, float *ws, int64_t ws_bytes);
int stream3x3_head_upfeat(const float *x, int B, int Cin, int H, int W, int64_t bsx,
                          const float *hw, const float *hbias, float *hy, int64_t bshy,
                          const float *uw, const float *ubias, float *uy, int64_t bsuy, hipStream_t st, float *ws, int64_t ws_bytes);
}  // namespace pwc_conv

namespace {

constexpr int kCK = pwc_conv::kPackCK;     // packing granularity (kernels consume 4- or 8-channel chunks of it)

inline int cout_padded(int Cout) { return (Cout + 31) / 32 * 32; }
inline int conv_chunks(int Cin) { return (Cin + kCK - 1) / kCK; }
// floats of the MFMA image [chunk][c][tap][CoutP]; 2-channel heads append the raw [Cout][Cin][9] filters
// (read by the VALU head kernel) after it
inline int64_t mfma_image_floats(int Cin, int Cout) { return (int64_t)conv_chunks(Cin) * kCK * 9 * cout_padded(Cout); }
//   tail layout: [ci][20] = {co0: 9 taps, 0, co1: 9 taps, 0}: 80-byte rows so a 4-channel chunk is 20 aligned 16-byte pieces
inline int64_t raw_tail_floats(int Cin, int Cout) { return Cout == 2 ? (int64_t)Cin * 20 : 0; }

// wp[chunk][c][tap][co] <- w[co][chunk*8 + c][tap], zero padded to CoutP columns / 8-channel chunks
__global__ void __launch_bounds__(256)
pack3x3_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cin, int Cout, int CoutP, int64_t image,
               int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (i >= image) {                          // raw tail
        const int64_t k = i - image;
        const int ci = (int)(k / 20), r = (int)(k % 20);
        const int co = r / 10, tap = r % 10;
        wp[i] = (tap < 9) ? w[((int64_t)co * Cin + ci) * 9 + tap] : 0.f;
        return;
    }
    const int co = (int)(i % CoutP);
    int64_t t = i / CoutP;
    const int tap = (int)(t % 9);
    t /= 9;
    const int cin = (int)t;                    // chunk*8 + c
    float v = 0.f;
    if (co < Cout && cin < Cin) v = w[((int64_t)co * Cin + cin) * 9 + tap];
    wp[i] = v;
}

// y = act(bias + sum_z partial[z]) (+ residual): fixed z order, one thread per output element
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ bias, const float *__restrict__ residual,
                     float *__restrict__ y, int Cout, int oplane, int ksplit, int64_t per_image, int64_t total,
                     int64_t bsy, int64_t bsr, float slope, int do_leaky) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t b = i / per_image;
    const int64_t r = i - b * per_image;                 // co*oplane + pixel
    float v = bias[(int)(r / oplane)];
    for (int z = 0; z < ksplit; ++z) v += partial[(int64_t)z * total + i];
    if (do_leaky) v = pwc::leaky(v, slope);
    if (residual) v += residual[b * bsr + r];
    y[b * bsy + r] = v;
}

// does this layer take the split-K route?  (stride 1, dilation 1 only)
inline pwc_conv::SplitPlan split_for(int B, int Cin, int H, int W, int Cout, int stride, int dilation) {
    if (stride != 1 || dilation != 1) return {1, 0};
    return pwc_conv::plan_split(B, Cin, H, W, cout_padded(Cout));
}

}  // namespace

namespace pwc_conv {
// partial [ksplit][B][Cout][plane] -> y (bias, LeakyReLU, optional residual), fixed z order; shared with pwc_conv_wino.hip
int splitk_reduce(const float *partial, const float *bias, const float *residual, float *y, int B, int Cout, int plane, int ksplit,
                  int64_t bsy, int64_t bsr, float slope, int do_leaky, hipStream_t st) {
    const int64_t per_image = (int64_t)Cout * plane, total = per_image * B;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       partial, bias, residual, y, Cout, plane, ksplit, per_image, total, bsy, bsr, slope, do_leaky);
    return pwc::check_launch("splitk_reduce_kernel");
}
}  // namespace pwc_conv

extern "C" int64_t pwc_conv2d_workspace_bytes(int B, int Cin, int H, int W, int Cout, int stride, int dilation) {
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return -1;
    const pwc_conv::SplitPlan sp = split_for(B, Cin, H, W, Cout, stride, dilation);
    int64_t need = sp.ksplit > 1 ? (int64_t)sp.ksplit * B * Cout * H * W * (int64_t)sizeof(float) : 0;
    if (Cout == 2 && stride == 1 && dilation == 1) need = std::max(need, pwc_conv::stream3x3_head_workspace_bytes(B, Cin, H, W));   // Cin slices of the streaming head
    return need;
}


extern "C" int64_t pwc_conv3x3_packed_bytes(int Cin, int Cout, int dtype) {
    if (Cin <= 0 || Cout <= 0 || dtype != PWC_F32) return -1;
    return (mfma_image_floats(Cin, Cout) + raw_tail_floats(Cin, Cout)) * (int64_t)sizeof(float);
}

extern "C" int pwc_conv3x3_pack(const void *w, void *wp, int Cin, int Cout, int dtype, void *stream) {
    if (!w || !wp) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_pack: null pointer");
    if (Cin <= 0 || Cout <= 0) PWC_FAIL(PWC_EINVAL, "pwc_conv3x3_pack: bad shape");
    if (dtype != PWC_F32) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv3x3_pack: weights must be f32");
    if (!pwc::aligned16(wp)) PWC_FAIL(PWC_EALIGN, "pwc_conv3x3_pack: packed buffer must be 16-byte aligned");
    const int64_t total = pwc_conv3x3_packed_bytes(Cin, Cout, dtype) / (int64_t)sizeof(float);
    const int64_t nblk = (total + 255) / 256;
    hipLaunchKernelGGL(pack3x3_kernel, dim3((unsigned)nblk), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(w), static_cast<float *>(wp), Cin, Cout, cout_padded(Cout),
                       mfma_image_floats(Cin, Cout), total);
    return pwc::check_launch("pack3x3_kernel");
}

extern "C" int pwc_conv2d_fwd(const void *x, const void *wp, const void *bias, const void *residual, void *y,
                              int B, int Cin, int H, int W, int Cout,
                              int stride, int dilation, int dtype, unsigned flags, float leaky_slope,
                              int64_t x_bstride, int64_t y_bstride, int64_t res_bstride,
                              void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !wp || !bias || !y) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: null pointer");
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: bad shape");
    if (dtype != PWC_F32) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv2d_fwd: dtype %d", dtype);
    if (!pwc::aligned16(wp)) PWC_FAIL(PWC_EALIGN, "pwc_conv2d_fwd: packed weights must be 16-byte aligned");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 3u)
        PWC_FAIL(PWC_EALIGN, "pwc_conv2d_fwd: tensors must be 4-byte aligned");
    if ((flags & PWC_CONV_RESIDUAL) && !residual) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: residual flag without pointer");
    const int64_t plane = (int64_t)H * W;
    if (x_bstride < Cin * plane) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: x batch stride < Cin*H*W");
    if (plane * kCK * 4 >= 0x7fffffffLL) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_conv2d_fwd: image plane too large for 32-bit DMA offsets");
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
    if (Cin == 3 && Cout == 16 && stride == 2 && dilation == 1 && !a.residual) {
        // conv1a on the image: its own kernel (16-byte patch loads, filters in registers) where the geometry allows it
        const int rc = pwc_conv::image_conv_s2(a.x, a.wp, a.bias, a.y, B, Cin, H, W, Cout, a.CoutP, a.bsx, a.bsy, a.slope, a.do_leaky, a.stream);
        if (rc != PWC_EUNSUPPORTED) return rc;
    }
    // split-K: few output tiles and a long Cin -> partial sums over Cin ranges into the caller's workspace, then
    // a fixed-order reduction.  Without a (large enough) workspace the layer runs unsplit.
    const pwc_conv::SplitPlan sp = split_for(B, Cin, H, W, Cout, stride, dilation);
    const bool split = sp.ksplit > 1 && workspace &&
                       workspace_bytes >= (int64_t)sp.ksplit * B * Cout * plane * (int64_t)sizeof(float) &&
                       !(reinterpret_cast<uintptr_t>(workspace) & 3u);
    float *ws = (workspace && !(reinterpret_cast<uintptr_t>(workspace) & 15u)) ? static_cast<float *>(workspace) : nullptr;
    const bool head_sliced = Cout == 2 && stride == 1 && dilation == 1 && pwc_conv::stream3x3_head_sliced_ok(B, Cin, H, W, a.x, a.bsx, ws, workspace_bytes);
    if (Cout == 2 && stride == 1 && dilation == 1 && !(split && !head_sliced && !pwc_conv::stream3x3_ok(B, Cin, H, W, a.x, a.bsx))) {
        // 2-channel heads: stream the arena through the LDS ring when the image is wide enough, split Cin over
        // waves when the level is tiny; in between the MFMA kernel (MT=1) is still the fastest
        const float *w_raw = a.wp + mfma_image_floats(Cin, Cout);
        int rc = PWC_EUNSUPPORTED;
        if (head_sliced || pwc_conv::stream3x3_ok(B, Cin, H, W, a.x, a.bsx))
            rc = pwc_conv::stream3x3_head(a.x, w_raw, a.bias, a.residual, a.y, B, Cin, H, W, a.bsx, a.bsy, a.bsr,
                                          a.slope, a.do_leaky, a.stream, ws, workspace_bytes);
        else if ((int64_t)B * H * W <= 4096)
            rc = pwc_conv::run_head(a.x, w_raw, a.bias, a.residual, a.y, B, Cin, H, W, Cout, a.bsx, a.bsy, a.bsr,
                                    a.slope, a.do_leaky, a.stream);
        if (rc != PWC_EUNSUPPORTED) return rc;
    }
    if (split) {
        a.partial = static_cast<float *>(workspace);
        a.ksplit = sp.ksplit;
        a.cps = sp.cps;
        if (const int rc = pwc_conv::run_s1d1(a)) return rc;
        return pwc_conv::splitk_reduce(a.partial, a.bias, a.residual, a.y, B, Cout, (int)plane, sp.ksplit, a.bsy, a.bsr, a.slope,
                                       a.do_leaky, a.stream);
    }
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

// predict_flowL + upfeatL in ONE pass over the level's arena (they read the same 3x3 window of the same
// [B,Cin,H,W] tensor; reference models/PWCNet.py:207+209, 221+223, 235+237, 251+253).  Returns
// PWC_EUNSUPPORTED when the geometry is outside the streaming kernel's range: the caller then issues
// pwc_conv2d_fwd and pwc_deconv4x4s2_fwd separately.
extern "C" int pwc_head_upfeat_fwd(const void *x, const void *head_wp, const void *head_bias, void *flow,
                                   const void *up_w, const void *up_bias, void *up_out,
                                   int B, int Cin, int H, int W, int dtype,
                                   int64_t x_bstride, int64_t flow_bstride, int64_t up_bstride, void *stream) {
    return pwc_head_upfeat_ws_fwd(x, head_wp, head_bias, flow, up_w, up_bias, up_out, B, Cin, H, W, dtype, x_bstride, flow_bstride, up_bstride,
                                  nullptr, 0, stream);
}

extern "C" int64_t pwc_head_upfeat_workspace_bytes(int B, int Cin, int H, int W) {
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0) return -1;
    return pwc_conv::stream3x3_head_upfeat_workspace_bytes(B, Cin, H, W);
}

// The same with a scratch buffer (pwc_head_upfeat_workspace_bytes): launches of fewer tiles than the chip has CUs are cut along Cin into
// slices (pwc_stream3x3.hip: slice_plan), partial sums in the workspace, fixed-order reduction.  Without one (or too small): one pass.
extern "C" int pwc_head_upfeat_ws_fwd(const void *x, const void *head_wp, const void *head_bias, void *flow,
                                      const void *up_w, const void *up_bias, void *up_out,
                                      int B, int Cin, int H, int W, int dtype,
                                      int64_t x_bstride, int64_t flow_bstride, int64_t up_bstride,
                                      void *workspace, int64_t workspace_bytes, void *stream) {
    if (!x || !head_wp || !head_bias || !flow || !up_w || !up_bias || !up_out) PWC_FAIL(PWC_EINVAL, "pwc_head_upfeat_fwd: null pointer");
    if (B <= 0 || Cin <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_head_upfeat_fwd: bad shape");
    if (dtype != PWC_F32) PWC_FAIL(PWC_EUNSUPPORTED, "pwc_head_upfeat_fwd: dtype %d", dtype);
    if (x_bstride < (int64_t)Cin * H * W) PWC_FAIL(PWC_EINVAL, "pwc_head_upfeat_fwd: x batch stride < Cin*H*W");
    const float *xf = static_cast<const float *>(x);
    if (!pwc_conv::stream3x3_ok(B, Cin, H, W, xf, x_bstride) &&
        !pwc_conv::stream3x3_head_upfeat_sliced_ok(B, Cin, H, W, xf, x_bstride, workspace, workspace_bytes)) {
        pwc::set_error("pwc_head_upfeat_fwd: geometry %dx%dx%dx%d outside the streaming kernel (needs W %% 4 == 0, W >= 64)", B, Cin, H, W);
        return PWC_EUNSUPPORTED;
    }
    const float *w_raw = static_cast<const float *>(head_wp) + mfma_image_floats(Cin, 2);
    const int rc = pwc_conv::stream3x3_head_upfeat(xf, B, Cin, H, W, x_bstride, w_raw, static_cast<const float *>(head_bias),
                                                   static_cast<float *>(flow), flow_bstride, static_cast<const float *>(up_w),
                                                   static_cast<const float *>(up_bias), static_cast<float *>(up_out), up_bstride,
                                                   static_cast<hipStream_t>(stream), static_cast<float *>(workspace), workspace_bytes);
    if (rc == PWC_EUNSUPPORTED) pwc::set_error("pwc_head_upfeat_fwd: outputs must be 16-byte aligned");
    return rc;
}
