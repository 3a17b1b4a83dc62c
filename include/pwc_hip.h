/*
 * pwc_hip.h -- C ABI of libpwc_hip.so, the MI355X (gfx950) replacement for the
 * native half of the reference's PWC-Net inference path.
 *
 * Every entry point
 *   - takes plain device pointers, sizes and an opaque HIP stream handle
 *     (void* == hipStream_t; NULL = the null stream),
 *   - never allocates, frees or synchronises (graph-capturable),
 *   - returns 0 on success, a negative PWC_E* code for an argument the kernel
 *     cannot honour (nothing launched), or a positive hipError_t when the
 *     launch failed.  pwc_last_error() returns a thread-local message.
 *
 * Reference interfaces replaced (paths relative to the reference root):
 *   pwc_corr_fwd        correlation_cuda.forward  models/correlation_package/correlation_cuda.cc:10-87
 *                       (+ channels_first / correlation_forward kernels,
 *                        correlation_cuda_kernel.cu:46-147, launcher :336-427)
 *                       and the fallback semantics of correlation.py:12-40
 *   pwc_corr_bwd        correlation_cuda.backward correlation_cuda.cc:89-167, kernels .cu:150-334
 *   pwc_warp_fwd        PWCDCNet.warp             models/PWCNet.py:141-177
 *   pwc_warp_bwd        autograd of the same (grid_sample backward as used by the training scripts)
 *   pwc_conv2d_fwd      conv()/predict_flow()     models/PWCNet.py:26-33 (nn.Conv2d 3x3 + LeakyReLU(0.1))
 *   pwc_deconv4x4s2_fwd deconv()                  models/PWCNet.py:35-36 (nn.ConvTranspose2d k4 s2 p1)
 *
 * All tensors are NCHW with contiguous C,H,W planes; only the batch stride is
 * free (in ELEMENTS), so an operand may be a channel slice of a wider
 * [B, Ctot, H, W] arena (this is how the DenseNet concatenations of
 * PWCNet.py:202-264 are done without copies).
 */
#ifndef PWC_HIP_H_
#define PWC_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PWC_ABI_VERSION 12

/* element types */
#define PWC_F32 0
#define PWC_F16 1

/* error codes (negative = argument error, positive = hipError_t) */
#define PWC_OK 0
#define PWC_EINVAL (-1)     /* bad shape / size / null pointer            */
#define PWC_EUNSUPPORTED (-2) /* valid request this build has no kernel for */
#define PWC_EALIGN (-3)     /* pointer or stride alignment not met        */

/* pwc_corr_fwd / pwc_conv2d_fwd flags */
#define PWC_CORR_NORMALIZE 1u /* divide by kernel_size^2*C (correlation_cuda_kernel.cu:104,143) instead of
                                 multiplying by corr_multiply (correlation.py:35-36)              */
#define PWC_ACT_LEAKY 2u      /* fuse LeakyReLU(slope) into the epilogue (PWCNet.py:72,199)          */
#define PWC_CONV_RESIDUAL 4u  /* y += residual (flow2 + dc_conv7(...), PWCNet.py:268)                */
/* pwc_conv3x3_wino4_fwd only */
#define PWC_CONV_SPLIT2 32u   /* y is [4B][Cout][H/2][W/2] (y_bstride = its image stride): image b is written as its four pixel lattices
                                 4b + 2 (oy & 1) + (ox & 1), so that the next, twice-as-dilated layer of the context network runs as
                                 a dilation-1 convolution on 4B small images; pwc_lattice_unsplit_f32 brings a tensor back */
/* pwc_conv2d_f16_fwd only */
#define PWC_CONV_OUT_F32 8u   /* y is float [B][ceil(Cout/8)][Ho][Wo][8] instead of half (flow heads: the values that
                                 carry the flow from level to level stay in fp32)                    */
#define PWC_CONV_SPLIT_W 16u  /* wp comes from pwc_conv3x3_f16_pack_split: hi + lo filters (~22 bits).  Free for Cout <= 16 (the
                                 2-channel heads: the residuals ride in the idle half of the 32-row cout tile); twice the
                                 MFMA passes otherwise (the strict half-precision mode's level-2 / context layers)        */

int pwc_abi_version(void);
/* 0 for the product.  Bits: a kernel source was built with a timing-experiment switch (1 F(4x4) conv, 2 correlation, 4 pipelined
 * correlation, 8 streaming heads): such a library skips work and its results are invalid; never ship or benchmark one. */
int pwc_experiment_mask(void);
const char *pwc_last_error(void);
/* Run-time switches of the kernel selection (process-wide; tests and A/B benchmarks flip them instead of relying on an
 * environment variable being read before first use).  Each option's default comes from the environment variable in brackets:
 *   "conv_wino4" [PWC_CONV_WINO4] 1, "w4_tailsplit" [PWC_W4_TAILSPLIT] 1, "w4_smallsplit" [PWC_W4_SMALLSPLIT] 1,
 *   "w4_small_min_wgs" [PWC_W4_SMALL_MIN_WGS] 160 (workgroups -- tiles x cout groups x Cin slices -- a launch smaller than the chip must reach for
 *   pwc_conv3x3_wino4_preferred to take it; 96 / 64 measured within +-0.6 % / slower on the whole forward),
 *   "corr_pipe" [PWC_CORR_PIPE] 0 (plain correlation on the round-4 pipelined kernels), "corr_roll" [PWC_CORR_ROLL] 1 (their rolling form),
 *   "corr_pipe_min_tiles" [PWC_CORR_PIPE_MIN_TILES] 1024 (8x32 tiles a launch needs for the round-4 kernels),
 *   "corr_small_tiles" [PWC_CORR_SMALL_TILES] 48 (launches of at most this many 8x32 tiles use the small-map correlation kernel, and
 *   pwc_warp_corr81_preferred sends them to pwc_warp_fwd + pwc_corr_fwd; 0: the tiled kernels always),
 *   "head10" [PWC_HEAD10] 1 (fp32 plans: levels too small for pwc_head_upfeat_fwd run predict_flowL + upfeatL as one 10-channel
 *   convolution followed by pwc_upsample_entry_f32; 0: pwc_conv2d_fwd + two pwc_deconv4x4s2_fwd),
 *   "f16_level_corr" [PWC_F16_LEVEL_CORR] 0 (1: the half-precision plans enter a level through pwc_level_corr81_c8_f16 instead of the two
 *   calls it fuses -- same bits, measured slower at batch 16),
 *   "warpcorr_window" [PWC_WARPCORR_WINDOW] 2 (fused warp+correlation on the LDS-window kernel: 2 = C in (28,32] and (60,64], 1 = C <= 32 only, 0 = off),
 *   "stream_slice_wgs" [PWC_STREAM_SLICE_WGS] 512 (pwc_conv2d_fwd, 2-channel flow head on a map of 8..63 8-row x 128-column tiles --
 *   predict_flow2 of one or two pairs, PWCNet.py:263 -- and pwc_head_upfeat_ws_fwd on fewer than 256 tiles: the streaming kernel runs
 *   on Cin slices, as many as bring the launch to this many workgroups, partial sums in the caller's workspace
 *   (pwc_conv2d_workspace_bytes / pwc_head_upfeat_workspace_bytes), fixed-order reduction; 0: one pass / the split-K MFMA kernel),
 *   "head_sliced_min_tiles" [PWC_HEAD_SLICED_MIN_TILES] 14 (fp32 plans: levels of at least this many 8-row x 128-column tiles run predict_flowL +
 *   upfeatL through pwc_head_upfeat_ws_fwd -- on Cin slices below the 64 tiles its one-pass form needs -- instead of the 10-channel
 *   convolution of "head10"; 64: only where the one-pass kernel runs),
 *   "c1_in_arena" [PWC_C1_IN_ARENA] 1 (fp32 plans: the level features of both images live at the arena's batch stride, so that the pyramid's
 *   last convolution writes the first image's straight into their arena slot; 0: dense pyramid buffers and one copy per level).
 * Unknown name: PWC_EINVAL.  A captured HIP graph keeps the kernels chosen at capture time. */
int pwc_set_option(const char *name, int value);
int pwc_get_option(const char *name, int *value);
/* Name and template arguments of the MFMA convolution variant this thread launched last -- what the tile cost
 * model picked for that layer: "conv3x3_mfma_kernel<MT, NT, stride, dilation, two-per-CU, 0>" (fp32) or
 * "conv3x3_f16_kernel<MT, NT, stride, dilation, ring, 0>" (fp16).  For benchmarks and profiles. */
const char *pwc_last_conv_kernel(void);

/* Cost volume.  in1,in2: [B,C,H,W]; out: [B,(2*(max_disp/stride2)+1)^2,outH,outW] with
 * outH = ceil((H + 2*pad - 2*((k-1)/2 + max_disp)) / stride1)   (correlation_cuda.cc:25-38).
 * out[b,(tj+r)*D+(ti+r),y,x] = sum_{k x k} sum_c in1p[..]*in2p[..], displacement order tj(dy) outer,
 * ti(dx) inner (correlation_cuda_kernel.cu:107-141; identical to correlation.py:28-29). */
int pwc_corr_fwd(const void *in1, const void *in2, void *out,
                 int B, int C, int H, int W,
                 int pad_size, int kernel_size, int max_disp, int stride1, int stride2,
                 float corr_multiply, int dtype, unsigned flags, float leaky_slope,
                 int64_t in1_bstride, int64_t in2_bstride, int64_t out_bstride,
                 void *stream);

/* warp + cost volume in one kernel for the decoder levels below the coarsest (PWCNet.py:212-213, 226-227, 240-241, 256-257:
 * corr = self.corr(c1, self.warp(c2, up_flow * s)); the warped tensor has no other consumer):
 *   out[b, (dy+4)*9 + (dx+4), y, x] = scale * sum_c in1[b,c,y,x] * warp(x2, flow_scale * flo)[b,c,y+dy,x+dx]   (+ LeakyReLU)
 * with pwc_warp_fwd's sampling / mask rule and pwc_corr_fwd's PWC configuration (pad 4, kernel 1, max displacement 4, strides 1);
 * bit-identical to pwc_warp_fwd followed by pwc_corr_fwd on its tiled kernels (launches of more than "corr_small_tiles" 8x32 tiles;
 * below that pwc_corr_fwd runs its small-map kernel, which sums the channels in another order).  f32 only.  Returns PWC_EUNSUPPORTED (nothing launched) unless
 * W % 4 == 0 and in1 / x2 / out are 16-byte aligned: call the two separate entry points then. */
int pwc_warp_corr81_fwd(const void *in1, const void *x2, const void *flo, void *out, int B, int C, int H, int W,
                        float flow_scale, int align_corners, float mask_threshold,
                        float corr_multiply, unsigned flags, float leaky_slope,
                        int64_t in1_bstride, int64_t x2_bstride, int64_t flo_bstride, int64_t out_bstride, void *stream);
/* 1 when the fused kernel is the faster way to warp + correlate this geometry, 0 for maps of a few tiles (levels 6-4, one-pair
 * inference), where pwc_warp_fwd + pwc_corr_fwd (small-map kernel) win by ~3x, the extra launch included. */
int pwc_warp_corr81_preferred(int B, int C, int H, int W);

/* Gradients of pwc_corr_fwd w.r.t. in1 and in2 (no fused activation; same scale rule as forward), for ANY
 * (pad_size, kernel_size, max_disp, stride1, stride2) like the reference's backward (correlation_cuda_kernel.cu:150-334);
 * grad_out: [B, D*D, outH, outW] contiguous, in*, grad_in*: [B,C,H,W] contiguous.  Gather form with a fixed summation
 * order (deterministic, no atomics).  PWC-Net's configuration in fp32 runs an LDS-tiled kernel (one thread per pixel,
 * 81 + 81 grad_out values in registers, both inputs streamed through LDS with their halos). */
int pwc_corr_bwd(const void *in1, const void *in2, const void *grad_out, void *grad_in1, void *grad_in2,
                 int B, int C, int H, int W,
                 int pad_size, int kernel_size, int max_disp, int stride1, int stride2,
                 float corr_multiply, int dtype, unsigned flags,
                 void *stream);

/* Backward warp of x by (flow_scale * flo): bilinear, zero padding, times the validity mask
 * [sum of in-bounds bilinear weights >= mask_threshold]  (PWCNet.py:141-177).
 * x,out: [B,C,H,W]; flo: [B,2,H,W] (u then v).  align_corners=0 reproduces the reference as executed
 * by torch>=1.3: x_src = (x+u)*W/(W-1) - 0.5. */
int pwc_warp_fwd(const void *x, const void *flo, void *out,
                 int B, int C, int H, int W,
                 float flow_scale, int align_corners, float mask_threshold, int dtype,
                 int64_t x_bstride, int64_t flo_bstride, int64_t out_bstride,
                 void *stream);

/* Gradients of pwc_warp_fwd w.r.t. x and flo (contiguous f32 tensors).  The validity mask is a constant, as in the
 * reference, whose in-place thresholding (PWCNet.py:174-175) cuts the mask's graph; what autograd derives for
 * PWCNet.py:141-177 is otherwise reproduced: d/dx through the bilinear taps (a scatter: which output pixels sample a source
 * pixel depends on the flow), d/dflo through the sample coordinates.
 * workspace (device, 8-byte aligned, >= pwc_warp_bwd_workspace_bytes): the scatter accumulates 64-bit FIXED-POINT integers
 * there (integer addition is associative -> bit-reproducible grad_x whatever order the atomics arrive in; resolution 2^-40
 * of the largest |grad_out|) and grad_x is written once at the end.  workspace == NULL: grad_x is zeroed and accumulated
 * with float atomics like torch's grid_sample backward (summation order, hence the last bits, not fixed).
 * Limits of the fixed-point form: (1) one contribution is at most 2^41 in magnitude, so a source pixel that collects more
 * than 2^22 (4 194 304) contributions of the largest |grad_out| would wrap its int64 -- more output pixels than that sampling
 * ONE source pixel; (2) a non-finite grad_out (Inf / NaN) has no fixed-point form: the call then falls back, on the device and
 * without synchronising, to the float-atomic accumulation for the whole tensor, so Inf / NaN reach grad_x exactly as in the
 * float path (that call is not bit-reproducible). */
int64_t pwc_warp_bwd_workspace_bytes(int B, int C, int H, int W);
int pwc_warp_bwd(const void *x, const void *flo, const void *grad_out, void *grad_x, void *grad_flo,
                 int B, int C, int H, int W,
                 float flow_scale, int align_corners, float mask_threshold, int dtype,
                 void *workspace, int64_t workspace_bytes, void *stream);

/* Bytes needed for the packed (kernel-native) form of a [Cout,Cin,3,3] filter bank. */
int64_t pwc_conv3x3_packed_bytes(int Cin, int Cout, int dtype);
/* Repack w:[Cout,Cin,3,3] (device, NCHW filter layout of nn.Conv2d) into wp (device). */
int pwc_conv3x3_pack(const void *w, void *wp, int Cin, int Cout, int dtype, void *stream);

/* 3x3 convolution, padding == dilation (so H,W are preserved at stride 1; Hout = (H-1)/stride+1),
 * + bias, optional LeakyReLU and residual.  x:[B,Cin,H,W], y:[B,Cout,Hout,Wout],
 * wp = output of pwc_conv3x3_pack, bias:[Cout] f32, residual: same geometry as y or NULL.
 * workspace (device, 4-byte aligned, may be NULL): scratch for the split-K route taken by layers with few
 * output tiles and many input channels (pyramid levels 6-4, batch-1 inference): partial sums per Cin range,
 * then a fixed-order reduction (deterministic).  A layer whose pwc_conv2d_workspace_bytes() exceeds
 * workspace_bytes runs unsplit -- same result up to fp32 summation order.  One workspace may be shared by all
 * layers launched on the same stream. */
int pwc_conv2d_fwd(const void *x, const void *wp, const void *bias, const void *residual, void *y,
                   int B, int Cin, int H, int W, int Cout,
                   int stride, int dilation, int dtype, unsigned flags, float leaky_slope,
                   int64_t x_bstride, int64_t y_bstride, int64_t res_bstride,
                   void *workspace, int64_t workspace_bytes,
                   void *stream);
/* Bytes of workspace the split-K route of this layer needs (0: the layer never splits; <0: bad shape). */
int64_t pwc_conv2d_workspace_bytes(int B, int Cin, int H, int W, int Cout, int stride, int dilation);

/* ---- Winograd F(2x2,3x3) route of the same operator (nn.Conv2d 3x3, stride 1, padding = dilation + LeakyReLU,
 * PWCNet.py:26-33), fp32 in / fp32 MFMA accumulation / fp32 out: 16 multiplications per 2x2 outputs instead of 36.
 * The result differs from pwc_conv2d_fwd only by fp32 rounding (the transforms add and halve; tests bound it).
 * up = pwc_conv3x3_wino_pack(w) holds G g Gt per (cout, cin) in the kernel's LDS order [chunk of 4 cin][16][2][CoutP][2].
 * A dilated layer runs as dilation^2 ordinary convolutions on the pixel lattices (y mod D, x mod D).
 * x:[B,Cin,H,W], y:[B,Cout,H,W] with free batch strides (elements); flags: PWC_ACT_LEAKY only. */
int64_t pwc_conv3x3_wino_packed_bytes(int Cin, int Cout);
/* 1 when this route is expected to beat pwc_conv2d_fwd for the layer (enough workgroups for 256 CUs, Cout >= 32), else 0 */
int pwc_conv3x3_wino_preferred(int B, int Cin, int H, int W, int Cout, int dilation);
int pwc_conv3x3_wino_pack(const void *w, void *up, int Cin, int Cout, void *stream);
int pwc_conv3x3_wino_fwd(const void *x, const void *up, const void *bias, void *y,
                         int B, int Cin, int H, int W, int Cout, int dilation, unsigned flags, float leaky_slope,
                         int64_t x_bstride, int64_t y_bstride, void *workspace, int64_t workspace_bytes, void *stream);
/* workspace (device, may be NULL): scratch for the split-K form taken by launches that would leave most CUs idle (levels 5-4);
 * pwc_conv3x3_wino_workspace_bytes() bytes, shareable with pwc_conv2d_fwd's workspace on one stream; without it the layer runs unsplit */
int64_t pwc_conv3x3_wino_workspace_bytes(int B, int Cin, int H, int W, int Cout, int dilation);

/* The same operator by Winograd F(4x4,3x3) (csrc/pwc_conv_wino4.hip, round 3): 36 multiplications per 4x4 outputs -- 1.78x fewer
 * MFMA passes than F(2x2,3x3) -- at ~6x (rms) the fp32 rounding error of F(2x2) (4e-5 instead of 2e-6 at the largest on unit-scale
 * data with 565 input channels).  Dilation 1, W % 4 == 0, 16-byte aligned x / y with batch strides that are multiples of 4
 * (PWC_EUNSUPPORTED / PWC_EALIGN otherwise).  up = pwc_conv3x3_wino4_pack(w): G g Gt (6x6 per filter, computed in double) in the
 * kernel's LDS order [chunk of 4 cin][9 groups of 4 positions][cin][CoutP][4], pwc_conv3x3_wino4_packed_bytes() bytes (4x the filter).
 * pwc_conv3x3_wino4_preferred: the measured rule for when this route beats pwc_conv3x3_wino_fwd (large, well-filled maps).
 * flags: PWC_ACT_LEAKY. */
int64_t pwc_conv3x3_wino4_packed_bytes(int Cin, int Cout);
int pwc_conv3x3_wino4_preferred(int B, int Cin, int H, int W, int Cout, int dilation);
int pwc_conv3x3_wino4_pack(const void *w, void *up, int Cin, int Cout, void *stream);
int pwc_conv3x3_wino4_fwd(const void *x, const void *up, const void *bias, void *y,
                          int B, int Cin, int H, int W, int Cout, int dilation, unsigned flags, float leaky_slope,
                          int64_t x_bstride, int64_t y_bstride, void *workspace, int64_t workspace_bytes, void *stream);
/* workspace (device, 16-byte aligned, may be NULL): pwc_conv3x3_wino4_workspace_bytes() bytes, shareable with the other convolutions'
 * workspaces on one stream.  With it, a launch whose last round of workgroups would leave most CUs idle (n workgroups on 256 CUs cost
 * ceil(n / 256) rounds: 896 -> 4 instead of 3.5) runs the tiles of that round as input-channel slices that fill the chip and adds the
 * slices in a fixed order (deterministic); without it the layer runs unsplit.  Not combined with PWC_CONV_SPLIT2. */
int64_t pwc_conv3x3_wino4_workspace_bytes(int B, int Cin, int H, int W, int Cout);
/* Inverse of `levels` nested PWC_CONV_SPLIT2 stores: x [B * 4^levels][C][h][w] (contiguous) -> y [B][C][h << levels][w << levels]
 * (dense planes, free batch stride); image index ((b*4 + s1)*4 + s2)... with s_i = 2 (y_i & 1) + (x_i & 1), coarsest split first. */
int pwc_lattice_unsplit_f32(const void *x, void *y, int B, int C, int h, int w, int levels, int64_t y_bstride, void *stream);

/* ---- image-space pre / post of the KITTI evaluation loop (reference inference_kitti.py:53-91,175-178,208-224; csrc/pwc_kitti.hip) ----
 * pwc_kitti_ingest_u8: uint8 RGB pairs [n][2][H][W][3] -> x float [n][6][Hp][Wp] (Hp, Wp = H, W rounded up to multiples of 64; free batch
 *   stride in elements, multiple of 4; 16-byte aligned): ToTensor (/ 255), (v - mean3[c]) / std3[c] (host pointers, read at the call),
 *   the two images concatenated along the channels, replicate padding at the bottom / right -- what the reference does with
 *   torchvision transforms, torch.cat and F.pad(mode="replicate") before the model.
 * pwc_flow_upsample_f32: the model's quarter-resolution flow [n][2][Hq][Wq] (free batch stride) -> out [n][2][out_h][out_w] (dense):
 *   crop to the top-left crop_h x crop_w, bilinear resize with align_corners = True (F.interpolate's arithmetic), u * (out_w / crop_w),
 *   v * (out_h / crop_h) -- `unpad` + `flow_resize` of the reference.  Neither allocates nor synchronises. */
int pwc_kitti_ingest_u8(const void *pairs_u8, void *x, int n, int H, int W, const float *mean3, const float *std3,
                        int64_t x_bstride, void *stream);
int pwc_flow_upsample_f32(const void *flow_q, void *out, int n, int Hq, int Wq, int crop_h, int crop_w, int out_h, int out_w,
                          int64_t q_bstride, void *stream);

/* ---- fp16 convolution (first piece of the half-precision path, BASELINE configs 3-4) --------------------------
 * Activations are channel-blocked "c8": [B][ceil(C/8)][H][W][8] halves, channels past C zero; only the batch
 * stride (in halves, multiple of 8) is free, so a tensor may be a channel-group slice of an arena.  fp32
 * accumulation on v_mfma_f32_32x32x16_f16; bias fp32; optional LeakyReLU; output rounded to half with saturation
 * (|v| > 65504 -> +-65504, never inf), or left in fp32 with PWC_CONV_OUT_F32.
 * Same operator as pwc_conv2d_fwd (nn.Conv2d 3x3 + LeakyReLU, PWCNet.py:26-33): stride 1 with dilation 1,2,4,8,16 and
 * stride 2 with dilation 1 (PWC_EUNSUPPORTED otherwise; no residual flag). */
int64_t pwc_conv3x3_f16_packed_bytes(int Cin, int Cout);
/* w: [Cout,Cin,3,3] f32 (nn.Conv2d layout, device) -> wp: packed halves [Cg/2][tap][2][CoutP][8]. */
int pwc_conv3x3_f16_pack(const void *w, void *wp, int Cin, int Cout, void *stream);
/* Split filters: the same layout with CoutP = 32 * ceil(Cout / 16) -- every 32-row cout tile carries 16 filters rounded to half
 * (rows 0..15) and their rounding residuals times 2^11 (rows 16..31): y = sum(hi) + sum(lo)/2^11 in the epilogue of
 * pwc_conv2d_f16_fwd(PWC_CONV_SPLIT_W).  pwc_conv3x3_f16_packed_bytes_split bytes (= the plain size for Cout <= 16). */
int64_t pwc_conv3x3_f16_packed_bytes_split(int Cin, int Cout);
int pwc_conv3x3_f16_pack_split(const void *w, void *wp, int Cin, int Cout, void *stream);
int pwc_conv2d_f16_fwd(const void *x, const void *wp, const void *bias, void *y,
                       int B, int Cin, int H, int W, int Cout, int stride, int dilation,
                       unsigned flags, float leaky_slope, int64_t x_bstride, int64_t y_bstride, void *stream);
/* layout conversions at the edges of an fp16 pipeline: NCHW f32 <-> c8 f16 (batch strides in elements) */
int pwc_nchw_to_c8_f16(const void *x, void *y, int B, int C, int H, int W, int64_t x_bstride, int64_t y_bstride, void *stream);
/* the same keeping the rounding residual: y_hi = half(x), y_lo = half(x - y_hi), two c8 tensors of ceil(C/8) groups (strict
 * half-precision mode: a consumer whose filters are duplicated over both channel sets reads ~22-bit activations) */
int pwc_nchw_to_c8_f16_hilo(const void *x, void *y_hi, void *y_lo, int B, int C, int H, int W, int64_t x_bstride,
                            int64_t hi_bstride, int64_t lo_bstride, void *stream);
int pwc_c8_f16_to_nchw(const void *x, void *y, int B, int C, int H, int W, int64_t x_bstride, int64_t y_bstride, void *stream);

/* First pyramid layer (conv1a: Conv2d(3,16,3,stride 2,pad 1) + LeakyReLU, PWCNet.py:52) from a float32 NCHW image
 * x:[B,3,H,W] (batch stride free: a pair tensor [B,6,H,W] is two calls) straight to c8 halves y:[B,2,H/2,W/2,8];
 * w:[16,3,3,3] f32 (nn layout), bias:[16] f32. */
int pwc_image_conv_s2_c8_f16(const void *x, const void *w, const void *bias, void *y, int B, int H, int W,
                             float leaky_slope, int64_t x_bstride, int64_t y_bstride, void *stream);

/* conv1a -> conv1aa -> conv1b -> conv2a (PWCNet.py:52-55 as used at :184-187: the level-1 features feed conv2a only) in ONE
 * launch: x float32 [B,3,H,W] -> y c8 halves [B][4][H2][W2][8] with H1 = (H-1)/2+1, H2 = (H1-1)/2+1 (same for W).  The three
 * level-1 maps live in LDS per 8x16 output tile.  wpack: pwc_pyramid1_f16_packed_bytes() bytes of halves = conv1a as
 * [k/8][cout 16][k%8] (k = ci*9+ky*3+kx, 27 padded to 32), then conv1aa, conv1b ([20 rows = tap*2+kh, last two zero][16][8])
 * and conv2a ([20][32][8]); bias: float[80] = conv1a | conv1aa | conv1b | conv2a.  LeakyReLU(leaky_slope) after every layer. */
int64_t pwc_pyramid1_f16_packed_bytes(void);
int pwc_pyramid1_fused_f16(const void *x, const void *wpack, const void *bias, void *y, int B, int H, int W,
                           float leaky_slope, int64_t x_bstride, int64_t y_bstride, void *stream);

/* PWC-Net's cost volume (pad 4, kernel 1, max displacement 4, strides 1) on c8 f16 tensors, fp32 accumulation:
 * in1,in2: [B][ceil(C/8)][H][W][8]; out: [B][11][H][W][8] = 81 displacement channels ((dy+4)*9+(dx+4)) + 7 zeros.
 * flags: PWC_CORR_NORMALIZE (divide by C instead of multiplying by corr_multiply), PWC_ACT_LEAKY. */
int pwc_corr81_c8_f16(const void *in1, const void *in2, void *out, int B, int C, int H, int W,
                      float corr_multiply, unsigned flags, float leaky_slope,
                      int64_t in1_bstride, int64_t in2_bstride, int64_t out_bstride, void *stream);
/* PWCDCNet.warp on c8 f16 tensors.  flo: a c8 tensor whose channels flo_channel, flo_channel+1 (same group) hold
 * (u, v) -- e.g. the arena group that carries up_flow; coordinates and weights in fp32 as in pwc_warp_fwd. */
int pwc_warp_c8_f16(const void *x, const void *flo, void *out, int B, int C, int H, int W, int flo_channel,
                    float flow_scale, int align_corners, float mask_threshold,
                    int64_t x_bstride, int64_t flo_bstride, int64_t out_bstride, void *stream);
/* Entry of decoder level L < 6 on c8 f16 tensors in one pass (reference models/PWCNet.py:208-212, 222-226, 236-240,
 * 252-256: up_flow = deconv(flow), up_feat = upfeat(x), warp = self.warp(c2L, up_flow * s), then the concat).
 * The flow stays in fp32 from level to level:
 *   flow32: float [B][>=1][H/2][W/2][8] -- the level above's head convolution run with PWC_CONV_OUT_F32, flow (u,v) in
 *     channels 0,1;  deconv_w [2,2,4,4] / deconv_b [2] float: deconvL's nn.ConvTranspose2d parameters, applied here in
 *     fp32 (16 fma per output value);
 *   feat_phases: float [B][1][H/2][W/2][8] -- upfeatL computed as a 3x3 conv with 4 output phases per channel
 *     (pwc_conv2d_f16_fwd), channel index co*4 + py*2 + px;
 *   flow_group [B][1][H][W][8] halves: channels 0,1 <- up_flow, 2,3 <- up_feat (4..7 untouched) -- the arena's last group;
 *   c1_dst <- c1 (ceil(C/8) groups: the arena's c1 slot);  warped <- warp(c2, up_flow * flow_scale) with the fp32
 *     up_flow, taps and mask as pwc_warp_c8_f16.
 * H and W must be even.  Batch strides in elements of the tensor's own type. */
int pwc_level_entry_c8_f16(const void *c1, const void *c2, const void *flow32, const void *feat_phases,
                           const void *deconv_w, const void *deconv_b,
                           void *c1_dst, void *flow_group, void *warped, int B, int C, int H, int W,
                           float flow_scale, int align_corners, float mask_threshold,
                           int64_t c1_bstride, int64_t c2_bstride, int64_t flow32_bstride,
                           int64_t feat_phases_bstride, int64_t c1_dst_bstride, int64_t flow_group_bstride,
                           int64_t warped_bstride, void *stream);

/* pwc_level_entry_c8_f16 + pwc_corr81_c8_f16 in ONE launch, the warped features staying in LDS (PWCNet.py:208-214 for a level below
 * the coarsest): the flow group and the c1 slot of the arena are written as by pwc_level_entry_c8_f16, corr_out
 * [B][11][H][W][8] (81 channels + 7 zeros, LeakyReLU with PWC_ACT_LEAKY, PWC_CORR_NORMALIZE as in pwc_corr81_c8_f16) as by
 * pwc_corr81_c8_f16 on the warped tensor -- bit-identical to the two calls; no `warped` tensor is needed.  One launch less and no
 * round trip of the warped features through HBM, but every tile gathers its 16 x 40 halo (2.5x the pixels): 97 vs 81 us at level 2
 * of a batch of 16, faster only for single small launches; the plans keep the two calls unless option "f16_level_corr" is set. */
int pwc_level_corr81_c8_f16(const void *c1, const void *c2, const void *flow32, const void *feat_phases,
                            const void *deconv_w, const void *deconv_b,
                            void *c1_dst, void *flow_group, void *corr_out, int B, int C, int H, int W,
                            float flow_scale, int align_corners, float mask_threshold,
                            float corr_multiply, unsigned flags, float leaky_slope,
                            int64_t c1_bstride, int64_t c2_bstride, int64_t flow32_bstride,
                            int64_t feat_phases_bstride, int64_t c1_dst_bstride, int64_t flow_group_bstride,
                            int64_t corr_bstride, void *stream);

/* ConvTranspose2d(kernel 4, stride 2, padding 1) + bias.  x:[B,Cin,H,W], w:[Cin,Cout,4,4] (nn layout),
 * y:[B,Cout,2H,2W]. */
int pwc_deconv4x4s2_fwd(const void *x, const void *w, const void *bias, void *y,
                        int B, int Cin, int H, int W, int Cout, int dtype,
                        int64_t x_bstride, int64_t y_bstride,
                        void *stream);
/* Small levels (ABI v11): the host runs predict_flowL and upfeatL as ONE 3x3 convolution with 10 output channels -- ConvTranspose2d
 * (k4, s2, p1) is a 3x3 convolution with four output phases per channel -- through pwc_conv2d_fwd (matrix cores, split-K over the
 * chip) instead of pwc_deconv4x4s2_fwd, which is VALU-bound on the few CUs a small map gives it.  This is the level's exit:
 * head [B,10,h,w] = [flow u, v | upfeat phases co*4 + py*2 + px] -> out [B,4,2h,2w] = [deconvL(flow) | up_feat] (PWCNet.py:208-209,
 * 222-223, 236-237, 252-253); deconv_w [2,2,4,4], deconv_b [2] = deconvL's nn.ConvTranspose2d parameters.  Batch strides in elements. */
int pwc_upsample_entry_f32(const void *head, const void *deconv_w, const void *deconv_b, void *out, int B, int h, int w,
                           int64_t head_bstride, int64_t out_bstride, void *stream);

/* predict_flowL (Conv2d Cin->2, 3x3) and upfeatL (ConvTranspose2d Cin->2, k4 s2 p1) in ONE pass over x:
 * both read the same 3x3 window of the same [B,Cin,H,W] arena (PWCNet.py:207+209, 221+223, 235+237, 251+253).
 * head_wp = pwc_conv3x3_pack of the [2,Cin,3,3] head filters; up_w = [Cin,2,4,4] (nn layout).
 * flow:[B,2,H,W], up_out:[B,2,2H,2W].  Returns PWC_EUNSUPPORTED (nothing launched) when the geometry is
 * outside the streaming kernel (W % 4 != 0, W < 128, unaligned): call the two separate entry points then. */
int pwc_head_upfeat_fwd(const void *x, const void *head_wp, const void *head_bias, void *flow,
                        const void *up_w, const void *up_bias, void *up_out,
                        int B, int Cin, int H, int W, int dtype,
                        int64_t x_bstride, int64_t flow_bstride, int64_t up_bstride, void *stream);
/* The same with a scratch buffer of pwc_head_upfeat_workspace_bytes(B, Cin, H, W) bytes (0: this geometry never uses one; ABI v12):
 * launches of fewer than 256 8-row x 128-column tiles -- fewer workgroups than the chip has CUs, each VALU-bound on its own CU -- are
 * cut along Cin into slices (option "stream_slice_wgs") whose partial sums meet in the workspace and are added in fixed slice order:
 * deterministic; the fp32 summation order differs from the one-pass form.  workspace NULL / too small: the one-pass form.  With the
 * workspace the entry also takes launches of 4..63 tiles (PWC_EUNSUPPORTED without it, as pwc_head_upfeat_fwd). */
int64_t pwc_head_upfeat_workspace_bytes(int B, int Cin, int H, int W);
int pwc_head_upfeat_ws_fwd(const void *x, const void *head_wp, const void *head_bias, void *flow,
                           const void *up_w, const void *up_bias, void *up_out,
                           int B, int Cin, int H, int W, int dtype,
                           int64_t x_bstride, int64_t flow_bstride, int64_t up_bstride,
                           void *workspace, int64_t workspace_bytes, void *stream);

/* Profiling calibration, not on the product path: streams `nbytes` (a multiple of 64*width) of src through LDS with the
 * kernels' own LDS-DMA instruction (width 4: buffer_load_dword ... lds, width 16: buffer_load_dwordx4 ... lds) so that rocprofv3's
 * FETCH_SIZE can be calibrated on a known byte count in this access pattern (MI355X guide, HBM section).  sums: blocks*256 floats. */
int pwc_calib_lds_dma_read(const void *src, void *sums, int64_t nbytes, int width, int blocks, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PWC_HIP_H_ */
