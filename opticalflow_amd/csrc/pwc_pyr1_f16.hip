// First pyramid level + the stride-2 entry of level 2 in ONE kernel (half-precision plan):
//     image (float32 NCHW) -> conv1a (3->16, s2) -> conv1aa (16->16) -> conv1b (16->16) -> conv2a (16->32, s2) -> c8 halves
// = reference models/PWCNet.py:52-55,184-187 (the level-1 features c11 / c21 are consumed by conv2a only, PWCNet.py:186-187,
// so they never need to exist in HBM).  As four launches these layers moved 0.94 GB per batch-16 step and took 330 us of a
// 3.9 ms forward at 2-3 TB/s (K = 16 channels: one or two LDS-DMA round trips per workgroup, nothing to pipeline); fused, the
// step reads the image once (176 MB) and writes the level-2 map (59 MB).
//
// One workgroup (8 waves) owns an 8 x 16 tile of the level-2 output and recomputes the halos of the three level-1 maps it
// needs (21x37 -> 19x35 -> 17x33 pixels: 1.2-1.5x redundant work, all of it on chip).  Every map is a c8 image in LDS
// ([kh 2][rows][cols] 16-byte pieces), every layer runs on v_mfma_f32_16x16x32_f16 with 16 couts on the rows and 16 pixels
// on the columns; a lane quarter kq = lane >> 4 carries k-group (tap, kh) = (2m + (kq >> 1), kq & 1) of MFMA step m = 0..4
// (row 4m + kq of the [tap][kh][cout] filter image; rows 18, 19 are zero: the ninth tap has no partner).  conv1a's K = 27 is
// padded to 32 and its operand is gathered from the image patch (kept as halves in LDS).  Values of a map outside the image
// are stored as ZERO: they are the next convolution's zero padding, not convolution results.
#include <stdlib.h>

#include "pwc_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kT2H = 8, kT2W = 16;                               // level-2 output tile
constexpr int kA3H = 2 * kT2H + 1, kA3W = 2 * kT2W + 1;          // conv1b outputs needed: 17 x 33
constexpr int kA2H = kA3H + 2, kA2W = kA3W + 2;                  // conv1aa: 19 x 35
constexpr int kA1H = kA2H + 2, kA1W = kA2W + 2;                  // conv1a: 21 x 37
constexpr int kImH = 2 * kA1H + 1, kImW = 2 * kA1W + 1;          // image patch: 43 x 75
constexpr int kImP = kImW + 1;                                    // row pitch in halves
constexpr int kBytesA1 = 2 * kA1H * kA1W * 16;                   // 24864  (later reused by A3: 17952)
constexpr int kBytesA2 = 2 * kA2H * kA2W * 16;                   // 21280  (earlier: the image patch as halves, 19608)
constexpr int kWPieces1a = 4 * 16, kWPieces16 = 20 * 16, kWPieces32 = 20 * 32;
constexpr int kWPieces = kWPieces1a + 2 * kWPieces16 + kWPieces32;   // 1344 16-byte pieces = 21504 B
constexpr int kOffA1 = 0, kOffA2 = kBytesA1, kOffW = kOffA2 + kBytesA2, kOffB = kOffW + kWPieces * 16;
constexpr int kSmem = kOffB + 80 * 4;                            // 67968 B: two workgroups per CU
static_assert(3 * kImH * kImP * 2 <= kBytesA2 && 2 * kA3H * kA3W * 16 <= kBytesA1, "aliased LDS regions");
static_assert(kBytesA1 % 16 == 0 && kBytesA2 % 16 == 0, "16-byte aligned regions");

__device__ __forceinline__ f32x4 mfma16(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// bias + LeakyReLU (0 <= slope <= 1: max(v, slope*v)) + saturating conversion of one lane's four couts, zeroed outside the map
__device__ __forceinline__ h4 finish4(f32x4 acc, const float (&bv)[4], float slope, bool inside) {
    h4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float v = acc[i] + bv[i];
        o[i] = pwc::sat_half(fmaxf(v, v * slope));
    }
    uint2 bits = *reinterpret_cast<uint2 *>(&o);
    const unsigned m = inside ? 0xffffffffu : 0u;
    bits.x &= m;
    bits.y &= m;
    return *reinterpret_cast<h4 *>(&bits);
}

// One stride-1 3x3 layer (16 -> 16) from LDS map `in` [2][IH][IW] to LDS map `out` [2][OH][OW] (OH = IH - 2, OW = IW - 2):
// out(r, c) = act(bias + sum_taps w * in(r + ky, c + kx)), zero where (gy0 + r, gx0 + c) lies outside the Hm x Wm map.
// A wave takes two 16-pixel tiles per iteration (shared filter fragments, two independent MFMA chains).
template <int IH, int IW>
__device__ __forceinline__ void layer_s1(const h8 *in, h8 *out, const h8 *w, const float *bias, int wave, int p, int kq,
                                         int gy0, int gx0, int Hm, int Wm, float slope) {
    constexpr int OH = IH - 2, OW = IW - 2, NPIX = OH * OW, NT = (NPIX + 15) / 16;
    const int odd = kq >> 1;
    float bv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = bias[4 * kq + i];
    const h8 *wl = w + kq * 16 + p;
    for (int t = wave; t < NT; t += 16) {
        int q[2], r[2], c[2];
        const h8 *inl[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            q[u] = 16 * (t + 8 * u) + p;
            const int qq = min(q[u], NPIX - 1);
            r[u] = qq / OW;
            c[u] = qq - r[u] * OW;
            inl[u] = in + ((kq & 1) * IH + r[u]) * IW + c[u];
        }
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            const int t0 = 2 * m, t1 = (2 * m + 1 > 8) ? 8 : 2 * m + 1;      // constants after unrolling
            const int o0 = (t0 / 3) * IW + t0 % 3, o1 = (t1 / 3) * IW + t1 % 3;
            const int o = odd ? o1 : o0;
            const h8 a = wl[4 * m * 16];
            acc[0] = mfma16(a, inl[0][o], acc[0]);
            acc[1] = mfma16(a, inl[1][o], acc[1]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bool inside = (unsigned)(gy0 + r[u]) < (unsigned)Hm && (unsigned)(gx0 + c[u]) < (unsigned)Wm;
            const h4 o = finish4(acc[u], bv, slope, inside);
            if (q[u] < NPIX)
                *reinterpret_cast<h4 *>(reinterpret_cast<_Float16 *>(out + (odd * OH + r[u]) * OW + c[u]) + (kq & 1) * 4) = o;
        }
    }
}

__global__ void __launch_bounds__(512, 4)
pyr1_fused_kernel(const float *__restrict__ img, const uint4 *__restrict__ wpack, const float *__restrict__ bias,
                  _Float16 *__restrict__ y, int H, int W, int H1, int W1, int H2, int W2, int tiles_x, int tiles_y,
                  int64_t bs_img, int64_t bsy, float slope, int stop) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    h8 *A1 = reinterpret_cast<h8 *>(smem + kOffA1);
    h8 *A2 = reinterpret_cast<h8 *>(smem + kOffA2);
    h8 *A3 = A1;                                                        // A1 is dead once conv1aa has run
    _Float16 *simg = reinterpret_cast<_Float16 *>(smem + kOffA2);       // the image patch is dead once conv1a has run
    const h8 *w1a = reinterpret_cast<const h8 *>(smem + kOffW);
    const h8 *w1aa = w1a + kWPieces1a, *w1b = w1aa + kWPieces16, *w2a = w1b + kWPieces16;
    float *sbias = reinterpret_cast<float *>(smem + kOffB);             // [1a 16 | 1aa 16 | 1b 16 | 2a 32]

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int p = lane & 15, kq = lane >> 4;
    int bid = blockIdx.x;
    if ((gridDim.x & 7u) == 0) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);      // XCD-contiguous tile runs (halo reuse in L2)
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int Y0 = ty * kT2H, X0 = tx * kT2W;

    // ---- stage 0: filters, biases and the image patch (as halves) into LDS --------------------------------------
    {
        // every global load of the stage is issued before the first one is used (a load-use-store loop would expose one
        // memory latency per iteration: 19 x ~2 us per workgroup, measured)
        constexpr int kNW = (kWPieces + 511) / 512, kNI = (3 * kImH * kImW + 511) / 512;
        uint4 wv[kNW];
#pragma unroll
        for (int k = 0; k < kNW; ++k) {
            const int i = tid + 512 * k;
            wv[k] = wpack[min(i, kWPieces - 1)];
        }
        const float bval = bias[min(tid, 79)];
        const float *ib = img + (int64_t)b * bs_img;
        const int iy0 = 4 * Y0 - 7, ix0 = 4 * X0 - 7;
        float iv[kNI];
        int dst[kNI];
#pragma unroll
        for (int k = 0; k < kNI; ++k) {
            const int i = tid + 512 * k;
            const int ci = i / (kImH * kImW);
            const int rem = i - ci * (kImH * kImW);
            const int r = rem / kImW, c = rem - r * kImW;
            const int gy = iy0 + r, gx = ix0 + c;
            const bool ok = (i < 3 * kImH * kImW) && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            // unconditional load from a clamped address + select (no divergent branch around the load)
            const float ld = ib[((int64_t)min(ci, 2) * H + min(max(gy, 0), H - 1)) * W + min(max(gx, 0), W - 1)];
            iv[k] = ok ? ld : 0.f;
            dst[k] = (i < 3 * kImH * kImW) ? (ci * kImH + r) * kImP + c : -1;
        }
#pragma unroll
        for (int k = 0; k < kNW; ++k)
            if (tid + 512 * k < kWPieces) reinterpret_cast<uint4 *>(smem + kOffW)[tid + 512 * k] = wv[k];
        if (tid < 80) sbias[tid] = bval;
#pragma unroll
        for (int k = 0; k < kNI; ++k)
            if (dst[k] >= 0) simg[dst[k]] = (_Float16)iv[k];
    }
    __syncthreads();
    if (stop == 0) return;

    // ---- stage 1: conv1a (stride 2) on the 21 x 37 region: one MFMA per 16 pixels, operand gathered from the patch -----
    {
        int koff[8];                                   // patch offset of k = 8*kq + j = (ci, ky, kx); k >= 27 meets zero filters
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * kq + j;
            koff[j] = (k < 27) ? ((k / 9) * kImH + (k % 9) / 3) * kImP + k % 3 : 0;
        }
        const h8 a = w1a[kq * 16 + p];
        float bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = sbias[4 * kq + i];
        constexpr int NPIX = kA1H * kA1W, NT = (NPIX + 15) / 16;
        for (int t = wave; t < NT; t += 8) {
            const int q = 16 * t + p;
            const int qq = min(q, NPIX - 1);
            const int r = qq / kA1W, c = qq - r * kA1W;
            const _Float16 *src = simg + (2 * r) * kImP + 2 * c;
            h8 bf;
#pragma unroll
            for (int j = 0; j < 8; ++j) bf[j] = src[koff[j]];
            const f32x4 acc = mfma16(a, bf, f32x4{0.f, 0.f, 0.f, 0.f});
            const bool inside = (unsigned)(2 * Y0 - 3 + r) < (unsigned)H1 && (unsigned)(2 * X0 - 3 + c) < (unsigned)W1;
            const h4 o = finish4(acc, bv, slope, inside);
            if (q < NPIX) *reinterpret_cast<h4 *>(reinterpret_cast<_Float16 *>(A1 + ((kq >> 1) * kA1H + r) * kA1W + c) + (kq & 1) * 4) = o;
        }
    }
    __syncthreads();
    if (stop == 1) return;
    // ---- stage 2: conv1aa 21x37 -> 19x35 (overwrites the image patch), stage 3: conv1b 19x35 -> 17x33 (overwrites A1) ----
    layer_s1<kA1H, kA1W>(A1, A2, w1aa, sbias + 16, wave, p, kq, 2 * Y0 - 2, 2 * X0 - 2, H1, W1, slope);
    __syncthreads();
    if (stop == 2) return;
    layer_s1<kA2H, kA2W>(A2, A3, w1b, sbias + 32, wave, p, kq, 2 * Y0 - 1, 2 * X0 - 1, H1, W1, slope);
    __syncthreads();
    if (stop == 3) return;
    // ---- stage 4: conv2a (stride 2, 32 couts): wave = tile row, 16 pixels x 2 cout tiles, straight to the c8 output -------
    {
        const int odd = kq >> 1;
        const h8 *inl = A3 + ((kq & 1) * kA3H + 2 * wave) * kA3W + 2 * p;
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            const int t0 = 2 * m, t1 = (2 * m + 1 > 8) ? 8 : 2 * m + 1;
            const int o0 = (t0 / 3) * kA3W + t0 % 3, o1 = (t1 / 3) * kA3W + t1 % 3;
            const h8 bf = inl[odd ? o1 : o0];
            acc[0] = mfma16(w2a[(4 * m + kq) * 32 + p], bf, acc[0]);
            acc[1] = mfma16(w2a[(4 * m + kq) * 32 + 16 + p], bf, acc[1]);
        }
        const int gy = Y0 + wave, gx = X0 + p;
        if (gy < H2 && gx < W2) {
            const int64_t oplane = (int64_t)H2 * W2;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                float bv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[i] = sbias[48 + ct * 16 + 4 * kq + i];
                const h4 o = finish4(acc[ct], bv, slope, true);
                const int cg = 2 * ct + odd;
                *reinterpret_cast<h4 *>(y + (int64_t)b * bsy + ((int64_t)cg * oplane + (int64_t)gy * W2 + gx) * 8 + (kq & 1) * 4) = o;
            }
        }
    }
}

}  // namespace

extern "C" int64_t pwc_pyramid1_f16_packed_bytes(void) { return (int64_t)kWPieces * 16; }

extern "C" int pwc_pyramid1_fused_f16(const void *img, const void *wpack, const void *bias, void *y, int B, int H, int W,
                                      float leaky_slope, int64_t img_bstride, int64_t y_bstride, void *stream) {
    if (!img || !wpack || !bias || !y || B <= 0 || H <= 0 || W <= 0) PWC_FAIL(PWC_EINVAL, "pwc_pyramid1_fused_f16: bad argument");
    if (!pwc::aligned16(wpack) || !pwc::aligned16(y) || (y_bstride % 8) || (reinterpret_cast<uintptr_t>(img) & 3u))
        PWC_FAIL(PWC_EALIGN, "pwc_pyramid1_fused_f16: packed filters / output must be 16-byte aligned, the image 4-byte aligned");
    const int H1 = (H - 1) / 2 + 1, W1 = (W - 1) / 2 + 1, H2 = (H1 - 1) / 2 + 1, W2 = (W1 - 1) / 2 + 1;
    if (!(leaky_slope >= 0.f && leaky_slope <= 1.f)) PWC_FAIL(PWC_EINVAL, "pwc_pyramid1_fused_f16: leaky_slope must be in [0, 1]");
    if (img_bstride < (int64_t)3 * H * W || y_bstride < (int64_t)4 * H2 * W2 * 8)
        PWC_FAIL(PWC_EINVAL, "pwc_pyramid1_fused_f16: batch stride smaller than the tensor");
    const int tiles_x = (W2 + kT2W - 1) / kT2W, tiles_y = (H2 + kT2H - 1) / kT2H;
    const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_pyramid1_fused_f16: grid too large");
    static pwc::LdsAttrOnce attr;
    if (const int rc = pwc::ensure_lds_attr(attr, reinterpret_cast<const void *>(pyr1_fused_kernel), kSmem, "pwc_pyramid1_fused_f16")) return rc;
    hipLaunchKernelGGL(pyr1_fused_kernel, dim3((unsigned)nblk), dim3(512), kSmem, static_cast<hipStream_t>(stream),
                       static_cast<const float *>(img), static_cast<const uint4 *>(wpack), static_cast<const float *>(bias),
                       static_cast<_Float16 *>(y), H, W, H1, W1, H2, W2, tiles_x, tiles_y, img_bstride, y_bstride, leaky_slope,
                       getenv("PWC_PYR1_STOP") ? atoi(getenv("PWC_PYR1_STOP")) : -1);
    return pwc::check_launch("pyr1_fused_kernel");
}
