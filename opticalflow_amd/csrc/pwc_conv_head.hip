// 3x3 convolution with a tiny Cout (the 2-channel flow heads predict_flow{6..2} and dc_conv7,
// reference models/PWCNet.py:32-33, used :207,221,235,251,265,268).
//
// With Cout = 2 an MFMA tile would be 94 % padding, and the fp32 VALU runs at the same FLOP rate as the
// fp32 MFMA on gfx950 -- so this is a direct VALU convolution:
//   * workgroup = 256 threads = an 8-row x 128-col output tile; thread = 4 consecutive pixels x CO couts;
//   * the input tile (+1 halo, starting 4 columns left of the tile so every 16-byte piece is aligned) is
//     streamed per 4-channel chunk into a double-buffered LDS image by buffer_load_dwordx4 ... lds; zero
//     padding and the ragged last chunk come from the buffer range check;
//   * per channel a thread reads its 3 x 6 window (one ds_read_b128 + two ds_read_b32 per row) and does
//     9 taps x 4 px x CO fma; weights are wave-uniform (scalar loads);
//   * epilogue: bias / optional LeakyReLU / optional residual (flow2 + dc_conv7(..), PWCNet.py:268),
//     16-byte stores.
// Needs W % 4 == 0 and 16-byte aligned tensors (the dispatcher falls back to the MFMA kernel otherwise).
#include "pwc_common.h"

namespace {

using pwc::leaky;

typedef __attribute__((address_space(3))) void lds_void;

constexpr int kHCK = 4;                 // channels per chunk
constexpr int kHTH = 8;                 // tile rows
constexpr int kHTW = 128;               // tile cols
constexpr int kHThreads = 256;
constexpr int kHRows = kHTH + 2;        // with halo
constexpr int kHPitch = kHTW + 8;       // floats: cols x0-4 .. x0+131
constexpr int kHQuads = kHPitch / 4;    // 34 pieces per row
constexpr int kHPieces = kHCK * kHRows * kHQuads;                 // 1360
constexpr int kHSlots = (kHPieces + kHThreads - 1) / kHThreads;   // 6
constexpr int kHBuf = kHSlots * kHThreads * 4;                    // floats per buffer (6144)
constexpr unsigned kHOOB = 0x80000000u;

__device__ __forceinline__ void *uniform_ptr(const void *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<void *>(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ void head_issue(const float *xb, int chunk, int Cin, int plane, int wave, float *buf,
                                           const unsigned (&off)[kHSlots]) {
    const int c0 = chunk * kHCK;
    const int cvalid = min(kHCK, Cin - c0);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(xb + (int64_t)c0 * plane), 0, __builtin_amdgcn_readfirstlane(cvalid * plane * 4), 0x00020000);
    float *dst = buf + wave * 256;
#pragma unroll
    for (int j = 0; j < kHSlots; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)(dst + j * kHThreads * 4), 16, off[j], 0, 0, 0);
}

template <int CO>
__global__ void __launch_bounds__(kHThreads)
conv3x3_head_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                    const float *__restrict__ residual, float *__restrict__ y,
                    int Cin, int H, int W, int tiles_x, int tiles_y,
                    int64_t bsx, int64_t bsy, int64_t bsr, float slope, int do_leaky) {
    __shared__ __attribute__((aligned(16))) float smem[2 * kHBuf];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty = tid >> 5;            // 0..7  output row inside the tile
    const int tx = tid & 31;            // 0..31 group of 4 pixels

    int bid = blockIdx.x;
    const int bx = bid % tiles_x;
    bid /= tiles_x;
    const int by = bid % tiles_y;
    const int b = bid / tiles_y;
    const int x0 = bx * kHTW;
    const int y0 = by * kHTH;
    const int plane = H * W;

    unsigned off[kHSlots];
#pragma unroll
    for (int j = 0; j < kHSlots; ++j) {
        const int p = j * kHThreads + tid;
        const int c = p / (kHRows * kHQuads);
        const int rem = p % (kHRows * kHQuads);
        const int r = rem / kHQuads;
        const int q = rem % kHQuads;
        const int iy = y0 - 1 + r;
        const int ix = x0 - 4 + 4 * q;
        const bool ok = (p < kHPieces) && (iy >= 0) && (iy < H) && (ix >= 0) && (ix < W);   // W % 4 == 0: all-in or all-out
        off[j] = ok ? (unsigned)(c * plane + iy * W + ix) * 4u : kHOOB;
    }

    float acc[CO][4];
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const float bv = bias[co];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[co][p] = bv;
    }

    const float *xb = x + (int64_t)b * bsx;
    const int nchunks = (Cin + kHCK - 1) / kHCK;
    head_issue(xb, 0, Cin, plane, wave, smem, off);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const float *cur = smem + (chunk & 1) * kHBuf;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (chunk + 1 < nchunks) head_issue(xb, chunk + 1, Cin, plane, wave, smem + ((chunk + 1) & 1) * kHBuf, off);
        const int c0 = chunk * kHCK;
        const int cvalid = min(kHCK, Cin - c0);
#pragma unroll
        for (int c = 0; c < kHCK; ++c) {
            if (c < cvalid) {
                const float *t = cur + (c * kHRows + ty) * kHPitch + 4 * tx + 3;     // window col -1
                float v[3][6];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float *row = t + r * kHPitch;
                    const float4 m = *reinterpret_cast<const float4 *>(row + 1);
                    v[r][0] = row[0];
                    v[r][1] = m.x; v[r][2] = m.y; v[r][3] = m.z; v[r][4] = m.w;
                    v[r][5] = row[5];
                }
#pragma unroll
                for (int co = 0; co < CO; ++co) {
                    const float *wk = w + ((int64_t)co * Cin + (c0 + c)) * 9;          // wave-uniform
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float wv = wk[ky * 3 + kx];
#pragma unroll
                            for (int p = 0; p < 4; ++p) acc[co][p] = fmaf(v[ky][p + kx], wv, acc[co][p]);
                        }
                }
            }
        }
    }

    const int oy = y0 + ty;
    const int ox = x0 + 4 * tx;
    if (oy >= H || ox >= W) return;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        const int64_t o = (int64_t)co * plane + (int64_t)oy * W + ox;
        float4 v = make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);
        if (do_leaky) { v.x = leaky(v.x, slope); v.y = leaky(v.y, slope); v.z = leaky(v.z, slope); v.w = leaky(v.w, slope); }
        if (residual) {
            const float4 rr = *reinterpret_cast<const float4 *>(residual + (int64_t)b * bsr + o);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        *reinterpret_cast<float4 *>(y + (int64_t)b * bsy + o) = v;
    }
}

}  // namespace

namespace pwc_conv {

// Returns PWC_EUNSUPPORTED when the fast path's preconditions do not hold (caller falls back).
int run_head(const float *x, const float *w_raw, const float *bias, const float *residual, float *y,
             int B, int Cin, int H, int W, int Cout, int64_t bsx, int64_t bsy, int64_t bsr,
             float slope, int do_leaky, hipStream_t st) {
    if (Cout != 2 || (W & 3)) return PWC_EUNSUPPORTED;
    if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15u) ||
        (bsx & 3) || (bsy & 3) || (bsr & 3))
        return PWC_EUNSUPPORTED;
    const int tiles_x = (W + kHTW - 1) / kHTW;
    const int tiles_y = (H + kHTH - 1) / kHTH;
    const int64_t nblk = (int64_t)B * tiles_x * tiles_y;
    if (nblk > 0x7fffffffLL) PWC_FAIL(PWC_EINVAL, "pwc_conv2d_fwd: grid too large");
    hipLaunchKernelGGL((conv3x3_head_kernel<2>), dim3((unsigned)nblk), dim3(kHThreads), 0, st,
                       x, w_raw, bias, residual, y, Cin, H, W, tiles_x, tiles_y, bsx, bsy, bsr, slope, do_leaky);
    return pwc::check_launch("conv3x3_head_kernel");
}

}  // namespace pwc_conv
