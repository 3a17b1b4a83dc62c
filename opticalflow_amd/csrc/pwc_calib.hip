// Profiling calibration (not on the product path): a byte-exact streaming read through the same LDS-DMA instructions the
// kernels use, so that rocprofv3's FETCH_SIZE can be calibrated for THIS access pattern as the MI355X guide asks ("calibrate on a
// known byte count in your own access pattern before trusting an absolute"): width 4 = buffer_load_dword ... lds (the fp32
// convolution's input staging), width 16 = buffer_load_dwordx4 ... lds (correlation, fp16 convolution).
#include "pwc_common.h"

namespace {

template <int WIDTH>
__global__ void __launch_bounds__(256)
calib_dma_read_kernel(const float *__restrict__ src, float *__restrict__ sums, int64_t nbytes) {
    __shared__ __attribute__((aligned(16))) float buf[4][64 * (WIDTH / 4)];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t per_instr = 64 * WIDTH;                               // bytes per wave-instruction
    const int64_t ninstr = nbytes / per_instr;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)pwc::lds_addr(&buf[wave][0]));
    float acc = 0.f;
    // 1 GiB windows: buffer descriptors take 32-bit offsets
    for (int64_t i = (int64_t)blockIdx.x * 4 + wave; i < ninstr; i += nwaves) {
        const int64_t byte0 = i * per_instr;
        const pwc::v4i32 rs = pwc::make_rsrc(reinterpret_cast<const char *>(src) + byte0, (int)per_instr);
        if (WIDTH == 4) pwc::dma_b32(rs, base, (unsigned)lane * 4u);
        else            pwc::dma_b128(rs, base, (unsigned)lane * 16u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += buf[wave][lane * (WIDTH / 4)];
    }
    sums[(int64_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

}  // namespace

extern "C" int pwc_calib_lds_dma_read(const void *src, void *sums, int64_t nbytes, int width, int blocks, void *stream) {
    if (!src || !sums || nbytes <= 0 || blocks <= 0 || (width != 4 && width != 16) || (nbytes % (64 * width)))
        PWC_FAIL(PWC_EINVAL, "pwc_calib_lds_dma_read: nbytes must be a positive multiple of 64*width, width 4 or 16");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (width == 4) hipLaunchKernelGGL(calib_dma_read_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const float *>(src), static_cast<float *>(sums), nbytes);
    else            hipLaunchKernelGGL(calib_dma_read_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const float *>(src), static_cast<float *>(sums), nbytes);
    return pwc::check_launch("calib_dma_read_kernel");
}
