// Does `buffer_load_dwordx4 ... lds` accept an LDS base (M0) that is only 4-byte aligned?  (Wanted: a 16-byte LDS-DMA of the
// Winograd kernel's raw input rows landing one float to the right, so that the 8-byte reads of the transform stay aligned.)
// build: hipcc --offload-arch=gfx950 -O2 -I opticalflow_amd/csrc tools/experiments/dma_align_test.hip -o tools/experiments/dma_align_test
#include <stdio.h>
#include <vector>
#include "pwc_common.h"

__global__ void k(const float *src, float *dst, int shift_bytes) {
    __shared__ __attribute__((aligned(16))) float buf[64 * 4 + 8];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 4 + 8; i += 64) buf[i] = -1.f;
    __syncthreads();
    const pwc::v4i32 rs = pwc::make_rsrc(src, 64 * 16);
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)(pwc::lds_addr(buf) + shift_bytes));
    pwc::dma_b128(rs, base, (unsigned)lane * 16u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 64 * 4 + 8; i += 64) dst[i] = buf[i];
}

int main() {
    const int n = 64 * 4;
    std::vector<float> h(n), out(n + 8);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *src, *dst;
    hipMalloc(&src, n * 4);
    hipMalloc(&dst, (n + 8) * 4);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int shift = 0; shift <= 12; shift += 4) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, dst, shift);
        hipMemcpy(out.data(), dst, (n + 8) * 4, hipMemcpyDeviceToHost);
        int bad = 0, first = -1;
        for (int i = 0; i < n; ++i)
            if (out[i + shift / 4] != h[i]) { if (first < 0) first = i; ++bad; }
        printf("shift %2d bytes: %d mismatches of %d (first at %d: got %g), guard before %g after %g\n", shift, bad, n, first,
               first >= 0 ? out[first + shift / 4] : 0.f, shift ? out[shift / 4 - 1] : -1.f, out[n + shift / 4]);
    }
    return 0;
}
