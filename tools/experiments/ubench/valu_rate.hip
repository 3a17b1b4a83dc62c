// Micro-benchmark: cycles per v_pk_fma_f32 / v_fma_f32 wave-instruction on one SIMD as a function of the waves sharing it.
// One workgroup of 64*W threads per CU (256 workgroups); every wave runs N iterations of 32 independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: v_pk_fma_f32 with op_sel broadcast, 1: v_fma_f32, 2: v_pk_fma_f32 plain pairs
__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *cyc, int iters, float x) {
    f32x2 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x2){(float)threadIdx.x, (float)i};
    f32x2 w[6];
    for (int i = 0; i < 6; ++i) w[i] = (f32x2){x + i, x - i};
    float a4[4] = {x, x * 2, x * 3, x * 4};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (MODE == 0) {
                    const f32x2 ap = {a4[p], a4[p]};
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[p * 4 + m] = __builtin_elementwise_fma(ap, w[(p + 1) / 2 + m], acc[p * 4 + m]);
                } else if (MODE == 2) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[p * 4 + m] = __builtin_elementwise_fma(w[(m + p) % 6], w[(p + 1) / 2 + m], acc[p * 4 + m]);
                } else {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        acc[p * 4 + m][0] = fmaf(a4[p], w[(p + 1) / 2 + m][0], acc[p * 4 + m][0]);
                        acc[p * 4 + m][1] = fmaf(a4[p], w[(p + 1) / 2 + m][1], acc[p * 4 + m][1]);
                    }
                }
            }
        }
        asm volatile("" : "+v"(a4[0]), "+v"(a4[1]), "+v"(a4[2]), "+v"(a4[3]));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode)
        for (int W : {1, 2, 3, 4, 8, 9, 12, 16}) {
            hipMemset(cyc, 0, 256 * 16 * 8);
            for (int r = 0; r < 2; ++r) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * W), 0, 0, out, cyc, iters, 1.0001f);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * W), 0, 0, out, cyc, iters, 1.0001f);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * W), 0, 0, out, cyc, iters, 1.0001f);
            }
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(256 * 16);
            hipMemcpy(h.data(), cyc, 256 * 16 * 8, hipMemcpyDeviceToHost);
            double mx = 0, sum = 0; int n = 0;
            for (int b = 0; b < 256; ++b) for (int w = 0; w < W; ++w) { double c = (double)h[b * 16 + w]; sum += c; ++n; if (c > mx) mx = c; }
            const double instr = (mode == 1 ? 128.0 : 64.0) * iters;     // wave-instructions per wave
            const double wps = (W + 3) / 4;                              // waves on the fullest SIMD
            printf("mode %d (%s) waves/WG %2d: mean %.2f max %.2f cycles per wave-instruction per wave; x waves on the fullest SIMD (%d) -> %.2f cycles per instruction per SIMD\n",
                   mode, mode == 0 ? "pk_fma bcast" : mode == 1 ? "v_fma" : "pk_fma pairs", W, sum / n / instr, mx / instr, (int)wps, mx / instr / wps);
        }
    return 0;
}
