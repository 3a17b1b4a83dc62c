// Micro-benchmark: practical HBM ceiling for the correlation's read/write mix -- a grid-stride float4 kernel that reads R MB and
// writes Wr MB (cold: three buffer sets in rotation), plus the 1:1 copy for reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void __launch_bounds__(256) mix(const float4 *__restrict__ a, float4 *__restrict__ o, long nq_out, long num, long den, int nt) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < nq_out; i += (long)gridDim.x * 256L) {
        float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
        const long j = i * num / den;
        if (((i + 1) * num / den) != j) v = a[j];           // each input quad is read by exactly one output quad
        if (nt) __builtin_nontemporal_store(v.x, &o[i].x), __builtin_nontemporal_store(v.y, &o[i].y), __builtin_nontemporal_store(v.z, &o[i].z), __builtin_nontemporal_store(v.w, &o[i].w);
        else o[i] = v;
    }
}

int main() {
    const long MB = 1 << 20;
    const long rbytes = 116 * MB, wbytes = 150 * MB;
    float4 *a[3], *o[3];
    for (int k = 0; k < 3; ++k) { hipMalloc(&a[k], 160 * MB); hipMalloc(&o[k], 160 * MB); hipMemset(a[k], 1, 160 * MB); hipMemset(o[k], 0, 160 * MB); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Case { const char *name; long r, w; } cases[] = {{"corr mix 116 R / 150 W", rbytes, wbytes}, {"copy 150 R / 150 W", wbytes, wbytes}, {"58 R / 150 W", 58 * MB, wbytes},
                                                            {"write only 0 R / 150 W", 0, wbytes}, {"read-heavy 140 R / 20 W", 20 * MB * 7, 20 * MB}};
    for (auto &c : cases)
        for (int nt = 0; nt < 2; ++nt)
        for (int grid : {1024, 4096, 16384}) {
            const long nq = c.w / 16;
            for (int it = 0; it < 6; ++it) hipLaunchKernelGGL(mix, dim3(grid), dim3(256), 0, 0, a[it % 3], o[it % 3], nq, c.r / 16, nq, nt);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const int reps = 30;
            for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(mix, dim3(grid), dim3(256), 0, 0, a[it % 3], o[it % 3], nq, c.r / 16, nq, nt);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps;
            printf("%-26s %s grid %5d: %.1f us  = %.2f TB/s total\n", c.name, nt ? "nt-stores" : "plain    ", grid, us, (c.r + c.w) / us / 1e6);
        }
    return 0;
}
