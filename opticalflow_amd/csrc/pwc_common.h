// Shared helpers for the libpwc_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include <atomic>

#include "../../include/pwc_hip.h"

namespace pwc {

void set_error(const char *fmt, ...);
void note_kernel(const char *name, int a, int b, int c, int d, int e, int f);

// run-time options (pwc_abi.hip): default from the environment variable, overridden by the C-ABI pwc_set_option()
enum Opt { OPT_CONV_WINO4, OPT_W4_TAILSPLIT, OPT_W4_SMALLSPLIT, OPT_W4_SMALL_MIN_WGS, OPT_CORR_PIPE, OPT_CORR_PIPE_MIN_TILES, OPT_CORR_ROLL, OPT_CORR_SMALL_TILES, OPT_HEAD10, OPT_F16_LEVEL_CORR, OPT_WARPCORR_WINDOW, OPT_STREAM_SLICE_WGS, OPT_C1_IN_ARENA, OPT_HEAD_SLICED_MIN_TILES, OPT_COUNT };
int option(Opt o);

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return PWC_OK;
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f32<__half>(float v) { return __float2half(v); }

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// float -> half with saturation: a value past the half range becomes +-65504 instead of inf (which the next layer's
// fp32 accumulation would turn into NaN through inf - inf).  NaN stays NaN: v_med3_f32 alone would launder a NaN operand into
// -65504 (with a quiet NaN it returns the min3 of its operands, and v_min returns the non-NaN one), so NaN is selected explicitly
// -- a NaN from bad weights or inputs must stay visible to the caller's isfinite checks.
__device__ __forceinline__ _Float16 sat_half(float v) {
    return (_Float16)(v != v ? v : __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f));
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Raise a kernel's dynamic-LDS limit once per (kernel instantiation, device).  `done` is a per-instantiation
// static array; relaxed atomics are enough (setting the attribute twice is harmless).
constexpr int kMaxDevices = 64;
struct LdsAttrOnce { std::atomic<unsigned char> done[kMaxDevices]; };
inline int ensure_lds_attr(LdsAttrOnce &state, const void *kernel, int bytes, const char *who) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { set_error("%s: hipGetDevice: %s", who, hipGetErrorString(e)); return (int)e; }
    if (dev >= 0 && dev < kMaxDevices && state.done[dev].load(std::memory_order_relaxed)) return PWC_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute(%d B LDS): %s", who, bytes, hipGetErrorString(e)); return (int)e; }
    if (dev >= 0 && dev < kMaxDevices) state.done[dev].store(1, std::memory_order_relaxed);
    return PWC_OK;
}

// ---- hand-issued LDS-DMA (buffer_load ... lds) ----------------------------------------------------------
// hipcc treats a builtin LDS-DMA as an LDS store that may alias every later ds_read and drains it with
// s_waitcnt vmcnt(0) before the first read, which serialises a multi-chunk ring.  Issued from inline asm the
// DMA is invisible to that bookkeeping; completion is then tracked ONLY by the caller's counted
// `s_waitcnt vmcnt(N)` + barrier (vmcnt retires in issue order).
typedef int v4i32 __attribute__((ext_vector_type(4)));

// raw buffer descriptor {base, num_records bytes}: built from readfirstlane'd words (wave-uniform SGPRs)
__device__ __forceinline__ v4i32 make_rsrc(const void *base, int num_bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    v4i32 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) & 0xffff;     // stride 0, no swizzle
    r.z = __builtin_amdgcn_readfirstlane(num_bytes);
    r.w = 0x00020000;
    return r;
}

// wave-uniform pointer the compiler can PROVE uniform (otherwise every buffer op is wrapped in a waterfall loop)
__device__ __forceinline__ void *uniform_ptr(const void *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<void *>(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ unsigned lds_addr(const void *p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

// 64 lanes x 16 B -> LDS[lds_base + lane*16 ..]; lanes whose voffset fails the range check deliver zeros.
// s_nop 4: SGPR operands may come straight from v_readfirstlane (VALU->VMEM SGPR hazard, not padded inside asm);
// s_nop 0: M0 write -> LDS-DMA.
__device__ __forceinline__ void dma_b128(v4i32 rsrc, unsigned lds_base, unsigned voffset) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(lds_base), "v"(voffset), "s"(rsrc) : "memory");
}
// same with the non-temporal cache policy (streamed-once bytes)
__device__ __forceinline__ void dma_b128_nt(v4i32 rsrc, unsigned lds_base, unsigned voffset) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds"
                 :: "s"(lds_base), "v"(voffset), "s"(rsrc) : "memory");
}
// 16-byte pieces with a wave-uniform extra byte offset in an SGPR (added to the address, NOT part of the range check):
// one per-lane offset register serves every instruction of a stream whose pieces are `soffset` apart
__device__ __forceinline__ void dma_b128_so(v4i32 rsrc, unsigned lds_base, unsigned voffset, unsigned soffset) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_base), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}
__device__ __forceinline__ void dma_b32(v4i32 rsrc, unsigned lds_base, unsigned voffset) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds"
                 :: "s"(lds_base), "v"(voffset), "s"(rsrc) : "memory");
}

}  // namespace pwc

#define PWC_FAIL(code, ...)            \
    do {                               \
        pwc::set_error(__VA_ARGS__);   \
        return (code);                 \
    } while (0)
