// Shared device/host helpers for libcodeps_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/codeps_hip.h"

#define CODEPS_WAVE 64

// ---- error plumbing: entry points never throw; they return a negative code and stash a message.
void codeps_set_error(const char* fmt, ...);

#define CODEPS_REQUIRE(cond, code, ...)          \
    do {                                         \
        if (!(cond)) {                           \
            codeps_set_error(__VA_ARGS__);       \
            return (code);                       \
        }                                        \
    } while (0)

// Raise a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize) once per (kernel, device); thread-safe.
// A failure sets the error string and is remembered for the calling thread: the next CODEPS_CHECK_LAUNCH returns it instead
// of the generic launch error the kernel would produce.
void codeps_allow_lds(const void* kernel, int bytes, const char* name);
bool codeps_take_attr_failure(void);

#define CODEPS_CHECK_LAUNCH(name)                                                      \
    do {                                                                               \
        if (codeps_take_attr_failure()) { (void)hipGetLastError(); return CODEPS_ERR_LAUNCH; } \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            codeps_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
            return CODEPS_ERR_LAUNCH;                                                  \
        }                                                                              \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#ifdef __HIPCC__
// ---- wave / block reductions (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
// Sum over a block of NT threads (NT multiple of 64, <= 1024). `red` = NT/64 floats of LDS.
// Result valid in every thread.  Deterministic order.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}
#endif
