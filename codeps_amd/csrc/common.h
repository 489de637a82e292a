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
// relu((v - mean) invstd gamma + beta): the ONE expression of the train-mode BatchNorm apply (bn_apply_mg_kernel) and of the kernels that
// normalise on their operand load instead (conv_igemm_kernel / wgrad_1x1s_kernel with in_aff) -- the same operations in the same order, so
// a product over the normalised-on-load operand equals the product over the materialised tensor bit for bit
typedef float codeps_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ codeps_f32x4 bn_norm_relu(const codeps_f32x4 v, const codeps_f32x4 mu, const codeps_f32x4 is, const codeps_f32x4 ga,
                                                     const codeps_f32x4 be) {
    codeps_f32x4 r = (v - mu) * is * ga + be;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.f ? r[e] : 0.f;
    return r;
}

// ---- in-kernel clock stamps: ONLY in the diagnostic variant build (tools/build_variant.sh stamps -DCODEPS_CLOCK_STAMPS; in the
// product library the macros are empty and no stamp executes).  MI355X_MICROARCH.md "DVFS give-back" item 6: s_memtime counts shader
// cycles, s_memrealtime a constant 100 MHz; wave 0 of every block writes (cycles, 100 MHz ticks, phase A cycles, phase B cycles) of its
// lifetime into a buffer of its own that no kernel reads (set per translation unit by codeps_stamp_buffer_*).
#ifdef CODEPS_CLOCK_STAMPS
#define CODEPS_STAMP_DECL(tag)                                                                                              \
    __device__ unsigned long long* codeps_stamp_buf_##tag = nullptr;                                                        \
    __device__ unsigned codeps_stamp_cap_##tag = 0;
__device__ __forceinline__ unsigned long long codeps_memtime() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__device__ __forceinline__ unsigned long long codeps_memrealtime() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define CODEPS_STAMP_BEGIN() unsigned long long st_pa_ = 0, st_pb_ = 0, st_ps_ = 0; (void)st_ps_; \
    const unsigned long long st_r0_ = codeps_memrealtime(), st_t0_ = codeps_memtime()
#define CODEPS_STAMP_PHASE_START() st_ps_ = codeps_memtime()
#define CODEPS_STAMP_PHASE_A() st_pa_ += codeps_memtime() - st_ps_
#define CODEPS_STAMP_PHASE_B() st_pb_ += codeps_memtime() - st_ps_
#define CODEPS_STAMP_END(tag)                                                                                               \
    do {                                                                                                                    \
        const unsigned long long st_t1_ = codeps_memtime(), st_r1_ = codeps_memrealtime();                                  \
        const unsigned st_slot_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                           \
        if (threadIdx.x == 0 && codeps_stamp_buf_##tag && st_slot_ < codeps_stamp_cap_##tag) {                              \
            unsigned long long* st_d_ = codeps_stamp_buf_##tag + (size_t)st_slot_ * 4;                                      \
            st_d_[0] = st_t1_ - st_t0_; st_d_[1] = st_r1_ - st_r0_; st_d_[2] = st_pa_; st_d_[3] = st_pb_;                   \
        }                                                                                                                   \
    } while (0)
// one record per WAVE (slot = 4 * block + wave): which wave of a block is the late one at its barriers
#define CODEPS_STAMP_END_WAVES(tag)                                                                                         \
    do {                                                                                                                    \
        const unsigned long long st_t1_ = codeps_memtime(), st_r1_ = codeps_memrealtime();                                  \
        const unsigned st_slot_ = 4 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + (threadIdx.x >> 6);  \
        if ((threadIdx.x & 63) == 0 && codeps_stamp_buf_##tag && st_slot_ < codeps_stamp_cap_##tag) {                       \
            unsigned long long* st_d_ = codeps_stamp_buf_##tag + (size_t)st_slot_ * 4;                                      \
            st_d_[0] = st_t1_ - st_t0_; st_d_[1] = st_r1_ - st_r0_; st_d_[2] = st_pa_; st_d_[3] = st_pb_;                   \
        }                                                                                                                   \
    } while (0)
#define CODEPS_STAMP_SETTER(tag)                                                                                            \
    extern "C" int codeps_stamp_buffer_##tag(unsigned long long* buf, unsigned cap) {                                       \
        if (hipMemcpyToSymbol(HIP_SYMBOL(codeps_stamp_buf_##tag), &buf, sizeof(buf)) != hipSuccess) return -1;              \
        return hipMemcpyToSymbol(HIP_SYMBOL(codeps_stamp_cap_##tag), &cap, sizeof(cap)) == hipSuccess ? 0 : -1;             \
    }
#else
#define CODEPS_STAMP_DECL(tag)
#define CODEPS_STAMP_BEGIN() do {} while (0)
#define CODEPS_STAMP_PHASE_START() do {} while (0)
#define CODEPS_STAMP_PHASE_A() do {} while (0)
#define CODEPS_STAMP_PHASE_B() do {} while (0)
#define CODEPS_STAMP_END(tag) do {} while (0)
#define CODEPS_STAMP_END_WAVES(tag) do {} while (0)
#define CODEPS_STAMP_SETTER(tag)
#endif

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
