// Measured peaks of the box the library runs on ("re-measure on the box with a stream-triad and an MFMA
// micro-kernel and report both"): the two rooflines bench.py prices the step against, as THIS chip delivers them.
//   * codeps_diag_mfma_rate   -- register-resident MFMA loop, no memory traffic: the matrix pipe's issue rate under the clock
//                                the power limit allows.  mode 0: v_mfma_f32_32x32x16_bf16, mode 1: v_mfma_f32_32x32x2_f32,
//                                mode 2: v_mfma_f32_16x16x32_bf16.  data 0: all-zero operands (what vendor peaks are
//                                quoted on), data 1: a pool of pseudo-random operand fragments rotated between
//                                consecutive MFMAs (operand bits toggle as they do in a real contraction).
//   * codeps_diag_stream_triad -- a[i] = b[i] + s c[i] on float4, grid-stride: HBM read + write bandwidth.
// Both time themselves with HIP events on the given stream and return milliseconds; nothing here is on the hot path.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// a value in [-1, 1) with a full 8-bit bf16 significand
__device__ __forceinline__ float unit(unsigned h) { return (float)(int)(h & 0xFFFFu) * (1.f / 32768.f) - 1.f; }

template <int MODE>
__global__ __launch_bounds__(256) void mfma_rate_kernel(int iters, int random, unsigned seed, float* __restrict__ out) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    constexpr int POOL = 4;
    f32x16 acc32[4];
    f32x4 acc16[8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[j][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc16[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[POOL], b[POOL];
    float fa[POOL], fb[POOL];
#pragma unroll
    for (int q = 0; q < POOL; ++q) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[q][e] = (__bf16)(random ? unit(mix(seed + tid * 64 + q * 16 + e)) : 0.f);
            b[q][e] = (__bf16)(random ? unit(mix(seed * 3 + tid * 64 + q * 16 + e + 8)) : 0.f);
        }
        fa[q] = random ? unit(mix(seed + tid * 8 + q)) : 0.f;
        fb[q] = random ? unit(mix(seed * 5 + tid * 8 + q + 4)) : 0.f;
    }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 16; ++u)         // 16 MFMAs: 4 accumulators x 4 operand pairs, consecutive MFMAs change both operands
                acc32[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + (u >> 2)) & 3], b[(u >> 2) & 3], acc32[u & 3], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                acc32[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[(u + (u >> 2)) & 3], fb[(u >> 2) & 3], acc32[u & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                acc16[u & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u + (u >> 2)) & 3], b[(u >> 2) & 3], acc16[u & 7], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc32[j][r];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc16[j][0] + acc16[j][1] + acc16[j][2] + acc16[j][3];
    out[tid] = s;
}

__global__ __launch_bounds__(256) void triad_kernel(const f32x4* __restrict__ b, const f32x4* __restrict__ c, float s, long long n4,
                                                    f32x4* __restrict__ a) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) a[i] = b[i] + s * c[i];
}

template <typename F>
int timed(hipStream_t st, int reps, float* ms, F&& launch) {
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return CODEPS_ERR_LAUNCH;
    launch();                                                    // warm-up (code object load, clocks)
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1, st);
    hipError_t e = hipEventSynchronize(e1);
    float t = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) { codeps_set_error("diagnostic kernel failed: %s", hipGetErrorString(e)); return CODEPS_ERR_LAUNCH; }
    *ms = t / (float)reps;
    return CODEPS_OK;
}

}  // namespace

extern "C" int codeps_diag_mfma_rate(int mode, int random_data, int blocks, int iters, int reps, float* scratch, double* flops_per_launch,
                                     float* ms_per_launch, codeps_stream_t stream) {
    CODEPS_REQUIRE(mode >= 0 && mode <= 2 && blocks > 0 && iters > 0 && reps > 0 && scratch && flops_per_launch && ms_per_launch,
                   CODEPS_ERR_ARG, "codeps_diag_mfma_rate: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double per_mfma = mode == 0 ? 2.0 * 32 * 32 * 16 : (mode == 1 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 32);
    *flops_per_launch = per_mfma * 16.0 * iters * 4.0 * blocks;            // 16 MFMAs per iteration, 4 waves per block
    int rc = timed(st, reps, ms_per_launch, [&]() {
        if (mode == 0) hipLaunchKernelGGL(mfma_rate_kernel<0>, dim3(blocks), dim3(256), 0, st, iters, random_data, 0x9E3779B9u, scratch);
        else if (mode == 1) hipLaunchKernelGGL(mfma_rate_kernel<1>, dim3(blocks), dim3(256), 0, st, iters, random_data, 0x9E3779B9u, scratch);
        else hipLaunchKernelGGL(mfma_rate_kernel<2>, dim3(blocks), dim3(256), 0, st, iters, random_data, 0x9E3779B9u, scratch);
    });
    CODEPS_CHECK_LAUNCH("mfma_rate_kernel");
    return rc;
}

extern "C" int codeps_diag_stream_triad(float* a, const float* b, const float* c, long long n, int reps, float* ms_per_launch,
                                        codeps_stream_t stream) {
    CODEPS_REQUIRE(a && b && c && n > 0 && n % 4 == 0 && reps > 0 && ms_per_launch, CODEPS_ERR_ARG, "codeps_diag_stream_triad: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = timed(st, reps, ms_per_launch, [&]() {
        hipLaunchKernelGGL(triad_kernel, dim3(256 * 16), dim3(256), 0, st, reinterpret_cast<const f32x4*>(b), reinterpret_cast<const f32x4*>(c), 1.5f,
                           n / 4, reinterpret_cast<f32x4*>(a));
    });
    CODEPS_CHECK_LAUNCH("triad_kernel");
    return rc;
}
