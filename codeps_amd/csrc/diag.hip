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
#include "../../include/codeps_diag.h"

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

// In-kernel clock (MI355X_MICROARCH.md "DVFS give-back" item 6): s_memtime counts shader cycles, s_memrealtime a constant 100 MHz, so
// (t1 - t0) / (r1 - r0) x 100 MHz is the clock the chip HELD around the loop and (t1 - t0) / #MFMA the cycles per MFMA -- neither
// depends on an assumed frequency.  `stamps` (nullable; nothing else in the kernel reads it): per wave [4] = t0, r0, t1, r1.
__device__ __forceinline__ void clock_stamp(unsigned long long* __restrict__ dst) {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0): the stamps are back
    __builtin_amdgcn_sched_barrier(0);
    if ((threadIdx.x & 63) == 0) { dst[0] = t; dst[1] = r; }
}

template <int MODE>
__global__ __launch_bounds__(256) void mfma_rate_kernel(int iters, int random, unsigned seed, float* __restrict__ out,
                                                        unsigned long long* __restrict__ stamps = nullptr) {
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
    if (stamps) clock_stamp(stamps + (size_t)(tid >> 6) * 4);
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
            for (int u = 0; u < 16; ++u)         // accumulate IN PLACE (inline asm pins D = C): through the builtin the register allocator
                                                 // rotated the eight accumulators through overlapping windows (a[32:35] <- a[30:33] ...),
                                                 // which the hardware serialises -- 31.5 instead of 16 cycles per MFMA, a codegen artefact
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[u & 7]) : "v"(a[(u + (u >> 2)) & 3]), "v"(b[(u >> 2) & 3]));
        }
    }
    // (the closing stamp may precede the retirement of the last MFMA by its latency, <= 64 of the loop's ~10^6 cycles)
    if (stamps) clock_stamp(stamps + (size_t)(tid >> 6) * 4 + 2);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc32[j][r];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc16[j][0] + acc16[j][1] + acc16[j][2] + acc16[j][3];
    out[tid] = s;
}

// What a SIMD does with vector ALU work next to a matrix-core stream (codeps_diag_mfma_valu): waves 0-3 issue 16
// v_mfma_f32_32x32x16_bf16 per iteration on changing register operands; V v_fma_f32 per MFMA are issued EITHER by the same wave,
// interleaved (OTHER = 0), OR by a second wave on the same SIMD (waves 4-7 of a 512-thread block, OTHER = 1) that runs nothing else.
template <int V, int OTHER>
__global__ __launch_bounds__(OTHER ? 512 : 256) void mfma_valu_kernel(int iters, unsigned seed, float* __restrict__ out,
                                                                      unsigned long long* __restrict__ stamps = nullptr) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    float x[8], ca = unit(mix(seed + tid)), cb = unit(mix(seed * 7 + tid));
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = unit(mix(seed * 3 + tid * 8 + e));
    if (OTHER && threadIdx.x >= 256) {                       // the vector-ALU wave: 16 * V fma per iteration, nothing else
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[v & 7]) : "v"(ca), "v"(cb));
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += x[e];
        out[tid] = s;
        return;
    }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[q][e] = (__bf16)unit(mix(seed + tid * 64 + q * 16 + e));
            b[q][e] = (__bf16)unit(mix(seed * 3 + tid * 64 + q * 16 + e + 8));
        }
    // stamps: the MFMA waves only (wave w of block b -> slot 4 b + w)
    if (stamps) clock_stamp(stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + (u >> 2)) & 3], b[(u >> 2) & 3], acc[u & 3], 0, 0, 0);
            if (!OTHER) {
#pragma unroll
                for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[v & 7]) : "v"(ca), "v"(cb));
            }
        }
    if (stamps) clock_stamp(stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + 2);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
#pragma unroll
    for (int e = 0; e < 8; ++e) s += x[e];
    out[tid] = s;
}

__global__ __launch_bounds__(256) void triad_kernel(const f32x4* __restrict__ b, const f32x4* __restrict__ c, float s, long long n4,
                                                    f32x4* __restrict__ a) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) a[i] = b[i] + s * c[i];
}

// HBM copy, the form MI355X_MICROARCH.md quotes its measured roof on (float4 copy: 6.29 TB/s = 79 % of 8): every thread moves four 16-byte
// vectors per trip, all four loads issued before the first store (loads behind stores queue on vmcnt), consecutive lanes consecutive vectors.
__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ src, long long n4, f32x4* __restrict__ dst) {
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n4; i += stride) {
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i + 256 * k < n4 ? __builtin_nontemporal_load(src + i + 256 * k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + 256 * k < n4) __builtin_nontemporal_store(v[k], dst + i + 256 * k);
    }
}

// How many bytes per clock can a CU take in NEXT TO a full-rate MFMA stream?  One block of four waves per CU; per iteration
// every wave issues 96 v_mfma_f32_16x16x32_bf16 on register operands (a 128 x 128 x 32 six-product K-step) and the block
// brings `kb` KiB from an L2-resident window into LDS:
//   mode 0: no loads;  mode 1: buffer loads into VGPRs, ds_write_b128 after the MFMAs (how the conv kernels stage);
//   mode 2: buffer_load_dwordx4 ... lds (LDS DMA, no VGPR round trip), waited before the barrier.
// The loads of iteration i + 1 are issued before the MFMAs of iteration i (double-buffered LDS), as in the kernels.
typedef int i32x4d __attribute__((ext_vector_type(4)));
template <int MODE, int FRAGS>
__global__ __launch_bounds__(256) void feed_kernel(const char* __restrict__ src, unsigned window_bytes, int iters, int kb, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int MAXL = 16;                                     // 16-byte loads per thread and iteration: up to 64 KiB per block
    const int nl = kb * 1024 / (256 * 16);
    bf16x8 a[4], b[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[q][e] = (__bf16)unit(mix(0x1234u + tid * 64 + q * 16 + e));
            b[q][e] = (__bf16)unit(mix(0x9876u + tid * 64 + q * 16 + e));
        }
    f32x4 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    i32x4d rs;
    {
        const unsigned long long p = reinterpret_cast<unsigned long long>(src);
        rs[0] = (int)__builtin_amdgcn_readfirstlane((unsigned)p);
        rs[1] = (int)(__builtin_amdgcn_readfirstlane((unsigned)(p >> 32)) & 0xFFFFu);
        rs[2] = (int)window_bytes;
        rs[3] = 0x00020000;
    }
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)window_bytes, 0x00020000);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(lds);
    const unsigned span = (unsigned)kb * 1024u;
    f32x4 regs[MAXL];
    unsigned pos = (blockIdx.x * 7919u * 1024u) % (window_bytes - span);
    auto issue = [&](int buf) {
        const unsigned g0 = pos & ~1023u;
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < MAXL; ++i)
                if (i < nl) regs[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsb, g0 + (i * 256 + tid) * 16, 0, 0));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < MAXL; ++i)
                if (i < nl) {
                    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(lds_base + buf * span + (i * 4 + wave) * 1024);
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                                 :: "s"(dst), "v"((unsigned)(lane * 16)), "s"(rs), "s"(g0 + (i * 4 + wave) * 1024) : "memory");
                }
        }
        pos = (pos + span * 13u) % (window_bytes - span);
    };
    issue(0);
    float sink = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < MAXL; ++i)
                if (i < nl) *reinterpret_cast<f32x4*>(lds + buf * span + (i * 256 + tid) * 16) = regs[i];
        } else if (MODE == 2) {
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        __syncthreads();
        issue(buf ^ 1);
        if (FRAGS) {
            // the conv kernels' fragment traffic: 24 ds_read_b128 per wave and K-step (conflict-free 208-byte row pitch), read one
            // step ahead of the MFMAs that use them, one read per four MFMAs
            bf16x8 fa[12], fb[12];
            const char* fbase = lds + buf * span + ((lane & 15) * 208 + (lane >> 4) * 16) % (span - 4096);
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                fa[q] = *reinterpret_cast<const bf16x8*>(fbase + (q % 3) * 64 + (q / 3) * 16 * 208 % 2048);
                fb[q] = *reinterpret_cast<const bf16x8*>(fbase + 2048 + (q % 3) * 64 + (q / 3) * 16 * 208 % 2048);
            }
#pragma unroll
            for (int u = 0; u < 96; ++u)
                acc[u & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u + (u >> 4)) & 3], b[(u >> 2) & 3], acc[u & 15], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 24; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[q] = fa[q * 3]; b[q] = fb[q * 3 + 1]; }     // the next iteration's operands
#pragma unroll
            for (int q = 0; q < 12; ++q) sink += (float)fa[q][0] + (float)fb[q][0];
        } else {
#pragma unroll
            for (int u = 0; u < 96; ++u)
                acc[u & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u + (u >> 4)) & 3], b[(u >> 2) & 3], acc[u & 15], 0, 0, 0);
        }
        if (kb > 0) sink += *reinterpret_cast<const float*>(lds + buf * span + ((tid * 68 + it * 4) % span & ~3u));
        __syncthreads();
    }
    if (MODE == 2) __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int j = 0; j < 16; ++j) sink += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < MAXL; ++i)
            if (i < nl) sink += regs[i][0];
    }
    out[blockIdx.x * 256 + tid] = sink;
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

// One matrix instruction on caller-supplied operands: what the f16 / bf16 MFMA does with subnormal inputs, how exactly it forms and sums
// its 32 products (tools/f16_mfma_probe.py; the question behind the two-fp16-piece arithmetic of the conv engine).
// mode 0: v_mfma_f32_16x16x32_f16, 1: v_mfma_f32_16x16x32_bf16.  a [16][32], b [16][32] raw 16-bit patterns (row = M / N index, k contiguous),
// c, d [16][16] fp32 (d[m][n] = c[m][n] + sum_k a[m][k] b[n][k]).
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(64) void mfma_once_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                                                       const float* __restrict__ c, float* __restrict__ d) {
    const int lane = threadIdx.x, l16 = lane & 15, kg = lane >> 4;
    typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));
    const u16x8_t av = *reinterpret_cast<const u16x8_t*>(a + l16 * 32 + kg * 8), bv = *reinterpret_cast<const u16x8_t*>(b + l16 * 32 + kg * 8);
    f32x4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = c[(4 * kg + r) * 16 + l16];
    if (MODE == 0) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, av), __builtin_bit_cast(f16x8_t, bv), acc, 0, 0, 0);
    else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bv), acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) d[(4 * kg + r) * 16 + l16] = acc[r];
}

}  // namespace

extern "C" int codeps_diag_mfma_once(int mode, const void* a, const void* b, const float* c, float* d, codeps_stream_t stream) {
    CODEPS_REQUIRE((mode == 0 || mode == 1) && a && b && c && d, CODEPS_ERR_ARG, "codeps_diag_mfma_once: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mode == 0) hipLaunchKernelGGL(mfma_once_kernel<0>, dim3(1), dim3(64), 0, st, (const unsigned short*)a, (const unsigned short*)b, c, d);
    else hipLaunchKernelGGL(mfma_once_kernel<1>, dim3(1), dim3(64), 0, st, (const unsigned short*)a, (const unsigned short*)b, c, d);
    CODEPS_CHECK_LAUNCH("mfma_once_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_diag_mfma_rate(int mode, int random_data, int blocks, int iters, int reps, float* scratch, double* flops_per_launch,
                                     float* ms_per_launch, codeps_stream_t stream) {
    CODEPS_REQUIRE(mode >= 0 && mode <= 2 && blocks > 0 && iters > 0 && reps > 0 && scratch && flops_per_launch && ms_per_launch,
                   CODEPS_ERR_ARG, "codeps_diag_mfma_rate: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double per_mfma = mode == 0 ? 2.0 * 32 * 32 * 16 : (mode == 1 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 32);
    *flops_per_launch = per_mfma * 16.0 * iters * 4.0 * blocks;            // 16 MFMAs per iteration, 4 waves per block
    int rc = timed(st, reps, ms_per_launch, [&]() {
        if (mode == 0) hipLaunchKernelGGL(mfma_rate_kernel<0>, dim3(blocks), dim3(256), 0, st, iters, random_data, 0x9E3779B9u, scratch, (unsigned long long*)nullptr);
        else if (mode == 1) hipLaunchKernelGGL(mfma_rate_kernel<1>, dim3(blocks), dim3(256), 0, st, iters, random_data, 0x9E3779B9u, scratch, (unsigned long long*)nullptr);
        else hipLaunchKernelGGL(mfma_rate_kernel<2>, dim3(blocks), dim3(256), 0, st, iters, random_data, 0x9E3779B9u, scratch, (unsigned long long*)nullptr);
    });
    CODEPS_CHECK_LAUNCH("mfma_rate_kernel");
    return rc;
}

extern "C" int codeps_diag_mfma_rate_clock(int mode, int random_data, int blocks, int iters, int reps, float* scratch, unsigned long long* stamps,
                                           float* ms_per_launch, codeps_stream_t stream) {
    CODEPS_REQUIRE(mode >= 0 && mode <= 2 && blocks > 0 && iters > 0 && reps > 0 && scratch && stamps && ms_per_launch, CODEPS_ERR_ARG,
                   "codeps_diag_mfma_rate_clock: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = timed(st, reps, ms_per_launch, [&]() {
        if (mode == 0) hipLaunchKernelGGL(mfma_rate_kernel<0>, dim3(blocks), dim3(256), 0, st, iters, random_data, 0x9E3779B9u, scratch, stamps);
        else if (mode == 1) hipLaunchKernelGGL(mfma_rate_kernel<1>, dim3(blocks), dim3(256), 0, st, iters, random_data, 0x9E3779B9u, scratch, stamps);
        else hipLaunchKernelGGL(mfma_rate_kernel<2>, dim3(blocks), dim3(256), 0, st, iters, random_data, 0x9E3779B9u, scratch, stamps);
    });
    CODEPS_CHECK_LAUNCH("mfma_rate_kernel (stamped)");
    return rc;
}

extern "C" int codeps_diag_mfma_valu_clock(int valu_per_mfma, int other_wave, int blocks, int iters, int reps, float* scratch,
                                           unsigned long long* stamps, float* ms_per_launch, codeps_stream_t stream) {
    CODEPS_REQUIRE((valu_per_mfma == 0 || valu_per_mfma == 2 || valu_per_mfma == 4 || valu_per_mfma == 8 || valu_per_mfma == 16) && blocks > 0 &&
                       iters > 0 && reps > 0 && scratch && stamps && ms_per_launch,
                   CODEPS_ERR_ARG, "codeps_diag_mfma_valu_clock: bad argument (valu_per_mfma in {0, 2, 4, 8, 16})");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = timed(st, reps, ms_per_launch, [&]() {
#define MV(V_, O_) hipLaunchKernelGGL((mfma_valu_kernel<V_, O_>), dim3(blocks), dim3(O_ ? 512 : 256), 0, st, iters, 0x9E3779B9u, scratch, stamps)
        if (!other_wave) {
            switch (valu_per_mfma) { case 0: MV(0, 0); break; case 2: MV(2, 0); break; case 4: MV(4, 0); break; case 8: MV(8, 0); break; default: MV(16, 0); }
        } else {
            switch (valu_per_mfma) { case 0: MV(0, 1); break; case 2: MV(2, 1); break; case 4: MV(4, 1); break; case 8: MV(8, 1); break; default: MV(16, 1); }
        }
#undef MV
    });
    CODEPS_CHECK_LAUNCH("mfma_valu_kernel (stamped)");
    return rc;
}

extern "C" int codeps_diag_mfma_valu(int valu_per_mfma, int other_wave, int blocks, int iters, int reps, float* scratch, float* ms_per_launch,
                                     codeps_stream_t stream) {
    CODEPS_REQUIRE((valu_per_mfma == 0 || valu_per_mfma == 2 || valu_per_mfma == 4 || valu_per_mfma == 8 || valu_per_mfma == 16) && blocks > 0 &&
                       iters > 0 && reps > 0 && scratch && ms_per_launch,
                   CODEPS_ERR_ARG, "codeps_diag_mfma_valu: bad argument (valu_per_mfma in {0, 2, 4, 8, 16})");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = timed(st, reps, ms_per_launch, [&]() {
#define MV(V_, O_) hipLaunchKernelGGL((mfma_valu_kernel<V_, O_>), dim3(blocks), dim3(O_ ? 512 : 256), 0, st, iters, 0x9E3779B9u, scratch, (unsigned long long*)nullptr)
        if (!other_wave) {
            switch (valu_per_mfma) { case 0: MV(0, 0); break; case 2: MV(2, 0); break; case 4: MV(4, 0); break; case 8: MV(8, 0); break; default: MV(16, 0); }
        } else {
            switch (valu_per_mfma) { case 0: MV(0, 1); break; case 2: MV(2, 1); break; case 4: MV(4, 1); break; case 8: MV(8, 1); break; default: MV(16, 1); }
        }
#undef MV
    });
    CODEPS_CHECK_LAUNCH("mfma_valu_kernel");
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

extern "C" int codeps_diag_stream_copy(float* dst, const float* src, long long n, int reps, float* ms_per_launch, codeps_stream_t stream) {
    CODEPS_REQUIRE(dst && src && n > 0 && n % 4 == 0 && reps > 0 && ms_per_launch, CODEPS_ERR_ARG, "codeps_diag_stream_copy: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = timed(st, reps, ms_per_launch, [&]() {
        hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<const f32x4*>(src), n / 4, reinterpret_cast<f32x4*>(dst));
    });
    CODEPS_CHECK_LAUNCH("copy_kernel");
    return rc;
}

extern "C" int codeps_diag_feed_rate(int mode, int kb, int blocks, int iters, int reps, const void* window, long long window_bytes,
                                     float* scratch, float* ms_per_launch, codeps_stream_t stream) {
    CODEPS_REQUIRE(mode >= 0 && (mode & 3) <= 2 && mode < 8 && kb >= 0 && kb <= 64 && (kb % 4) == 0 && blocks > 0 && iters > 0 && reps > 0 && window &&
                       window_bytes >= (1 << 20) && window_bytes < (1LL << 31) && scratch && ms_per_launch,
                   CODEPS_ERR_ARG, "codeps_diag_feed_rate: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = (size_t)2 * (kb ? kb : 4) * 1024;
    CODEPS_REQUIRE(!(mode & 4) || kb >= 8, CODEPS_ERR_ARG, "codeps_diag_feed_rate: the fragment reads need kb >= 8");
    codeps_allow_lds((const void*)feed_kernel<0, 0>, 160 * 1024, "feed_kernel<0,0>");
    codeps_allow_lds((const void*)feed_kernel<1, 0>, 160 * 1024, "feed_kernel<1,0>");
    codeps_allow_lds((const void*)feed_kernel<2, 0>, 160 * 1024, "feed_kernel<2,0>");
    codeps_allow_lds((const void*)feed_kernel<0, 1>, 160 * 1024, "feed_kernel<0,1>");
    codeps_allow_lds((const void*)feed_kernel<1, 1>, 160 * 1024, "feed_kernel<1,1>");
    codeps_allow_lds((const void*)feed_kernel<2, 1>, 160 * 1024, "feed_kernel<2,1>");
    const char* w = static_cast<const char*>(window);
    int rc = timed(st, reps, ms_per_launch, [&]() {
#define FEED(M, F) hipLaunchKernelGGL((feed_kernel<M, F>), dim3(blocks), dim3(256), lds, st, w, (unsigned)window_bytes, iters, kb, scratch)
        switch (mode) {
            case 0: FEED(0, 0); break;
            case 1: FEED(1, 0); break;
            case 2: FEED(2, 0); break;
            case 4: FEED(0, 1); break;
            case 5: FEED(1, 1); break;
            default: FEED(2, 1); break;
        }
#undef FEED
    });
    CODEPS_CHECK_LAUNCH("feed_kernel");
    return rc;
}
