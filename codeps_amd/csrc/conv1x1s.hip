// 1x1 convolution with BOTH operands pre-split (three bf16 pieces per fp32 value) and brought in by LDS DMA: the measurement
// behind DESIGN.md 8 "what would move the 1x1 layers".  tools/feed_rate.py shows that a CU takes the 48 KiB of a
// 128 x 128 x 32 six-product K-step by `buffer_load_dwordx4 ... lds` at the MFMA rate itself, against 1.34 us through VGPRs --
// provided the activations already ARE in the matrix cores' operand form.  This file holds that form and its first consumer:
//   * codeps_split_pack_rows    fp32 [M][C] -> [M][C / 32][piece 0..2][32] bf16 (the layout of the conv engine's pre-split
//                               weight pack, codeps_conv2d_pack_weights: a row's 32-channel group is 192 contiguous bytes)
//   * codeps_conv1x1_split_fwd  y[M][N] = act(scale * (A . W^T) + shift + residual), A in that layout, W the pre-split pack
// Kernel: 128 x 64 tiles, four waves of 64 x 32, 32-deep K-steps of 18 fragment reads + 48 v_mfma_f32_16x16x32_bf16 per wave.
// Both operand tiles of step k + 2 are DMA'd into the LDS slot whose fragments were read during step k - 1; step k + 1's
// fragments are read under step k's MFMAs (one read per two or three MFMAs, pinned).  No VALU, no ds_write, no staging
// registers in the K loop.  80 KB of LDS: two blocks per CU, the second block's MFMAs cover the first one's DMA wait.
// Arithmetic: identical to conv.hip (six products of order <= 2 per multiply, fp32 accumulation).
#include "common.h"
#include "../../include/codeps_diag.h"
#include <algorithm>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr unsigned OOB = 0xFFFF0000u;

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk / 8, r = nblk % 8, x = bid % 8, i = bid / 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
__device__ __forceinline__ float act_of(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : expm1f(v);
    return v;
}
__device__ __forceinline__ i32x4 rsrc4(const void* base, size_t bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    i32x4 r;
    r[0] = (int)__builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = (int)(__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xFFFFu);
    r[2] = (int)__builtin_amdgcn_readfirstlane((unsigned)(bytes < (size_t)OOB ? bytes : (size_t)OOB));
    r[3] = 0x00020000;
    return r;
}

// one thread per (row, 32-channel group, 8-channel chunk): 32 bytes in, three 16-byte pieces out
__global__ __launch_bounds__(256) void split_pack_rows_kernel(const float* __restrict__ x, long long rows, int C, int ldx,
                                                              unsigned short* __restrict__ out) {
    const long long total = rows * (C / 8);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / (C / 8);
        const int c8 = (int)(i % (C / 8)), grp = c8 >> 2, chunk = c8 & 3;
        const float* src = x + row * ldx + c8 * 8;
        f32x4 v[2] = {*reinterpret_cast<const f32x4*>(src), *reinterpret_cast<const f32x4*>(src + 4)};
        unsigned short pc[3][8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float r = v[h][e];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const unsigned short b = __builtin_bit_cast(unsigned short, (__bf16)r);
                    pc[q][h * 4 + e] = b;
                    r -= __builtin_bit_cast(float, (unsigned)b << 16);
                }
            }
        unsigned short* dst = out + (row * (C / 32) + grp) * 96 + chunk * 8;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            *reinterpret_cast<i32x4*>(dst + q * 32) =
                i32x4{(int)(pc[q][0] | ((unsigned)pc[q][1] << 16)), (int)(pc[q][2] | ((unsigned)pc[q][3] << 16)),
                      (int)(pc[q][4] | ((unsigned)pc[q][5] << 16)), (int)(pc[q][6] | ((unsigned)pc[q][7] << 16))};
    }
}

struct C1P {
    int M, N, K, n_mt, n_nt, act;
};

__global__ __launch_bounds__(256) void conv1x1s_kernel(C1P p, const unsigned short* __restrict__ a, const unsigned short* __restrict__ w,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ residual, float* __restrict__ out) {
    constexpr int BM = 128, BN = 64, BK = 32, NP = 3;
    constexpr int ROWB = 208;                                      // LDS row: 192 bytes + 16 (conflict-free 16-byte fragment reads)
    constexpr int STAGE = (BM + BN) * ROWB;                        // 39 936 bytes
    constexpr int AINS = BM * 13 / 64, BINS = BN * 13 / 64, INS = AINS + BINS, IPW = (INS + 3) / 4;   // 26 + 13 wave-level loads per stage
    static_assert((BM * 13) % 64 == 0 && (BN * 13) % 64 == 0, "linear DMA layout");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int logical = xcd_remap(blockIdx.x, p.n_mt * p.n_nt);
    const int nt = logical % p.n_nt, mt = logical / p.n_nt;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk = p.K / BK;
    const size_t rowbytes = (size_t)p.K * 6;

    // rows rebased to the tile so that the per-lane byte offsets fit 32 bits
    const i32x4 rsA = rsrc4(reinterpret_cast<const char*>(a) + (size_t)m0 * rowbytes, (size_t)(p.M - m0) * rowbytes);
    const i32x4 rsB = rsrc4(reinterpret_cast<const char*>(w) + (size_t)n0 * rowbytes, (size_t)(p.N - n0) * rowbytes);
    unsigned goff[IPW];
    auto ins_of = [&](int i) { return wave + 4 * i < INS ? wave + 4 * i : wave + 4 * (i - 1); };       // wave-uniform
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int ins = ins_of(i);
        const bool isb = ins >= AINS;
        const int L = (isb ? ins - AINS : ins) * 64 + lane, row = L / 13, seg = L % 13;
        const int rmax = isb ? min(BN, p.N - n0) - 1 : min(BM, p.M - m0) - 1;
        goff[i] = (unsigned)((size_t)min(row, rmax) * rowbytes + (seg < 12 ? seg : 0) * 16);
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(lds);
    auto dma = [&](int k, int slot) {
        const unsigned koff = (unsigned)__builtin_amdgcn_readfirstlane(k * 192);
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int ins = ins_of(i);
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(lds_base + slot * STAGE + ins * 1024);
            const i32x4 rs = ins >= AINS ? rsB : rsA;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                         :: "s"(dst), "v"(goff[i]), "s"(rs), "s"(koff) : "memory");
        }
    };

    constexpr int NA = 4, NB = 2;
    const int l16 = lane & 15, kg = lane >> 4;
    f32x4 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* arow = lds + (wm * 64 + l16) * ROWB + kg * 16;
    const char* brow = lds + BM * ROWB + (wn * 32 + l16) * ROWB + kg * 16;
    struct Frag { bf16x8 a[NA][NP], b[NB][NP]; };
    auto read_frags = [&](int slot, Frag& f) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int q = 0; q < NP; ++q) f.a[i][q] = *reinterpret_cast<const bf16x8*>(arow + slot * STAGE + i * 16 * ROWB + q * 64);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int q = 0; q < NP; ++q) f.b[j][q] = *reinterpret_cast<const bf16x8*>(brow + slot * STAGE + j * 16 * ROWB + q * 64);
    };
    auto mfmas = [&](const Frag& f) {
#pragma unroll
        for (int ord = NP - 1; ord >= 0; --ord)
#pragma unroll
            for (int qa = 0; qa <= ord; ++qa)
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[i][qa], f.b[j][ord - qa], acc[i][j], 0, 0, 0);
    };

    dma(0, 0);
    dma(1, 1);                                                     // (past the end of K: the range check returns zeros)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    Frag fr[2];
    read_frags(0, fr[0]);
    __syncthreads();                                               // every wave holds step 0's fragments: slot 0 may be refilled
    for (int kb = 0; kb < nk; kb += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = kb + u;
            dma(k + 2, u);                                         // slot of step k, read during step k - 1
            __builtin_amdgcn_sched_barrier(0);
            read_frags(u ^ 1, fr[u ^ 1]);                          // step k + 1 (landed before the barrier that ended step k - 1)
            mfmas(fr[u]);
#pragma unroll
            for (int g = 0; g < 6; ++g) {                           // 18 reads, 48 MFMAs: 3 + 3 + 2 MFMAs behind the reads
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0F70);                    // step k + 2 has landed (the other block of the CU runs meanwhile)
            __syncthreads();
            if (k + 1 >= nk) break;
        }
    }

    // ---- epilogue: accumulators -> LDS -> 16-byte rows
    constexpr int LDC = BN + 4;
    float* Cs = reinterpret_cast<float*>(lds);
    constexpr int V4 = BN / 4, RPASS = 256 / V4;
    const int c4 = tid % V4, r0 = tid / V4;
    const int n = n0 + c4 * 4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (n < p.N) {
        if (scale) sc = *reinterpret_cast<const f32x4*>(scale + n);
        if (shift) sh = *reinterpret_cast<const f32x4*>(shift + n);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i > 0) __syncthreads();
#pragma unroll
        for (int ih = 0; ih < 2; ++ih)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Cs[(wm * 32 + ih * 16 + 4 * kg + r) * LDC + (wn * NB + j) * 16 + l16] = acc[2 * i + ih][j][r];
        __syncthreads();
        if (n < p.N) {
#pragma unroll 4
            for (int rr = r0; rr < 64; rr += RPASS) {
                const int m = m0 + ((rr >> 5) * 2 + i) * 32 + (rr & 31);
                if (m >= p.M) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(Cs + rr * LDC + c4 * 4);
                v = v * sc + sh;
                if (residual) v += *reinterpret_cast<const f32x4*>(residual + (size_t)m * p.N + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_of(v[e], p.act);
                *reinterpret_cast<f32x4*>(out + (size_t)m * p.N + n) = v;
            }
        }
    }
}

}  // namespace

extern "C" int codeps_split_pack_rows(const float* x, long long rows, int channels, int row_stride, void* out, codeps_stream_t stream) {
    CODEPS_REQUIRE(x && out && rows > 0 && channels > 0 && channels % 32 == 0 && row_stride >= channels && row_stride % 4 == 0, CODEPS_ERR_ARG,
                   "codeps_split_pack_rows: channels must be a multiple of 32 (got %d), row_stride >= channels", channels);
    const long long total = rows * (channels / 8);
    hipLaunchKernelGGL(split_pack_rows_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, x, rows,
                       channels, row_stride, static_cast<unsigned short*>(out));
    CODEPS_CHECK_LAUNCH("split_pack_rows_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_conv1x1_split_fwd(const void* a_split, long long rows, int in_c, int out_c, const void* w_split, const float* scale,
                                        const float* shift, const float* residual, int act, float* y, codeps_stream_t stream) {
    CODEPS_REQUIRE(a_split && w_split && y && rows > 0 && rows < (1LL << 31) && in_c >= 64 && in_c % 32 == 0 && out_c >= 64 && out_c % 64 == 0 &&
                       act >= 0 && act <= 2,
                   CODEPS_ERR_ARG, "codeps_conv1x1_split_fwd: in_c %d must be a multiple of 32 (>= 64), out_c %d a multiple of 64", in_c, out_c);
    C1P p;
    p.M = (int)rows; p.N = out_c; p.K = in_c; p.act = act;
    p.n_mt = cdiv(p.M, 128); p.n_nt = out_c / 64;
    codeps_allow_lds((const void*)conv1x1s_kernel, 160 * 1024, "conv1x1s_kernel");
    hipLaunchKernelGGL(conv1x1s_kernel, dim3(p.n_mt * p.n_nt), dim3(256), 2 * (128 + 64) * 208, (hipStream_t)stream, p,
                       static_cast<const unsigned short*>(a_split), static_cast<const unsigned short*>(w_split), scale, shift, residual, y);
    CODEPS_CHECK_LAUNCH("conv1x1s_kernel");
    return CODEPS_OK;
}
