// HBM-bound NHWC building blocks around the conv engine (all fp32, 16-byte vector accesses, no atomics):
// layout conversion, activation backward + bias gradient, BatchNorm (train-mode statistics as
// exchangeable sums for SyncBN, apply, backward), 3x3/2 max-pool, 5x5 depthwise conv, bilinear resize
// (both align_corners conventions) writing into channel slices (concat-by-pointer), channel softmax,
// sigmoid + disparity->depth.
//
// Reference ops replaced: nn.BatchNorm2d / SyncBatchNorm (misc/train_utils.py:50-51, torchvision ResNet,
// models/instance_head.py:26-49), nn.MaxPool2d(3,2,1) (models/resnet_encoder.py:31), depthwise 5x5
// (models/instance_head.py:38-46,90-98), F.interpolate bilinear (models/semantic_head.py:66-69,
// models/instance_head.py:65-68,116-118), nn.Softmax(dim=1) (semantic_head.py:74, instance_head.py:146),
// sigmoid + disp_to_depth (models/depth_head.py:48-54,71-74), ELU/ReLU backward.
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GRID_CAP = 256 * 8;          // memory-bound grids: <= 8 blocks per CU, grid-stride the rest

static inline int grid_for(long long work_items) {
    return (int)std::max<long long>(1, std::min<long long>((work_items + 255) / 256, GRID_CAP));
}

// ---------------------------------------------------------------------------------- layout
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, int B, int C, int H, int W,
                                                           float* __restrict__ y, int Cp, int c_off, int zero_rest, int pad_l, int Wp) {
    // one thread per (destination pixel, 4-channel group); rows of the destination are Wp >= pad_l + W pixels wide, the
    // border columns are zero (written when zero_rest)
    const int groups = zero_rest ? Cp / 4 : (C + 3) / 4;
    const int Wd = zero_rest ? Wp : W;                           // columns this launch walks
    const long long total = (long long)B * H * Wd * groups;
    const long long HW = (long long)H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % groups);
        long long r = i / groups;
        const int xd = (int)(r % Wd); r /= Wd;
        const int yy = (int)(r % H);
        const int b = (int)(r / H);
        const int xs = zero_rest ? xd - pad_l : xd;              // source column
        float* dst = y + (((size_t)b * H + yy) * Wp + (zero_rest ? xd : xd + pad_l)) * Cp;
        const long long hw = (long long)yy * W + xs;
        if (zero_rest) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = g * 4 + e - c_off;
                v[e] = (c >= 0 && c < C && xs >= 0 && xs < W) ? x[((size_t)b * C + c) * HW + hw] : 0.f;
            }
            *reinterpret_cast<f32x4*>(dst + g * 4) = v;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = g * 4 + e;
                if (c < C) dst[c_off + c] = x[((size_t)b * C + c) * HW + hw];
            }
        }
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, int B, int C, int H, int W, int Cp,
                                                           float* __restrict__ y) {
    const long long HW = (long long)H * W, total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long hw = i % HW;
        const int c = (int)((i / HW) % C);
        const int b = (int)(i / (HW * C));
        y[i] = x[((size_t)b * HW + hw) * Cp + c];
    }
}

// ---------------------------------------------------------------------------------- column sums (per-channel)
// Generic two-stage deterministic per-channel reduction over M rows of an [M][C] matrix.
// stage 1: block (rows chunk) -> partial[blk][K][C];  stage 2: sum over blocks in order.

__device__ __forceinline__ float act_grad(float y, int act) {
    if (act == 1) return y > 0.f ? 1.f : 0.f;
    if (act == 2) return y > 0.f ? 1.f : y + 1.f;
    return 1.f;
}

// dz = dy * act'(y) * scale;  col partial of dy*act'(y)  (bias / shift gradient)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      const float* __restrict__ scale, int act, long long M, int C,
                                                      float* __restrict__ dz, float* __restrict__ part, int rows_per_blk) {
    // thread = (row lane r = tid / (C/4 clipped), channel group); C/4 groups must divide 256 or be <= 256
    const int Gall = C / 4, G = Gall < 256 ? Gall : 256;          // blockIdx.y tiles the channel groups
    const int tl = threadIdx.x % G, tr = threadIdx.x / G, TR = 256 / G;
    const int tg = blockIdx.y * G + tl;
    const bool cok = tg < Gall && tr < TR;
    const long long r0 = (long long)blockIdx.x * rows_per_blk;
    const long long r1 = r0 + rows_per_blk < M ? r0 + rows_per_blk : M;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 sc = {1.f, 1.f, 1.f, 1.f};
    if (scale && cok) sc = *reinterpret_cast<const f32x4*>(scale + tg * 4);
    // four rows per trip, every load issued before the first use; `act` / `dz` are launch-uniform: one copy of the loop each
    auto body = [&](auto has_act, auto has_dz) {
        for (long long r = r0 + tr; r < r1; r += 4 * TR) {
            f32x4 g[4], yy[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t o = (size_t)min(r + (long long)u * TR, r1 - 1) * C + tg * 4;
                g[u] = *reinterpret_cast<const f32x4*>(dy + o);
                if constexpr (decltype(has_act)::value) yy[u] = *reinterpret_cast<const f32x4*>(y + o);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long rr = r + (long long)u * TR;
                if (rr >= r1) break;
                f32x4 d = g[u];
                if constexpr (decltype(has_act)::value) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = g[u][e] * act_grad(yy[u][e], act);
                }
                acc += d;
                if constexpr (decltype(has_dz)::value) *reinterpret_cast<f32x4*>(dz + (size_t)rr * C + tg * 4) = d * sc;
            }
        }
    };
    if (cok && r0 < r1) {
        if (act) { if (dz) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{}); }
        else { if (dz) body(std::false_type{}, std::true_type{}); else body(std::false_type{}, std::false_type{}); }
    }
    if (!part) return;
    __shared__ f32x4 red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < G && blockIdx.y * G + threadIdx.x < Gall) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < TR; ++k) t += red[k * G + threadIdx.x];
        *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * C + (blockIdx.y * G + threadIdx.x) * 4) = t;
    }
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int nblk, int K, int C,
                                                            float* __restrict__ out, int c_real, int accumulate) {
    // part: [nblk][K][C] -> out [K][c_real]; one wave per output walks the blocks (fixed assignment: lane l takes blocks l, l + 64, ...
    // in order, then a fixed shuffle tree) -- with <= 1024 blocks that is <= 16 dependent loads per lane (16 lanes per output
    // walked 64 each and the kernel took 11 us of pure latency, 25 times per step)
    const int total = K * C;
    const int i = blockIdx.x * 4 + threadIdx.x / 64, sub = threadIdx.x % 64;
    const bool in = i < total;
    float a = 0.f;
    if (in) {
#pragma unroll 4
        for (int b = sub; b < nblk; b += 64) a += part[(size_t)b * total + i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (!in || sub != 0) return;
    const int c = i % C, k = i / C;
    if (c >= c_real) return;
    float* o = out + (size_t)k * c_real + c;
    *o = accumulate ? *o + a : a;
}

// Several column-sum finishes in one launch (the bias gradients of a backward pass, accumulated into their slots of the flat gradient
// buffer): item k owns blocks [blk0[k], blk0[k + 1]); per item exactly colsum_finish_kernel's arithmetic (K = 1, accumulate).
struct ColsumItems { const float* part[32]; float* out[32]; int tiles[32], c[32], c_real[32], blk0[33]; int n; };
__global__ __launch_bounds__(256) void colsum_finish_multi_kernel(ColsumItems it) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < 32; ++j) k += (j < it.n && (int)blockIdx.x >= it.blk0[j]) ? 1 : 0;
    const float* part = it.part[0];
    float* out = it.out[0];
    int nblk = it.tiles[0], C = it.c[0], c_real = it.c_real[0], b0 = it.blk0[0];
#pragma unroll
    for (int j = 1; j < 32; ++j)
        if (j == k) { part = it.part[j]; out = it.out[j]; nblk = it.tiles[j]; C = it.c[j]; c_real = it.c_real[j]; b0 = it.blk0[j]; }     // (select chain: no indexed access)
    const int i = ((int)blockIdx.x - b0) * 4 + threadIdx.x / 64, sub = threadIdx.x % 64;
    const bool in = i < C;
    float a = 0.f;
    if (in) {
#pragma unroll 4
        for (int b = sub; b < nblk; b += 64) a += part[(size_t)b * C + i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (!in || sub != 0 || i >= c_real) return;
    out[i] += a;
}

// ---------------------------------------------------------------------------------- BatchNorm (train)
// forward statistics: partial (sum (x - s), sum (x - s)^2) per channel with the shift s = x[row 0][c]
// ("shifted data" variance: no catastrophic cancellation when |mean| >> std)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, long long M, int C,
                                                       float* __restrict__ part, int rows_per_blk) {
    const int Gall = C / 4, G = Gall < 256 ? Gall : 256;          // blockIdx.y tiles the channel groups
    const int tl = threadIdx.x % G, tr = threadIdx.x / G, TR = 256 / G;
    const int tg = blockIdx.y * G + tl;
    const bool cok = tg < Gall && tr < TR;
    const long long r0 = (long long)blockIdx.x * rows_per_blk;
    const long long r1 = r0 + rows_per_blk < M ? r0 + rows_per_blk : M;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (cok) {
        const f32x4 sh = *reinterpret_cast<const f32x4*>(x + tg * 4);
        for (long long r = r0 + tr; r < r1; r += TR) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r * C + tg * 4) - sh;
            s += v; q += v * v;
        }
    }
    __shared__ f32x4 red[2][256];
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < G && blockIdx.y * G + threadIdx.x < Gall) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < TR; ++k) { a += red[0][k * G + threadIdx.x]; b += red[1][k * G + threadIdx.x]; }
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + 0) * C + (blockIdx.y * G + threadIdx.x) * 4) = a;
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + 1) * C + (blockIdx.y * G + threadIdx.x) * 4) = b;
    }
}

// y = act((x - mean) * invstd * gamma + beta + residual)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ residual,
                                                       int act, long long M, int C, float* __restrict__ y) {
    const int G = C / 4;
    const long long total = M * G;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % G);
        const size_t o = (size_t)(i / G) * C + g * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + o);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + g * 4), is = *reinterpret_cast<const f32x4*>(invstd + g * 4);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + g * 4), be = *reinterpret_cast<const f32x4*>(beta + g * 4);
        f32x4 r = (v - mu) * is * ga + be;
        if (residual) r += *reinterpret_cast<const f32x4*>(residual + o);
        if (act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.f ? r[e] : 0.f;
        }
        *reinterpret_cast<f32x4*>(y + o) = r;
    }
}

// backward reduce: g = dy * act'(y); partial (sum g, sum g * xhat); optionally writes g (masked grad) to dres
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, int act, long long M, int C,
                                                            float* __restrict__ part, int rows_per_blk) {
    const int Gall = C / 4, G = Gall < 256 ? Gall : 256;          // blockIdx.y tiles the channel groups
    const int tl = threadIdx.x % G, tr = threadIdx.x / G, TR = 256 / G;
    const int tg = blockIdx.y * G + tl;
    const bool cok = tg < Gall && tr < TR;
    const long long r0 = (long long)blockIdx.x * rows_per_blk;
    const long long r1 = r0 + rows_per_blk < M ? r0 + rows_per_blk : M;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (cok) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + tg * 4), is = *reinterpret_cast<const f32x4*>(invstd + tg * 4);
        for (long long r = r0 + tr; r < r1; r += TR) {
            const size_t o = (size_t)r * C + tg * 4;
            f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
            if (act == 1) {
                const f32x4 yy = *reinterpret_cast<const f32x4*>(y + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = yy[e] > 0.f ? g[e] : 0.f;
            }
            const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + o) - mu) * is;
            s += g; q += g * xh;
        }
    }
    __shared__ f32x4 red[2][256];
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < G && blockIdx.y * G + threadIdx.x < Gall) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < TR; ++k) { a += red[0][k * G + threadIdx.x]; b += red[1][k * G + threadIdx.x]; }
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + 0) * C + (blockIdx.y * G + threadIdx.x) * 4) = a;
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 2 + 1) * C + (blockIdx.y * G + threadIdx.x) * 4) = b;
    }
}

// dx = gamma*invstd * (g - sum_g/N - xhat * sum_gx/N);  dres = g
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ sums, float inv_count, int act,
                                                           long long M, int C, float* __restrict__ dx,
                                                           float* __restrict__ dres) {
    const int G = C / 4;
    const long long total = M * G;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g4 = (int)(i % G);
        const size_t o = (size_t)(i / G) * C + g4 * 4;
        f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
        if (act == 1) {
            const f32x4 yy = *reinterpret_cast<const f32x4*>(y + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = yy[e] > 0.f ? g[e] : 0.f;
        }
        if (dres) *reinterpret_cast<f32x4*>(dres + o) = g;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + g4 * 4), is = *reinterpret_cast<const f32x4*>(invstd + g4 * 4);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + g4 * 4);
        const f32x4 sg = *reinterpret_cast<const f32x4*>(sums + g4 * 4), sgx = *reinterpret_cast<const f32x4*>(sums + C + g4 * 4);
        const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + o) - mu) * is;
        *reinterpret_cast<f32x4*>(dx + o) = ga * is * (g - sg * inv_count - xh * sgx * inv_count);
    }
}

// ---------------------------------------------------------------------------------- BatchNorm, multi-group launches
// Several independent statistic groups (contiguous row ranges of one [rows][C] tensor, e.g. the six pose-net calls of
// an adaptation step batched into one forward) handled by ONE launch per phase: blockIdx.z = group.
struct BnGroups { int n; long long row0[8]; long long rows[8]; float inv_count[8]; };

// partial (sum (x - s_g), sum (x - s_g)^2) with s_g = first row of the group: part[g][blk][2][C]
__global__ __launch_bounds__(256) void bn_stats_mg_kernel(const float* __restrict__ x, BnGroups gr, int C, int nblk,
                                                          float* __restrict__ part) {
    const int g = blockIdx.z;
    const int Gall = C / 4, G = Gall < 256 ? Gall : 256;
    const int tl = threadIdx.x % G, tr = threadIdx.x / G, TR = 256 / G;
    const int tg = blockIdx.y * G + tl;
    const bool cok = tg < Gall && tr < TR;
    const long long M = gr.rows[g], rpb = (M + nblk - 1) / nblk;
    const long long r0 = gr.row0[g] + (long long)blockIdx.x * rpb;
    const long long r1 = min(r0 + rpb, gr.row0[g] + M);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (cok && r0 < r1) {
        const f32x4 sh = *reinterpret_cast<const f32x4*>(x + (size_t)gr.row0[g] * C + tg * 4);
        // four rows per trip, all loads issued before the first use, no branch in the body (rows past the range re-read the
        // last row with weight 0): the loop as first written waited for every load before issuing the next
        for (long long r = r0 + tr; r < r1; r += 4 * TR) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (size_t)min(r + (long long)u * TR, r1 - 1) * C + tg * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float wgt = (r + (long long)u * TR < r1) ? 1.f : 0.f;
                const f32x4 d = (v[u] - sh) * wgt;
                s += d; q += d * d;
            }
        }
    }
    __shared__ f32x4 red[2][256];
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < G && blockIdx.y * G + threadIdx.x < Gall) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < TR; ++k) { a += red[0][k * G + threadIdx.x]; b += red[1][k * G + threadIdx.x]; }
        float* dst = part + (((size_t)g * nblk + blockIdx.x) * 2) * C + (blockIdx.y * G + threadIdx.x) * 4;
        *reinterpret_cast<f32x4*>(dst) = a;
        *reinterpret_cast<f32x4*>(dst + C) = b;
    }
}

// one thread per channel: finish the partial sums of every group (fixed order), mean / biased var / invstd per group,
// and the running-stat momentum updates applied group after group (= the order of the reference's separate calls)
__global__ __launch_bounds__(256) void bn_finalize_mg_kernel(const float* __restrict__ x, const float* __restrict__ part, BnGroups gr,
                                                             int C, int c_real, int nblk, float eps, float momentum,
                                                             float* __restrict__ mean, float* __restrict__ var,
                                                             float* __restrict__ invstd, float* __restrict__ running_mean,
                                                             float* __restrict__ running_var) {
    // one wave per channel walks the partial blocks (fixed assignment + fixed-order shuffle tree -> deterministic)
    const int c = blockIdx.x * 4 + threadIdx.x / 64, sub = threadIdx.x % 64;
    const bool cin = c < C;
    float rm = 0.f, rv = 0.f;
    const bool upd = running_mean != nullptr && cin && c < c_real;
    if (upd) { rm = running_mean[c]; rv = running_var[c]; }
    for (int g = 0; g < gr.n; ++g) {
        float s1 = 0.f, s2 = 0.f;
        if (cin) {
            const float* p = part + (size_t)g * nblk * 2 * C + c;
            for (int b = sub; b < nblk; b += 64) { s1 += p[(size_t)b * 2 * C]; s2 += p[(size_t)b * 2 * C + C]; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (!cin || sub != 0) continue;
        const float n = (float)gr.rows[g];
        const float d = s1 / n;
        const float m = x[(size_t)gr.row0[g] * C + c] + d;
        const float v = fmaxf(s2 / n - d * d, 0.f);
        mean[g * C + c] = m; var[g * C + c] = v; invstd[g * C + c] = rsqrtf(v + eps);
        if (upd) {
            rm = rm * (1.f - momentum) + momentum * m;
            rv = rv * (1.f - momentum) + momentum * (v * (n / fmaxf(n - 1.f, 1.f)));
        }
    }
    if (upd && sub == 0) { running_mean[c] = rm; running_var[c] = rv; }
}

// SyncBatchNorm: merge the per-rank (mean, biased var) of every statistic group after the all-gather with the parallel-variance
// formula and apply the running-stat momentum updates group after group with the GLOBAL count (torch.nn.SyncBatchNorm
// semantics).  gathered: [world][2][G][C].  Equal element counts on all ranks (rc.on == 0) reduce the merge to plain averages;
// otherwise rank r weighs in with rc.n[r * G + g] elements (0 = the rank holds nothing of that group: its slot is ignored).
// One thread per channel; replaces ~45 tiny tensor ops per layer.
struct RankCounts { int on; float n[64]; };
__global__ __launch_bounds__(256) void bn_sync_combine_kernel(const float* __restrict__ gathered, int world, BnGroups gr, RankCounts rc, int C, int c_real,
                                                              float eps, float momentum, float* __restrict__ mean, float* __restrict__ var,
                                                              float* __restrict__ invstd, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const bool upd = running_mean != nullptr && c < c_real;
    float rm = upd ? running_mean[c] : 0.f, rv = upd ? running_var[c] : 0.f;
    const float inv_w = 1.f / (float)world;
    const size_t rank_stride = (size_t)2 * gr.n * C;
    for (int g = 0; g < gr.n; ++g) {
        float m = 0.f, v = 0.f;
        if (!rc.on) {
            for (int r = 0; r < world; ++r) m += gathered[r * rank_stride + (size_t)g * C + c];
            m *= inv_w;
            for (int r = 0; r < world; ++r) {
                const float mr = gathered[r * rank_stride + (size_t)g * C + c], vr = gathered[r * rank_stride + (size_t)(gr.n + g) * C + c];
                v += vr + (mr - m) * (mr - m);
            }
            v *= inv_w;
        } else {
            float tot = 0.f;
            for (int r = 0; r < world; ++r) {
                const float w = rc.n[r * gr.n + g];
                if (w > 0.f) { m += w * gathered[r * rank_stride + (size_t)g * C + c]; tot += w; }
            }
            m /= tot;
            for (int r = 0; r < world; ++r) {
                const float w = rc.n[r * gr.n + g];
                if (w > 0.f) {
                    const float mr = gathered[r * rank_stride + (size_t)g * C + c], vr = gathered[r * rank_stride + (size_t)(gr.n + g) * C + c];
                    v += w * (vr + (mr - m) * (mr - m));
                }
            }
            v /= tot;
        }
        mean[g * C + c] = m; var[g * C + c] = v; invstd[g * C + c] = rsqrtf(v + eps);
        if (upd) {
            const float n = gr.inv_count[g];                     // here: the GLOBAL element count of the group
            rm = rm * (1.f - momentum) + momentum * m;
            rv = rv * (1.f - momentum) + momentum * (v * (n / fmaxf(n - 1.f, 1.f)));
        }
    }
    if (upd) { running_mean[c] = rm; running_var[c] = rv; }
}

// ---- BatchNorm statistics from the producing / consuming convolution's epilogue (ConvP.ex_stat, conv.hip): per M TILE partials
// instead of a pass over the tensor.  A tile lies inside one image; group g = images [b0, b1); segment k (the stride-2 data gradient is
// four launches) holds tiles tile0[k] + image * tpi[k] + (0 .. tpi[k]), rows[k] rows each.
struct TileSegs { int n; int tile0[4]; int tpi[4]; int rows[4]; };
struct ImgGroups { int n; int b0[8]; int b1[8]; };

// One block per (64 channels, statistic group), 1024 threads = 16 waves: wave w walks tiles w, w + 16, ... with its lanes on 64
// CONSECUTIVE channels (256-byte rows of the partials), the 16 waves meet in LDS in a fixed order.  (First form, round 5: one wave per
// channel with the lanes on different tiles -- 4-byte reads 768 bytes apart -- and a last-arrival ticket (__threadfence() + an integer
// atomic per block, right behind a convolution that had filled the L2) for the cross-group part: 45 us per launch against 6.5 us for
// this form.  The parts that need every group -- running statistics, dgamma / dbeta -- ride in the apply kernels, which run after this
// one anyway.)
// forward: part[tile][3][C] = (shift, sum (z - shift), sum (z - shift)^2) -> mean / biased var / invstd of group blockIdx.y
// (bn_finalize_mg_kernel's arithmetic on sums re-based to the group's first tile's shift)
// tab (codeps_bn_finalize_tiles_table): the [image][mean | invstd | gamma | beta][C] table of a BatchNorm whose apply pass is left to its consumer's
// operand load, written for the group's images, + the running-statistics updates of a ONE-group launch (bn_apply_mg_kernel's expressions)
struct BnTableOut { float* table; const float* gamma; const float* beta; float* run_mean; float* run_var; float momentum; int c_real; };
__global__ __launch_bounds__(1024) void bn_finalize_tiles_kernel(const float* __restrict__ part, TileSegs sg, ImgGroups ig, int C, float eps,
                                                                 float* __restrict__ mean, float* __restrict__ var, float* __restrict__ invstd,
                                                                 BnTableOut tab) {
    const int g = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const bool cin = c < C;
    const int t0 = sg.tile0[0] + ig.b0[g] * sg.tpi[0], nt = (ig.b1[g] - ig.b0[g]) * sg.tpi[0];
    const float rows = (float)sg.rows[0];
    float s1 = 0.f, s2 = 0.f, sh0 = 0.f;
    if (cin) {
        sh0 = part[(size_t)t0 * 3 * C + c];
#pragma unroll 4
        for (int t = w; t < nt; t += 16) {
            const float* q = part + (size_t)(t0 + t) * 3 * C + c;
            const float d = q[0] - sh0, a = q[C], b = q[2 * (size_t)C];
            s1 += a + rows * d;
            s2 += b + 2.f * d * a + rows * d * d;
        }
    }
    __shared__ float red[2][16][64];
    red[0][w][threadIdx.x & 63] = s1; red[1][w][threadIdx.x & 63] = s2;
    __syncthreads();
    if (w == 0 && cin) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { a += red[0][k][threadIdx.x]; b += red[1][k][threadIdx.x]; }
        const float n = rows * (float)nt;
        const float d = a / n;
        const float v = fmaxf(b / n - d * d, 0.f);
        mean[g * C + c] = sh0 + d; var[g * C + c] = v; invstd[g * C + c] = rsqrtf(v + eps);
        if (tab.table) {
            const float ga = tab.gamma[c], be = tab.beta[c];
            for (int b = ig.b0[g]; b < ig.b1[g]; ++b) {
                float* t = tab.table + (size_t)b * 4 * C + c;
                t[0] = sh0 + d; t[C] = rsqrtf(v + eps); t[2 * C] = ga; t[3 * C] = be;
            }
            if (tab.run_mean && c < tab.c_real) {
                tab.run_mean[c] = tab.run_mean[c] * (1.f - tab.momentum) + tab.momentum * (sh0 + d);
                tab.run_var[c] = tab.run_var[c] * (1.f - tab.momentum) + tab.momentum * (v * (n / fmaxf(n - 1.f, 1.f)));
            }
        }
    }
}

// backward: part[tile][2][C] = (sum g, sum g * xhat) over every segment's tiles of group blockIdx.y -> sums[g][2][C]
__global__ __launch_bounds__(1024) void bn_bwd_finish_tiles_kernel(const float* __restrict__ part, TileSegs sg, ImgGroups ig, int C,
                                                                   float* __restrict__ sums) {
    const int g = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const bool cin = c < C;
    float s1 = 0.f, s2 = 0.f;
    if (cin)
        for (int k = 0; k < sg.n; ++k) {
            const int t0 = sg.tile0[k] + ig.b0[g] * sg.tpi[k], nt = (ig.b1[g] - ig.b0[g]) * sg.tpi[k];
#pragma unroll 4
            for (int t = w; t < nt; t += 16) {
                const float* q = part + (size_t)(t0 + t) * 2 * C + c;
                s1 += q[0]; s2 += q[C];
            }
        }
    __shared__ float red[2][16][64];
    red[0][w][threadIdx.x & 63] = s1; red[1][w][threadIdx.x & 63] = s2;
    __syncthreads();
    if (w == 0 && cin) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { a += red[0][k][threadIdx.x]; b += red[1][k][threadIdx.x]; }
        sums[(size_t)g * 2 * C + c] = a; sums[(size_t)g * 2 * C + C + c] = b;
    }
}

__device__ __forceinline__ int bn_group_of(const BnGroups& gr, long long row) {
    int g = 0;
#pragma unroll
    for (int i = 1; i < 8; ++i) g += (i < gr.n && row >= gr.row0[i]) ? 1 : 0;
    return g;
}

// running statistics of a launch whose per-group (mean, biased var) come from codeps_bn_finalize_tiles: the momentum updates, group
// after group (bn_finalize_mg_kernel's expressions), by the first block of the apply pass
struct BnRunning { float* mean; float* var; const float* batch_var; float momentum; int c_real; };

__global__ __launch_bounds__(256) void bn_apply_mg_kernel(const float* __restrict__ x, BnGroups gr, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ residual, int act,
                                                          long long M, int C, float* __restrict__ y, BnRunning run) {
    if (run.mean && blockIdx.x == 0)
        for (int c = threadIdx.x; c < run.c_real; c += 256) {
            float rm = run.mean[c], rv = run.var[c];
            for (int g = 0; g < gr.n; ++g) {
                const float n = (float)gr.rows[g];
                rm = rm * (1.f - run.momentum) + run.momentum * mean[g * C + c];
                rv = rv * (1.f - run.momentum) + run.momentum * (run.batch_var[g * C + c] * (n / fmaxf(n - 1.f, 1.f)));
            }
            run.mean[c] = rm; run.var[c] = rv;
        }
    const int G = C / 4;
    const long long total = M * G;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g4 = (int)(i % G);
        const long long row = i / G;
        const int grp = bn_group_of(gr, row);
        const size_t o = (size_t)row * C + g4 * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + o);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + grp * C + g4 * 4), is = *reinterpret_cast<const f32x4*>(invstd + grp * C + g4 * 4);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + g4 * 4), be = *reinterpret_cast<const f32x4*>(beta + g4 * 4);
        f32x4 r = (v - mu) * is * ga + be;
        if (residual) r += *reinterpret_cast<const f32x4*>(residual + o);
        if (act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.f ? r[e] : 0.f;
        }
        *reinterpret_cast<f32x4*>(y + o) = r;
    }
}

// Gradient source "through a 3 x 3 / stride-2 / pad-1 max-pool" (the stem: bn1 -> relu -> maxpool, models/resnet_encoder.py:118-121): the
// BatchNorm backward kernels then gather dy of a pixel from the pooled gradient and the arg-max bytes of the (up to four) windows
// that contain it -- exactly maxpool_bwd_kernel's sum, in its order -- instead of reading a materialised full-resolution dy.
struct PoolSrc { const float* dp; const uint8_t* arg; int H, W, Ho, Wo; float inv_hw, inv_w; };
// n / d for 0 <= n < 2^24 * d-ish through a float reciprocal + one correction step (the kernels below divide per element; the
// hardware has no integer divide: ~35 instructions each)
__device__ __forceinline__ int fast_div(long long n, long long d, float inv_d) {
    if (n >> 23) return (int)(n / d);                             // (beyond the float's exact integers: the plain divide)
    int q = (int)((float)n * inv_d);
    long long r = n - (long long)q * d;
    q += r >= d ? 1 : 0;
    q -= r < 0 ? 1 : 0;
    return q;
}
// branch-free: the (up to) 2 x 2 windows that contain pixel (iy, ix) -- an even coordinate lies in one window (tap 1), an odd one in
// two (taps 2 and 0) -- are all loaded (clamped addresses), masked afterwards, and added in maxpool_bwd_kernel's order (oy, then ox)
__device__ __forceinline__ f32x4 pool_grad(const PoolSrc& ps, long long row, int C, int g4) {
    const long long hw = (long long)ps.H * ps.W;
    const int b = fast_div(row, hw, ps.inv_hw);
    const int rem = (int)(row - (long long)b * hw), iy = fast_div(rem, ps.W, ps.inv_w), ix = rem - iy * ps.W;
    const int oy0 = iy >> 1, ox0 = ix >> 1;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    uint32_t pk[2][2];
    f32x4 gd[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int oy = min(oy0 + dy, ps.Ho - 1), ox = min(ox0 + dx, ps.Wo - 1);
            const size_t o = (((size_t)b * ps.Ho + oy) * ps.Wo + ox) * C + g4 * 4;
            pk[dy][dx] = *reinterpret_cast<const uint32_t*>(ps.arg + o);
            gd[dy][dx] = *reinterpret_cast<const f32x4*>(ps.dp + o);
        }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const bool in = (dy == 0 || ((iy & 1) && oy0 + 1 < ps.Ho)) && (dx == 0 || ((ix & 1) && ox0 + 1 < ps.Wo));
            const uint32_t me = (uint32_t)((iy - ((oy0 + dy) * 2 - 1)) * 3 + ix - ((ox0 + dx) * 2 - 1));
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (in && ((pk[dy][dx] >> (8 * e)) & 0xFF) == me) a[e] += gd[dy][dx][e];
        }
    return a;
}

// backward partials per group: (sum g, sum g*xhat) with g = dy * relu'(y): part[g][blk][2][C]
// y == nullptr with act == 1: the ReLU mask is recomputed as (x - mean) * invstd * gamma + beta > 0 (the forward expression;
// only valid without a residual) instead of reading the saved output -- one tensor less per pass
template <bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_reduce_mg_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                               const float* __restrict__ x, BnGroups gr, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int act, int C, int nblk,
                                                               float* __restrict__ part, PoolSrc ps) {
    const int g = blockIdx.z;
    const int Gall = C / 4, G = Gall < 256 ? Gall : 256;
    const int tl = threadIdx.x % G, tr = threadIdx.x / G, TR = 256 / G;
    const int tg = blockIdx.y * G + tl;
    const bool cok = tg < Gall && tr < TR;
    const long long M = gr.rows[g], rpb = (M + nblk - 1) / nblk;
    const long long r0 = gr.row0[g] + (long long)blockIdx.x * rpb;
    const long long r1 = min(r0 + rpb, gr.row0[g] + M);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (cok) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + g * C + tg * 4), is = *reinterpret_cast<const f32x4*>(invstd + g * C + tg * 4);
        f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
        if (!y && act == 1) { ga = *reinterpret_cast<const f32x4*>(gamma + tg * 4); be = *reinterpret_cast<const f32x4*>(beta + tg * 4); }
        // four rows per trip: every load of the trip is issued before the first use and the body has no branch (rows past the
        // range re-read the last row with weight 0; `y` is block-uniform: two copies of the loop) -- as first written the loop
        // waited for each pair of loads before issuing the next (2.8 TB/s)
        auto body = [&](auto has_y) {
            for (long long r = r0 + tr; r < r1; r += 4 * TR) {
                f32x4 gv[4], xv[4], yv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long ru = min(r + (long long)u * TR, r1 - 1);
                    const size_t o = (size_t)ru * C + tg * 4;
                    gv[u] = POOL ? pool_grad(ps, ru, C, tg) : *reinterpret_cast<const f32x4*>(dy + o);
                    xv[u] = *reinterpret_cast<const f32x4*>(x + o);
                    if constexpr (decltype(has_y)::value) yv[u] = *reinterpret_cast<const f32x4*>(y + o);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float wgt = (r + (long long)u * TR < r1) ? 1.f : 0.f;
                    const f32x4 xh = (xv[u] - mu) * is;
                    f32x4 yy;
                    if constexpr (decltype(has_y)::value) yy = yv[u]; else yy = xh * ga + be;
                    f32x4 gg = gv[u] * wgt;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gg[e] = (act != 1 || yy[e] > 0.f) ? gg[e] : 0.f;
                    s += gg; q += gg * xh;
                }
            }
        };
        if (r0 < r1) { if (y) body(std::true_type{}); else body(std::false_type{}); }
    }
    __shared__ f32x4 red[2][256];
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < G && blockIdx.y * G + threadIdx.x < Gall) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < TR; ++k) { a += red[0][k * G + threadIdx.x]; b += red[1][k * G + threadIdx.x]; }
        float* dst = part + (((size_t)g * nblk + blockIdx.x) * 2) * C + (blockIdx.y * G + threadIdx.x) * 4;
        *reinterpret_cast<f32x4*>(dst) = a;
        *reinterpret_cast<f32x4*>(dst + C) = b;
    }
}

// finish backward partials: sums[g][2][C] per group, dgamma / dbeta (= / +=) summed over groups
__global__ __launch_bounds__(256) void bn_bwd_finish_mg_kernel(const float* __restrict__ part, BnGroups gr, int C, int c_real, int nblk,
                                                               float* __restrict__ sums, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int accumulate) {
    const int c = blockIdx.x * 4 + threadIdx.x / 64, sub = threadIdx.x % 64;
    const bool cin = c < C;
    float tg = 0.f, tb = 0.f;
    for (int g = 0; g < gr.n; ++g) {
        float s1 = 0.f, s2 = 0.f;
        if (cin) {
            const float* p = part + (size_t)g * nblk * 2 * C + c;
            for (int b = sub; b < nblk; b += 64) { s1 += p[(size_t)b * 2 * C]; s2 += p[(size_t)b * 2 * C + C]; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (!cin || sub != 0) continue;
        sums[(size_t)g * 2 * C + c] = s1; sums[(size_t)g * 2 * C + C + c] = s2;
        tb += s1; tg += s2;
    }
    if (cin && sub == 0 && c < c_real) {
        dgamma[c] = accumulate ? dgamma[c] + tg : tg;
        dbeta[c] = accumulate ? dbeta[c] + tb : tb;
    }
}

// affine gradients of a backward whose per-group sums come from codeps_bn_bwd_finish_tiles (local_sums: THIS rank's, before any all-reduce)
struct BnAffineGrad { float* dgamma; float* dbeta; const float* local_sums; int accumulate, c_real; };

template <bool POOL>
__global__ __launch_bounds__(256) void bn_bwd_apply_mg_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              const float* __restrict__ x, BnGroups gr, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ sums, int act,
                                                              long long M, int C, float* __restrict__ dx, float* __restrict__ dres, PoolSrc ps,
                                                              BnAffineGrad ag) {
    if (ag.dgamma && blockIdx.x == 0)                             // dgamma / dbeta from this rank's per-group sums, in group order (bn_bwd_finish_mg_kernel)
        for (int c = threadIdx.x; c < ag.c_real; c += 256) {
            float tb = 0.f, tg = 0.f;
            for (int g = 0; g < gr.n; ++g) { tb += ag.local_sums[(size_t)g * 2 * C + c]; tg += ag.local_sums[(size_t)g * 2 * C + C + c]; }
            ag.dgamma[c] = ag.accumulate ? ag.dgamma[c] + tg : tg;
            ag.dbeta[c] = ag.accumulate ? ag.dbeta[c] + tb : tb;
        }
    const int G = C / 4;
    const long long total = M * G;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g4 = (int)(i % G);
        const long long row = i / G;
        const int grp = bn_group_of(gr, row);
        const size_t o = (size_t)row * C + g4 * 4;
        f32x4 g = POOL ? pool_grad(ps, row, C, g4) : *reinterpret_cast<const f32x4*>(dy + o);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + grp * C + g4 * 4), is = *reinterpret_cast<const f32x4*>(invstd + grp * C + g4 * 4);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + g4 * 4);
        const f32x4 xh = (*reinterpret_cast<const f32x4*>(x + o) - mu) * is;
        if (act == 1) {
            const f32x4 yy = y ? *reinterpret_cast<const f32x4*>(y + o) : xh * ga + *reinterpret_cast<const f32x4*>(beta + g4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = yy[e] > 0.f ? g[e] : 0.f;
        }
        if (dres) *reinterpret_cast<f32x4*>(dres + o) = g;
        const f32x4 sg = *reinterpret_cast<const f32x4*>(sums + (size_t)grp * 2 * C + g4 * 4);
        const f32x4 sgx = *reinterpret_cast<const f32x4*>(sums + (size_t)grp * 2 * C + C + g4 * 4);
        const float ic = gr.inv_count[grp];
        *reinterpret_cast<f32x4*>(dx + o) = ga * is * (g - sg * ic - xh * sgx * ic);
    }
}

// bn_bwd_apply_mg_kernel<POOL = true> for even H and W: one thread per 2 x 2 block of pixels and channel quad.  The block's four
// windows (rows k, k + 1; columns j, j + 1 of the pooled map) are loaded ONCE and serve its four pixels (3 loads per pixel instead of 9);
// every pixel's gradient is summed in pool_grad's order, so the result is bit-identical.
__global__ __launch_bounds__(256) void bn_bwd_apply_pool2x2_kernel(const float* __restrict__ x, BnGroups gr, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, const float* __restrict__ sums, int B, int C,
                                                                   float* __restrict__ dx, PoolSrc ps) {
    const int G = C / 4, H2 = ps.H / 2, W2 = ps.W / 2;
    const long long total = (long long)B * H2 * W2 * G;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g4 = (int)(i % G);
        long long t = i / G;
        const int j = (int)(t % W2); t /= W2;
        const int k = (int)(t % H2);
        const int b = (int)(t / H2);
        const int grp = bn_group_of(gr, (long long)b * ps.H * ps.W);
        uint32_t pk[2][2];
        f32x4 gd[2][2], xv[2][2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dxx = 0; dxx < 2; ++dxx) {
                const int oy = min(k + dy, ps.Ho - 1), ox = min(j + dxx, ps.Wo - 1);
                const size_t o = (((size_t)b * ps.Ho + oy) * ps.Wo + ox) * C + g4 * 4;
                pk[dy][dxx] = *reinterpret_cast<const uint32_t*>(ps.arg + o);
                gd[dy][dxx] = *reinterpret_cast<const f32x4*>(ps.dp + o);
                xv[dy][dxx] = *reinterpret_cast<const f32x4*>(x + (((size_t)b * ps.H + 2 * k + dy) * ps.W + 2 * j + dxx) * C + g4 * 4);
            }
        const bool oky = k + 1 < ps.Ho, okx = j + 1 < ps.Wo;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + grp * C + g4 * 4), is = *reinterpret_cast<const f32x4*>(invstd + grp * C + g4 * 4);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + g4 * 4), be = *reinterpret_cast<const f32x4*>(beta + g4 * 4);
        const f32x4 sg = *reinterpret_cast<const f32x4*>(sums + (size_t)grp * 2 * C + g4 * 4);
        const f32x4 sgx = *reinterpret_cast<const f32x4*>(sums + (size_t)grp * 2 * C + C + g4 * 4);
        const float ic = gr.inv_count[grp];
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                // pixel (2 k + py, 2 j + px): windows (k + dy, j + dx) with dy <= py, dx <= px; its tap there = (py + 1 - 2 dy, px + 1 - 2 dx)
                f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int dy = 0; dy <= py; ++dy)
#pragma unroll
                    for (int dxx = 0; dxx <= px; ++dxx) {
                        const bool in = (dy == 0 || oky) && (dxx == 0 || okx);
                        const uint32_t me = (uint32_t)((py + 1 - 2 * dy) * 3 + px + 1 - 2 * dxx);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (in && ((pk[dy][dxx] >> (8 * e)) & 0xFF) == me) g[e] += gd[dy][dxx][e];
                    }
                const f32x4 xh = (xv[py][px] - mu) * is;
                const f32x4 yy = xh * ga + be;
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = yy[e] > 0.f ? g[e] : 0.f;
                *reinterpret_cast<f32x4*>(dx + (((size_t)b * ps.H + 2 * k + py) * ps.W + 2 * j + px) * C + g4 * 4) = ga * is * (g - sg * ic - xh * sgx * ic);
            }
    }
}

// ---------------------------------------------------------------------------------- max-pool 3x3 / stride 2 / pad 1
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, int B, int H, int W, int C, int Ho, int Wo,
                                                          float* __restrict__ y, uint8_t* __restrict__ arg) {
    const int G = C / 4;
    const long long total = (long long)B * Ho * Wo * G;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % G);
        long long pix = i / G;
        const int ox = (int)(pix % Wo); pix /= Wo;
        const int oy = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + iy) * W + ix) * C + g * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > best[e] || (v[e] != v[e])) { best[e] = v[e]; bi[e] = ky * 3 + kx; }
            }
        const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * C + g * 4;
        *reinterpret_cast<f32x4*>(y + o) = best;
        if (arg) *reinterpret_cast<uint32_t*>(arg + o) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    }
}

// relu(batch-norm(z)) and the 3 x 3 / stride-2 / pad-1 max-pool over it in one pass (the stem in train mode): the full-resolution
// activation is never written.  Same expression as bn_apply_mg_kernel, same comparison and tie rule as maxpool_fwd_kernel.
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const float* __restrict__ z, BnGroups gr, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int B, int H, int W, int C, int Ho, int Wo,
                                                               float* __restrict__ y, uint8_t* __restrict__ arg, BnRunning run) {
    if (run.mean && blockIdx.x == 0)            // batch statistics from codeps_bn_finalize_tiles: the momentum updates ride here (as in bn_apply_mg_kernel)
        for (int c = threadIdx.x; c < run.c_real; c += 256) {
            float rm = run.mean[c], rv = run.var[c];
            for (int g = 0; g < gr.n; ++g) {
                const float n = (float)gr.rows[g];
                rm = rm * (1.f - run.momentum) + run.momentum * mean[g * C + c];
                rv = rv * (1.f - run.momentum) + run.momentum * (run.batch_var[g * C + c] * (n / fmaxf(n - 1.f, 1.f)));
            }
            run.mean[c] = rm; run.var[c] = rv;
        }
    const int G = C / 4;
    const long long total = (long long)B * Ho * Wo * G;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % G);
        long long pix = i / G;
        const int ox = (int)(pix % Wo); pix /= Wo;
        const int oy = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        const int grp = bn_group_of(gr, (long long)b * H * W);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + grp * C + g * 4), is = *reinterpret_cast<const f32x4*>(invstd + grp * C + g * 4);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + g * 4), be = *reinterpret_cast<const f32x4*>(beta + g * 4);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                f32x4 v = (*reinterpret_cast<const f32x4*>(z + (((size_t)b * H + iy) * W + ix) * C + g * 4) - mu) * is * ga + be;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = v[e] > 0.f ? v[e] : 0.f;
                    if (v[e] > best[e] || (v[e] != v[e])) { best[e] = v[e]; bi[e] = ky * 3 + kx; }
                }
            }
        const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * C + g * 4;
        *reinterpret_cast<f32x4*>(y + o) = best;
        *reinterpret_cast<uint32_t*>(arg + o) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ arg, int B,
                                                          int H, int W, int C, int Ho, int Wo, float* __restrict__ dx) {
    const int G = C / 4;
    const long long total = (long long)B * H * W * G;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % G);
        long long pix = i / G;
        const int ix = (int)(pix % W); pix /= W;
        const int iy = (int)(pix % H);
        const int b = (int)(pix / H);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        // outputs whose window [2o-1, 2o+1] contains this pixel
        for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
            if (oy >= Ho) continue;
            const int ky = iy - (oy * 2 - 1);
            for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
                if (ox >= Wo) continue;
                const int kx = ix - (ox * 2 - 1);
                const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * C + g * 4;
                const uint32_t pk = *reinterpret_cast<const uint32_t*>(arg + o);
                const f32x4 gd = *reinterpret_cast<const f32x4*>(dy + o);
                const uint32_t me = (uint32_t)(ky * 3 + kx);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((pk >> (8 * e)) & 0xFF) == me) a[e] += gd[e];
            }
        }
        *reinterpret_cast<f32x4*>(dx + (((size_t)b * H + iy) * W + ix) * C + g * 4) = a;
    }
}

// ---------------------------------------------------------------------------------- depthwise 5x5, stride 1, pad 2
// weights repacked [25][C].  mode 0: forward (optionally affine + ReLU), mode 1: data gradient (flipped taps)
__global__ __launch_bounds__(256) void dwconv5_kernel(const float* __restrict__ x, const float* __restrict__ w25c,
                                                      const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                      int B, int H, int W, int C, int flip, float* __restrict__ y) {
    // one thread per 4-channel group and RS x CS block of output pixels: the (RS + 4) x (CS + 4) input patch is loaded once
    // (48 loads + 25 weights for 8 outputs); branch-free: out-of-image taps load a clamped address and are zeroed afterwards,
    // so the loads of a patch column are issued back to back
    constexpr int RS = 4, CS = 2;
    const int G = C / 4, strips = (H + RS - 1) / RS, cols = (W + CS - 1) / CS;
    const long long total = (long long)B * strips * cols * G;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % G);
        long long r = i / G;
        const int ox0 = (int)(r % cols) * CS; r /= cols;
        const int oy0 = (int)(r % strips) * RS;
        const int b = (int)(r / strips);
        f32x4 a[RS][CS];
#pragma unroll
        for (int o = 0; o < RS; ++o)
#pragma unroll
            for (int q = 0; q < CS; ++q) a[o][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 wprev[5];                                         // weights of tap column jx - 1 (for the second output column)
#pragma unroll
        for (int jx = 0; jx < CS + 4; ++jx) {                   // input patch column ox0 - 2 + jx
            const int ix = ox0 - 2 + jx;
            const bool xok = (unsigned)ix < (unsigned)W;
            const int ixc = min(max(ix, 0), W - 1);
            f32x4 wk[5];
            if (jx < 5) {
#pragma unroll
                for (int ky = 0; ky < 5; ++ky)
                    wk[ky] = *reinterpret_cast<const f32x4*>(w25c + (size_t)(flip ? (4 - ky) * 5 + (4 - jx) : ky * 5 + jx) * C + g * 4);
            }
            f32x4 v[RS + 4];
#pragma unroll
            for (int ry = 0; ry < RS + 4; ++ry)
                v[ry] = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + min(max(oy0 - 2 + ry, 0), H - 1)) * W + ixc) * C + g * 4);
#pragma unroll
            for (int ry = 0; ry < RS + 4; ++ry) {
                const int iy = oy0 - 2 + ry;
                const f32x4 vv = (xok && (unsigned)iy < (unsigned)H) ? v[ry] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 5; ++ky) {
                    const int o = ry - ky;                      // output row (oy0 + o) reads input row ry through tap ky
                    if (o < 0 || o >= RS) continue;
                    if (jx < 5) a[o][0] += vv * wk[ky];         // output column ox0:     tap kx = jx
                    if (jx >= 1) a[o][1] += vv * wprev[ky];     // output column ox0 + 1: tap kx = jx - 1
                }
            }
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) wprev[ky] = wk[ky];
        }
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (scale) sc = *reinterpret_cast<const f32x4*>(scale + g * 4);
        if (shift) sh = *reinterpret_cast<const f32x4*>(shift + g * 4);
#pragma unroll
        for (int o = 0; o < RS; ++o)
#pragma unroll
            for (int q = 0; q < CS; ++q) {
                if (oy0 + o >= H || ox0 + q >= W) continue;
                f32x4 v = a[o][q] * sc + sh;
                if (act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                *reinterpret_cast<f32x4*>(y + (((size_t)b * H + oy0 + o) * W + ox0 + q) * C + g * 4) = v;
            }
    }
}

// depthwise weight gradient partials: part[blk][25][C]
__global__ __launch_bounds__(256) void dwconv5_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz, int B, int H,
                                                            int W, int C, float* __restrict__ part, int items_per_blk) {
    // dW[tap][c] = sum_pixels dz[p][c] * x[p + delta_tap][c].  One pass over the data: a thread owns a 4-channel group and
    // walks items = (image, strip of RS output rows, column); per item the RS dz values and, per tap column, the RS + 4
    // x values are loaded once and feed all 25 tap accumulators held in registers.  Block partials part[blk][25][C].
    constexpr int RS = 4;
    const int Gall = C / 4, G = Gall < 256 ? Gall : 256;          // blockIdx.y tiles the channel groups
    const int tl = threadIdx.x % G, tr = threadIdx.x / G, TR = 256 / G;
    const int tg = blockIdx.y * G + tl;
    const bool cok = tg < Gall && tr < TR;
    const int strips = (H + RS - 1) / RS;
    const long long items = (long long)B * strips * W;
    const long long i0 = (long long)blockIdx.x * items_per_blk;
    const long long i1 = i0 + items_per_blk < items ? i0 + items_per_blk : items;
    f32x4 acc[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (cok)
        for (long long it = i0 + tr; it < i1; it += TR) {
            const int ox = (int)(it % W);
            const int oy0 = (int)((it / W) % strips) * RS;
            const int b = (int)(it / ((long long)W * strips));
            f32x4 g[RS];
#pragma unroll
            for (int o = 0; o < RS; ++o) g[o] = *reinterpret_cast<const f32x4*>(dz + (((size_t)b * H + min(oy0 + o, H - 1)) * W + ox) * C + tg * 4);
#pragma unroll
            for (int o = 0; o < RS; ++o) if (oy0 + o >= H) g[o] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
                const int ix = ox - 2 + kx;
                const bool xok = (unsigned)ix < (unsigned)W;
                const int ixc = min(max(ix, 0), W - 1);
                f32x4 v[RS + 4];
#pragma unroll
                for (int ry = 0; ry < RS + 4; ++ry)          // unconditional loads from clamped addresses, zeroed below
                    v[ry] = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + min(max(oy0 - 2 + ry, 0), H - 1)) * W + ixc) * C + tg * 4);
#pragma unroll
                for (int ry = 0; ry < RS + 4; ++ry) {
                    const int iy = oy0 - 2 + ry;
                    const f32x4 vv = (xok && (unsigned)iy < (unsigned)H) ? v[ry] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ky = 0; ky < 5; ++ky) {
                        const int o = ry - ky;                  // output row oy0 + o reads input row ry through tap ky
                        if (o >= 0 && o < RS) acc[ky * 5 + kx] += g[o] * vv;
                    }
                }
            }
        }
    __shared__ f32x4 red[256];
    for (int tap = 0; tap < 25; ++tap) {
        __syncthreads();
        red[threadIdx.x] = acc[tap];
        __syncthreads();
        if (threadIdx.x < G && blockIdx.y * G + threadIdx.x < Gall) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < TR; ++k) t += red[k * G + threadIdx.x];
            *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 25 + tap) * C + (blockIdx.y * G + threadIdx.x) * 4) = t;
        }
    }
}

// ---------------------------------------------------------------------------------- depthwise 5x5 on LDS tiles (round 6)
// dwconv5_kernel gathers every thread's (RS + 4) x (CS + 4) patch from global memory: 6 loads per output quad (11 in the weight gradient), and
// a block's footprint (82 KB at 128 channels) does not stay in the 32 KB L1 -- the loads are served by the L2 and the kernel moves 2.4 TB/s of
// algorithmic bytes (tools/dwconv_rate.py).  Here a block stages ONE halo tile of 20 x 20 pixels x 32 channels in LDS (every input element
// fetched once per block: 1.56x with the halo, 128-byte segments) and the threads -- (channel quad, column pair, 4-row strip) of a 16 x 16
// output tile -- run dwconv5_kernel's register loops on LDS reads.  Same products, same summation order per output: bit-identical results.
constexpr int DWT = 16, DWH = DWT + 4, DWCG = 32, DWPITCH = DWCG + 4;        // tile, halo tile, channels per block, floats per staged pixel
// blocks with equal blockIdx.x % 8 share an XCD (its L2): they take CONSECUTIVE tiles, so that the halos neighbouring tiles share are fetched from
// HBM once (conv.hip: xcd_remap)
__device__ __forceinline__ int dw_xcd_remap(int bid, int nblk) {
    const int q = nblk / 8, r = nblk % 8, x = bid % 8, i = bid / 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
constexpr int DWLOADS = (DWH * DWH * (DWCG / 4) + 255) / 256;

template <int U0, int U1>
__device__ __forceinline__ void dw_stage(const float* __restrict__ x, int b, int ty0, int tx0, int c0, int H, int W, int C, f32x4 (&r)[U1 - U0]) {
#pragma unroll
    for (int u = U0; u < U1; ++u) {
        const int i = threadIdx.x + 256 * u, pos = min(i, DWH * DWH * (DWCG / 4) - 1) / (DWCG / 4), q = i % (DWCG / 4);
        const int iy = ty0 - 2 + pos / DWH, ix = tx0 - 2 + pos % DWH, c = c0 + q * 4;
        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && c < C;
        // branch-free: a clamped address is loaded and zeroed afterwards, so the loads of a part are issued back to back
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)) * C + min(c, C - 4));
        r[u - U0] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
template <int U0, int U1>
__device__ __forceinline__ void dw_store(float* xs, const f32x4 (&r)[U1 - U0]) {
#pragma unroll
    for (int u = U0; u < U1; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i < DWH * DWH * (DWCG / 4)) *reinterpret_cast<f32x4*>(xs + (i / (DWCG / 4)) * DWPITCH + (i % (DWCG / 4)) * 4) = r[u - U0];
    }
}

__global__ __launch_bounds__(256, 2) void dwconv5_tile_kernel(const float* __restrict__ x, const float* __restrict__ w25c, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int act, int B, int H, int W, int C, int flip,
                                                           float* __restrict__ y, int tiles_x, int tiles_y) {
    constexpr int RS = 4, CS = 2;
    __shared__ __attribute__((aligned(16))) float xs[DWH * DWH * DWPITCH];
    const int tile = dw_xcd_remap(blockIdx.x, gridDim.x), c0 = blockIdx.y * DWCG;
    const int b = tile / (tiles_x * tiles_y), ty0 = ((tile / tiles_x) % tiles_y) * DWT, tx0 = (tile % tiles_x) * DWT;
    {
        f32x4 r[DWLOADS];
        dw_stage<0, DWLOADS>(x, b, ty0, tx0, c0, H, W, C, r);
        dw_store<0, DWLOADS>(xs, r);
    }
    __syncthreads();
    const int q = threadIdx.x % 8, cp = (threadIdx.x / 8) % 8, rs = threadIdx.x / 64;
    const int c = c0 + q * 4;
    if (c >= C) return;
    const int oy0 = ty0 + rs * RS, ox0 = tx0 + cp * CS;
    f32x4 a[RS][CS];
#pragma unroll
    for (int o = 0; o < RS; ++o)
#pragma unroll
        for (int k = 0; k < CS; ++k) a[o][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 wprev[5];
#pragma unroll
    for (int jx = 0; jx < CS + 4; ++jx) {                       // (dwconv5_kernel's loops: the zero padding is in the staged tile)
        f32x4 wk[5];
        if (jx < 5) {
#pragma unroll
            for (int ky = 0; ky < 5; ++ky)
                wk[ky] = *reinterpret_cast<const f32x4*>(w25c + (size_t)(flip ? (4 - ky) * 5 + (4 - jx) : ky * 5 + jx) * C + c);
        }
        f32x4 v[RS + 4];
#pragma unroll
        for (int ry = 0; ry < RS + 4; ++ry) v[ry] = *reinterpret_cast<const f32x4*>(xs + ((rs * RS + ry) * DWH + cp * CS + jx) * DWPITCH + q * 4);
#pragma unroll
        for (int ry = 0; ry < RS + 4; ++ry) {
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const int o = ry - ky;
                if (o < 0 || o >= RS) continue;
                if (jx < 5) a[o][0] += v[ry] * wk[ky];
                if (jx >= 1) a[o][1] += v[ry] * wprev[ky];
            }
        }
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) wprev[ky] = wk[ky];
        // one patch column in registers at a time: the sums pass through an empty asm, so the products of this column are placed here (the
        // instruction selector would otherwise put all 48 LDS reads in front of all the arithmetic: 60 spilled registers)
#pragma unroll
        for (int o = 0; o < RS; ++o) asm volatile("" : "+v"(a[o][0]), "+v"(a[o][1]));
    }
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (scale) sc = *reinterpret_cast<const f32x4*>(scale + c);
    if (shift) sh = *reinterpret_cast<const f32x4*>(shift + c);
#pragma unroll
    for (int o = 0; o < RS; ++o)
#pragma unroll
        for (int k = 0; k < CS; ++k) {
            if (oy0 + o >= H || ox0 + k >= W) continue;
            f32x4 v = a[o][k] * sc + sh;
            if (act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(y + (((size_t)b * H + oy0 + o) * W + ox0 + k) * C + c) = v;
        }
}

// ... and the weight gradient: a block walks `tiles_per_blk` tiles of one 32-channel group with the 25 tap sums of every thread in registers,
// then the 32 threads of a channel quad meet in LDS in a fixed order: part[blk][25][C] (summed over blocks by colsum_finish_kernel)
__global__ __launch_bounds__(256, 2) void dwconv5_wgrad_tile_kernel(const float* __restrict__ x, const float* __restrict__ dz, int B, int H, int W, int C,
                                                                 float* __restrict__ part, int tiles_total, int tiles_per_blk, int tiles_x, int tiles_y) {
    constexpr int RS = 4, CS = 2;
    __shared__ __attribute__((aligned(16))) float xs[DWH * DWH * DWPITCH];
    const int c0 = blockIdx.y * DWCG;
    const int q = threadIdx.x % 8, cp = (threadIdx.x / 8) % 8, rs = threadIdx.x / 64;
    const int c = c0 + q * 4;
    const bool cok = c < C;
    f32x4 acc[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int t_beg = dw_xcd_remap(blockIdx.x, gridDim.x) * tiles_per_blk, t_end = min(t_beg + tiles_per_blk, tiles_total);
    for (int tile = t_beg; tile < t_end; ++tile) {
        const int b = tile / (tiles_x * tiles_y), ty0 = ((tile / tiles_x) % tiles_y) * DWT, tx0 = (tile % tiles_x) * DWT;
        constexpr int T1 = (DWLOADS + 2) / 3, T2 = 2 * T1 < DWLOADS ? 2 * T1 : DWLOADS;     // the tile in three parts: a part's registers are the previous part's
        const int oy0 = ty0 + rs * RS, ox0 = tx0 + cp * CS;
        f32x4 g[RS][CS];
        {
            f32x4 r[T1];
            dw_stage<0, T1>(x, b, ty0, tx0, c0, H, W, C, r);
            __syncthreads();                                    // the previous tile's reads are done
            dw_store<0, T1>(xs, r);
        }
        {
            f32x4 r[T2 - T1];
            dw_stage<T1, T2>(x, b, ty0, tx0, c0, H, W, C, r);
            dw_store<T1, T2>(xs, r);
        }
        {
            f32x4 r[DWLOADS - T2];
            dw_stage<T2, DWLOADS>(x, b, ty0, tx0, c0, H, W, C, r);
#pragma unroll
            for (int o = 0; o < RS; ++o)
#pragma unroll
                for (int k = 0; k < CS; ++k)
                    g[o][k] = (cok && oy0 + o < H && ox0 + k < W) ? *reinterpret_cast<const f32x4*>(dz + (((size_t)b * H + oy0 + o) * W + ox0 + k) * C + c)
                                                                 : f32x4{0.f, 0.f, 0.f, 0.f};
            dw_store<T2, DWLOADS>(xs, r);
        }
        __syncthreads();
#pragma unroll
        for (int jx = 0; jx < CS + 4; ++jx) {
            asm volatile("" ::: "memory");                      // (this column's LDS reads stay behind the previous column's products)
            f32x4 v[RS + 4];
#pragma unroll
            for (int ry = 0; ry < RS + 4; ++ry) v[ry] = *reinterpret_cast<const f32x4*>(xs + ((rs * RS + ry) * DWH + cp * CS + jx) * DWPITCH + q * 4);
#pragma unroll
            for (int ry = 0; ry < RS + 4; ++ry)
#pragma unroll
                for (int ky = 0; ky < 5; ++ky) {
                    const int o = ry - ky;
                    if (o < 0 || o >= RS) continue;
                    if (jx < 5) acc[ky * 5 + jx] += g[o][0] * v[ry];
                    if (jx >= 1) acc[ky * 5 + jx - 1] += g[o][1] * v[ry];
                }
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {                    // (as in dwconv5_tile_kernel: this column's products stay with this column's reads)
                if (jx < 5) asm volatile("" : "+v"(acc[ky * 5 + (jx < 5 ? jx : 0)]));
                if (jx >= 1) asm volatile("" : "+v"(acc[ky * 5 + (jx >= 1 ? jx - 1 : 0)]));
            }
        }
    }
    // the 32 threads of a channel quad meet in LDS, five taps (one kernel row) per round: [tap][thread] quads, thread 8 t + q sums tap t of quad q
    f32x4* red = reinterpret_cast<f32x4*>(xs);                  // [5][256] quads = 20 KB of the 57 KB tile
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        __syncthreads();
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) red[kx * 256 + threadIdx.x] = acc[ky * 5 + kx];
        __syncthreads();
        if (threadIdx.x < 40) {
            const int kx = threadIdx.x / 8, qq = threadIdx.x % 8, cc = c0 + qq * 4;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int k = 0; k < 32; ++k) t += red[kx * 256 + k * 8 + qq];
            if (cc < C) *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 25 + ky * 5 + kx) * C + cc) = t;
        }
    }
}

// ---------------------------------------------------------------------------------- bilinear resize (NHWC)
__device__ __forceinline__ void bil_src(int align, float ratio, int dst, int in_size, int out_size, int& i0, int& i1, float& l0,
                                        float& l1) {
    float r;
    if (align) r = ratio * (float)dst;                          // ratio = (in-1)/(out-1)
    else {
        if (in_size == out_size) { i0 = i1 = dst; l0 = 1.f; l1 = 0.f; return; }
        r = ratio * ((float)dst + 0.5f) - 0.5f;                 // ratio = in/out
        r = r < 0.f ? 0.f : r;
    }
    i0 = min((int)floorf(r), in_size - 1);
    l1 = fminf(fmaxf(r - (float)i0, 0.f), 1.f);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l0 = 1.f - l1;
}

// Block = (output row b * Ho + oy, a run of 256 (column, channel quad) items): the row's two source rows and their weights are block-uniform
// (until round 6 every thread split a 64-bit linear index with three 64-bit divisions: 20 launches, 0.34 ms of the adaptation step).
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, int B, int H, int W, int C, int Ho, int Wo,
                                                           int align, float rh, float rw, float* __restrict__ y, int Cy, int c_off) {
    const int G = C / 4;
    const int row = blockIdx.x, b = row / Ho, oy = row % Ho;
    int y0, y1; float ly0, ly1;
    bil_src(align, rh, oy, H, Ho, y0, y1, ly0, ly1);
    const float* r0 = x + ((size_t)b * H + y0) * W * C;
    const float* r1 = x + ((size_t)b * H + y1) * W * C;
    float* dst = y + (size_t)row * Wo * Cy + c_off;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < Wo * G; i += gridDim.y * 256) {
        const int ox = i / G, g = i - ox * G;
        int x0, x1; float lx0, lx1;
        bil_src(align, rw, ox, W, Wo, x0, x1, lx0, lx1);
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(r0 + (size_t)x0 * C + g * 4);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(r0 + (size_t)x1 * C + g * 4);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(r1 + (size_t)x0 * C + g * 4);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(r1 + (size_t)x1 * C + g * 4);
        const f32x4 r = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
        *reinterpret_cast<f32x4*>(dst + (size_t)ox * Cy + g * 4) = r;
    }
}

// transposed resize as a gather over INPUT pixels (deterministic): dx[i] = sum_o w(o,i) dy[o]
// Block = (input row b * H + iy, a run of 256 (column, channel quad) items): the output rows that touch the row and their weights are
// block-uniform; same sums in the same order as before round 6 (when every thread split a 64-bit linear index with three 64-bit divisions).
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, int B, int H, int W, int C, int Ho, int Wo,
                                                           int align, float rh, float rw, int Cy, int c_off,
                                                           float* __restrict__ dx) {
    const int G = C / 4;
    const float inv_rh = rh > 0.f ? 1.f / rh : (float)Ho, inv_rw = rw > 0.f ? 1.f / rw : (float)Wo;
    const int row = blockIdx.x, b = row / H, iy = row % H;
    // output rows that can touch input row iy: src(o) in (iy-1, iy+1)  =>  o in ((iy-1.5)/r, (iy+1.5)/r) (+-1 slack)
    const int oy_lo = max(0, (int)floorf(((float)iy - 1.5f) * inv_rh) - 1), oy_hi = min(Ho - 1, (int)ceilf(((float)iy + 1.5f) * inv_rh) + 1);
    for (int i = blockIdx.y * 256 + threadIdx.x; i < W * G; i += gridDim.y * 256) {
        const int ix = i / G, g = i - ix * G;
        const int ox_lo = max(0, (int)floorf(((float)ix - 1.5f) * inv_rw) - 1), ox_hi = min(Wo - 1, (int)ceilf(((float)ix + 1.5f) * inv_rw) + 1);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1; float ly0, ly1;
            bil_src(align, rh, oy, H, Ho, y0, y1, ly0, ly1);
            const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
            if (wy == 0.f) continue;
            const float* drow = dy + ((size_t)b * Ho + oy) * Wo * Cy + c_off + g * 4;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1; float lx0, lx1;
                bil_src(align, rw, ox, W, Wo, x0, x1, lx0, lx1);
                const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
                if (wx == 0.f) continue;
                a += (wy * wx) * *reinterpret_cast<const f32x4*>(drow + (size_t)ox * Cy);
            }
        }
        *reinterpret_cast<f32x4*>(dx + ((size_t)row * W + ix) * C + g * 4) = a;
    }
}

// ---------------------------------------------------------------------------------- GUDA fusion without the 240-channel concat
// SemanticHead's GUDA block (models/semantic_head.py:64-72) is conv3x3(cat_k bilinear_up(d_k)).  Both operators are linear and
// the up-sampling acts per channel, so   conv3x3(up(d_k))(p) = sum_t up(W_t . d_k)(p + delta_t):
// the channel contraction (W_t . d_k, a 1x1 conv to 9 x 16 tap-channels) is done at the LOW resolution of each decoder
// level (1/4 ... 1/64 of the pixels, on the MFMA conv engine) and only 16-channel tap planes are interpolated at full
// resolution here.  Zero padding of the 3x3 conv = taps whose position p + delta_t leaves the image contribute nothing.
struct GudaBranch { const float* y; int h, w; float rh, rw; };      // y: [B][h][w][144] = 9 taps x 16 channels
struct GudaP { int B, H, W, nb; GudaBranch br[3]; };

// One thread per (4-pixel strip along x, 4-channel group): the six positions x - 1 .. x + 4 that the strip's taps read fall
// into at most three low-res columns of a branch (ratio >= 2), so each (tap, source row) costs three 16-byte loads for the
// whole strip instead of two per pixel -- the kernel is bound by the L1 / texture-address path (108 loads per pixel before).
__global__ __launch_bounds__(256) void guda_gather_fwd_kernel(GudaP p, const float* __restrict__ base, float* __restrict__ out) {
    const int W4 = (p.W + 3) / 4;
    const long long total = (long long)p.B * p.H * W4 * 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i & 3);
        long long s = i >> 2;
        const int xs = (int)(s % W4) * 4; s /= W4;
        const int y = (int)(s % p.H);
        const int b = (int)(s / p.H);
        const int npx = min(4, p.W - xs);
        f32x4 a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            a[j] = j < npx ? *reinterpret_cast<const f32x4*>(base + ((((size_t)b * p.H + y) * p.W + xs + j) << 4) + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < p.nb; ++k) {
            const GudaBranch br = p.br[k];
            const float* yb = br.y + (size_t)b * br.h * br.w * 144 + g * 4;
            // bilinear sources of the six columns xs - 1 .. xs + 4 (clamped like a zero-padded tap never is: invalid ones get weight 0)
            int c0[6]; float l1[6]; bool ok[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int qx = xs + q - 1;
                ok[q] = qx >= 0 && qx < p.W;
                int i0, i1; float w0, w1;
                bil_src(0, br.rw, min(max(qx, 0), p.W - 1), br.w, p.W, i0, i1, w0, w1);
                c0[q] = i0;
                l1[q] = i1 > i0 ? w1 : 0.f;              // weight on column i0 + 1; the rest (incl. the clamped last column) on i0
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int qy = y + ky - 1;
                if (qy < 0 || qy >= p.H) continue;
                int y0, y1; float ly0, ly1;
                bil_src(0, br.rh, qy, br.h, p.H, y0, y1, ly0, ly1);
                const float* r0 = yb + (size_t)y0 * br.w * 144 + ky * 48;
                const float* r1 = yb + (size_t)y1 * br.w * 144 + ky * 48;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int cb = c0[kx];                // first low-res column of this tap's window (pixel 0 of the strip)
                    if (c0[kx + 3] + 1 - cb > 2) {        // ratio < 2 would need more than three columns: per-pixel path
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (!ok[j + kx]) continue;
                            const int x0 = c0[j + kx], x1 = min(x0 + 1, br.w - 1);
                            const float w1 = l1[j + kx], w0 = 1.f - w1;
                            a[j] += ly0 * (w0 * *reinterpret_cast<const f32x4*>(r0 + x0 * 144 + kx * 16) + w1 * *reinterpret_cast<const f32x4*>(r0 + x1 * 144 + kx * 16)) +
                                    ly1 * (w0 * *reinterpret_cast<const f32x4*>(r1 + x0 * 144 + kx * 16) + w1 * *reinterpret_cast<const f32x4*>(r1 + x1 * 144 + kx * 16));
                        }
                        continue;
                    }
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        const int col = min(cb + cc, br.w - 1);
                        const f32x4 v = ly0 * *reinterpret_cast<const f32x4*>(r0 + col * 144 + kx * 16) + ly1 * *reinterpret_cast<const f32x4*>(r1 + col * 144 + kx * 16);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int q = j + kx, d = cb + cc - c0[q];        // 0: the pixel's left column, 1: its right column
                            const float w = !ok[q] ? 0.f : (d == 0 ? 1.f - l1[q] : (d == 1 ? l1[q] : 0.f));
                            a[j] += w * v;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < npx) *reinterpret_cast<f32x4*>(out + ((((size_t)b * p.H + y) * p.W + xs + j) << 4) + g * 4) = a[j];
    }
}

// Tiled form of guda_gather_fwd_kernel for branches at <= 1/2 of the output resolution.  The bilinear weights are separable, so
// for an 8 x 64 output tile and one branch the block first blends VERTICALLY and sums the three ky taps at the branch's own
// column resolution:  R[ty][col][kx][c] = sum_ky (wy0 Y[y0][col][ky, kx, c] + wy1 Y[y1][col][ky, kx, c])   (rows of tap ky outside
// the image contribute nothing) -- 8 x <= 36 x 48 values in LDS, read from Y in 192-byte runs -- and every output pixel then
// blends HORIZONTALLY: out += sum_kx (wx0 R[ty][x0][kx] + wx1 R[ty][x1][kx]) at position x + kx - 1 (outside the image: nothing).
// 6 Y loads per R element instead of 36 per output pixel and channel quad; column pitch 52 floats keeps the 16-byte LDS reads of
// neighbouring pixels on different banks.  Rounding differs from the direct form in the order of the blends only.
#ifndef CODEPS_GUDA_TW
#define CODEPS_GUDA_TW 32            // A/B builds: 64 = the 8 x 64 tile of rounds 2-5 (60 KB of LDS, two blocks per CU)
#endif
constexpr int GT_TH = 8, GT_TW = CODEPS_GUDA_TW, GT_PXT = GT_TW / 32, GT_COLS = GT_TW / 2 + 4, GT_PITCH = 52;    // 8 x 32: 33 KB, four blocks per CU
__global__ __launch_bounds__(256) void guda_gather_fwd_tiled_kernel(GudaP p, int tiles_x, int tiles_y, const float* __restrict__ base,
                                                                    float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float R[GT_TH * GT_COLS * GT_PITCH];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x;
    const int b = tile / (tiles_x * tiles_y), tr = tile % (tiles_x * tiles_y);
    const int ty0 = (tr / tiles_x) * GT_TH, tx0 = (tr % tiles_x) * GT_TW;
    const int prow = tid >> 5, px0 = (tid & 31) * GT_PXT;          // this thread's pixel(s): row ty0 + prow, columns tx0 + px0 (+ 1)
    const int oy = ty0 + prow;
    f32x4 acc[GT_PXT][4];
#pragma unroll
    for (int j = 0; j < GT_PXT; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < p.nb; ++k) {
        const GudaBranch br = p.br[k];
        const float* yb = br.y + (size_t)b * br.h * br.w * 144;
        // low-res columns the tile's taps can touch: sources of output columns tx0 - 1 .. tx0 + GT_TW (clamped to the image)
        int c_lo, c_hi, dummy;
        float fd0, fd1;
        bil_src(0, br.rw, max(tx0 - 1, 0), br.w, p.W, c_lo, dummy, fd0, fd1);
        bil_src(0, br.rw, min(tx0 + GT_TW, p.W - 1), br.w, p.W, dummy, c_hi, fd0, fd1);
        const int ncols = c_hi - c_lo + 1;                          // <= GT_COLS (checked by the launcher)
        __syncthreads();                                            // the previous branch's reads of R are done
        // one item per (output row, low-res column, channel quad of the 48): its six Y values (3 ky x 2 rows) are loaded TOGETHER and blended.
        // (Until round 6 a thread walked the tile's 8 rows keeping the row pair of the previous output row: a sixth of the loads, but every load
        // waited for the one before it -- 9-12 memory latencies in a row per branch, 25 us per tile and block, 2.0 TB/s of algorithmic bytes.
        // Neighbouring rows' items read the same Y rows: the repeats hit in the L1.)
        const int nitems = GT_TH * ncols * 12;
#pragma unroll 2
        for (int e = tid; e < nitems; e += 256) {
            const int k4 = e % 12, col = (e / 12) % ncols, ty = e / (12 * ncols);
            const float* src = yb + (size_t)(c_lo + col) * 144 + k4 * 4;
            const int y = ty0 + ty;
            f32x4 v0[3], v1[3];
            float l0[3], l1[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int qy = y + ky - 1;
                const bool ok = y < p.H && qy >= 0 && qy < p.H;
                int y0, y1;
                bil_src(0, br.rh, min(max(qy, 0), p.H - 1), br.h, p.H, y0, y1, l0[ky], l1[ky]);
                if (!ok) l0[ky] = l1[ky] = 0.f;
                v0[ky] = *reinterpret_cast<const f32x4*>(src + (size_t)y0 * br.w * 144 + ky * 48);
                v1[ky] = *reinterpret_cast<const f32x4*>(src + (size_t)y1 * br.w * 144 + ky * 48);
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) v += l0[ky] * v0[ky] + l1[ky] * v1[ky];
            *reinterpret_cast<f32x4*>(R + (ty * GT_COLS + col) * GT_PITCH + k4 * 4) = v;
        }
        __syncthreads();
        if (oy < p.H) {
#pragma unroll
            for (int j = 0; j < GT_PXT; ++j) {
                const int ox = tx0 + px0 + j;
                if (ox >= p.W) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int qx = ox + kx - 1;
                    if (qx < 0 || qx >= p.W) continue;
                    int x0, x1; float lx0, lx1;
                    bil_src(0, br.rw, qx, br.w, p.W, x0, x1, lx0, lx1);
                    const float* r0 = R + (prow * GT_COLS + (x0 - c_lo)) * GT_PITCH + kx * 16;
                    const float* r1 = R + (prow * GT_COLS + (x1 - c_lo)) * GT_PITCH + kx * 16;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        acc[j][c] += lx0 * *reinterpret_cast<const f32x4*>(r0 + c * 4) + lx1 * *reinterpret_cast<const f32x4*>(r1 + c * 4);
                }
            }
        }
    }
    if (oy < p.H) {
#pragma unroll
        for (int j = 0; j < GT_PXT; ++j) {
            const int ox = tx0 + px0 + j;
            if (ox >= p.W) continue;
            const size_t o = (((size_t)b * p.H + oy) * p.W + ox) << 4;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                *reinterpret_cast<f32x4*>(out + o + c * 4) = *reinterpret_cast<const f32x4*>(base + o + c * 4) + acc[j][c];
        }
    }
}

// d y_branch[b][qy][qx][tap][c] = sum_oy wy(oy; qy) sum_ox wx(ox; qx) dout[b][oy - dky][ox - dkx][c]   (dk = tap offset -1..1,
// w = bilinear weight of low-res row / column q for full-res row / column o).  The weights are separable, so the transpose
// runs as two 1-D passes (4x fewer loads than the 2-D footprint sum at x8):
//   pass 1: tmp[b][py][qx][kx][c] = sum_ox wx(ox; qx) dout[b][py][ox - dkx][c]        (full-res rows, low-res columns)
//   pass 2: dy[b][qy][qx][ky*3+kx][c] = sum_oy wy(oy; qy) tmp[b][oy - dky][qx][kx][c]
// The interpolation weight of low-res index q for full-res index o is max(0, 1 - |clamp(r (o + 0.5) - 0.5, 0, n - 1) - q|)
// (same as bil_src's l0 / l1).  Deterministic gathers.
__device__ __forceinline__ void guda_range(int q, float inv_r, int n_full, int& lo, int& hi) {
    lo = max(0, (int)ceilf(((float)q - 0.5f) * inv_r - 0.5f) - 1);
    hi = min(n_full - 1, (int)floorf(((float)q + 1.5f) * inv_r - 0.5f) + 1);
}
__device__ __forceinline__ float guda_weight(int o, int q, float r, int n_low) {
    const float s = fminf(fmaxf(r * ((float)o + 0.5f) - 0.5f, 0.f), (float)(n_low - 1));
    return fmaxf(0.f, 1.f - fabsf(s - (float)q));
}

// (Both passes: block = one row of the result, a run of 256 items of that row -- until round 6 every thread split a 64-bit linear index with
// three or four 64-bit divisions in front of a 4 ... 18-term sum.)
__global__ __launch_bounds__(256) void guda_gather_bwd_x_kernel(int B, int H, int W, int w, float rw, const float* __restrict__ dout,
                                                                float* __restrict__ tmp) {
    const float inv_rw = 1.f / rw;
    const size_t r = blockIdx.x;                                  // b * H + py
    const float* drow = dout + ((r * W) << 4);
    float* trow = tmp + r * (size_t)w * 48;                        // tmp[b][py][qx][kx][16]
    for (int i = blockIdx.y * 256 + threadIdx.x; i < w * 12; i += gridDim.y * 256) {
        const int g = i & 3, q3 = i >> 2, qx = q3 / 3, kx = q3 - qx * 3;
        int lo, hi;
        guda_range(qx, inv_rw, W, lo, hi);
        const float* row = drow + g * 4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        // four terms' loads in flight at a time (a `for (ox ...) a += w * load` loop waits for every load in turn: 4 ... 18 memory latencies
        // per item); the terms are added in the same order
        const int o0 = max(lo, kx - 1), o1 = min(hi, W - 2 + kx);               // px = ox - (kx - 1) in [0, W)
        for (int ob = o0; ob <= o1; ob += 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(row + ((size_t)(min(ob + u, o1) - kx + 1) << 4));
#pragma unroll
            for (int u = 0; u < 4; ++u) if (ob + u <= o1) a += guda_weight(ob + u, qx, rw, w) * v[u];
        }
        *reinterpret_cast<f32x4*>(trow + ((size_t)i << 2)) = a;
    }
}

__global__ __launch_bounds__(256) void guda_gather_bwd_y_kernel(int B, int H, int h, int w, float rh, const float* __restrict__ tmp,
                                                                float* __restrict__ dy) {
    const float inv_rh = 1.f / rh;
    const int rowi = blockIdx.x, b = rowi / h, qy = rowi % h;
    int lo, hi;
    guda_range(qy, inv_rh, H, lo, hi);
    float* yrow = dy + (size_t)rowi * w * 144;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < w * 36; i += gridDim.y * 256) {
        const int g = i & 3, q9 = i >> 2, qx = q9 / 9, tap = q9 - qx * 9;
        const int ky = tap / 3, kx = tap - ky * 3;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        const int o0 = max(lo, ky - 1), o1 = min(hi, H - 2 + ky);
        for (int ob = o0; ob <= o1; ob += 4) {                  // (four loads in flight, as in the x pass)
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                v[u] = *reinterpret_cast<const f32x4*>(tmp + (((((size_t)b * H + (min(ob + u, o1) - ky + 1)) * w + qx) * 3 + kx) << 4) + g * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u) if (ob + u <= o1) a += guda_weight(ob + u, qy, rh, h) * v[u];
        }
        *reinterpret_cast<f32x4*>(yrow + ((size_t)i << 2)) = a;
    }
}

// Both passes in ONE launch (round 6): a block owns TQ x TX low-resolution outputs (all nine taps), runs the x pass for the <= (TQ + 1) s + 6
// full-resolution rows they touch into LDS (T[row][qx][kx][16]) and the y pass from there -- the [B][H][w][3][16] intermediate (151 MB at
// 1/2 resolution and B = 3) is neither written nor read back; the rows two vertically neighbouring blocks share are computed twice (x 1.4 ...
// 2 of the x pass, whose reads hit the L2).  Same terms in the same order as the two-pass form: bit-identical.
template <int TX>
__global__ __launch_bounds__(256) void guda_gather_bwd_fused_kernel(int B, int H, int W, int h, int w, float rh, float rw, int TQ, int tiles_x, int tiles_y,
                                                                    const float* __restrict__ dout, float* __restrict__ dy) {
    extern __shared__ __attribute__((aligned(16))) float T[];       // [rows][TX][3][16]
    const float inv_rh = 1.f / rh, inv_rw = 1.f / rw;
    const int tile = blockIdx.x;
    const int b = tile / (tiles_x * tiles_y), tr = tile % (tiles_x * tiles_y);
    const int qy0 = (tr / tiles_x) * TQ, qx0 = (tr % tiles_x) * TX;
    int lo0, hi0, lo1, hi1;
    guda_range(qy0, inv_rh, H, lo0, hi0);
    guda_range(min(qy0 + TQ, h) - 1, inv_rh, H, lo1, hi1);
    const int py_lo = max(lo0 - 1, 0), py_hi = min(hi1 + 1, H - 1), nrows = py_hi - py_lo + 1;
    // x pass: T[r][qxl][kx] = sum_ox wx(ox; qx) dout[py_lo + r][ox - kx + 1]
    for (int e = threadIdx.x; e < nrows * TX * 12; e += 256) {
        const int g = e & 3, q3 = e >> 2, kx = q3 % 3, t = q3 / 3, qxl = t % TX, r = t / TX;
        const int qx = qx0 + qxl;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (qx < w) {
            int lo, hi;
            guda_range(qx, inv_rw, W, lo, hi);
            const float* row = dout + ((((size_t)b * H + py_lo + r) * W) << 4) + g * 4;
            const int o0 = max(lo, kx - 1), o1 = min(hi, W - 2 + kx);
            for (int ob = o0; ob <= o1; ob += 4) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(row + ((size_t)(min(ob + u, o1) - kx + 1) << 4));
#pragma unroll
                for (int u = 0; u < 4; ++u) if (ob + u <= o1) a += guda_weight(ob + u, qx, rw, w) * v[u];
            }
        }
        *reinterpret_cast<f32x4*>(T + ((size_t)e << 2)) = a;         // (e enumerates [r][qxl][kx][g] in T's order)
    }
    __syncthreads();
    // y pass: dy[qy][qx][ky][kx] = sum_oy wy(oy; qy) T[oy - ky + 1 - py_lo][qxl][kx]
    for (int e = threadIdx.x; e < TQ * TX * 36; e += 256) {
        const int g = e & 3, q9 = e >> 2, tap = q9 % 9, t = q9 / 9, qxl = t % TX, qyl = t / TX;
        const int qy = qy0 + qyl, qx = qx0 + qxl;
        if (qy >= h || qx >= w) continue;
        const int ky = tap / 3, kx = tap - ky * 3;
        int lo, hi;
        guda_range(qy, inv_rh, H, lo, hi);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int oy = max(lo, ky - 1); oy <= min(hi, H - 2 + ky); ++oy)
            a += guda_weight(oy, qy, rh, h) * *reinterpret_cast<const f32x4*>(T + ((((oy - ky + 1 - py_lo) * TX + qxl) * 3 + kx) << 4) + g * 4);
        *reinterpret_cast<f32x4*>(dy + ((((size_t)b * h + qy) * w + qx) * 9 + tap) * 16 + g * 4) = a;
    }
}

// ---------------------------------------------------------------------------------- channel softmax / sigmoid-depth
// x, y: [M][Cp] with C real channels (padding channels of y are written as 0)
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, long long M, int C, int Cp,
                                                          float* __restrict__ y) {
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long long)gridDim.x * 256) {
        const float* px = x + (size_t)r * Cp;
        float v[32];
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) { v[c] = px[c]; mx = fmaxf(mx, v[c]); }
        float s = 0.f;
        for (int c = 0; c < C; ++c) { v[c] = expf(v[c] - mx); s += v[c]; }
        float* py = y + (size_t)r * Cp;
        for (int c = 0; c < C; ++c) py[c] = v[c] / s;
        for (int c = C; c < Cp; ++c) py[c] = 0.f;
    }
}

// dx = y * (dy - sum_c dy*y)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, long long M,
                                                          int C, int Cp, float* __restrict__ dx) {
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long long)gridDim.x * 256) {
        const float* pd = dy + (size_t)r * Cp;
        const float* py = y + (size_t)r * Cp;
        float dot = 0.f;
        for (int c = 0; c < C; ++c) dot += pd[c] * py[c];
        float* po = dx + (size_t)r * Cp;
        for (int c = 0; c < C; ++c) po[c] = py[c] * (pd[c] - dot);
        for (int c = C; c < Cp; ++c) po[c] = 0.f;
    }
}

// Cp == 16 (every class head of the path): one thread per pixel row held in four float4 registers (64 contiguous bytes
// per lane, whole cache lines per wave) instead of scalar loads through an indexed local array
// Round 5: the rows travel through an LDS tile -- the block's 256 rows are read and written with 16-byte accesses in memory order (a wave
// = 1 KB contiguous) and a thread picks its row out of LDS (68-byte pitch).  A thread reading its own 64 bytes straight from memory issued
// four instructions of 16 bytes at a 64-byte lane stride: 3.3 TB/s.  Same arithmetic per row.
constexpr int SM_PITCH = 17;
__device__ __forceinline__ void rows16_in(const float* __restrict__ x, long long r0, int nrow, float* tile) {
    for (int e = threadIdx.x; e < nrow * 4; e += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)r0 * 16 + (size_t)e * 4);
        float* d = tile + (e >> 2) * SM_PITCH + (e & 3) * 4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
}
__device__ __forceinline__ void rows16_out(const float* tile, long long r0, int nrow, float* __restrict__ y) {
    for (int e = threadIdx.x; e < nrow * 4; e += 256) {
        const float* s = tile + (e >> 2) * SM_PITCH + (e & 3) * 4;
        *reinterpret_cast<f32x4*>(y + (size_t)r0 * 16 + (size_t)e * 4) = f32x4{s[0], s[1], s[2], s[3]};
    }
}

__global__ __launch_bounds__(256) void softmax16_fwd_kernel(const float* __restrict__ x, long long M, int C, float* __restrict__ y) {
    __shared__ float tile[256 * SM_PITCH];
    const long long chunks = (M + 255) / 256;
    for (long long ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
        const long long r0 = ch * 256;
        const int nrow = (int)min((long long)256, M - r0);
        __syncthreads();
        rows16_in(x, r0, nrow, tile);
        __syncthreads();
        if ((int)threadIdx.x < nrow) {
            float* row = tile + threadIdx.x * SM_PITCH;
            float v[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = row[c];
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < 16; ++c) if (c < C) mx = fmaxf(mx, v[c]);
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float e = c < C ? expf(v[c] - mx) : 0.f;
                v[c] = e; s += e;
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) row[c] = v[c] / s;          // same rounding as the generic kernel (v / s)
        }
        __syncthreads();
        rows16_out(tile, r0, nrow, y);
    }
}

__global__ __launch_bounds__(256) void softmax16_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, long long M, int C,
                                                            float* __restrict__ dx) {
    __shared__ float td[256 * SM_PITCH];
    __shared__ float tp[256 * SM_PITCH];
    const long long chunks = (M + 255) / 256;
    for (long long ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
        const long long r0 = ch * 256;
        const int nrow = (int)min((long long)256, M - r0);
        __syncthreads();
        rows16_in(dy, r0, nrow, td);
        rows16_in(y, r0, nrow, tp);
        __syncthreads();
        if ((int)threadIdx.x < nrow) {
            float* d = td + threadIdx.x * SM_PITCH;
            const float* p = tp + threadIdx.x * SM_PITCH;
            float dv[16], pv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) { dv[c] = d[c]; pv[c] = p[c]; }
            float dot = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) if (c < C) dot += dv[c] * pv[c];
#pragma unroll
            for (int c = 0; c < 16; ++c) d[c] = c < C ? pv[c] * (dv[c] - dot) : 0.f;
        }
        __syncthreads();
        rows16_out(td, r0, nrow, dx);
    }
}

// logits [M][Cp] channel 0 -> disp = sigmoid, depth = 1 / (min_disp + (max_disp - min_disp) * disp)   (planar outputs [M])
__global__ __launch_bounds__(256) void disp_depth_fwd_kernel(const float* __restrict__ x, long long M, int Cp, float min_disp,
                                                             float max_disp, float* __restrict__ disp, float* __restrict__ depth) {
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long long)gridDim.x * 256) {
        const float s = 1.f / (1.f + expf(-x[(size_t)r * Cp]));
        disp[r] = s;
        depth[r] = 1.f / (min_disp + (max_disp - min_disp) * s);
    }
}

// dlogit (written into channel 0 of a zeroed [M][Cp] row) = (ddisp + ddepth * d depth/d disp) * s(1-s)
__global__ __launch_bounds__(256) void disp_depth_bwd_kernel(const float* __restrict__ ddisp, const float* __restrict__ ddepth,
                                                             const float* __restrict__ disp, long long M, int Cp, float min_disp,
                                                             float max_disp, float* __restrict__ dx) {
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long long)gridDim.x * 256) {
        const float s = disp[r];
        const float sd = min_disp + (max_disp - min_disp) * s;
        float g = 0.f;
        if (ddisp) g += ddisp[r];
        if (ddepth) g += ddepth[r] * (-(max_disp - min_disp) / (sd * sd));
        float* po = dx + (size_t)r * Cp;
        po[0] = g * s * (1.f - s);
        for (int c = 1; c < Cp; ++c) po[c] = 0.f;
    }
}

static int rows_per_block(long long M, int& nblk) {
    // <= 1024 partial blocks (enough loads in flight to cover HBM latency; the finish pass walks them with 8 lanes per
    // output in a fixed order), >= 128 rows each
    long long rpb = std::max<long long>(128, (M + 1023) / 1024);
    nblk = (int)((M + rpb - 1) / rpb);
    return (int)rpb;
}

static int check_colsum_c(int C, const char* who) {
    CODEPS_REQUIRE(C >= 4 && (C % 4) == 0, CODEPS_ERR_ARG, "%s: channel count %d must be a positive multiple of 4", who, C);
    return CODEPS_OK;
}
static inline int col_tiles(int C) { return cdiv(C / 4, 256); }

// Several NCHW frame batches -> ONE NHWC tensor in one launch (round 5: the pose net's six frame pairs were twelve launches, the encoder's
// four image keys four): slab s covers nb[s] consecutive batch items of the destination; its pixels take the c channels of frame a at
// channel 0 and (b != null) of frame b at channel c, zeros elsewhere -- what nchw_to_nhwc_kernel wrote with (zero_rest = 1) for a and
// (zero_rest = 0, c_off = c) for b.  One thread per (destination pixel, 4-channel group).
struct FrameSlabs { const float* a[16]; const float* b[16]; int b0[17]; int n; };
// Block = (destination row bb * H + yy, a run of 256 (column, channel group) items of that row): the slab lookup and the row's addresses are
// block-uniform scalars, a thread's own arithmetic is one small division.  (Until round 6 every thread split a 64-bit linear index with three
// 64-bit divisions and two 15-step select chains -- ~400 instructions per 16-byte store: 122 us for the adaptation step's 8-frame batch, 2.0 TB/s.)
__global__ __launch_bounds__(256) void frames_to_nhwc_kernel(FrameSlabs fs, int C, int H, int W, float* __restrict__ y, int Cp, int pad_l, int Wp) {
    const int groups = Cp / 4;
    const int row = blockIdx.x, bb = row / H, yy = row % H;
    int s = 0;
#pragma unroll
    for (int k = 1; k < 16; ++k) s += (k < fs.n && bb >= fs.b0[k]) ? 1 : 0;
    const float* fa = fs.a[0];
    const float* fb = fs.b[0];
    int bs = fs.b0[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) if (k == s) { fa = fs.a[k]; fb = fs.b[k]; bs = fs.b0[k]; }        // (select chain: no indexed access into the by-value struct)
    const size_t HW = (size_t)H * W;
    const size_t src_row = (size_t)(bb - bs) * C * HW + (size_t)yy * W;
    float* dst_row = y + (size_t)row * Wp * Cp;
    for (int i = blockIdx.y * 256 + threadIdx.x; i < Wp * groups; i += gridDim.y * 256) {
        const int xd = i / groups, g = i - xd * groups;
        const int xs = xd - pad_l;
        const bool in = xs >= 0 && xs < W;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = g * 4 + e;
            float t = 0.f;
            if (in && c < C) t = fa[src_row + (size_t)c * HW + xs];
            else if (in && fb && c < 2 * C) t = fb[src_row + (size_t)(c - C) * HW + xs];
            v[e] = t;
        }
        *reinterpret_cast<f32x4*>(dst_row + (size_t)i * 4) = v;
    }
}

}  // namespace

// =============================================================================================== C ABI
extern "C" int codeps_frames_to_nhwc(const float* const* frames_a, const float* const* frames_b, const int* slab_batch, int n_slabs, int c, int h,
                                     int w, float* y, int c_pad, int pad_left, int pad_right, codeps_stream_t stream) {
    CODEPS_REQUIRE(frames_a && slab_batch && y && n_slabs >= 1 && n_slabs <= 16 && c > 0 && h > 0 && w > 0 && (c_pad % 4) == 0 && pad_left >= 0 &&
                   pad_right >= 0, CODEPS_ERR_ARG, "frames_to_nhwc: bad arguments (1..16 slabs)");
    FrameSlabs fs;
    memset(&fs, 0, sizeof(fs));
    fs.n = n_slabs;
    bool two = false;
    for (int s = 0; s < n_slabs; ++s) {
        CODEPS_REQUIRE(frames_a[s] && slab_batch[s] > 0, CODEPS_ERR_ARG, "frames_to_nhwc: null frame / empty slab %d", s);
        fs.a[s] = frames_a[s]; fs.b[s] = frames_b ? frames_b[s] : nullptr;
        two = two || fs.b[s];
        fs.b0[s + 1] = fs.b0[s] + slab_batch[s];
    }
    for (int s = n_slabs + 1; s <= 16; ++s) fs.b0[s] = fs.b0[n_slabs];
    CODEPS_REQUIRE((two ? 2 : 1) * c <= c_pad, CODEPS_ERR_ARG, "frames_to_nhwc: %d channels do not fit c_pad %d", (two ? 2 : 1) * c, c_pad);
    const int wp = pad_left + w + pad_right;
    const long long rows = (long long)fs.b0[n_slabs] * h;
    CODEPS_REQUIRE(rows < (1ll << 31), CODEPS_ERR_ARG, "frames_to_nhwc: too many rows");
    hipLaunchKernelGGL(frames_to_nhwc_kernel, dim3((unsigned)rows, std::min(cdiv(wp * (c_pad / 4), 256), 64)), dim3(256), 0, (hipStream_t)stream, fs, c, h, w,
                       y, c_pad, pad_left, wp);
    CODEPS_CHECK_LAUNCH("frames_to_nhwc_kernel");
    return CODEPS_OK;
}

static int nchw_to_nhwc_impl(const float* x, int batch, int c, int h, int w, float* y, int c_pad, int c_off, int zero_rest, int pad_l,
                             int pad_r, codeps_stream_t stream, const char* who) {
    CODEPS_REQUIRE(x && y && batch > 0 && c > 0 && h > 0 && w > 0 && (c_pad % 4) == 0 && c_off >= 0 && c_off + c <= c_pad && pad_l >= 0 &&
                       pad_r >= 0,
                   CODEPS_ERR_ARG, "%s: bad arguments", who);
    const int wp = pad_l + w + pad_r;
    const long long items = (long long)batch * h * (zero_rest ? wp : w) * (zero_rest ? c_pad / 4 : (c + 3) / 4);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(items)), dim3(256), 0, (hipStream_t)stream, x, batch, c, h, w, y, c_pad,
                       c_off, zero_rest, pad_l, wp);
    CODEPS_CHECK_LAUNCH("nchw_to_nhwc_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_nchw_to_nhwc(const float* x, int batch, int c, int h, int w, float* y, int c_pad, int c_off, int zero_rest,
                                   codeps_stream_t stream) {
    return nchw_to_nhwc_impl(x, batch, c, h, w, y, c_pad, c_off, zero_rest, 0, 0, stream, "nchw_to_nhwc");
}

extern "C" int codeps_nchw_to_nhwc_xpad(const float* x, int batch, int c, int h, int w, float* y, int c_pad, int c_off, int zero_rest,
                                        int pad_left, int pad_right, codeps_stream_t stream) {
    return nchw_to_nhwc_impl(x, batch, c, h, w, y, c_pad, c_off, zero_rest, pad_left, pad_right, stream, "nchw_to_nhwc_xpad");
}

extern "C" int codeps_nhwc_to_nchw(const float* x, int batch, int c, int h, int w, int c_pad, float* y, codeps_stream_t stream) {
    CODEPS_REQUIRE(x && y && batch > 0 && c > 0 && c <= c_pad && h > 0 && w > 0, CODEPS_ERR_ARG, "nhwc_to_nchw: bad arguments");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((long long)batch * c * h * w)), dim3(256), 0, (hipStream_t)stream, x, batch,
                       c, h, w, c_pad, y);
    CODEPS_CHECK_LAUNCH("nhwc_to_nchw_kernel");
    return CODEPS_OK;
}

extern "C" size_t codeps_colsum_workspace_bytes(long long rows, int c, int k) {
    int nblk;
    rows_per_block(rows, nblk);
    return (size_t)nblk * k * c * sizeof(float);
}

extern "C" int codeps_act_bwd(const float* dy, const float* y, const float* scale, int act, long long rows, int c, float* dz,
                              float* dbias, int c_real, int accumulate, void* workspace, size_t workspace_bytes,
                              codeps_stream_t stream) {
    CODEPS_REQUIRE(dy && rows > 0 && (dz || dbias) && (act == 0 || y), CODEPS_ERR_ARG, "act_bwd: bad arguments");
    int rc = check_colsum_c(c, "act_bwd");
    if (rc) return rc;
    int nblk;
    const int rpb = rows_per_block(rows, nblk);
    float* part = nullptr;
    if (dbias) {
        CODEPS_REQUIRE(workspace && workspace_bytes >= (size_t)nblk * c * 4, CODEPS_ERR_WORKSPACE, "act_bwd: workspace too small");
        part = (float*)workspace;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(nblk, col_tiles(c)), dim3(256), 0, st, dy, y, scale, act, rows, c, dz, part, rpb);
    CODEPS_CHECK_LAUNCH("act_bwd_kernel");
    if (dbias) {
        hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(c, 4)), dim3(256), 0, st, part, nblk, 1, c, dbias, c_real, accumulate);
        CODEPS_CHECK_LAUNCH("colsum_finish_kernel");
    }
    return CODEPS_OK;
}

extern "C" int codeps_bn_stats(const float* x, long long rows, int c, float* sums /*[2][c]*/, void* workspace, size_t workspace_bytes,
                               codeps_stream_t stream) {
    CODEPS_REQUIRE(x && sums && rows > 0, CODEPS_ERR_ARG, "bn_stats: bad arguments");
    int rc = check_colsum_c(c, "bn_stats");
    if (rc) return rc;
    int nblk;
    const int rpb = rows_per_block(rows, nblk);
    CODEPS_REQUIRE(workspace && workspace_bytes >= (size_t)nblk * 2 * c * 4, CODEPS_ERR_WORKSPACE, "bn_stats: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk, col_tiles(c)), dim3(256), 0, st, x, rows, c, (float*)workspace, rpb);
    CODEPS_CHECK_LAUNCH("bn_stats_kernel");
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(2 * c, 4)), dim3(256), 0, st, (const float*)workspace, nblk, 2, c, sums, c, 0);
    CODEPS_CHECK_LAUNCH("colsum_finish_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                               const float* residual, int act, long long rows, int c, float* y, codeps_stream_t stream) {
    CODEPS_REQUIRE(x && mean && invstd && gamma && beta && y && rows > 0 && (c % 4) == 0 && act >= 0 && act <= 1, CODEPS_ERR_ARG,
                   "bn_apply: bad arguments");
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(rows * (c / 4))), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta,
                       residual, act, rows, c, y);
    CODEPS_CHECK_LAUNCH("bn_apply_kernel");
    return CODEPS_OK;
}

// bias gradient from the per-tile column sums a data-gradient launch left (codeps_conv_epilogue stat_mode 2): part [tiles][c]
extern "C" int codeps_colsum_finish(const float* part, int tiles, int c, float* out, int c_real, int accumulate, codeps_stream_t stream) {
    CODEPS_REQUIRE(part && out && tiles > 0 && c > 0 && c_real > 0 && c_real <= c, CODEPS_ERR_ARG, "colsum_finish: bad arguments");
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(c, 4)), dim3(256), 0, (hipStream_t)stream, part, tiles, 1, c, out, c_real, accumulate);
    CODEPS_CHECK_LAUNCH("colsum_finish_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_colsum_finish_multi(const float* const* parts, const int* tiles, const int* c, float* const* outs, const int* c_real, int n,
                                          codeps_stream_t stream) {
    CODEPS_REQUIRE(parts && tiles && c && outs && c_real && n >= 1 && n <= 32, CODEPS_ERR_ARG, "colsum_finish_multi: 1..32 items");
    ColsumItems it;
    memset(&it, 0, sizeof(it));
    it.n = n;
    for (int k = 0; k < n; ++k) {
        CODEPS_REQUIRE(parts[k] && outs[k] && tiles[k] > 0 && c[k] > 0 && c_real[k] > 0 && c_real[k] <= c[k], CODEPS_ERR_ARG, "colsum_finish_multi: bad item %d", k);
        it.part[k] = parts[k]; it.out[k] = outs[k]; it.tiles[k] = tiles[k]; it.c[k] = c[k]; it.c_real[k] = c_real[k];
        it.blk0[k + 1] = it.blk0[k] + cdiv(c[k], 4);
    }
    hipLaunchKernelGGL(colsum_finish_multi_kernel, dim3(it.blk0[n]), dim3(256), 0, (hipStream_t)stream, it);
    CODEPS_CHECK_LAUNCH("colsum_finish_multi_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_bn_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean, const float* invstd, int act,
                                    long long rows, int c, float* sums /*[2][c]*/, void* workspace, size_t workspace_bytes,
                                    codeps_stream_t stream) {
    CODEPS_REQUIRE(dy && x && mean && invstd && sums && rows > 0 && (act == 0 || y), CODEPS_ERR_ARG, "bn_bwd_reduce: bad arguments");
    int rc = check_colsum_c(c, "bn_bwd_reduce");
    if (rc) return rc;
    int nblk;
    const int rpb = rows_per_block(rows, nblk);
    CODEPS_REQUIRE(workspace && workspace_bytes >= (size_t)nblk * 2 * c * 4, CODEPS_ERR_WORKSPACE, "bn_bwd_reduce: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nblk, col_tiles(c)), dim3(256), 0, st, dy, y, x, mean, invstd, act, rows, c, (float*)workspace, rpb);
    CODEPS_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(2 * c, 4)), dim3(256), 0, st, (const float*)workspace, nblk, 2, c, sums, c, 0);
    CODEPS_CHECK_LAUNCH("colsum_finish_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* mean, const float* invstd,
                                   const float* gamma, const float* sums, float inv_count, int act, long long rows, int c, float* dx,
                                   float* dres, codeps_stream_t stream) {
    CODEPS_REQUIRE(dy && x && mean && invstd && gamma && sums && dx && rows > 0 && (c % 4) == 0 && (act == 0 || y), CODEPS_ERR_ARG,
                   "bn_bwd_apply: bad arguments");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * (c / 4))), dim3(256), 0, (hipStream_t)stream, dy, y, x, mean, invstd,
                       gamma, sums, inv_count, act, rows, c, dx, dres);
    CODEPS_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return CODEPS_OK;
}

static int make_groups(int n_groups, const long long* row0, const long long* rows, BnGroups& gr, long long& total, const char* who) {
    CODEPS_REQUIRE(n_groups >= 1 && n_groups <= 8 && row0 && rows, CODEPS_ERR_ARG, "%s: 1..8 groups supported", who);
    gr.n = n_groups;
    total = 0;
    for (int i = 0; i < 8; ++i) { gr.row0[i] = 0; gr.rows[i] = 0; gr.inv_count[i] = 0.f; }
    for (int i = 0; i < n_groups; ++i) {
        CODEPS_REQUIRE(rows[i] > 0 && row0[i] == total, CODEPS_ERR_ARG, "%s: groups must be non-empty, contiguous and in order", who);
        gr.row0[i] = row0[i]; gr.rows[i] = rows[i];
        total += rows[i];
    }
    return CODEPS_OK;
}

static int mg_blocks(const BnGroups& gr) {
    // partial blocks per group: ~2048 blocks in total (8 per CU: enough loads in flight for a streaming reduction even when a
    // single group = plain BatchNorm owns the launch), >= 128 rows each, <= 1024 per group (= the workspace slabs)
    long long mx = 0;
    for (int i = 0; i < gr.n; ++i) mx = std::max(mx, gr.rows[i]);
    const long long cap = std::min<long long>(1024, std::max<long long>(256, 2048 / std::max(gr.n, 1)));
    return (int)std::max<long long>(1, std::min<long long>(cap, (mx + 127) / 128));
}

extern "C" size_t codeps_bn_mg_workspace_bytes(int n_groups, int c) { return (size_t)n_groups * 1024 * 2 * c * sizeof(float); }

extern "C" int codeps_bn_fwd_mg(const float* x, int n_groups, const long long* row0, const long long* rows, int c, int c_real,
                                const float* gamma, const float* beta, const float* residual, int act, float eps, float momentum,
                                float* running_mean, float* running_var, float* mean, float* var, float* invstd, float* y,
                                int stats_only, int apply_only, void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    BnGroups gr;
    long long M;
    int rc = make_groups(n_groups, row0, rows, gr, M, "bn_fwd_mg");
    if (rc) return rc;
    CODEPS_REQUIRE(x && mean && var && invstd && (c % 4) == 0 && c_real <= c && act >= 0 && act <= 1, CODEPS_ERR_ARG, "bn_fwd_mg: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = mg_blocks(gr);
    if (!apply_only) {
        CODEPS_REQUIRE(workspace && workspace_bytes >= (size_t)n_groups * nblk * 2 * c * 4, CODEPS_ERR_WORKSPACE, "bn_fwd_mg: workspace too small");
        hipLaunchKernelGGL(bn_stats_mg_kernel, dim3(nblk, col_tiles(c), n_groups), dim3(256), 0, st, x, gr, c, nblk, (float*)workspace);
        hipLaunchKernelGGL(bn_finalize_mg_kernel, dim3(cdiv(c, 4)), dim3(256), 0, st, x, (const float*)workspace, gr, c, c_real, nblk, eps,
                           momentum, mean, var, invstd, running_mean, running_var);
        CODEPS_CHECK_LAUNCH("bn_stats_mg_kernel");
    }
    if (!stats_only) {
        CODEPS_REQUIRE(gamma && beta && y, CODEPS_ERR_ARG, "bn_fwd_mg: null affine / output");
        // apply_only with running statistics: the batch statistics came from codeps_bn_finalize_tiles -- the momentum updates ride here
        BnRunning run{apply_only ? running_mean : nullptr, running_var, var, momentum, c_real};
        hipLaunchKernelGGL(bn_apply_mg_kernel, dim3(grid_for(M * (c / 4))), dim3(256), 0, st, x, gr, (const float*)mean, (const float*)invstd,
                           gamma, beta, residual, act, M, c, y, run);
        CODEPS_CHECK_LAUNCH("bn_apply_mg_kernel");
    }
    return CODEPS_OK;
}

static int make_tile_args(int n_seg, const int* seg_tile0, const int* seg_tiles_per_image, const int* seg_tile_rows, int n_groups,
                          const int* group_b0, const int* group_b1, TileSegs& sg, ImgGroups& ig, const char* who) {
    CODEPS_REQUIRE(n_seg >= 1 && n_seg <= 4 && seg_tile0 && seg_tiles_per_image && seg_tile_rows && n_groups >= 1 && n_groups <= 8 && group_b0 &&
                   group_b1, CODEPS_ERR_ARG, "%s: 1..4 tile segments, 1..8 groups", who);
    memset(&sg, 0, sizeof(sg)); memset(&ig, 0, sizeof(ig));
    sg.n = n_seg; ig.n = n_groups;
    for (int k = 0; k < n_seg; ++k) {
        CODEPS_REQUIRE(seg_tile0[k] >= 0 && seg_tiles_per_image[k] > 0 && seg_tile_rows[k] > 0, CODEPS_ERR_ARG, "%s: bad tile segment %d", who, k);
        sg.tile0[k] = seg_tile0[k]; sg.tpi[k] = seg_tiles_per_image[k]; sg.rows[k] = seg_tile_rows[k];
    }
    for (int g = 0; g < n_groups; ++g) {
        CODEPS_REQUIRE(group_b1[g] > group_b0[g] && group_b0[g] >= 0, CODEPS_ERR_ARG, "%s: empty statistic group %d", who, g);
        ig.b0[g] = group_b0[g]; ig.b1[g] = group_b1[g];
    }
    return CODEPS_OK;
}

extern "C" int codeps_bn_finalize_tiles(const float* part, int tile0, int tiles_per_image, int tile_rows, int n_groups, const int* group_b0,
                                        const int* group_b1, int c, float eps, float* mean, float* var, float* invstd, codeps_stream_t stream) {
    TileSegs sg; ImgGroups ig;
    int rc = make_tile_args(1, &tile0, &tiles_per_image, &tile_rows, n_groups, group_b0, group_b1, sg, ig, "bn_finalize_tiles");
    if (rc) return rc;
    CODEPS_REQUIRE(part && mean && var && invstd && c > 0, CODEPS_ERR_ARG, "bn_finalize_tiles: bad arguments");
    hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3(cdiv(c, 64), n_groups), dim3(1024), 0, (hipStream_t)stream, part, sg, ig, c, eps, mean, var, invstd,
                       BnTableOut{});
    CODEPS_CHECK_LAUNCH("bn_finalize_tiles_kernel");
    return CODEPS_OK;
}

// codeps_bn_finalize_tiles + codeps_bn_in_affine in one launch: statistics from the producing convolution's per-tile partials AND the table of a
// BatchNorm whose apply pass is left to its consumer (gamma / beta [c]; table [batch][4][c]).  running_mean / running_var: one statistic group
// only (the updates of several groups are ordered: codeps_bn_in_affine).
extern "C" int codeps_bn_finalize_tiles_table(const float* part, int tile0, int tiles_per_image, int tile_rows, int n_groups, const int* group_b0,
                                              const int* group_b1, int c, int c_real, float eps, float momentum, const float* gamma, const float* beta,
                                              float* mean, float* var, float* invstd, float* running_mean, float* running_var, float* table,
                                              codeps_stream_t stream) {
    TileSegs sg; ImgGroups ig;
    int rc = make_tile_args(1, &tile0, &tiles_per_image, &tile_rows, n_groups, group_b0, group_b1, sg, ig, "bn_finalize_tiles_table");
    if (rc) return rc;
    CODEPS_REQUIRE(part && mean && var && invstd && gamma && beta && table && c > 0 && c_real <= c && (!running_mean || (running_var && n_groups == 1)),
                   CODEPS_ERR_ARG, "bn_finalize_tiles_table: bad arguments (running statistics: one group only)");
    hipLaunchKernelGGL(bn_finalize_tiles_kernel, dim3(cdiv(c, 64), n_groups), dim3(1024), 0, (hipStream_t)stream, part, sg, ig, c, eps, mean, var, invstd,
                       BnTableOut{table, gamma, beta, running_mean, running_var, momentum, c_real});
    CODEPS_CHECK_LAUNCH("bn_finalize_tiles_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_bn_bwd_tiles(const float* part, int n_seg, const int* seg_tile0, const int* seg_tiles_per_image, const int* seg_tile_rows,
                                   int n_groups, const long long* row0, const long long* rows, const int* group_b0, const int* group_b1,
                                   const float* g, const float* x, int c, int c_real, const float* mean, const float* invstd, const float* gamma,
                                   const float* inv_count, float* local_sums, const float* sums, float* dgamma, float* dbeta, int accumulate,
                                   float* dx, int phase, codeps_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    if (phase == 0 || phase == 1) {
        TileSegs sg; ImgGroups ig;
        int rc = make_tile_args(n_seg, seg_tile0, seg_tiles_per_image, seg_tile_rows, n_groups, group_b0, group_b1, sg, ig, "bn_bwd_tiles");
        if (rc) return rc;
        CODEPS_REQUIRE(part && local_sums && c > 0, CODEPS_ERR_ARG, "bn_bwd_tiles: bad arguments");
        hipLaunchKernelGGL(bn_bwd_finish_tiles_kernel, dim3(cdiv(c, 64), n_groups), dim3(1024), 0, st, part, sg, ig, c, local_sums);
        CODEPS_CHECK_LAUNCH("bn_bwd_finish_tiles_kernel");
    }
    if (phase == 0 || phase == 2) {
        BnGroups gr;
        long long M;
        int rc = make_groups(n_groups, row0, rows, gr, M, "bn_bwd_tiles");
        if (rc) return rc;
        CODEPS_REQUIRE(g && x && mean && invstd && gamma && inv_count && local_sums && dgamma && dbeta && dx && (c % 4) == 0 && c_real <= c,
                       CODEPS_ERR_ARG, "bn_bwd_tiles: bad arguments");
        for (int i = 0; i < n_groups; ++i) gr.inv_count[i] = inv_count[i];
        hipLaunchKernelGGL(bn_bwd_apply_mg_kernel<false>, dim3(grid_for(M * (c / 4))), dim3(256), 0, st, g, (const float*)nullptr, x, gr, mean, invstd, gamma,
                           (const float*)nullptr, sums ? sums : (const float*)local_sums, 0, M, c, dx, (float*)nullptr, PoolSrc{},
                           BnAffineGrad{dgamma, dbeta, local_sums, accumulate, c_real});
        CODEPS_CHECK_LAUNCH("bn_bwd_apply_mg_kernel");
    }
    return CODEPS_OK;
}

// ---- train-mode BatchNorm + ReLU applied on the CONSUMER's operand load (round 6): the per-image table the consumer reads
// (codeps_conv2d_fwd_in / codeps_conv2d_wgrad_in) + the running-statistics update that otherwise rides in the apply pass
namespace {
__global__ __launch_bounds__(256) void bn_in_affine_kernel(const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, ImgGroups ig, int batch, int C, float* __restrict__ table,
                                                           BnRunning run, int rows_per_image) {
    if (run.mean && blockIdx.x == 0)                             // (bn_apply_mg_kernel's expressions, group after group)
        for (int c = threadIdx.x; c < run.c_real; c += 256) {
            float rm = run.mean[c], rv = run.var[c];
            for (int g = 0; g < ig.n; ++g) {
                const float n = (float)((long long)(ig.b1[g] - ig.b0[g]) * rows_per_image);
                rm = rm * (1.f - run.momentum) + run.momentum * mean[g * C + c];
                rv = rv * (1.f - run.momentum) + run.momentum * (run.batch_var[g * C + c] * (n / fmaxf(n - 1.f, 1.f)));
            }
            run.mean[c] = rm; run.var[c] = rv;
        }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < batch * C; i += gridDim.x * 256) {
        const int b = i / C, c = i % C;
        int g = 0;
        for (int k = 1; k < ig.n; ++k) g += b >= ig.b0[k] ? 1 : 0;
        float* t = table + (size_t)b * 4 * C + c;
        t[0] = mean[g * C + c]; t[C] = invstd[g * C + c]; t[2 * C] = gamma[c]; t[3 * C] = beta[c];
    }
}

__global__ __launch_bounds__(256) void in_affine_apply_kernel(const float* __restrict__ x, const float* __restrict__ table, long long total4, int rows_per_image,
                                                              int C, float* __restrict__ y) {
    const int G = C / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int g4 = (int)(i % G);
        const long long row = i / G;
        const float* t = table + (size_t)(row / rows_per_image) * 4 * C + g4 * 4;
        *reinterpret_cast<f32x4*>(y + i * 4) = bn_norm_relu(*reinterpret_cast<const f32x4*>(x + i * 4), *reinterpret_cast<const f32x4*>(t),
                                                            *reinterpret_cast<const f32x4*>(t + C), *reinterpret_cast<const f32x4*>(t + 2 * C),
                                                            *reinterpret_cast<const f32x4*>(t + 3 * C));
    }
}
}  // namespace

extern "C" int codeps_bn_in_affine(const float* mean, const float* var, const float* invstd, const float* gamma, const float* beta, int n_groups,
                                   const int* group_b0, const int* group_b1, int batch, int rows_per_image, int c, int c_real, float momentum,
                                   float* running_mean, float* running_var, float* table, codeps_stream_t stream) {
    CODEPS_REQUIRE(mean && invstd && gamma && beta && table && n_groups >= 1 && n_groups <= 8 && group_b0 && group_b1 && batch > 0 && rows_per_image > 0 &&
                   c > 0 && (c % 4) == 0 && c_real <= c && (!running_mean || (running_var && var)), CODEPS_ERR_ARG, "bn_in_affine: bad arguments");
    ImgGroups ig;
    memset(&ig, 0, sizeof(ig));
    ig.n = n_groups;
    for (int g = 0; g < n_groups; ++g) {
        CODEPS_REQUIRE(group_b1[g] > group_b0[g] && group_b0[g] >= 0 && group_b1[g] <= batch, CODEPS_ERR_ARG, "bn_in_affine: bad statistic group %d", g);
        ig.b0[g] = group_b0[g]; ig.b1[g] = group_b1[g];
    }
    BnRunning run{running_mean, running_var, var, momentum, c_real};
    hipLaunchKernelGGL(bn_in_affine_kernel, dim3(std::max(1, std::min(64, cdiv(batch * c, 256)))), dim3(256), 0, (hipStream_t)stream, mean, invstd, gamma, beta, ig,
                       batch, c, table, run, rows_per_image);
    CODEPS_CHECK_LAUNCH("bn_in_affine_kernel");
    return CODEPS_OK;
}

// y = relu(BatchNorm(x)) from the table of codeps_bn_in_affine: the pass the consumers of codeps_conv2d_in_affine_ok shapes do without
// (for a consumer that cannot normalise on its load)
extern "C" int codeps_in_affine_apply(const float* x, const float* table, int batch, int rows_per_image, int c, float* y, codeps_stream_t stream) {
    CODEPS_REQUIRE(x && table && y && batch > 0 && rows_per_image > 0 && c > 0 && (c % 4) == 0, CODEPS_ERR_ARG, "in_affine_apply: bad arguments");
    const long long total4 = (long long)batch * rows_per_image * (c / 4);
    hipLaunchKernelGGL(in_affine_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, table, total4, rows_per_image, c, y);
    CODEPS_CHECK_LAUNCH("in_affine_apply_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_bn_sync_combine(const float* gathered, int world, int n_groups, const float* global_counts, const float* rank_counts,
                                      int c, int c_real, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                                      float* var, float* invstd, codeps_stream_t stream) {
    CODEPS_REQUIRE(gathered && global_counts && mean && var && invstd && world >= 1 && n_groups >= 1 && n_groups <= 8 && c > 0 && c_real <= c,
                   CODEPS_ERR_ARG, "bn_sync_combine: bad arguments");
    CODEPS_REQUIRE(!rank_counts || world * n_groups <= 64, CODEPS_ERR_ARG, "bn_sync_combine: per-rank counts support world x groups <= 64");
    RankCounts rc;
    rc.on = rank_counts ? 1 : 0;
    for (int i = 0; i < 64; ++i) rc.n[i] = (rank_counts && i < world * n_groups) ? rank_counts[i] : 0.f;
    if (rank_counts)
        for (int g = 0; g < n_groups; ++g) {
            float tot = 0.f;
            for (int r = 0; r < world; ++r) tot += rank_counts[r * n_groups + g];
            CODEPS_REQUIRE(tot > 0.f, CODEPS_ERR_ARG, "bn_sync_combine: statistic group %d is empty on every rank", g);
        }
    BnGroups gr;
    gr.n = n_groups;
    for (int i = 0; i < 8; ++i) { gr.row0[i] = 0; gr.rows[i] = 0; gr.inv_count[i] = 0.f; }
    for (int i = 0; i < n_groups; ++i) gr.inv_count[i] = global_counts[i];
    hipLaunchKernelGGL(bn_sync_combine_kernel, dim3(cdiv(c, 256)), dim3(256), 0, (hipStream_t)stream, gathered, world, gr, rc, c, c_real, eps, momentum,
                       mean, var, invstd, running_mean, running_var);
    CODEPS_CHECK_LAUNCH("bn_sync_combine_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_bn_bwd_mg(const float* dy, const float* y, const float* x, int n_groups, const long long* row0, const long long* rows,
                                int c, int c_real, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                const float* inv_count, int act, float* sums, float* dgamma, float* dbeta, int accumulate, float* dx, float* dres,
                                int reduce_only, int apply_only, void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    BnGroups gr;
    long long M;
    int rc = make_groups(n_groups, row0, rows, gr, M, "bn_bwd_mg");
    if (rc) return rc;
    CODEPS_REQUIRE(dy && x && mean && invstd && sums && (c % 4) == 0 && (act == 0 || y || (gamma && beta && !dres)), CODEPS_ERR_ARG,
                   "bn_bwd_mg: bad arguments (the ReLU mask needs y, or gamma + beta and no residual)");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = mg_blocks(gr);
    if (!apply_only) {
        CODEPS_REQUIRE(workspace && workspace_bytes >= (size_t)n_groups * nblk * 2 * c * 4 && dgamma && dbeta, CODEPS_ERR_WORKSPACE,
                       "bn_bwd_mg: workspace too small / null grads");
        hipLaunchKernelGGL(bn_bwd_reduce_mg_kernel<false>, dim3(nblk, col_tiles(c), n_groups), dim3(256), 0, st, dy, y, x, gr, mean, invstd, gamma, beta, act, c, nblk,
                           (float*)workspace, PoolSrc{});
        hipLaunchKernelGGL(bn_bwd_finish_mg_kernel, dim3(cdiv(c, 4)), dim3(256), 0, st, (const float*)workspace, gr, c, c_real, nblk, sums, dgamma,
                           dbeta, accumulate);
        CODEPS_CHECK_LAUNCH("bn_bwd_reduce_mg_kernel");
    }
    if (!reduce_only) {
        CODEPS_REQUIRE(gamma && inv_count && dx, CODEPS_ERR_ARG, "bn_bwd_mg: null gamma / inv_count / dx");
        for (int i = 0; i < n_groups; ++i) gr.inv_count[i] = inv_count[i];
        hipLaunchKernelGGL(bn_bwd_apply_mg_kernel<false>, dim3(grid_for(M * (c / 4))), dim3(256), 0, st, dy, y, x, gr, mean, invstd, gamma, beta, (const float*)sums,
                           act, M, c, dx, dres, PoolSrc{}, BnAffineGrad{});
        CODEPS_CHECK_LAUNCH("bn_bwd_apply_mg_kernel");
    }
    return CODEPS_OK;
}

// Stem in train mode, fused (models/resnet_encoder.py:118-121: bn1 -> relu -> maxpool): forward = relu(BN(z)) max-pooled in one pass, the
// statistics come from codeps_bn_fwd_mg(stats_only = 1); backward = BatchNorm's two passes reading the POOLED gradient + arg-max bytes.
// Groups are whole images (row0 / rows in pixels, multiples of h * w).  Bit-identical to bn_fwd_mg + maxpool3x3s2_fwd resp.
// maxpool3x3s2_bwd + bn_bwd_mg: same expressions, same orders of summation.
extern "C" int codeps_bn_relu_pool_fwd(const float* z, int n_groups, const long long* row0, const long long* rows, int batch, int h, int w, int c,
                                       const float* mean, const float* invstd, const float* gamma, const float* beta, float* pooled,
                                       uint8_t* argmax, float* running_mean, float* running_var, const float* batch_var, float momentum,
                                       int c_real, codeps_stream_t stream) {
    BnGroups gr;
    long long M;
    int rc = make_groups(n_groups, row0, rows, gr, M, "bn_relu_pool_fwd");
    if (rc) return rc;
    CODEPS_REQUIRE(z && mean && invstd && gamma && beta && pooled && argmax && batch > 0 && h > 0 && w > 0 && (c % 4) == 0 &&
                   M == (long long)batch * h * w, CODEPS_ERR_ARG, "bn_relu_pool_fwd: bad arguments (groups must cover batch * h * w rows)");
    for (int i = 0; i < n_groups; ++i)
        CODEPS_REQUIRE(row0[i] % ((long long)h * w) == 0 && rows[i] % ((long long)h * w) == 0, CODEPS_ERR_ARG, "bn_relu_pool_fwd: groups are whole images");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    CODEPS_REQUIRE(!running_mean || (running_var && batch_var && c_real > 0 && c_real <= c), CODEPS_ERR_ARG,
                   "bn_relu_pool_fwd: running statistics need running_var, the batch variances and c_real");
    BnRunning run{running_mean, running_var, batch_var, momentum, c_real};
    hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, dim3(grid_for((long long)batch * ho * wo * (c / 4))), dim3(256), 0, (hipStream_t)stream, z, gr, mean, invstd,
                       gamma, beta, batch, h, w, c, ho, wo, pooled, argmax, run);
    CODEPS_CHECK_LAUNCH("bn_relu_pool_fwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_bn_relu_pool_bwd(const float* dpooled, const uint8_t* argmax, const float* z, int n_groups, const long long* row0,
                                       const long long* rows, int batch, int h, int w, int c, int c_real, const float* mean, const float* invstd,
                                       const float* gamma, const float* beta, const float* inv_count, float* sums, float* dgamma, float* dbeta,
                                       int accumulate, float* dz, void* workspace, size_t workspace_bytes, int phase, codeps_stream_t stream) {
    BnGroups gr;
    long long M;
    int rc = make_groups(n_groups, row0, rows, gr, M, "bn_relu_pool_bwd");
    if (rc) return rc;
    CODEPS_REQUIRE(dpooled && argmax && z && mean && invstd && gamma && beta && inv_count && sums && dgamma && dbeta && dz && (c % 4) == 0 &&
                   M == (long long)batch * h * w && phase >= 0 && phase <= 2, CODEPS_ERR_ARG, "bn_relu_pool_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = mg_blocks(gr);
    CODEPS_REQUIRE(workspace && workspace_bytes >= (size_t)n_groups * nblk * 2 * c * 4, CODEPS_ERR_WORKSPACE, "bn_relu_pool_bwd: workspace too small");
    PoolSrc ps;
    ps.dp = dpooled; ps.arg = argmax; ps.H = h; ps.W = w; ps.Ho = (h + 2 - 3) / 2 + 1; ps.Wo = (w + 2 - 3) / 2 + 1;
    ps.inv_hw = 1.f / ((float)h * (float)w); ps.inv_w = 1.f / (float)w;
    // phase 0: everything; 1: the sums (+ dgamma / dbeta, from THIS rank's sums) only; 2: dz from `sums` only -- SyncBatchNorm all-reduces
    // the sums in between (misc/train_utils.py:50-51)
    if (phase != 2) {
        hipLaunchKernelGGL(bn_bwd_reduce_mg_kernel<true>, dim3(nblk, col_tiles(c), n_groups), dim3(256), 0, st, (const float*)nullptr, (const float*)nullptr, z, gr,
                           mean, invstd, gamma, beta, 1, c, nblk, (float*)workspace, ps);
        hipLaunchKernelGGL(bn_bwd_finish_mg_kernel, dim3(cdiv(c, 4)), dim3(256), 0, st, (const float*)workspace, gr, c, c_real, nblk, sums, dgamma, dbeta,
                           accumulate);
    }
    if (phase == 1) { CODEPS_CHECK_LAUNCH("bn_relu_pool_bwd"); return CODEPS_OK; }
    for (int i = 0; i < n_groups; ++i) gr.inv_count[i] = inv_count[i];
    if (h % 2 == 0 && w % 2 == 0)
        hipLaunchKernelGGL(bn_bwd_apply_pool2x2_kernel, dim3(grid_for(M / 4 * (c / 4))), dim3(256), 0, st, z, gr, mean, invstd, gamma, beta, (const float*)sums,
                           batch, c, dz, ps);
    else
    hipLaunchKernelGGL(bn_bwd_apply_mg_kernel<true>, dim3(grid_for(M * (c / 4))), dim3(256), 0, st, (const float*)nullptr, (const float*)nullptr, z, gr, mean,
                       invstd, gamma, beta, (const float*)sums, 1, M, c, dz, (float*)nullptr, ps, BnAffineGrad{});
    CODEPS_CHECK_LAUNCH("bn_relu_pool_bwd");
    return CODEPS_OK;
}

extern "C" int codeps_maxpool3x3s2_fwd(const float* x, int batch, int h, int w, int c, float* y, uint8_t* argmax, codeps_stream_t stream) {
    CODEPS_REQUIRE(x && y && batch > 0 && h > 0 && w > 0 && (c % 4) == 0, CODEPS_ERR_ARG, "maxpool_fwd: bad arguments");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long long)batch * ho * wo * (c / 4))), dim3(256), 0, (hipStream_t)stream, x, batch,
                       h, w, c, ho, wo, y, argmax);
    CODEPS_CHECK_LAUNCH("maxpool_fwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_maxpool3x3s2_bwd(const float* dy, const uint8_t* argmax, int batch, int h, int w, int c, float* dx,
                                       codeps_stream_t stream) {
    CODEPS_REQUIRE(dy && argmax && dx && batch > 0 && (c % 4) == 0, CODEPS_ERR_ARG, "maxpool_bwd: bad arguments");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long long)batch * h * w * (c / 4))), dim3(256), 0, (hipStream_t)stream, dy, argmax,
                       batch, h, w, c, ho, wo, dx);
    CODEPS_CHECK_LAUNCH("maxpool_bwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_dwconv5x5(const float* x, const float* w25c, const float* scale, const float* shift, int act, int batch, int h,
                                int w, int c, int flip, float* y, codeps_stream_t stream) {
    CODEPS_REQUIRE(x && w25c && y && batch > 0 && h > 0 && w > 0 && (c % 4) == 0, CODEPS_ERR_ARG, "dwconv5x5: bad arguments");
    static const int dw_tile = getenv("CODEPS_DW_TILE") ? atoi(getenv("CODEPS_DW_TILE")) : 1;       // A/B switch: 0 = the gather kernels
    const int tiles_x = cdiv(w, DWT), tiles_y = cdiv(h, DWT);
    if (dw_tile && (long long)batch * tiles_x * tiles_y < (1ll << 31)) {
        hipLaunchKernelGGL(dwconv5_tile_kernel, dim3(batch * tiles_x * tiles_y, cdiv(c, DWCG)), dim3(256), 0, (hipStream_t)stream, x, w25c, scale, shift, act,
                           batch, h, w, c, flip, y, tiles_x, tiles_y);
        CODEPS_CHECK_LAUNCH("dwconv5_tile_kernel");
        return CODEPS_OK;
    }
    hipLaunchKernelGGL(dwconv5_kernel, dim3(grid_for((long long)batch * ((h + 3) / 4) * ((w + 1) / 2) * (c / 4))), dim3(256), 0, (hipStream_t)stream, x, w25c, scale,
                       shift, act, batch, h, w, c, flip, y);
    CODEPS_CHECK_LAUNCH("dwconv5_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_dwconv5x5_wgrad(const float* x, const float* dz, int batch, int h, int w, int c, float* dw25c, void* workspace,
                                      size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(x && dz && dw25c && batch > 0, CODEPS_ERR_ARG, "dwconv5x5_wgrad: bad arguments");
    int rc = check_colsum_c(c, "dwconv5x5_wgrad");
    if (rc) return rc;
    int nblk;
    (void)rows_per_block((long long)batch * h * w, nblk);        // block count (= workspace slabs) as codeps_colsum_workspace_bytes sizes it
    CODEPS_REQUIRE(workspace && workspace_bytes >= (size_t)nblk * 25 * c * 4, CODEPS_ERR_WORKSPACE, "dwconv5x5_wgrad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    static const int dw_tile = getenv("CODEPS_DW_TILE") ? atoi(getenv("CODEPS_DW_TILE")) : 1;       // A/B switch: 0 = the gather kernel
    const int tiles_x = cdiv(w, DWT), tiles_y = cdiv(h, DWT);
    const long long tiles = (long long)batch * tiles_x * tiles_y;
    if (dw_tile && tiles < (1ll << 31)) {
        // <= nblk slabs (the workspace's size); ~2 blocks per CU and channel group in flight, whole tiles per block
        const int groups = cdiv(c, DWCG);
        const int want = std::max(1, std::min<int>(nblk, std::max(1, 1024 / groups)));
        const int tpb = (int)((tiles + want - 1) / want);
        nblk = (int)((tiles + tpb - 1) / tpb);
        hipLaunchKernelGGL(dwconv5_wgrad_tile_kernel, dim3(nblk, groups), dim3(256), 0, st, x, dz, batch, h, w, c, (float*)workspace, (int)tiles, tpb, tiles_x, tiles_y);
        CODEPS_CHECK_LAUNCH("dwconv5_wgrad_tile_kernel");
    } else {
        const long long items = (long long)batch * ((h + 3) / 4) * w;  // (image, 4-row strip, column)
        const int ipb = (int)((items + nblk - 1) / nblk);
        hipLaunchKernelGGL(dwconv5_wgrad_kernel, dim3(nblk, col_tiles(c)), dim3(256), 0, st, x, dz, batch, h, w, c, (float*)workspace, ipb);
        CODEPS_CHECK_LAUNCH("dwconv5_wgrad_kernel");
    }
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(25 * c, 4)), dim3(256), 0, st, (const float*)workspace, nblk, 25, c, dw25c, c, 0);
    CODEPS_CHECK_LAUNCH("colsum_finish_kernel");
    return CODEPS_OK;
}

static void resize_ratios(int align, int h, int w, int ho, int wo, float& rh, float& rw) {
    if (align) {
        rh = ho > 1 ? (float)(h - 1) / (float)(ho - 1) : 0.f;
        rw = wo > 1 ? (float)(w - 1) / (float)(wo - 1) : 0.f;
    } else {
        rh = (float)h / (float)ho;
        rw = (float)w / (float)wo;
    }
}

extern "C" int codeps_bilinear_fwd(const float* x, int batch, int h, int w, int c, int out_h, int out_w, int align_corners, float* y,
                                   int y_c, int y_c_off, codeps_stream_t stream) {
    CODEPS_REQUIRE(x && y && batch > 0 && h > 0 && w > 0 && out_h > 0 && out_w > 0 && (c % 4) == 0 && (y_c % 4) == 0 &&
                       (y_c_off % 4) == 0 && y_c_off + c <= y_c,
                   CODEPS_ERR_ARG, "bilinear_fwd: bad arguments");
    float rh, rw;
    resize_ratios(align_corners, h, w, out_h, out_w, rh, rw);
    CODEPS_REQUIRE((long long)batch * out_h < (1ll << 31), CODEPS_ERR_ARG, "bilinear_fwd: too many rows");
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3((unsigned)(batch * out_h), std::min(cdiv(out_w * (c / 4), 256), 64)), dim3(256), 0, (hipStream_t)stream, x,
                       batch, h, w, c, out_h, out_w, align_corners, rh, rw, y, y_c, y_c_off);
    CODEPS_CHECK_LAUNCH("bilinear_fwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_bilinear_bwd(const float* dy, int batch, int h, int w, int c, int out_h, int out_w, int align_corners, int y_c,
                                   int y_c_off, float* dx, codeps_stream_t stream) {
    CODEPS_REQUIRE(dy && dx && batch > 0 && h > 0 && w > 0 && out_h >= h && out_w >= w && (c % 4) == 0 && (y_c % 4) == 0 &&
                       (y_c_off % 4) == 0 && y_c_off + c <= y_c,
                   CODEPS_ERR_ARG, "bilinear_bwd: bad arguments (only up-sampling is differentiated)");
    float rh, rw;
    resize_ratios(align_corners, h, w, out_h, out_w, rh, rw);
    CODEPS_REQUIRE((long long)batch * h < (1ll << 31), CODEPS_ERR_ARG, "bilinear_bwd: too many rows");
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3((unsigned)(batch * h), std::min(cdiv(w * (c / 4), 256), 64)), dim3(256), 0, (hipStream_t)stream, dy, batch, h,
                       w, c, out_h, out_w, align_corners, rh, rw, y_c, y_c_off, dx);
    CODEPS_CHECK_LAUNCH("bilinear_bwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_guda_gather_fwd(const float* base, int batch, int h, int w, int n_branches, const float* const* y, const int* bh,
                                      const int* bw, float* out, codeps_stream_t stream) {
    CODEPS_REQUIRE(base && out && y && bh && bw && n_branches >= 1 && n_branches <= 3 && batch > 0 && h > 0 && w > 0, CODEPS_ERR_ARG,
                   "guda_gather_fwd: bad arguments");
    GudaP p;
    p.B = batch; p.H = h; p.W = w; p.nb = n_branches;
    for (int k = 0; k < 3; ++k) { p.br[k].y = nullptr; p.br[k].h = p.br[k].w = 1; p.br[k].rh = p.br[k].rw = 1.f; }
    for (int k = 0; k < n_branches; ++k) {
        CODEPS_REQUIRE(y[k] && bh[k] > 0 && bw[k] > 0 && bh[k] < h && bw[k] < w, CODEPS_ERR_ARG, "guda_gather_fwd: branch %d must be coarser than the output", k);
        p.br[k].y = y[k]; p.br[k].h = bh[k]; p.br[k].w = bw[k];
        p.br[k].rh = (float)bh[k] / (float)h; p.br[k].rw = (float)bw[k] / (float)w;
    }
    // tiled form when every branch's column footprint of a 64-pixel tile (+ the taps' halo) fits the LDS tile: ratios <= ~1/2
    static const int tiled_env = getenv("CODEPS_GUDA_TILED") ? atoi(getenv("CODEPS_GUDA_TILED")) : 1;
    bool tiled = tiled_env != 0;
    for (int k = 0; k < n_branches; ++k)
        if ((int)ceilf((float)(GT_TW + 2) * p.br[k].rw) + 3 > GT_COLS) tiled = false;
    if (tiled) {
        const int tiles_x = (w + GT_TW - 1) / GT_TW, tiles_y = (h + GT_TH - 1) / GT_TH;
        hipLaunchKernelGGL(guda_gather_fwd_tiled_kernel, dim3(batch * tiles_x * tiles_y), dim3(256), 0, (hipStream_t)stream, p, tiles_x, tiles_y, base, out);
        CODEPS_CHECK_LAUNCH("guda_gather_fwd_tiled_kernel");
        return CODEPS_OK;
    }
    hipLaunchKernelGGL(guda_gather_fwd_kernel, dim3(grid_for((long long)batch * h * ((w + 3) / 4) * 4)), dim3(256), 0, (hipStream_t)stream, p, base, out);
    CODEPS_CHECK_LAUNCH("guda_gather_fwd_kernel");
    return CODEPS_OK;
}

extern "C" size_t codeps_guda_gather_bwd_workspace_bytes(int batch, int h, int bw) { return (size_t)batch * h * bw * 3 * 16 * sizeof(float); }

extern "C" int codeps_guda_gather_bwd(const float* grad_out, int batch, int h, int w, int bh, int bw, float* grad_y, void* workspace,
                                      size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(grad_out && grad_y && batch > 0 && bh > 0 && bw > 0 && bh < h && bw < w, CODEPS_ERR_ARG, "guda_gather_bwd: bad arguments");
    CODEPS_REQUIRE(workspace && workspace_bytes >= codeps_guda_gather_bwd_workspace_bytes(batch, h, bw), CODEPS_ERR_WORKSPACE,
                   "guda_gather_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    {
        // the fused form: TQ low-resolution rows per block so that (TQ + 1) s + 6 full-resolution rows x 8 columns x 192 bytes fit 48 KB of LDS
        static const int fused = getenv("CODEPS_GUDA_BWD_FUSED") ? atoi(getenv("CODEPS_GUDA_BWD_FUSED")) : 1;       // A/B switch: 0 = two passes
        constexpr int TX = 8;
        const int sy = cdiv(h, bh);
        const int TQ = std::max(1, std::min(8, 16 / sy));
        const int max_rows = (TQ + 1) * sy + 6;
        const size_t lds = (size_t)max_rows * TX * 48 * sizeof(float);
        const long long tiles = (long long)batch * cdiv(bh, TQ) * cdiv(bw, TX);
        // (measured at B = 3, 512 x 1024: 1/2 resolution 206 -> 140 us; 1/4: 76 -> 77; 1/8: 49 -> 80 -- the shared rows are 2.1 x the x pass there)
        if (fused && sy <= 2 && lds <= 48 * 1024 && tiles < (1ll << 31)) {
            hipLaunchKernelGGL(guda_gather_bwd_fused_kernel<TX>, dim3((unsigned)tiles), dim3(256), lds, st, batch, h, w, bh, bw, (float)bh / (float)h,
                               (float)bw / (float)w, TQ, cdiv(bw, TX), cdiv(bh, TQ), grad_out, grad_y);
            CODEPS_CHECK_LAUNCH("guda_gather_bwd_fused_kernel");
            return CODEPS_OK;
        }
    }
    hipLaunchKernelGGL(guda_gather_bwd_x_kernel, dim3((unsigned)(batch * h), std::min(cdiv(bw * 12, 256), 64)), dim3(256), 0, st, batch, h, w, bw,
                       (float)bw / (float)w, grad_out, (float*)workspace);
    hipLaunchKernelGGL(guda_gather_bwd_y_kernel, dim3((unsigned)(batch * bh), std::min(cdiv(bw * 36, 256), 64)), dim3(256), 0, st, batch, h, bh, bw,
                       (float)bh / (float)h, (const float*)workspace, grad_y);
    CODEPS_CHECK_LAUNCH("guda_gather_bwd kernels");
    return CODEPS_OK;
}

// dst = src[0] ++ src[1] ++ ... (up to 8 contiguous fp32 sources, pointers and lengths by value): torch.cat along the leading dimension
// and plain copies (the padded per-channel vectors, the pieces of a split batch's gradient) without a library kernel
struct CatSet { const float* src[8]; long long end[8]; };       // end[i] = elements of sources 0 .. i
__global__ __launch_bounds__(256) void concat_kernel(CatSet s, int n, float* __restrict__ dst) {
    const long long total = s.end[n - 1];
    const bool vec = (total & 3) == 0 && (reinterpret_cast<size_t>(dst) & 15) == 0;
    bool al = vec;
    for (int k = 0; k < n; ++k) al = al && (s.end[k] & 3) == 0 && (reinterpret_cast<size_t>(s.src[k]) & 15) == 0;
    if (al) {
        for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long long)gridDim.x * 1024) {
            int k = 0;
            while (i >= s.end[k]) ++k;
            const long long off = i - (k ? s.end[k - 1] : 0);
            *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<const f32x4*>(s.src[k] + off);
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            int k = 0;
            while (i >= s.end[k]) ++k;
            dst[i] = s.src[k][i - (k ? s.end[k - 1] : 0)];
        }
    }
}

extern "C" int codeps_concat_f32(const float* const* sources, const long long* counts, int n, float* dst, codeps_stream_t stream) {
    CODEPS_REQUIRE(sources && counts && dst && n >= 1 && n <= 8, CODEPS_ERR_ARG, "concat_f32: 1 .. 8 sources");
    CatSet s;
    long long total = 0;
    for (int i = 0; i < 8; ++i) {
        if (i < n) { CODEPS_REQUIRE(sources[i] && counts[i] >= 0, CODEPS_ERR_ARG, "concat_f32: bad source %d", i); total += counts[i]; }
        s.src[i] = i < n ? sources[i] : nullptr;
        s.end[i] = total;
    }
    if (total == 0) return CODEPS_OK;
    hipLaunchKernelGGL(concat_kernel, dim3(grid_for((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, s, n, dst);
    CODEPS_CHECK_LAUNCH("concat_kernel");
    return CODEPS_OK;
}

// arg-max over the first C channels of every NHWC row (torch.argmax semantics: first index of the maximum, a NaN is the maximum),
// written as int64 and / or uint8 in one pass
__global__ __launch_bounds__(256) void channel_argmax_kernel(const float* __restrict__ x, long long M, int C, int Cp, long long* __restrict__ o64,
                                                             uint8_t* __restrict__ o8) {
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long long)gridDim.x * 256) {
        const float* row = x + (size_t)r * Cp;
        float best = row[0];
        int bi = 0;
        for (int c0 = 0; c0 < C; c0 += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + c0);          // Cp % 4 == 0: the padding lanes are never compared
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = c0 + e;
                if (c == 0 || c >= C) continue;
                const float t = v[e];
                if ((t > best || t != t) && best == best) { best = t; bi = c; }
            }
        }
        if (o64) o64[r] = bi;
        if (o8) o8[r] = (uint8_t)bi;
    }
}

extern "C" int codeps_channel_argmax(const float* x, long long rows, int c, int c_pad, long long* out_i64, unsigned char* out_u8,
                                     codeps_stream_t stream) {
    CODEPS_REQUIRE(x && (out_i64 || out_u8) && rows > 0 && c > 0 && c <= c_pad && (c_pad % 4) == 0 && (!out_u8 || c <= 256), CODEPS_ERR_ARG,
                   "channel_argmax: bad arguments (c_pad %% 4 == 0; uint8 output: c <= 256)");
    hipLaunchKernelGGL(channel_argmax_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)stream, x, rows, c, c_pad, out_i64, out_u8);
    CODEPS_CHECK_LAUNCH("channel_argmax_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_softmax_fwd(const float* x, long long rows, int c, int c_pad, float* y, codeps_stream_t stream) {
    CODEPS_REQUIRE(x && y && rows > 0 && c > 0 && c <= 32 && c <= c_pad, CODEPS_ERR_ARG, "softmax_fwd: bad arguments (c <= 32)");
    if (c_pad == 16) hipLaunchKernelGGL(softmax16_fwd_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)stream, x, rows, c, y);
    else hipLaunchKernelGGL(softmax_fwd_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)stream, x, rows, c, c_pad, y);
    CODEPS_CHECK_LAUNCH("softmax_fwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_softmax_bwd(const float* dy, const float* y, long long rows, int c, int c_pad, float* dx, codeps_stream_t stream) {
    CODEPS_REQUIRE(dy && y && dx && rows > 0 && c > 0 && c <= c_pad, CODEPS_ERR_ARG, "softmax_bwd: bad arguments");
    if (c_pad == 16) hipLaunchKernelGGL(softmax16_bwd_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)stream, dy, y, rows, c, dx);
    else hipLaunchKernelGGL(softmax_bwd_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)stream, dy, y, rows, c, c_pad, dx);
    CODEPS_CHECK_LAUNCH("softmax_bwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_disp_depth_fwd(const float* logits, long long rows, int c_pad, float min_depth, float max_depth, float* disp,
                                     float* depth, codeps_stream_t stream) {
    CODEPS_REQUIRE(logits && disp && depth && rows > 0 && c_pad > 0 && min_depth > 0 && max_depth > min_depth, CODEPS_ERR_ARG,
                   "disp_depth_fwd: bad arguments");
    hipLaunchKernelGGL(disp_depth_fwd_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)stream, logits, rows, c_pad,
                       (float)(1.0 / (double)max_depth), (float)(1.0 / (double)min_depth), disp, depth);
    CODEPS_CHECK_LAUNCH("disp_depth_fwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_disp_depth_bwd(const float* ddisp, const float* ddepth, const float* disp, long long rows, int c_pad,
                                     float min_depth, float max_depth, float* dlogits, codeps_stream_t stream) {
    CODEPS_REQUIRE(disp && dlogits && (ddisp || ddepth) && rows > 0, CODEPS_ERR_ARG, "disp_depth_bwd: bad arguments");
    hipLaunchKernelGGL(disp_depth_bwd_kernel, dim3(grid_for(rows)), dim3(256), 0, (hipStream_t)stream, ddisp, ddepth, disp, rows, c_pad,
                       (float)(1.0 / (double)max_depth), (float)(1.0 / (double)min_depth), dlogits);
    CODEPS_CHECK_LAUNCH("disp_depth_bwd_kernel");
    return CODEPS_OK;
}
