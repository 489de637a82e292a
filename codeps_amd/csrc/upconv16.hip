// The last up-convolution of the depth / semantic decoders (models/depth_head.py:60-71: nearest x2 -> Conv3x3(16, 16) -> ELU, no
// skip tensor at scale 0) as a convolution of the HALF-resolution tensor.  Nearest up-sampling makes neighbouring taps read the
// same source pixel: for an output pixel of row parity py the taps ky = {0 | 1, 2} (py = 0) or {0, 1 | 2} (py = 1) fall on two
// source rows, likewise in x, so each of the four parity classes is a 2 x 2 convolution with pre-summed weights
//     Wc[py][px][a][b] = sum of W[ky][kx] over the taps that land on source offset (a - 1 + py, b - 1 + px),
// zero padding of the full-resolution image = zero padding of the source.  16 (class, tap) pairs instead of 36 (pixel, tap)
// pairs: 2.25 x fewer FLOPs than the fused-gather form of the generic engine, the source tensor is read once from a halo tile
// in LDS (the generic 16-column kernel fetches every tap from L1: nine times the bytes through the texture path), and the
// arithmetic is exact fp32 on v_mfma_f32_16x16x4_f32 (K order: channel c = 4 i + j is the i-th k of the j-th MFMA, so that a
// lane's four k values are one 16-byte LDS read).  HBM-bound by construction: forward reads 1/4 + writes 1, data gradient
// reads 1 + writes 1/4, weight gradient reads 5/4 of the full-resolution tensor.
#include <algorithm>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 16;
constexpr int SH = 8, SW = 16;                         // source tile: 8 x 16 pixels -> 16 x 32 outputs
constexpr int XH = SH + 2, XW = SW + 2, XPIX = XH * XW;
constexpr int P4 = 5;                                  // pixel pitch in float4 (80 bytes: conflict-free 16-byte reads)
constexpr int NWC = 4 * 4 * C * C;                     // collapsed weights [class][tap][ci][co]

// source-row offset of collapsed tap a for parity p, and the set of 3x3 taps it sums
__device__ __forceinline__ int off_of(int p, int a) { return a - 1 + p; }
__device__ __forceinline__ bool tap_in(int p, int a, int k) { return p == 0 ? (a == 0 ? k == 0 : k >= 1) : (a == 0 ? k <= 1 : k == 2); }

// OIHW [16][16][3][3] -> wc[class = py * 2 + px][tap = a * 2 + b][ci][co] (one small launch per call; the blocks read the 16 KB
// from L2)
__global__ __launch_bounds__(256) void upconv16_collapse_kernel(const float* __restrict__ w, float* __restrict__ wc) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < NWC; i += gridDim.x * 256) {
        const int co = i % C, ci = (i / C) % C, tap = (i / (C * C)) % 4, cls = i / (C * C * 4);
        const int py = cls >> 1, px = cls & 1, a = tap >> 1, b = tap & 1;
        float s = 0.f;
        for (int ky = 0; ky < 3; ++ky) {
            if (!tap_in(py, a, ky)) continue;
            for (int kx = 0; kx < 3; ++kx)
                if (tap_in(px, b, kx)) s += w[((size_t)co * C + ci) * 9 + ky * 3 + kx];
        }
        wc[i] = s;
    }
}

// source halo tile (rows sy0 - 1 .., columns sx0 - 1 ..), zero outside the image
// (round 6: every staging loop of this file issues ALL its loads -- from clamped addresses, zeroed afterwards -- into registers before the first
// LDS store.  As `if (inside) v = load; store` loops each element waited for its own memory latency: 12 in a row per tile in the data
// gradient, 8 + 3 in the weight gradient -- 140 / 104 us per launch for 300 / 250 MB.)
__device__ __forceinline__ void load_src_halo(const float* __restrict__ x, int b, int Hs, int Ws, int sy0, int sx0, f32x4* xs) {
    constexpr int NL = (XPIX * 4 + 255) / 256;
    f32x4 r[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int i = min((int)threadIdx.x + 256 * u, XPIX * 4 - 1);
        const int px = i >> 2, q = i & 3;
        const int iy = sy0 - 1 + px / XW, ix = sx0 - 1 + px % XW;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * Hs + min(max(iy, 0), Hs - 1)) * Ws + min(max(ix, 0), Ws - 1)) * C + q * 4);
        r[u] = (iy >= 0 && iy < Hs && ix >= 0 && ix < Ws) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i < XPIX * 4) xs[(i >> 2) * P4 + (i & 3)] = r[u];
    }
}

__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : expm1f(v); }

// ------------------------------------------------------------------------------------------------ forward
// one wave per parity class; per source row one 16-pixel M tile: 4 taps x 4 MFMAs
__global__ __launch_bounds__(256) void upconv16_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wc, const float* __restrict__ bias,
                                                           int Hs, int Ws, int tiles_x, int act, float* __restrict__ y) {
    __shared__ f32x4 xs[XPIX * P4];
    const int b = blockIdx.y, sy0 = (blockIdx.x / tiles_x) * SH, sx0 = (blockIdx.x % tiles_x) * SW;
    const int lane = threadIdx.x & 63, cls = threadIdx.x >> 6, py = cls >> 1, px = cls & 1;
    const int m = lane & 15, i = lane >> 4;
    float bf[4][4];                                     // B fragments: [tap][j] = Wc[cls][tap][ci = 4 i + j][co = m]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[t][j] = wc[((cls * 4 + t) * C + 4 * i + j) * C + m];
    const float bv = bias ? bias[m] : 0.f;
    load_src_halo(x, b, Hs, Ws, sy0, sx0, xs);
    __syncthreads();
    const int H = 2 * Hs, W = 2 * Ws;
#pragma unroll 1
    for (int r0 = 0; r0 < SH; r0 += 4) {
        f32x4 acc[4];
        f32x4 a[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int yy = r0 + u + 1 + off_of(py, t >> 1), xx = m + 1 + off_of(px, t & 1);
                a[u][t] = xs[(yy * XW + xx) * P4 + i];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t][j], bf[t][j], acc[u], 0, 0, 0);
        // C layout: lane (column n = m, rows 4 i + e): output pixel of source column sx0 + 4 i + e, channel m
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int sy = sy0 + r0 + u;
            if (sy >= Hs) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int sx = sx0 + 4 * i + e;
                if (sx >= Ws) continue;
                float v = acc[u][e] + bv;
                if (act == 2) v = elu1(v);
                else if (act == 1) v = v > 0.f ? v : 0.f;
                y[(((size_t)b * H + 2 * sy + py) * W + 2 * sx + px) * C + m] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ data gradient
// dx[q][ci] = sum over (class, tap, co) of dz(2 (q - off) + parity)[co] Wc[class][tap][ci][co]: the dz tile of the 10 x 18
// source neighbourhood (20 x 36 output pixels) sits in LDS; two source rows per wave, 16 (class, tap) x 4 MFMAs per row
constexpr int ZH = 2 * XH, ZW = 2 * XW, ZPIX = ZH * ZW;

// x_act != 0 (round 5): the source tensor x is the output of an activation read by this layer alone: dx *= act'(x) in the output stage and
// the block's column sums of the gated dx go to col_part[block][16] (the producer's activation backward and bias gradient)
__global__ __launch_bounds__(256) void upconv16_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ wc, int Hs, int Ws, int tiles_x,
                                                             float* __restrict__ dx, const float* __restrict__ x, int x_act,
                                                             float* __restrict__ col_part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* zs = reinterpret_cast<f32x4*>(smem);        // [ZPIX][P4]
    float* os = smem + ZPIX * P4 * 4;                  // [SH x SW source pixels][P4 * 4] output staging
    const int b = blockIdx.y, sy0 = (blockIdx.x / tiles_x) * SH, sx0 = (blockIdx.x % tiles_x) * SW;
    const int H = 2 * Hs, W = 2 * Ws;
    {
        constexpr int NL = (ZPIX * 4 + 255) / 256;
        f32x4 r[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int idx = min((int)threadIdx.x + 256 * u, ZPIX * 4 - 1);
            const int p = idx >> 2, q = idx & 3;
            const int oy = 2 * (sy0 - 1) + p / ZW, ox = 2 * (sx0 - 1) + p % ZW;
            const f32x4 v = *reinterpret_cast<const f32x4*>(dz + (((size_t)b * H + min(max(oy, 0), H - 1)) * W + min(max(ox, 0), W - 1)) * C + q * 4);
            r[u] = (oy >= 0 && oy < H && ox >= 0 && ox < W) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int idx = threadIdx.x + 256 * u;
            if (idx < ZPIX * 4) zs[(idx >> 2) * P4 + (idx & 3)] = r[u];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, m = lane & 15, i = lane >> 4;
    f32x4 bf[16];                                       // B[k = co = 4 i + j][n = ci = m] = Wc[class][tap][ci][co]: one 16-byte load per (class, tap)
#pragma unroll
    for (int ct = 0; ct < 16; ++ct) bf[ct] = *reinterpret_cast<const f32x4*>(wc + (ct * C + m) * C + 4 * i);
    __syncthreads();
#pragma unroll 1
    for (int rr = 0; rr < 2; ++rr) {
        const int r = wave * 2 + rr;                    // source row of the tile
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                // the output pixel of this class whose tap t reads source pixel (r, m): source position s = q - off
                const int sy = r - off_of(py, t >> 1), sx = m - off_of(px, t & 1);
                const int zy = 2 * (sy + 1) + py, zx = 2 * (sx + 1) + px;          // tile-local (the tile starts at source row -1)
                const f32x4 a = zs[(zy * ZW + zx) * P4 + i];                      // dz channels co = 4 i + j
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bf[cls * 4 + t][j], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) os[(r * SW + 4 * i + e) * (P4 * 4) + m] = acc[e];
    }
    __syncthreads();
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};                  // (a thread keeps its channel quad q = idx & 3 over the two passes: 256 % 4 == 0)
    for (int idx = threadIdx.x; idx < SH * SW * 4; idx += 256) {
        const int p = idx >> 2, q = idx & 3;
        const int sy = sy0 + p / SW, sx = sx0 + p % SW;
        if (sy < Hs && sx < Ws) {
            const size_t o = (((size_t)b * Hs + sy) * Ws + sx) * C + q * 4;
            f32x4 v = *reinterpret_cast<const f32x4*>(os + p * (P4 * 4) + q * 4);
            if (x_act) {
                const f32x4 y = *reinterpret_cast<const f32x4*>(x + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= y[e] > 0.f ? 1.f : (x_act == 2 ? y[e] + 1.f : 0.f);
                csum += v;
            }
            *reinterpret_cast<f32x4*>(dx + o) = v;
        }
    }
    if (!col_part) return;
    __syncthreads();                                    // `os` is free again: the 64 threads of a channel quad meet there in a fixed order
    f32x4* red = reinterpret_cast<f32x4*>(os);
    red[threadIdx.x] = csum;
    __syncthreads();
    if (threadIdx.x < 4) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < 64; ++k) a += red[k * 4 + threadIdx.x];
        *reinterpret_cast<f32x4*>(col_part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * C + threadIdx.x * 4) = a;
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// G[class][tap][ci][co] = sum over source pixels of x[s + off][ci] dz(2 s + parity)[co]: one wave per class, persistent blocks,
// 4 accumulators (taps) per wave; K = 4 pixels per MFMA
constexpr int OH = 2 * SH, OW = 2 * SW, OPIX = OH * OW;

__global__ __launch_bounds__(256) void upconv16_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz, int B, int Hs, int Ws,
                                                             int tiles_x, int tiles_y, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* xs = reinterpret_cast<f32x4*>(smem);        // [XPIX][P4]
    float* zs = smem + XPIX * P4 * 4;                  // [OPIX][17]: pitch 17 floats (scalar reads, 16 channels of a pixel)
    const int lane = threadIdx.x & 63, cls = threadIdx.x >> 6, py = cls >> 1, px = cls & 1;
    const int m = lane & 15, i = lane >> 4;
    const int H = 2 * Hs, W = 2 * Ws;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xf = reinterpret_cast<const float*>(xs);
    const int ntiles = B * tiles_y * tiles_x;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / (tiles_y * tiles_x), tr = tile % (tiles_y * tiles_x);
        const int sy0 = (tr / tiles_x) * SH, sx0 = (tr % tiles_x) * SW;
        __syncthreads();
        {
            static_assert((OPIX * 4) % 256 == 0, "whole passes");
            constexpr int NL = OPIX * 4 / 256;
            f32x4 r[NL];
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int idx = threadIdx.x + 256 * u;
                const int p = idx >> 2, q = idx & 3;
                const int oy = 2 * sy0 + p / OW, ox = 2 * sx0 + p % OW;
                const f32x4 v = *reinterpret_cast<const f32x4*>(dz + (((size_t)b * H + min(oy, H - 1)) * W + min(ox, W - 1)) * C + q * 4);
                r[u] = (oy < H && ox < W) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            load_src_halo(x, b, Hs, Ws, sy0, sx0, xs);
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int idx = threadIdx.x + 256 * u;
                float* d = zs + (idx >> 2) * 17 + (idx & 3) * 4;
                d[0] = r[u][0]; d[1] = r[u][1]; d[2] = r[u][2]; d[3] = r[u][3];
            }
        }
        __syncthreads();
#pragma unroll 2
        for (int kk = 0; kk < SH * SW / 4; ++kk) {      // 4 source pixels per MFMA: pixel 4 kk + i
            const int p = 4 * kk + i, r = p / SW, c = p % SW;
            const float zb = zs[((2 * r + py) * OW + 2 * c + px) * 17 + m];        // B[k = pixel][n = co = m]
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int yy = r + 1 + off_of(py, t >> 1), xx = c + 1 + off_of(px, t & 1);
                const float xa = xf[(yy * XW + xx) * (P4 * 4) + m];                // A[m = ci][k = pixel]
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, zb, acc[t], 0, 0, 0);
            }
        }
    }
    // C layout: rows ci = 4 i + e, column co = m
    float* dst = part + (size_t)blockIdx.x * NWC + (size_t)cls * 4 * C * C;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[(t * C + 4 * i + e) * C + m] = acc[t][e];
}

// partials [nblk][class][tap][ci][co] -> G[class][tap][ci][co]: 16 lanes walk the blocks of one element, fixed-order tree
__global__ __launch_bounds__(256) void upconv16_wgrad_sum_kernel(const float* __restrict__ part, int nblk, float* __restrict__ gsum) {
    __shared__ float red[16][17];
    const int e = threadIdx.x & 15, kl = threadIdx.x >> 4, idx = blockIdx.x * 16 + e;
    float a = 0.f;
    for (int k = kl; k < nblk; k += 16) a += part[(size_t)k * NWC + idx];
    red[kl][e] = a;
    __syncthreads();
    if (kl == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][e];
        gsum[idx] = t;
    }
}

// G -> OIHW [16][16][3][3]: tap (ky, kx) collects from every class the collapsed tap it fed
__global__ __launch_bounds__(256) void upconv16_wgrad_finish_kernel(const float* __restrict__ gsum, int accumulate, float* __restrict__ dw) {
    for (int o = blockIdx.x * 256 + threadIdx.x; o < C * C * 9; o += gridDim.x * 256) {
        const int kx = o % 3, ky = (o / 3) % 3, ci = (o / 9) % C, co = o / (9 * C);
        float s = 0.f;
        for (int cls = 0; cls < 4; ++cls) {
            const int py = cls >> 1, px = cls & 1;
            const int a = tap_in(py, 0, ky) ? 0 : 1, bb = tap_in(px, 0, kx) ? 0 : 1;
            s += gsum[((size_t)(cls * 4 + a * 2 + bb) * C + ci) * C + co];
        }
        dw[o] = accumulate ? dw[o] + s : s;
    }
}

}  // namespace

extern "C" size_t codeps_upconv16_workspace_bytes(void) { return (size_t)NWC * sizeof(float); }

extern "C" int codeps_upconv16_fwd(const float* x, const float* weight, const float* bias, int batch, int hs, int ws, int act, float* y,
                                   void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(x && weight && y && workspace && batch > 0 && hs > 0 && ws > 0 && act >= 0 && act <= 2, CODEPS_ERR_ARG, "codeps_upconv16_fwd: bad arguments");
    CODEPS_REQUIRE(workspace_bytes >= codeps_upconv16_workspace_bytes(), CODEPS_ERR_WORKSPACE, "codeps_upconv16_fwd: workspace too small");
    const int tiles_x = cdiv(ws, SW), tiles_y = cdiv(hs, SH);
    hipStream_t st = (hipStream_t)stream;
    float* wc = static_cast<float*>(workspace);
    hipLaunchKernelGGL(upconv16_collapse_kernel, dim3(16), dim3(256), 0, st, weight, wc);
    hipLaunchKernelGGL(upconv16_fwd_kernel, dim3(tiles_x * tiles_y, batch), dim3(256), 0, st, x, (const float*)wc, bias, hs, ws, tiles_x, act, y);
    CODEPS_CHECK_LAUNCH("upconv16_fwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_upconv16_dgrad_blocks(int batch, int hs, int ws) { return batch * cdiv(ws, SW) * cdiv(hs, SH); }

extern "C" int codeps_upconv16_dgrad(const float* dz, const float* weight, int batch, int hs, int ws, float* dx, void* workspace,
                                     size_t workspace_bytes, codeps_stream_t stream) {
    return codeps_upconv16_dgrad_gated(dz, weight, batch, hs, ws, dx, nullptr, 0, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int codeps_upconv16_dgrad_gated(const float* dz, const float* weight, int batch, int hs, int ws, float* dx, const float* x, int x_act,
                                           float* x_colsum_part, void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(x_act >= 0 && x_act <= 2 && (!x_act || x) && (!x_colsum_part || x_act), CODEPS_ERR_ARG, "codeps_upconv16_dgrad: bad gate arguments");
    CODEPS_REQUIRE(dz && weight && dx && workspace && batch > 0 && hs > 0 && ws > 0, CODEPS_ERR_ARG, "codeps_upconv16_dgrad: bad arguments");
    CODEPS_REQUIRE(workspace_bytes >= codeps_upconv16_workspace_bytes(), CODEPS_ERR_WORKSPACE, "codeps_upconv16_dgrad: workspace too small");
    const int tiles_x = cdiv(ws, SW), tiles_y = cdiv(hs, SH);
    const size_t lds = ((size_t)ZPIX * P4 * 4 + (size_t)SH * SW * P4 * 4) * sizeof(float);
    codeps_allow_lds((const void*)upconv16_dgrad_kernel, 160 * 1024, "upconv16_dgrad_kernel");
    hipStream_t st = (hipStream_t)stream;
    float* wc = static_cast<float*>(workspace);
    hipLaunchKernelGGL(upconv16_collapse_kernel, dim3(16), dim3(256), 0, st, weight, wc);
    hipLaunchKernelGGL(upconv16_dgrad_kernel, dim3(tiles_x * tiles_y, batch), dim3(256), lds, st, dz, (const float*)wc, hs, ws, tiles_x, dx, x, x_act,
                       x_colsum_part);
    CODEPS_CHECK_LAUNCH("upconv16_dgrad_kernel");
    return CODEPS_OK;
}

extern "C" size_t codeps_upconv16_wgrad_workspace_bytes(void) { return (size_t)(512 + 1) * NWC * sizeof(float); }

extern "C" int codeps_upconv16_wgrad(const float* x, const float* dz, int batch, int hs, int ws, float* dweight, int accumulate, void* workspace,
                                     size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(x && dz && dweight && workspace && batch > 0 && hs > 0 && ws > 0, CODEPS_ERR_ARG, "codeps_upconv16_wgrad: bad arguments");
    CODEPS_REQUIRE(workspace_bytes >= codeps_upconv16_wgrad_workspace_bytes(), CODEPS_ERR_WORKSPACE, "codeps_upconv16_wgrad: workspace too small");
    const int tiles_x = cdiv(ws, SW), tiles_y = cdiv(hs, SH);
    const int nblk = std::min(batch * tiles_x * tiles_y, 512);
    const size_t lds = ((size_t)XPIX * P4 * 4 + (size_t)OPIX * 17) * sizeof(float);
    codeps_allow_lds((const void*)upconv16_wgrad_kernel, 160 * 1024, "upconv16_wgrad_kernel");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(upconv16_wgrad_kernel, dim3(nblk), dim3(256), lds, st, x, dz, batch, hs, ws, tiles_x, tiles_y, static_cast<float*>(workspace));
    float* gsum = static_cast<float*>(workspace) + (size_t)512 * NWC;
    hipLaunchKernelGGL(upconv16_wgrad_sum_kernel, dim3(NWC / 16), dim3(256), 0, st, (const float*)workspace, nblk, gsum);
    hipLaunchKernelGGL(upconv16_wgrad_finish_kernel, dim3(9), dim3(256), 0, st, (const float*)gsum, accumulate, dweight);
    CODEPS_CHECK_LAUNCH("upconv16_wgrad");
    return CODEPS_OK;
}
