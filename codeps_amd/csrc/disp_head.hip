// The disparity output of the depth decoder in one pass each way (models/depth_head.py:44-49,73-77: ReflectionPad2d(1) ->
// Conv2d(16, 1, 3) -> Sigmoid -> disp_to_depth).  Through the generic conv engine this layer wrote a 16-channel padded logit
// tensor of which one channel is real (201 MB per 6 images at 512 x 1024), read it back for the sigmoid, and mirrored all of
// it in the backward pass (padded logit gradient, data gradient w.r.t. the reflection-padded tensor + fold, weight gradient on
// 15/16 zero columns): 1.3 ms per adaptation step for 0.03 GFLOP of real work.  Here every kernel moves the 16-channel
// activation tensor exactly once (read forward, written in the data gradient, read in the weight gradient) and one float per
// pixel otherwise -- plain fp32 FMAs, HBM-bound:
//   disp_head_fwd_kernel   8 x 32 pixel tile with a reflected halo in LDS (80-byte pixel pitch: conflict-free 16-byte reads),
//                          one thread per pixel: 144 FMAs, sigmoid, disp_to_depth
//   disp_head_g_kernel     dlogit = (ddisp + ddepth d depth / d disp) s (1 - s)
//   disp_head_dx_kernel    gather form of the transposed reflection-padded convolution: a border pixel also collects what the
//                          reference adds back from the padding ring
//   disp_head_dw_kernel    persistent blocks, 144 + 1 register accumulators per thread over its tiles, one fixed-order block
//                          reduction, partials summed by disp_head_dw_finish_kernel in block order (deterministic)
#include <algorithm>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C = 16, C4 = C / 4;
constexpr int TH = 8, TW = 32, HH = TH + 2, HW = TW + 2, HPIX = HH * HW, PITCH4 = 5;      // pixel pitch: 5 float4 = 80 bytes
constexpr int NW = 9 * C;                                                                 // weights

__device__ __forceinline__ int reflect1(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * (n - 1) - i : i;
    return min(max(i, 0), n - 1);                   // tiles overhanging the image: any valid pixel (never used)
}

// x [B][H][W][16] halo tile of (y0 - 1 .., x0 - 1 ..) into LDS, reflected at the image border
// (all loads of a thread are issued into registers before the first LDS store: as a load -> store loop every element waited for its own
// memory latency, six in a row per tile -- disp_head_dw_kernel spent 129 us on 214 MB)
__device__ __forceinline__ void load_halo(const float* __restrict__ x, int b, int H, int W, int y0, int x0, f32x4* xs) {
    constexpr int NL = (HPIX * C4 + 255) / 256;
    f32x4 r[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int i = min((int)threadIdx.x + 256 * u, HPIX * C4 - 1);
        const int px = i / C4, q = i % C4;
        const int iy = reflect1(y0 - 1 + px / HW, H), ix = reflect1(x0 - 1 + px % HW, W);
        r[u] = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + iy) * W + ix) * C + q * 4);
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i < HPIX * C4) xs[(i / C4) * PITCH4 + i % C4] = r[u];
    }
}

// OIHW [1][16][3][3] -> ws[tap][16] in LDS
__device__ __forceinline__ void load_weights(const float* __restrict__ w, float* ws) {
    for (int i = threadIdx.x; i < NW; i += 256) ws[i] = w[(i % C) * 9 + i / C];
}

__global__ __launch_bounds__(256) void disp_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            int H, int W, int tiles_x, float min_disp, float max_disp,
                                                            float* __restrict__ disp, float* __restrict__ depth, float* __restrict__ logit) {
    __shared__ f32x4 xs[HPIX * PITCH4];
    __shared__ __attribute__((aligned(16))) float ws[NW];
    const int b = blockIdx.y, y0 = (blockIdx.x / tiles_x) * TH, x0 = (blockIdx.x % tiles_x) * TW;
    load_halo(x, b, H, W, y0, x0, xs);
    load_weights(w, ws);
    __syncthreads();
    const int ty = threadIdx.x / TW, tx = threadIdx.x % TW, y = y0 + ty, xx = x0 + tx;
    if (y >= H || xx >= W) return;
    float acc = bias ? bias[0] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const f32x4* px = xs + ((ty + t / 3) * HW + tx + t % 3) * PITCH4;
        const f32x4* wt = reinterpret_cast<const f32x4*>(ws + t * C);
#pragma unroll
        for (int q = 0; q < C4; ++q) {
            const f32x4 a = px[q], k = wt[q];
            acc += a[0] * k[0] + a[1] * k[1] + a[2] * k[2] + a[3] * k[3];
        }
    }
    const size_t r = ((size_t)b * H + y) * W + xx;
    const float s = 1.f / (1.f + expf(-acc));
    if (logit) logit[r] = acc;          // the pre-sigmoid value (models/depth_head.py:73), for parity checks in logit space
    disp[r] = s;
    depth[r] = 1.f / (min_disp + (max_disp - min_disp) * s);
}

__global__ __launch_bounds__(256) void disp_head_g_kernel(const float* __restrict__ ddisp, const float* __restrict__ ddepth,
                                                          const float* __restrict__ disp, long long M, float min_disp, float max_disp,
                                                          float* __restrict__ g) {
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < M; r += (long long)gridDim.x * 256) {
        const float s = disp[r];
        const float sd = min_disp + (max_disp - min_disp) * s;
        float v = 0.f;
        if (ddisp) v += ddisp[r];
        if (ddepth) v += ddepth[r] * (-(max_disp - min_disp) / (sd * sd));
        g[r] = v * s * (1.f - s);
    }
}

// dx[q][c] = sum over padded positions u that reflect onto q and taps t with p = u - t inside the image of g[p] w[t][c]
// x_act != 0 (round 5): x is the OUTPUT of an activation (the decoder's last ELU, models/depth_head.py:63-71) read by this head alone: dx is
// multiplied by act'(x) here -- the producer's activation backward needs no pass over the full-resolution tensor -- and the block's column
// sums of the gated dx (the producer's bias gradient) go to col_part[block][16]
__global__ __launch_bounds__(256) void disp_head_dx_kernel(const float* __restrict__ g, const float* __restrict__ w, int B, int H, int W,
                                                           float* __restrict__ dx, const float* __restrict__ x, int x_act,
                                                           float* __restrict__ col_part) {
    // one thread per (pixel, channel quad): four neighbouring lanes write one pixel's 64 bytes, a wave 1 KB in memory order (round 5; a
    // thread per pixel wrote its four quads as four instructions of 16 bytes at a 64-byte lane stride -- 194 us for 430 MB)
    __shared__ __attribute__((aligned(16))) float ws[NW];
    __shared__ float cred[4][C];
    load_weights(w, ws);
    __syncthreads();
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    const long long n4 = (long long)B * H * W * C4;
    const int q = threadIdx.x & (C4 - 1);                       // (256 and the grid stride are multiples of C4: a thread keeps its quad)
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < n4; j += (long long)gridDim.x * 256) {
        const long long i = j / C4;
        const int xq = (int)(i % W), yq = (int)((i / W) % H);
        const float* gb = g + (i / ((long long)H * W)) * (long long)H * W;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        // padded rows / columns that land on this pixel: itself, row -1 (on row 1), row H (on row H - 2)
        const int uys[3] = {yq, yq == 1 ? -1 : -2, yq == H - 2 ? H : -2};
        const int uxs[3] = {xq, xq == 1 ? -1 : -2, xq == W - 2 ? W : -2};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int uy = uys[a];
            if (uy == -2) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int ux = uxs[c];
                if (ux == -2) continue;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int py = uy - (t / 3 - 1), px = ux - (t % 3 - 1);
                    if (py < 0 || py >= H || px < 0 || px >= W) continue;
                    const float gv = gb[(size_t)py * W + px];
                    acc += gv * reinterpret_cast<const f32x4*>(ws + t * C)[q];
                }
            }
        }
        if (x_act) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(x + j * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] *= y[e] > 0.f ? 1.f : (x_act == 2 ? y[e] + 1.f : 0.f);       // = act_bwd_kernel's product
            csum += acc;
        }
        *reinterpret_cast<f32x4*>(dx + j * 4) = acc;
    }
    if (!col_part) return;
    // lanes with the same quad meet by xor shuffles (fixed order), then the four waves through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v = csum[e];
#pragma unroll
        for (int o = C4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        if (lane < C4) cred[wave][lane * 4 + e] = v;
    }
    __syncthreads();
    if (threadIdx.x < C) col_part[(size_t)blockIdx.x * C + threadIdx.x] = (cred[0][threadIdx.x] + cred[1][threadIdx.x]) + (cred[2][threadIdx.x] + cred[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void disp_head_dw_kernel(const float* __restrict__ x, const float* __restrict__ g, int B, int H, int W,
                                                           int tiles_x, int tiles_y, float* __restrict__ part) {
    __shared__ f32x4 xs[HPIX * PITCH4];
    __shared__ float red[4][NW + 1];
    const int ty = threadIdx.x / TW, tx = threadIdx.x % TW, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[9][C4];
    float accb = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < C4; ++q) acc[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntiles = B * tiles_y * tiles_x;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / (tiles_y * tiles_x), tr = tile % (tiles_y * tiles_x);
        const int y0 = (tr / tiles_x) * TH, x0 = (tr % tiles_x) * TW;
        __syncthreads();                                 // the previous tile's reads are done
        load_halo(x, b, H, W, y0, x0, xs);
        __syncthreads();
        const int y = y0 + ty, xx = x0 + tx;
        const float gv = (y < H && xx < W) ? g[((size_t)b * H + y) * W + xx] : 0.f;
        accb += gv;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const f32x4* px = xs + ((ty + t / 3) * HW + tx + t % 3) * PITCH4;
#pragma unroll
            for (int q = 0; q < C4; ++q) acc[t][q] += gv * px[q];
        }
    }
    // block reduction in a fixed order: 64 lanes by shuffles, then the four waves
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < C4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = wave_sum(acc[t][q][e]);
                if (lane == 0) red[wave][t * C + q * 4 + e] = v;
            }
    {
        const float v = wave_sum(accb);
        if (lane == 0) red[wave][NW] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= NW; i += 256) part[(size_t)blockIdx.x * (NW + 1) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// partials [nblk][145] (tap-major weights, then the bias) -> OIHW weight gradient [1][16][3][3] and the bias gradient
// One block per output (144 weights + the bias): thread t adds partials t, t + 256, ... in order, the 256 threads meet in a fixed-order tree.
// (Until round 6 ONE block of 145 threads walked all <= 512 partials serially, every load waiting for the one before: 112 us per step for
// 290 KB.)
__global__ __launch_bounds__(256) void disp_head_dw_finish_kernel(const float* __restrict__ part, int nblk, int accumulate, float* __restrict__ dw,
                                                                  float* __restrict__ dbias) {
    __shared__ float red[256];
    const int i = blockIdx.x;
    float a = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 256) a += part[(size_t)b * (NW + 1) + i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    a = red[0];
    if (i == NW) {
        if (dbias) dbias[0] = accumulate ? dbias[0] + a : a;
        return;
    }
    if (!dw) return;
    const int o = (i % C) * 9 + i / C;
    dw[o] = accumulate ? dw[o] + a : a;
}

}  // namespace

extern "C" int codeps_disp_head_fwd(const float* x, const float* weight, const float* bias, int batch, int h, int w, int c, float min_depth,
                                    float max_depth, float* disp, float* depth, float* logit, codeps_stream_t stream) {
    CODEPS_REQUIRE(x && weight && disp && depth && batch > 0 && h >= 2 && w >= 2, CODEPS_ERR_ARG, "codeps_disp_head_fwd: bad arguments");
    CODEPS_REQUIRE(c == C, CODEPS_ERR_ARG, "codeps_disp_head_fwd: %d input channels (the decoder's last stage has %d)", c, C);
    const int tiles_x = cdiv(w, TW), tiles_y = cdiv(h, TH);
    hipLaunchKernelGGL(disp_head_fwd_kernel, dim3(tiles_x * tiles_y, batch), dim3(256), 0, (hipStream_t)stream, x, weight, bias, h, w, tiles_x,
                       (float)(1.0 / (double)max_depth), (float)(1.0 / (double)min_depth), disp, depth, logit);
    CODEPS_CHECK_LAUNCH("disp_head_fwd_kernel");
    return CODEPS_OK;
}

extern "C" size_t codeps_disp_head_bwd_workspace_bytes(int batch, int h, int w) {
    return batch > 0 ? align_up((size_t)batch * h * w * sizeof(float), 256) + (size_t)512 * (NW + 1) * sizeof(float) : 0;
}

extern "C" int codeps_disp_head_dx_blocks(int batch, int h, int w) {
    return (int)std::min<long long>(((long long)batch * h * w + 255) / 256, 4096);
}

extern "C" int codeps_disp_head_bwd(const float* x, const float* weight, const float* disp, const float* ddisp, const float* ddepth, int batch,
                                    int h, int w, int c, float min_depth, float max_depth, float* dx, float* dweight, float* dbias,
                                    int accumulate, void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    return codeps_disp_head_bwd_gated(x, weight, disp, ddisp, ddepth, batch, h, w, c, min_depth, max_depth, dx, dweight, dbias, accumulate, 0, nullptr,
                                      workspace, workspace_bytes, stream);
}

extern "C" int codeps_disp_head_bwd_gated(const float* x, const float* weight, const float* disp, const float* ddisp, const float* ddepth, int batch,
                                          int h, int w, int c, float min_depth, float max_depth, float* dx, float* dweight, float* dbias,
                                          int accumulate, int x_act, float* x_colsum_part, void* workspace, size_t workspace_bytes,
                                          codeps_stream_t stream) {
    CODEPS_REQUIRE(x_act >= 0 && x_act <= 2 && (!x_colsum_part || (x_act && dx)), CODEPS_ERR_ARG, "codeps_disp_head_bwd: bad gate arguments");
    CODEPS_REQUIRE(x && weight && disp && (ddisp || ddepth) && workspace && batch > 0 && h >= 2 && w >= 2, CODEPS_ERR_ARG,
                   "codeps_disp_head_bwd: bad arguments");
    CODEPS_REQUIRE(c == C, CODEPS_ERR_ARG, "codeps_disp_head_bwd: %d input channels (the decoder's last stage has %d)", c, C);
    CODEPS_REQUIRE(workspace_bytes >= codeps_disp_head_bwd_workspace_bytes(batch, h, w), CODEPS_ERR_WORKSPACE, "codeps_disp_head_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const long long M = (long long)batch * h * w;
    float* g = static_cast<float*>(workspace);
    float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + align_up((size_t)M * sizeof(float), 256));
    const float mind = (float)(1.0 / (double)max_depth), maxd = (float)(1.0 / (double)min_depth);
    const int grid = (int)std::min<long long>((M + 255) / 256, 4096);
    hipLaunchKernelGGL(disp_head_g_kernel, dim3(grid), dim3(256), 0, st, ddisp, ddepth, disp, M, mind, maxd, g);
    if (dx) hipLaunchKernelGGL(disp_head_dx_kernel, dim3(grid), dim3(256), 0, st, (const float*)g, weight, batch, h, w, dx, x, x_act, x_colsum_part);
    if (dweight || dbias) {
        const int tiles_x = cdiv(w, TW), tiles_y = cdiv(h, TH);
        const int nblk = std::min(batch * tiles_x * tiles_y, 512);
        hipLaunchKernelGGL(disp_head_dw_kernel, dim3(nblk), dim3(256), 0, st, x, (const float*)g, batch, h, w, tiles_x, tiles_y, part);
        hipLaunchKernelGGL(disp_head_dw_finish_kernel, dim3(NW + 1), dim3(256), 0, st, (const float*)part, nblk, accumulate, dweight, dbias);
    }
    CODEPS_CHECK_LAUNCH("codeps_disp_head_bwd");
    return CODEPS_OK;
}
