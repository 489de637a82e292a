// Stand-alone image warper (misc/image_warper.py:100-184): back-project with depth, rigid transform,
// project, bilinear / nearest grid-sample with border padding and align_corners=True, for C-channel NCHW
// images.  The training path never calls this (the fused photometric kernel has its own copy of the same
// arithmetic); it exists for the reference's public ImageWarper API (visualisation, mixup warps).
#include "common.h"

namespace {

struct WarpPoint {                       // forward chain of one pixel (misc/image_warper.py:83-85,127-139, camera_model.py:37-71)
    float q0, q1, q2;                    // d point / d depth
    float px, py, pz;                    // back-projected point
    float Wh, xh, yh, zh, zc;            // homogeneous division and the clamped depth
    float u2, v2, un, vn;
};

__device__ __forceinline__ WarpPoint warp_point(const float* __restrict__ T, float fx, float fy, float cx, float cy, int gx, int gy,
                                                float depth, const float* __restrict__ motion, size_t mo, int HW, int W, int H) {
    WarpPoint r;
    float rx = ((float)gx - cx) / fx, ry = ((float)gy - cy) / fy;
    const float norm = sqrtf(rx * rx + ry * ry + 1.0f);
    rx = rx / norm; ry = ry / norm;
    const float rz = 1.0f / norm;
    const float inv = 1.0f / fabsf(rz);
    r.q0 = inv * rx; r.q1 = inv * ry; r.q2 = inv * rz;
    const float dz = depth / fabsf(rz);
    r.px = dz * rx; r.py = dz * ry; r.pz = dz * rz;
    float X = T[0] * r.px + T[1] * r.py + T[2] * r.pz + T[3], Y = T[4] * r.px + T[5] * r.py + T[6] * r.pz + T[7];
    float Z = T[8] * r.px + T[9] * r.py + T[10] * r.pz + T[11];
    r.Wh = T[12] * r.px + T[13] * r.py + T[14] * r.pz + T[15];
    if (motion) { X += motion[mo]; Y += motion[mo + HW]; Z += motion[mo + 2 * (size_t)HW]; }   // image_warper.py:133-134
    r.xh = X / r.Wh; r.yh = Y / r.Wh; r.zh = Z / r.Wh;
    r.zc = fmaxf(r.zh, 1e-5f);
    r.u2 = (r.xh / r.zc) * fx + cx; r.v2 = (r.yh / r.zc) * fy + cy;
    r.un = (r.u2 / (float)(W - 1) - 0.5f) * 2.f; r.vn = (r.v2 / (float)(H - 1) - 0.5f) * 2.f;
    return r;
}

__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                   const float* __restrict__ pose, const float* __restrict__ intr,
                                                   const float* __restrict__ motion, int B, int C, int H,
                                                   int W, int nearest, float* __restrict__ out, float* __restrict__ grid) {
    const int HW = H * W;
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / HW), p = (int)(i % HW), gx = p % W, gy = p / W;
        const float fx = intr[b * 4], fy = intr[b * 4 + 1], cx = intr[b * 4 + 2], cy = intr[b * 4 + 3];
        const WarpPoint r = warp_point(pose + b * 16, fx, fy, cx, cy, gx, gy, depth[i], motion, (size_t)b * 3 * HW + p, HW, W, H);
        const float un = r.un, vn = r.vn;
        if (grid) { grid[i * 2] = un; grid[i * 2 + 1] = vn; }
        if (!out) continue;
        float ix = ((un + 1.f) / 2.f) * (float)(W - 1), iy = ((vn + 1.f) / 2.f) * (float)(H - 1);
        ix = !(ix > 0.f) ? 0.f : fminf(ix, (float)(W - 1));
        iy = !(iy > 0.f) ? 0.f : fminf(iy, (float)(H - 1));
        const float* sb = src + (size_t)b * C * HW;
        float* ob = out + (size_t)b * C * HW;
        if (nearest) {
            const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
            for (int c = 0; c < C; ++c) ob[(size_t)c * HW + p] = sb[(size_t)c * HW + yn * W + xn];
        } else {
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const int x0 = (int)fx0, y0 = (int)fy0, x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
            const float tx = ix - fx0, ty = iy - fy0;
            const float wx1 = (x0 + 1 < W) ? tx : 0.f, wy1 = (y0 + 1 < H) ? ty : 0.f;
            for (int c = 0; c < C; ++c) {
                const float* s = sb + (size_t)c * HW;
                ob[(size_t)c * HW + p] = s[y0 * W + x0] * ((1.f - tx) * (1.f - ty)) + s[y0 * W + x1] * (wx1 * (1.f - ty)) +
                                         s[y1 * W + x0] * ((1.f - tx) * wy1) + s[y1 * W + x1] * (wx1 * wy1);
            }
        }
    }
}

// Transpose of the warp: d/d src (scatter: four atomic adds per pixel and channel -- the one place in the library where float
// atomics are used; this is the public ImageWarper API, the training path's warp lives in photometric.hip and has none),
// d/d depth, d/d object motion, and per-block partial sums of d/d T.  grid = (blocks per image, B).
constexpr int WB = 64;                   // blocks per image for the pose partials
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                       const float* __restrict__ pose, const float* __restrict__ intr,
                                                       const float* __restrict__ motion, const float* __restrict__ gout,
                                                       const float* __restrict__ ggrid, int C, int H, int W, int nearest,
                                                       float* __restrict__ gsrc, float* __restrict__ gdepth,
                                                       float* __restrict__ gmotion, float* __restrict__ pose_part) {
    __shared__ float red[4];
    const int HW = H * W, b = blockIdx.y;
    const float fx = intr[b * 4], fy = intr[b * 4 + 1], cx = intr[b * 4 + 2], cy = intr[b * 4 + 3];
    const float* T = pose + b * 16;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const int gx = p % W, gy = p / W;
        const size_t i = (size_t)b * HW + p;
        const WarpPoint r = warp_point(T, fx, fy, cx, cy, gx, gy, depth[i], motion, (size_t)b * 3 * HW + p, HW, W, H);
        float g_u2 = 0.f, g_v2 = 0.f;
        if (ggrid) { g_u2 += ggrid[i * 2] * (2.f / (float)(W - 1)); g_v2 += ggrid[i * 2 + 1] * (2.f / (float)(H - 1)); }
        if (gout) {
            float ix = ((r.un + 1.f) / 2.f) * (float)(W - 1), iy = ((r.vn + 1.f) / 2.f) * (float)(H - 1);
            const bool in_x = ix > 0.f && ix < (float)(W - 1), in_y = iy > 0.f && iy < (float)(H - 1);   // border clip: zero gradient outside
            ix = !(ix > 0.f) ? 0.f : fminf(ix, (float)(W - 1));
            iy = !(iy > 0.f) ? 0.f : fminf(iy, (float)(H - 1));
            const float* sb = src ? src + (size_t)b * C * HW : nullptr;
            const float* gb = gout + (size_t)b * C * HW;
            float* gs = gsrc ? gsrc + (size_t)b * C * HW : nullptr;
            if (nearest) {
                const int xn = (int)nearbyintf(ix), yn = (int)nearbyintf(iy);
                if (gs) for (int c = 0; c < C; ++c) atomicAdd(gs + (size_t)c * HW + yn * W + xn, gb[(size_t)c * HW + p]);
            } else {
                const float fx0 = floorf(ix), fy0 = floorf(iy);
                const int x0 = (int)fx0, y0 = (int)fy0, x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
                const float tx = ix - fx0, ty = iy - fy0;
                const bool hx = x0 + 1 < W, hy = y0 + 1 < H;
                const float wx1 = hx ? tx : 0.f, wy1 = hy ? ty : 0.f;
                float gix = 0.f, giy = 0.f;
                for (int c = 0; c < C; ++c) {
                    const float g = gb[(size_t)c * HW + p];
                    if (gs) {
                        float* d = gs + (size_t)c * HW;
                        atomicAdd(d + y0 * W + x0, g * ((1.f - tx) * (1.f - ty)));
                        if (hx) atomicAdd(d + y0 * W + x1, g * (wx1 * (1.f - ty)));
                        if (hy) atomicAdd(d + y1 * W + x0, g * ((1.f - tx) * wy1));
                        if (hx && hy) atomicAdd(d + y1 * W + x1, g * (wx1 * wy1));
                    }
                    if (sb) {
                        const float* s = sb + (size_t)c * HW;
                        const float s00 = s[y0 * W + x0], s01 = hx ? s[y0 * W + x1] : 0.f, s10 = hy ? s[y1 * W + x0] : 0.f,
                                    s11 = (hx && hy) ? s[y1 * W + x1] : 0.f;
                        // d out / d ix with the out-of-range corners contributing value 0 (torch grid_sampler_2d_backward)
                        gix += g * ((s01 - s00) * (1.f - ty) + (s11 - s10) * ty);
                        giy += g * ((s10 - s00) * (1.f - tx) + (s11 - s01) * tx);
                    }
                }
                // un -> ix is ((un + 1) / 2) (W - 1), u2 -> un is 2 / (W - 1): the product is 1 up to rounding
                if (in_x) g_u2 += gix * ((float)(W - 1) / 2.f) * (2.f / (float)(W - 1));
                if (in_y) g_v2 += giy * ((float)(H - 1) / 2.f) * (2.f / (float)(H - 1));
            }
        }
        // (u2, v2) <- (xh, yh, zc) <- (X, Y, Z, Wh) <- (T, point) <- depth
        const float g_xh = g_u2 * fx / r.zc, g_yh = g_v2 * fy / r.zc;
        const float g_zc = -(g_u2 * fx * r.xh + g_v2 * fy * r.yh) / (r.zc * r.zc);
        const float g_zh = (r.zh >= 1e-5f) ? g_zc : 0.f;                 // clamp(min=1e-5) passes the gradient where z >= min
        const float gX = g_xh / r.Wh, gY = g_yh / r.Wh, gZ = g_zh / r.Wh;
        const float gW = -(g_xh * r.xh + g_yh * r.yh + g_zh * r.zh) / r.Wh;
        if (gmotion) { const size_t mo = (size_t)b * 3 * HW + p; gmotion[mo] = gX; gmotion[mo + HW] = gY; gmotion[mo + 2 * (size_t)HW] = gZ; }
        const float gpx = T[0] * gX + T[4] * gY + T[8] * gZ + T[12] * gW, gpy = T[1] * gX + T[5] * gY + T[9] * gZ + T[13] * gW;
        const float gpz = T[2] * gX + T[6] * gY + T[10] * gZ + T[14] * gW;
        if (gdepth) gdepth[i] = gpx * r.q0 + gpy * r.q1 + gpz * r.q2;
        const float gr[4] = {gX, gY, gZ, gW}, pt[4] = {r.px, r.py, r.pz, 1.f};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[a * 4 + k] += gr[a] * pt[k];
    }
    if (!pose_part) return;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float t = block_sum<256>(acc[k], red);
        if (threadIdx.x == 0) pose_part[((size_t)b * gridDim.x + blockIdx.x) * 16 + k] = t;
    }
}

__global__ void warp_pose_reduce_kernel(const float* __restrict__ part, int nblk, float* __restrict__ gpose) {
    const int b = blockIdx.x, k = threadIdx.x;           // 16 threads, fixed summation order
    float s = 0.f;
    for (int j = 0; j < nblk; ++j) s += part[((size_t)b * nblk + j) * 16 + k];
    gpose[b * 16 + k] = s;
}

}  // namespace

extern "C" int codeps_image_warp(const float* src, const float* depth, const float* pose, const float* intr, const float* motion,
                                 int batch, int c, int h, int w, int nearest, float* out, float* grid, codeps_stream_t stream) {
    CODEPS_REQUIRE(depth && pose && intr && (out || grid) && (!out || src) && batch > 0 && c > 0 && h > 1 && w > 1, CODEPS_ERR_ARG,
                   "image_warp: bad arguments");
    const long long n = (long long)batch * h * w;
    hipLaunchKernelGGL(warp_kernel, dim3((int)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, src, depth, pose,
                       intr, motion, batch, c, h, w, nearest, out, grid);
    CODEPS_CHECK_LAUNCH("warp_kernel");
    return CODEPS_OK;
}

extern "C" size_t codeps_image_warp_bwd_workspace_bytes(int batch) { return batch > 0 ? (size_t)batch * WB * 16 * sizeof(float) : 0; }

extern "C" int codeps_image_warp_bwd(const float* src, const float* depth, const float* pose, const float* intr, const float* motion,
                                     const float* grad_out, const float* grad_grid, int batch, int c, int h, int w, int nearest,
                                     float* grad_src, float* grad_depth, float* grad_pose, float* grad_motion, void* workspace,
                                     size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(depth && pose && intr && (grad_out || grad_grid) && batch > 0 && c > 0 && h > 1 && w > 1, CODEPS_ERR_ARG,
                   "image_warp_bwd: bad arguments");
    CODEPS_REQUIRE(!grad_out || src || nearest || !(grad_depth || grad_pose || grad_motion), CODEPS_ERR_ARG,
                   "image_warp_bwd: the coordinate gradient of the bilinear sample needs the source image");
    CODEPS_REQUIRE(!grad_pose || (workspace && workspace_bytes >= codeps_image_warp_bwd_workspace_bytes(batch)), CODEPS_ERR_WORKSPACE,
                   "image_warp_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(warp_bwd_kernel, dim3(WB, batch), dim3(256), 0, st, src, depth, pose, intr, motion, grad_out, grad_grid, c, h, w,
                       nearest, grad_src, grad_depth, grad_motion, grad_pose ? (float*)workspace : nullptr);
    CODEPS_CHECK_LAUNCH("warp_bwd_kernel");
    if (grad_pose) {
        hipLaunchKernelGGL(warp_pose_reduce_kernel, dim3(batch), dim3(16), 0, st, (const float*)workspace, WB, grad_pose);
        CODEPS_CHECK_LAUNCH("warp_pose_reduce_kernel");
    }
    return CODEPS_OK;
}
