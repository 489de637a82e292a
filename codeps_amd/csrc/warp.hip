// Stand-alone image warper (misc/image_warper.py:100-184): back-project with depth, rigid transform,
// project, bilinear / nearest grid-sample with border padding and align_corners=True, for C-channel NCHW
// images.  The training path never calls this (the fused photometric kernel has its own copy of the same
// arithmetic); it exists for the reference's public ImageWarper API (visualisation, mixup warps).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ src, const float* __restrict__ depth,
                                                   const float* __restrict__ pose, const float* __restrict__ intr, int B, int C, int H,
                                                   int W, int nearest, float* __restrict__ out, float* __restrict__ grid) {
    const int HW = H * W;
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / HW), p = (int)(i % HW), gx = p % W, gy = p / W;
        const float fx = intr[b * 4], fy = intr[b * 4 + 1], cx = intr[b * 4 + 2], cy = intr[b * 4 + 3];
        const float* T = pose + b * 16;
        float rx = ((float)gx - cx) / fx, ry = ((float)gy - cy) / fy;
        const float norm = sqrtf(rx * rx + ry * ry + 1.0f);
        rx = rx / norm; ry = ry / norm;
        const float rz = 1.0f / norm;
        const float dz = depth[i] / fabsf(rz);
        const float px = dz * rx, py = dz * ry, pz = dz * rz;
        const float X = T[0] * px + T[1] * py + T[2] * pz + T[3], Y = T[4] * px + T[5] * py + T[6] * pz + T[7];
        const float Z = T[8] * px + T[9] * py + T[10] * pz + T[11], Wh = T[12] * px + T[13] * py + T[14] * pz + T[15];
        const float zc = fmaxf(Z / Wh, 1e-5f);
        const float u2 = ((X / Wh) / zc) * fx + cx, v2 = ((Y / Wh) / zc) * fy + cy;
        const float un = (u2 / (float)(W - 1) - 0.5f) * 2.f, vn = (v2 / (float)(H - 1) - 0.5f) * 2.f;
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

}  // namespace

extern "C" int codeps_image_warp(const float* src, const float* depth, const float* pose, const float* intr, int batch, int c, int h,
                                 int w, int nearest, float* out, float* grid, codeps_stream_t stream) {
    CODEPS_REQUIRE(depth && pose && intr && (out || grid) && (!out || src) && batch > 0 && c > 0 && h > 1 && w > 1, CODEPS_ERR_ARG,
                   "image_warp: bad arguments");
    const long long n = (long long)batch * h * w;
    hipLaunchKernelGGL(warp_kernel, dim3((int)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, src, depth, pose,
                       intr, batch, c, h, w, nearest, out, grid);
    CODEPS_CHECK_LAUNCH("warp_kernel");
    return CODEPS_OK;
}
