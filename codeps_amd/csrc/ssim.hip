// Stand-alone SSIMLoss (algos/depth.py:110-155): ReflectionPad2d(1) + 3x3 average pooling (stride 1) of x, y, x^2, y^2, xy;
// loss = clamp((1 - n / d) / 2, 0, 1) per pixel and channel, C1 = 1e-4, C2 = 9e-4.  The training path evaluates the same
// expression inside the fused photometric kernel (photometric.hip); this file serves the reference's public, callable and
// differentiable SSIMLoss.  HBM-bound: forward 8 B read + 4 B written per element; backward = one coefficient pass
// (12 B read, 16 B written) + one gather pass (9 x 16 B from L2 + 8 B read, 8 B written).
#include "common.h"

// The local variances are differences of nearly equal numbers (E[x^2] - mu^2 against C2 = 9e-4): a fused multiply-add in
// place of the reference's separately rounded products moves the result by 1e-5.  No contraction in this file -- every
// product and sum is rounded like the ATen ops it mirrors (x ** 2, avg_pool2d = sequential sum / 9, mu_x ** 2, ...).
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

struct SsimStats { float mx, my, n1, n2, d1, d2; };

__device__ __forceinline__ SsimStats ssim_stats(const float* __restrict__ x, const float* __restrict__ y, int H, int W, int py, int px,
                                                float c1, float c2) {
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int ry = reflect1(py + dy, H);
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int rx = reflect1(px + dx, W);
            const float a = x[ry * W + rx], b = y[ry * W + rx];
            sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
    }
    SsimStats s;
    s.mx = sx / 9.f; s.my = sy / 9.f;
    const float vx = sxx / 9.f - s.mx * s.mx, vy = syy / 9.f - s.my * s.my, cxy = sxy / 9.f - s.mx * s.my;
    s.n1 = 2.f * s.mx * s.my + c1; s.n2 = 2.f * cxy + c2;
    s.d1 = s.mx * s.mx + s.my * s.my + c1; s.d2 = vx + vy + c2;
    return s;
}

__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int planes, int H, int W,
                                                       float c1, float c2, float* __restrict__ out) {
    const int HW = H * W;
    const long long total = (long long)planes * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int pl = (int)(i / HW), p = (int)(i % HW);
        const SsimStats s = ssim_stats(x + (size_t)pl * HW, y + (size_t)pl * HW, H, W, p / W, p % W, c1, c2);
        const float v = (1.f - (s.n1 * s.n2) / (s.d1 * s.d2)) / 2.f;
        out[i] = fminf(fmaxf(v, 0.f), 1.f);
    }
}

// pass 1 of the backward: per output pixel the four adjoint coefficients (Ax, Ay, P, Q) with
//   d loss_p / d x_q = Ax + Q x_q + P y_q,   d loss_p / d y_q = Ay + P x_q + Q y_q     for every q in p's window
__global__ __launch_bounds__(256) void ssim_coef_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ gout, int planes, int H, int W, float c1, float c2,
                                                        float* __restrict__ coef) {
    const int HW = H * W;
    const long long total = (long long)planes * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int pl = (int)(i / HW), p = (int)(i % HW);
        const SsimStats s = ssim_stats(x + (size_t)pl * HW, y + (size_t)pl * HW, H, W, p / W, p % W, c1, c2);
        const float n = s.n1 * s.n2, d = s.d1 * s.d2;
        const float v = (1.f - n / d) / 2.f;
        const float k = (v >= 0.f && v <= 1.f) ? gout[i] * (-0.5f) / 9.f : 0.f;      // clamp passes the gradient on [0, 1]
        const float inv_d = 1.f / d, n_d2 = n * inv_d * inv_d;
        float4 c;
        c.x = k * (2.f * s.my * (s.n2 - s.n1) * inv_d - n_d2 * 2.f * s.mx * (s.d2 - s.d1));
        c.y = k * (2.f * s.mx * (s.n2 - s.n1) * inv_d - n_d2 * 2.f * s.my * (s.d2 - s.d1));
        c.z = k * 2.f * s.n1 * inv_d;
        c.w = -k * 2.f * n_d2 * s.d1;
        reinterpret_cast<float4*>(coef)[i] = c;
    }
}

// pass 2: every input pixel gathers the coefficients of the (up to 9) windows it belongs to, with the multiplicity the
// reflection padding gives it (row 1 is also row -1, row H-2 also row H)
__global__ __launch_bounds__(256) void ssim_gather_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ coef, int planes, int H, int W,
                                                          float* __restrict__ gx, float* __restrict__ gy) {
    const int HW = H * W;
    const long long total = (long long)planes * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int pl = (int)(i / HW), q = (int)(i % HW), qy = q / W, qx = q % W;
        float wy[3], wx[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int py = qy + a - 1, px = qx + a - 1;
            float cy = 0.f, cx = 0.f;
#pragma unroll
            for (int d = -1; d <= 1; ++d) {
                if (py >= 0 && py < H && reflect1(py + d, H) == qy) cy += 1.f;
                if (px >= 0 && px < W && reflect1(px + d, W) == qx) cx += 1.f;
            }
            wy[a] = cy; wx[a] = cx;
        }
        float sax = 0.f, say = 0.f, sp = 0.f, sq = 0.f;
        const float4* cp = reinterpret_cast<const float4*>(coef) + (size_t)pl * HW;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (wy[a] == 0.f) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float m = wy[a] * wx[b];
                if (m == 0.f) continue;
                const float4 c = cp[(qy + a - 1) * W + (qx + b - 1)];
                sax += m * c.x; say += m * c.y; sp += m * c.z; sq += m * c.w;
            }
        }
        const float xv = x[i], yv = y[i];
        if (gx) gx[i] = sax + sq * xv + sp * yv;
        if (gy) gy[i] = say + sp * xv + sq * yv;
    }
}

static inline int grid_of(long long n) { return (int)std::min<long long>((n + 255) / 256, 4096); }

}  // namespace

extern "C" int codeps_ssim_fwd(const float* x, const float* y, int planes, int h, int w, float c1, float c2, float* out,
                               codeps_stream_t stream) {
    CODEPS_REQUIRE(x && y && out && planes > 0 && h >= 2 && w >= 2, CODEPS_ERR_ARG, "ssim_fwd: bad arguments (reflection padding needs >= 2x2)");
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(grid_of((long long)planes * h * w)), dim3(256), 0, (hipStream_t)stream, x, y, planes, h, w, c1, c2, out);
    CODEPS_CHECK_LAUNCH("ssim_fwd_kernel");
    return CODEPS_OK;
}

extern "C" size_t codeps_ssim_bwd_workspace_bytes(int planes, int h, int w) {
    return planes > 0 && h > 0 && w > 0 ? (size_t)planes * h * w * 4 * sizeof(float) : 0;
}

extern "C" int codeps_ssim_bwd(const float* x, const float* y, const float* grad_out, int planes, int h, int w, float c1, float c2,
                               float* grad_x, float* grad_y, void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(x && y && grad_out && (grad_x || grad_y) && planes > 0 && h >= 2 && w >= 2, CODEPS_ERR_ARG, "ssim_bwd: bad arguments");
    CODEPS_REQUIRE(workspace && workspace_bytes >= codeps_ssim_bwd_workspace_bytes(planes, h, w), CODEPS_ERR_WORKSPACE, "ssim_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int g = grid_of((long long)planes * h * w);
    hipLaunchKernelGGL(ssim_coef_kernel, dim3(g), dim3(256), 0, st, x, y, grad_out, planes, h, w, c1, c2, (float*)workspace);
    hipLaunchKernelGGL(ssim_gather_kernel, dim3(g), dim3(256), 0, st, x, y, (const float*)workspace, planes, h, w, grad_x, grad_y);
    CODEPS_CHECK_LAUNCH("ssim_bwd kernels");
    return CODEPS_OK;
}
