// Weight re-layouts of the step as single launches (instead of pad / permute / clone chains of tensor ops, forward and backward):
//   * stem:  ResNet conv1 weight [co][ci <= 8][7][7] -> its double-pixel form [co][16][7][4]
//            W'[o][p*8 + c][ky][j] = W[o][c][ky][2j + p - 1]   (models/resnet_encoder.py:115 on the layout of F.stem_conv2d)
//   * GUDA:  semconv_guda weight [n][Ct][3][3] (models/semantic_head.py:64-72) -> per coarse branch k the 1x1 tap weights
//            [9 * 16][c_k] (row = tap * 16 + class, classes >= n zero) and the fine branch's own [n][c_f][3][3] block
// Both maps are injective on the source, so the backward is a gather as well (no atomics); `accumulate` adds into the
// flat gradient buffer of the step harness.
#include <algorithm>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void stem_weight_fwd_kernel(const float* __restrict__ w, int co, int ci, float* __restrict__ out) {
    const int total = co * 16 * 7 * 4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int j = i & 3, ky = (i >> 2) % 7, pc = (i / 28) & 15, o = i / (28 * 16);
        const int p = pc >> 3, c = pc & 7, kx = 2 * j + p - 1;
        out[i] = (kx >= 0 && c < ci) ? w[((o * ci + c) * 7 + ky) * 7 + kx] : 0.f;
    }
}

__global__ __launch_bounds__(256) void stem_weight_bwd_kernel(const float* __restrict__ gout, int co, int ci, int accumulate, float* __restrict__ gw) {
    const int total = co * ci * 49;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int kx = i % 7, ky = (i / 7) % 7, c = (i / 49) % ci, o = i / (49 * ci);
        const int p = (kx + 1) & 1, j = (kx + 1) >> 1;
        const float g = gout[((o * 16 + p * 8 + c) * 7 + ky) * 4 + j];
        gw[i] = accumulate ? gw[i] + g : g;
    }
}

constexpr int GMAX = 4;
struct GudaSeg { float* ptr[GMAX]; int c0[GMAX], c[GMAX]; int off[GMAX + 1]; int n_coarse; };

__global__ __launch_bounds__(256) void guda_weight_fwd_kernel(const float* __restrict__ w, int n, int ct, GudaSeg s) {
    const int total = s.off[s.n_coarse + 1];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int k = 0;
        while (k < s.n_coarse && i >= s.off[k + 1]) ++k;
        const int r = i - s.off[k];
        if (k < s.n_coarse) {                       // taps_k [tap * 16 + cls][c]
            const int c = r % s.c[k], row = r / s.c[k], cls = row & 15, tap = row >> 4;
            s.ptr[k][r] = cls < n ? w[((size_t)cls * ct + s.c0[k] + c) * 9 + tap] : 0.f;
        } else {                                    // base [cls][c][3][3]
            const int tap = r % 9, c = (r / 9) % s.c[k], cls = r / (9 * s.c[k]);
            s.ptr[k][r] = w[((size_t)cls * ct + s.c0[k] + c) * 9 + tap];
        }
    }
}

__global__ __launch_bounds__(256) void guda_weight_bwd_kernel(GudaSeg s, int n, int ct, int accumulate, float* __restrict__ gw) {
    const int total = n * ct * 9;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int tap = i % 9, ch = (i / 9) % ct, cls = i / (9 * ct);
        float g = 0.f;
        for (int k = 0; k <= s.n_coarse; ++k) {
            if (ch < s.c0[k] || ch >= s.c0[k] + s.c[k] || !s.ptr[k]) continue;
            const int c = ch - s.c0[k];
            g = k < s.n_coarse ? s.ptr[k][(size_t)(tap * 16 + cls) * s.c[k] + c] : s.ptr[k][((size_t)cls * s.c[k] + c) * 9 + tap];
        }
        gw[i] = accumulate ? gw[i] + g : g;
    }
}

static int fill_seg(GudaSeg& s, int n, int ct, int n_coarse, const int* channels, float* const* ptrs, const char* who) {
    CODEPS_REQUIRE(n >= 1 && n <= 16 && n_coarse >= 0 && n_coarse < GMAX && channels && ptrs, CODEPS_ERR_ARG, "%s: bad arguments", who);
    int c0 = 0;
    s.n_coarse = n_coarse;
    s.off[0] = 0;
    for (int k = 0; k <= n_coarse; ++k) {
        CODEPS_REQUIRE(channels[k] > 0, CODEPS_ERR_ARG, "%s: branch %d has no channels", who, k);
        s.ptr[k] = ptrs[k]; s.c0[k] = c0; s.c[k] = channels[k];
        s.off[k + 1] = s.off[k] + (k < n_coarse ? 144 * channels[k] : n * channels[k] * 9);
        c0 += channels[k];
    }
    for (int k = n_coarse + 1; k < GMAX; ++k) { s.ptr[k] = nullptr; s.c0[k] = 0; s.c[k] = 0; }
    CODEPS_REQUIRE(c0 == ct, CODEPS_ERR_ARG, "%s: branch channels sum to %d, the weight has %d", who, c0, ct);
    return CODEPS_OK;
}


// Data gradient of "nearest x2 -> conv3x3 (padding 1)" w.r.t. the half-resolution tensor = a 4 x 4, stride-2, padding-1 FORWARD
// convolution of dz: the 36 (sub-pixel, tap) pairs that reach a source pixel fall on 16 distinct output pixels 2 q + d,
// d in {-1, 0, 1, 2} per axis, with the taps that coincide pre-summed: d = -1: W[2]; 0: W[1] + W[2]; 1: W[0] + W[1]; 2: W[0].
// w [co][ci_total][3][3] (the up-sampled source are input channels 0 .. c0 - 1) -> wt [c0][co][4][4] (OIHW of that forward conv).
__global__ __launch_bounds__(256) void up_dgrad_weight_kernel(const float* __restrict__ w, int co, int ci_total, int c0, float* __restrict__ wt) {
    const long long n = (long long)c0 * co * 16;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int tx = (int)(i & 3), ty = (int)((i >> 2) & 3), o = (int)((i >> 4) % co), c = (int)(i / (16LL * co));
        const float* src = w + ((size_t)o * ci_total + c) * 9;
        const int ky0 = ty == 0 ? 2 : (ty == 1 ? 1 : 0), ky1 = ty == 0 ? 2 : (ty == 1 ? 2 : (ty == 2 ? 1 : 0));
        const int kx0 = tx == 0 ? 2 : (tx == 1 ? 1 : 0), kx1 = tx == 0 ? 2 : (tx == 1 ? 2 : (tx == 2 ? 1 : 0));
        float s = 0.f;
        for (int ky = ky0; ky <= ky1; ++ky)
            for (int kx = kx0; kx <= kx1; ++kx) s += src[ky * 3 + kx];
        wt[i] = s;
    }
}

}  // namespace

extern "C" int codeps_stem_weight_fwd(const float* w, int out_c, int in_c, float* w_double_pixel, codeps_stream_t stream) {
    CODEPS_REQUIRE(w && w_double_pixel && out_c > 0 && in_c > 0 && in_c <= 8, CODEPS_ERR_ARG, "stem_weight_fwd: a 7x7 kernel over <= 8 input channels");
    hipLaunchKernelGGL(stem_weight_fwd_kernel, dim3(cdiv(out_c * 16 * 28, 256)), dim3(256), 0, (hipStream_t)stream, w, out_c, in_c, w_double_pixel);
    CODEPS_CHECK_LAUNCH("stem_weight_fwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_stem_weight_bwd(const float* grad_double_pixel, int out_c, int in_c, int accumulate, float* grad_w, codeps_stream_t stream) {
    CODEPS_REQUIRE(grad_double_pixel && grad_w && out_c > 0 && in_c > 0 && in_c <= 8, CODEPS_ERR_ARG, "stem_weight_bwd: bad arguments");
    hipLaunchKernelGGL(stem_weight_bwd_kernel, dim3(cdiv(out_c * in_c * 49, 256)), dim3(256), 0, (hipStream_t)stream, grad_double_pixel, out_c, in_c,
                       accumulate, grad_w);
    CODEPS_CHECK_LAUNCH("stem_weight_bwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_guda_weight_fwd(const float* w, int n_classes, int in_c, int n_coarse, const int* channels, float* const* outs,
                                      codeps_stream_t stream) {
    GudaSeg s;
    int rc = fill_seg(s, n_classes, in_c, n_coarse, channels, outs, "guda_weight_fwd");
    if (rc != CODEPS_OK) return rc;
    CODEPS_REQUIRE(w, CODEPS_ERR_ARG, "guda_weight_fwd: null weight");
    for (int k = 0; k <= n_coarse; ++k) CODEPS_REQUIRE(outs[k], CODEPS_ERR_ARG, "guda_weight_fwd: output %d is null", k);
    hipLaunchKernelGGL(guda_weight_fwd_kernel, dim3(cdiv(s.off[n_coarse + 1], 256)), dim3(256), 0, (hipStream_t)stream, w, n_classes, in_c, s);
    CODEPS_CHECK_LAUNCH("guda_weight_fwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_guda_weight_bwd(float* const* grad_outs, int n_classes, int in_c, int n_coarse, const int* channels, int accumulate,
                                      float* grad_w, codeps_stream_t stream) {
    GudaSeg s;
    int rc = fill_seg(s, n_classes, in_c, n_coarse, channels, grad_outs, "guda_weight_bwd");
    if (rc != CODEPS_OK) return rc;
    CODEPS_REQUIRE(grad_w, CODEPS_ERR_ARG, "guda_weight_bwd: null gradient");
    hipLaunchKernelGGL(guda_weight_bwd_kernel, dim3(cdiv(n_classes * in_c * 9, 256)), dim3(256), 0, (hipStream_t)stream, s, n_classes, in_c,
                       accumulate, grad_w);
    CODEPS_CHECK_LAUNCH("guda_weight_bwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_up_dgrad_weight(const float* w, int out_c, int in_c_total, int c0, float* wt, codeps_stream_t stream) {
    CODEPS_REQUIRE(w && wt && out_c > 0 && c0 > 0 && c0 <= in_c_total, CODEPS_ERR_ARG, "codeps_up_dgrad_weight: bad arguments");
    const long long n = (long long)c0 * out_c * 16;
    hipLaunchKernelGGL(up_dgrad_weight_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, w, out_c,
                       in_c_total, c0, wt);
    CODEPS_CHECK_LAUNCH("up_dgrad_weight_kernel");
    return CODEPS_OK;
}
