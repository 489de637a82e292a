// Image-file data path on the device: what the reference's data loader does between the decoded PNG and the
// network input (datasets/dataset.py:66-67, datasets/cityscapes.py:137-290, datasets/preprocessing.py:12-105), as byte / integer
// kernels.  PNG inflate stays on the host (one serial zlib stream per file); everything after it runs here on uint8 images
// [B][H][W][3] as the decoder leaves them:
//   resample_u8_kernel        PIL Image.resize(LANCZOS) = transforms.Resize: separable 8.22 fixed-point passes with an 8-bit
//                             intermediate (Pillow Resample.c); coefficient tables from codeps_lanczos_coeffs (host, double)
//   jitter_kernel             torchvision ColorJitter pieces on PIL images = Pillow ImageEnhance blends (brightness / contrast /
//                             saturation) and the uint8 hue shift through Pillow's RGB <-> HSV conversion, in a per-image order
//   hist_u8 / match_lut / ... skimage match_histograms per channel (transfer_histogram_style mode "rgb")
//   to_tensor_kernel          horizontal flip + ToTensor + Normalize -> fp32 NCHW
//   cityscapes_labels_*       nearest resize of labelIds / instanceIds, label conversion, instance ids of the thing classes,
//                             thing mask, loss weights, small-instance weights by area
//   disparity_depth_kernel    16-bit disparity -> metric depth, nearest resize
// All HBM-bound at 3-16 bytes per pixel; bit-exact integer / table arithmetic (the float pieces mirror Pillow's C expressions
// operation by operation, contraction off).
#include "common.h"

#include <math.h>

#include <vector>

#pragma clang fp contract(off)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// ------------------------------------------------------------------------------------------------ resample
// One pass along x (axis 1) or y (axis 0) of [B][H][W][3] uint8.  bounds [out][2] = (first tap, taps), kk [out][ksize] int32.
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ src, int B, int H, int W, int axis, int out_size,
                                                          const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                          uint8_t* __restrict__ dst) {
    const int oh = axis == 0 ? out_size : H, ow = axis == 1 ? out_size : W;
    const long long total = (long long)B * oh * ow;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % ow), y = (int)((i / ow) % oh), b = (int)(i / ((long long)ow * oh));
        const int o = axis == 1 ? x : y;
        const int first = bounds[o * 2], cnt = bounds[o * 2 + 1];
        const int* k = kk + (size_t)o * ksize;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        const uint8_t* base = src + (size_t)b * H * W * 3;
        for (int t = 0; t < cnt; ++t) {
            const uint8_t* p = axis == 1 ? base + ((size_t)y * W + first + t) * 3 : base + ((size_t)(first + t) * W + x) * 3;
            const int c = k[t];
            s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
        }
        uint8_t* q = dst + (size_t)i * 3;
        q[0] = (uint8_t)min(max(s0 >> PRECISION_BITS, 0), 255);
        q[1] = (uint8_t)min(max(s1 >> PRECISION_BITS, 0), 255);
        q[2] = (uint8_t)min(max(s2 >> PRECISION_BITS, 0), 255);
    }
}

// ------------------------------------------------------------------------------------------------ colour jitter
constexpr int JMAXB = 16;
struct JitterP {
    int n_ops[JMAXB];
    int op[JMAXB][4];          // 0 brightness, 1 contrast, 2 saturation, 3 hue
    float factor[JMAXB][4];
    int first[JMAXB], last[JMAXB];   // op range [first, last) applied by this launch
    int mean[JMAXB];           // grey mean for a contrast op at `first` (computed by the previous launch)
    int want_sum[JMAXB];       // accumulate the grey sum of the result (a contrast op follows)
};

__device__ __forceinline__ int grey_of(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// Pillow ImagingBlend(im1 = degenerate, im2 = image, alpha) per channel
__device__ __forceinline__ int blend_u8(int d, int v, float alpha) {
    if (alpha == 0.f) return d;
    if (alpha == 1.f) return v;
    const float t = (float)d + alpha * (float)(v - d);
    if (alpha >= 0.f && alpha <= 1.f) return (int)t;
    return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}

__device__ __forceinline__ void rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    uv = maxc;
    if (minc == maxc) { uh = 0; us = 0; return; }
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    uh = min(max((int)((double)h * 255.0), 0), 255);
    us = min(max((int)((double)s * 255.0), 0), 255);
}

__device__ __forceinline__ int c_round(double x) { return (int)(x >= 0.0 ? floor(x + 0.5) : ceil(x - 0.5)); }

__device__ __forceinline__ void hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
    if (s == 0) { r = g = b = v; return; }
    const double hf = (double)(float)h * 6.0 / 255.0;
    const int i = (int)floor(hf);
    const float f = (float)(hf - (double)(float)i);
    const float fs = (float)((double)(float)s / 255.0);
    const double vf = (double)(float)v;
    const int p = min(max(c_round(vf * (1.0 - (double)fs)), 0), 255);
    const int q = min(max(c_round(vf * (1.0 - (double)fs * (double)f)), 0), 255);
    const int t = min(max(c_round(vf * (1.0 - (double)fs * (1.0 - (double)f))), 0), 255);
    switch (i % 6) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

// ------------------------------------------------------------------------------------------------ histogram matching
__global__ __launch_bounds__(256) void hist_u8_kernel(const uint8_t* __restrict__ img, int HW, unsigned* __restrict__ hist /* [B][3][256] */) {
    __shared__ unsigned h[3][256];
    for (int i = threadIdx.x; i < 768; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    const int b = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const uint8_t* p = img + ((size_t)b * HW + i) * 3;
        atomicAdd(&h[0][p[0]], 1u); atomicAdd(&h[1][p[1]], 1u); atomicAdd(&h[2][p[2]], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += 256)
        if ((&h[0][0])[i]) atomicAdd(hist + (size_t)b * 768 + i, (&h[0][0])[i]);
}

// lut[b][c][v] = uint8(np.interp(cdf_src(v), cdf_ref over the values present in ref, those values)); one thread per (b, c)
__global__ void match_lut_kernel(const unsigned* __restrict__ hs, const unsigned* __restrict__ hr, int n_src, int n_ref, int B,
                                 uint8_t* __restrict__ lut) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * 3) return;
    const unsigned* s = hs + (size_t)t * 256;
    const unsigned* r = hr + (size_t)t * 256;
    double xp[256], fp[256];
    int m = 0;
    unsigned long long cum = 0;
    for (int v = 0; v < 256; ++v)
        if (r[v]) { cum += r[v]; xp[m] = (double)cum / (double)n_ref; fp[m] = (double)v; ++m; }
    cum = 0;
    int j = 0;
    for (int v = 0; v < 256; ++v) {
        cum += s[v];
        const double x = (double)cum / (double)n_src;
        double y;
        if (m == 0) y = 0.0;
        else if (x <= xp[0]) y = fp[0];                       // np.interp: left = fp[0], right = fp[-1]
        else if (x >= xp[m - 1]) y = fp[m - 1];
        else {
            while (j + 1 < m && xp[j + 1] <= x) ++j;          // x is non-decreasing in v: resume the search
            const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
            y = slope * (x - xp[j]) + fp[j];
        }
        lut[(size_t)t * 256 + v] = (uint8_t)(int)y;           // truncating cast of the float64 result back to uint8
    }
}

__global__ __launch_bounds__(256) void apply_lut_kernel(const uint8_t* __restrict__ img, int HW, const uint8_t* __restrict__ lut,
                                                        uint8_t* __restrict__ out) {
    const int b = blockIdx.y;
    const uint8_t* l = lut + (size_t)b * 768;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const size_t o = ((size_t)b * HW + i) * 3;
        out[o] = l[img[o]]; out[o + 1] = l[256 + img[o + 1]]; out[o + 2] = l[512 + img[o + 2]];
    }
}

// ------------------------------------------------------------------------------------------------ to tensor
struct NormP { float mean[3], std[3]; int normalize; unsigned flip_mask; };

__global__ __launch_bounds__(256) void to_tensor_kernel(const uint8_t* __restrict__ img, int B, int H, int W, NormP P, float* __restrict__ out) {
    const long long total = (long long)B * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((long long)W * H));
        const int sx = ((P.flip_mask >> (b & 31)) & 1u) ? W - 1 - x : x;
        const uint8_t* p = img + (((size_t)b * H + y) * W + sx) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (float)p[c] / 255.f;                    // ToTensor: uint8 -> float32, div(255)
            if (P.normalize) v = (v - P.mean[c]) / P.std[c];          // Normalize: sub_(mean).div_(std)
            out[(((size_t)b * 3 + c) * H + y) * W + x] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ labels
struct LabelP { uint8_t map[256]; unsigned long long thing_mask; };
constexpr int ISLOTS = 4096;

__device__ __forceinline__ int nearest_src(int dst, int src_size, int dst_size) {
    const int s = (int)floor((double)dst * ((double)src_size / (double)dst_size));
    return min(s, src_size - 1);
}

// pass 1: maps + per-instance areas (open-addressing table of (id, count))
// flavor 0 = Cityscapes (cityscapes.py:166-246: "iscrowd" instances dropped, centre weights = not ignore, offset weights = instance mask),
// flavor 1 = KITTI-360 (kitti_360.py:252-300: every thing pixel is an instance pixel, centre and offset weights = thing mask)
__global__ __launch_bounds__(256) void city_labels_kernel(const uint8_t* __restrict__ sem_ids, const uint16_t* __restrict__ inst_ids, int H, int W,
                                                          int oh, int ow, LabelP P, int flavor, uint8_t* __restrict__ sem, int* __restrict__ inst,
                                                          uint8_t* __restrict__ thing, uint8_t* __restrict__ sem_w, uint8_t* __restrict__ center_w,
                                                          uint8_t* __restrict__ offset_w, uint8_t* __restrict__ sem_eval,
                                                          int* __restrict__ table /* [ISLOTS][2] */) {
    const int total = oh * ow;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int y = i / ow, x = i % ow;
        const size_t so = (size_t)nearest_src(y, H, oh) * W + nearest_src(x, W, ow);
        const int s_raw = sem_ids[so], i_raw = inst_ids[so];
        const int sc = P.map[s_raw];
        const bool th = sc < 64 && ((P.thing_mask >> sc) & 1ull);
        const bool msk = th && (flavor == 1 || i_raw >= 1000);                // Cityscapes: "remove iscrowd instances"
        const int ic = msk ? (int)(uint16_t)(sc * 1000 + (i_raw - s_raw * 1000)) : 0;
        sem[i] = (uint8_t)sc; inst[i] = ic; thing[i] = th ? 1 : 0;
        sem_w[i] = sc == 255 ? 0 : 1;
        center_w[i] = flavor == 1 ? (th ? 1 : 0) : (sc == 255 ? 0 : 1);
        offset_w[i] = flavor == 1 ? (th ? 1 : 0) : (msk ? 1 : 0);
        if (sem_eval) sem_eval[i] = (uint8_t)sc;
        if (ic != 0) {
            unsigned slot = ((unsigned)ic * 2654435761u) >> 20;               // 12 bits
            for (int probe = 0; probe < ISLOTS; ++probe, slot = (slot + 1) & (ISLOTS - 1)) {
                const int prev = atomicCAS(&table[slot * 2], 0, ic);
                if (prev == 0 || prev == ic) { atomicAdd(&table[slot * 2 + 1], 1); break; }
            }
        }
    }
}

// pass 2: pixels of instances smaller than the area threshold get the small-instance weight
__global__ __launch_bounds__(256) void city_small_kernel(const int* __restrict__ inst, int total, const int* __restrict__ table, float small_area,
                                                         int small_weight, uint8_t* __restrict__ sem_w, uint8_t* __restrict__ sem_eval) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int ic = inst[i];
        if (ic == 0) continue;
        unsigned slot = ((unsigned)ic * 2654435761u) >> 20;
        for (int probe = 0; probe < ISLOTS; ++probe, slot = (slot + 1) & (ISLOTS - 1)) {
            const int id = table[slot * 2];
            if (id == ic) {
                const float area = (float)table[slot * 2 + 1];
                if (area < small_area) sem_w[i] = (uint8_t)small_weight;
                if (sem_eval && area < small_area * .1f) sem_eval[i] = 255;           // kitti_360.py:283-285
                break;
            }
            if (id == 0) break;
        }
    }
}

__global__ __launch_bounds__(256) void disparity_depth_kernel(const uint16_t* __restrict__ disp, int H, int W, int oh, int ow, float bf,
                                                              float* __restrict__ depth) {
    const int total = oh * ow;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int y = i / ow, x = i % ow;
        const float d = (float)disp[(size_t)nearest_src(y, H, oh) * W + nearest_src(x, W, ow)];
        depth[i] = d > 0.f ? bf / ((d - 1.f) / 256.f) : 0.f;
    }
}

// KITTI-360 depth from the Velodyne sweep (kitti_360.py:303-330): project every point, keep 0 < depth < 80 inside the image, and let
// the LAST point in file order win a pixel (numpy fancy-index assignment) -- a 64-bit atomicMax on (point index << 32 | depth bits).
struct LidarP { double M[12]; double K[9]; };
__global__ __launch_bounds__(256) void lidar_scatter_kernel(const float* __restrict__ pts, int n, LidarP P, int W, int H,
                                                            unsigned long long* __restrict__ keys) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const double x = pts[i * 4], y = pts[i * 4 + 1], z = pts[i * 4 + 2];
        const double cx = P.M[0] * x + P.M[1] * y + P.M[2] * z + P.M[3], cy = P.M[4] * x + P.M[5] * y + P.M[6] * z + P.M[7];
        const double cz = P.M[8] * x + P.M[9] * y + P.M[10] * z + P.M[11];
        const double px = P.K[0] * cx + P.K[1] * cy + P.K[2] * cz, py = P.K[3] * cx + P.K[4] * cy + P.K[5] * cz;
        double d = P.K[6] * cx + P.K[7] * cy + P.K[8] * cz;
        if (d == 0.0) d = -1e-6;
        const long long u = (long long)rint(px / fabs(d)), v = (long long)rint(py / fabs(d));      // np.round: half to even
        if (u < 0 || u >= W || v < 0 || v >= H || !(d > 0.0) || !(d < 80.0)) continue;
        const unsigned long long key = ((unsigned long long)(unsigned)(i + 1) << 32) | __float_as_uint((float)d);
        atomicMax(keys + (size_t)v * W + u, key);
    }
}

__global__ __launch_bounds__(256) void lidar_gather_kernel(const unsigned long long* __restrict__ keys, int H, int W, int oh, int ow,
                                                           float* __restrict__ depth) {
    const int total = oh * ow;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const unsigned long long k = keys[(size_t)nearest_src(i / ow, H, oh) * W + nearest_src(i % ow, W, ow)];
        depth[i] = k ? __uint_as_float((unsigned)(k & 0xFFFFFFFFull)) : 0.f;
    }
}

static inline double sinc_d(double x) { if (x == 0.0) return 1.0; x = x * M_PI; return sin(x) / x; }
static inline double lanczos_d(double x) { return (-3.0 <= x && x < 3.0) ? sinc_d(x) * sinc_d(x / 3) : 0.0; }
static inline int grid1(long long n) { return (int)std::min<long long>((n + 255) / 256, 8192); }

}  // namespace

// HOST: Pillow precompute_coeffs + normalize_coeffs_8bpc for the Lanczos filter (support 3).  Returns ksize; with bounds / coeffs
// NULL only computes ksize.  bounds [out_size][2] = (first tap, taps); coeffs [out_size][ksize] int32 (8.22 fixed point).
extern "C" int codeps_lanczos_coeffs(int in_size, int out_size, int* bounds, int* coeffs) {
    if (in_size <= 0 || out_size <= 0) { codeps_set_error("lanczos_coeffs: bad sizes"); return CODEPS_ERR_ARG; }
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 3.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    if (!bounds || !coeffs) return ksize;
    const double ss = 1.0 / filterscale;
    std::vector<double> k(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { k[x] = lanczos_d((x + xmin - center + 0.5) * ss); ww += k[x]; }
        for (int x = 0; x < ksize; ++x) {
            double v = x < xmax ? (ww != 0.0 ? k[x] / ww : k[x]) : 0.0;
            coeffs[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[xx * 2] = xmin; bounds[xx * 2 + 1] = xmax;
    }
    return ksize;
}

extern "C" int codeps_resample_u8(const uint8_t* src, int batch, int h, int w, int axis, int out_size, const int* bounds, const int* coeffs,
                                  int ksize, uint8_t* dst, codeps_stream_t stream) {
    CODEPS_REQUIRE(src && dst && bounds && coeffs && batch > 0 && h > 0 && w > 0 && out_size > 0 && ksize > 0 && (axis == 0 || axis == 1),
                   CODEPS_ERR_ARG, "resample_u8: bad arguments");
    const long long total = (long long)batch * (axis == 0 ? out_size : h) * (axis == 1 ? out_size : w);
    hipLaunchKernelGGL(resample_u8_kernel, dim3(grid1(total)), dim3(256), 0, (hipStream_t)stream, src, batch, h, w, axis, out_size, bounds, coeffs,
                       ksize, dst);
    CODEPS_CHECK_LAUNCH("resample_u8_kernel");
    return CODEPS_OK;
}

// ops / factors: HOST arrays [batch][4] (op codes 0 brightness, 1 contrast, 2 saturation, 3 hue, in application order), n_ops [batch].
// A contrast op needs the grey mean of the image AS IT IS when the op is applied, so the chain is cut in front of every contrast
// op: one launch per segment; the mean goes from the integer grey sums to the next launch through device memory (no host sync).
// grey_sums: device scratch of 16 x 8 + 16 x 4 bytes; tmp: a second image buffer (ping-pong).
namespace {
__global__ void jitter_mean_kernel(const unsigned long long* sums, int B, int HW, int* means) {
    const int b = threadIdx.x;
    if (b < B) means[b] = (int)((double)sums[b] / (double)HW + 0.5);          // int(ImageStat.Stat(L).mean[0] + 0.5)
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void jitter_kernel_dev(const uint8_t* __restrict__ src, int HW, JitterP P, const int* __restrict__ means_dev,
                                                         uint8_t* __restrict__ dst, unsigned long long* __restrict__ grey_sum) {
    __shared__ unsigned long long red[4];
    const int b = blockIdx.y;
    const int mean = means_dev ? means_dev[b] : 0;
    unsigned long long acc = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const uint8_t* p = src + ((size_t)b * HW + i) * 3;
        int r = p[0], g = p[1], bl = p[2];
        for (int o = P.first[b]; o < P.last[b]; ++o) {
            const float f = P.factor[b][o];
            switch (P.op[b][o]) {
                case 0: r = blend_u8(0, r, f); g = blend_u8(0, g, f); bl = blend_u8(0, bl, f); break;
                case 1: r = blend_u8(mean, r, f); g = blend_u8(mean, g, f); bl = blend_u8(mean, bl, f); break;
                case 2: { const int m = grey_of(r, g, bl); r = blend_u8(m, r, f); g = blend_u8(m, g, f); bl = blend_u8(m, bl, f); break; }
                default: {
                    int h, s, v;
                    rgb2hsv(r, g, bl, h, s, v);
                    h = (h + ((int)(long long)((double)f * 255.0) & 0xFF)) & 0xFF;
                    hsv2rgb(h, s, v, r, g, bl);
                }
            }
        }
        uint8_t* q = dst + ((size_t)b * HW + i) * 3;
        q[0] = (uint8_t)r; q[1] = (uint8_t)g; q[2] = (uint8_t)bl;
        if (P.want_sum[b]) acc += (unsigned long long)grey_of(r, g, bl);
    }
    if (!P.want_sum[b]) return;
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(grey_sum + b, red[0] + red[1] + red[2] + red[3]);
}
}  // namespace

extern "C" int codeps_color_jitter_u8(const uint8_t* src, int batch, int h, int w, const int* n_ops, const int* ops, const float* factors,
                                      uint8_t* dst, uint8_t* tmp, unsigned long long* grey_sums, codeps_stream_t stream) {
    CODEPS_REQUIRE(src && dst && tmp && grey_sums && n_ops && ops && factors && batch > 0 && batch <= JMAXB && h > 0 && w > 0, CODEPS_ERR_ARG,
                   "color_jitter_u8: bad arguments (batch <= %d)", JMAXB);
    hipStream_t st = (hipStream_t)stream;
    const int HW = h * w;
    JitterP P;
    memset(&P, 0, sizeof(P));
    int max_ops = 0;
    for (int b = 0; b < batch; ++b) {
        CODEPS_REQUIRE(n_ops[b] >= 0 && n_ops[b] <= 4, CODEPS_ERR_ARG, "color_jitter_u8: 0..4 ops per image");
        P.n_ops[b] = n_ops[b];
        for (int o = 0; o < n_ops[b]; ++o) {
            CODEPS_REQUIRE(ops[b * 4 + o] >= 0 && ops[b * 4 + o] <= 3, CODEPS_ERR_ARG, "color_jitter_u8: unknown op code");
            P.op[b][o] = ops[b * 4 + o]; P.factor[b][o] = factors[b * 4 + o];
        }
        max_ops = std::max(max_ops, n_ops[b]);
    }
    int* means_dev = reinterpret_cast<int*>(grey_sums + JMAXB);               // scratch behind the sums
    int pos[JMAXB];
    bool ready[JMAXB];
    for (int b = 0; b < batch; ++b) { pos[b] = 0; ready[b] = false; }
    const uint8_t* cur = src;
    bool have_means = false;
    for (int guard = 0; guard < 12; ++guard) {
        bool work = false, any_sum = false, done = true;
        for (int b = 0; b < batch; ++b) {
            const int n = P.n_ops[b], s0 = pos[b];
            int e = s0;
            if (s0 < n) {
                if (P.op[b][s0] == 1 && !ready[b]) e = s0;                     // contrast next, mean unknown: reduce only
                else { e = s0 + 1; while (e < n && P.op[b][e] != 1) ++e; }     // apply up to (not including) the next contrast
            }
            const bool want = e < n && P.op[b][e] == 1;
            P.first[b] = s0; P.last[b] = e; P.want_sum[b] = want ? 1 : 0;
            work = work || e > s0 || want;
            any_sum = any_sum || want;
            pos[b] = e; ready[b] = want;
            done = done && e >= n && !want;
        }
        if (!work) break;
        uint8_t* out = (cur == dst) ? tmp : dst;
        if (any_sum) {
            hipError_t e_ = hipMemsetAsync(grey_sums, 0, sizeof(unsigned long long) * JMAXB, st);
            CODEPS_REQUIRE(e_ == hipSuccess, CODEPS_ERR_LAUNCH, "color_jitter_u8: memset failed");
        }
        hipLaunchKernelGGL(jitter_kernel_dev, dim3(std::min((HW + 255) / 256, 1024), batch), dim3(256), 0, st, cur, HW, P,
                           have_means ? means_dev : (const int*)nullptr, out, grey_sums);
        CODEPS_CHECK_LAUNCH("jitter_kernel");
        if (any_sum) { hipLaunchKernelGGL(jitter_mean_kernel, dim3(1), dim3(64), 0, st, grey_sums, batch, HW, means_dev); have_means = true; }
        cur = out;
        if (done) break;
    }
    if (cur != dst) {
        hipError_t e_ = hipMemcpyAsync(dst, cur, (size_t)batch * HW * 3, hipMemcpyDeviceToDevice, st);
        CODEPS_REQUIRE(e_ == hipSuccess, CODEPS_ERR_LAUNCH, "color_jitter_u8: copy failed");
    }
    return CODEPS_OK;
}

extern "C" int codeps_match_histograms_u8(const uint8_t* img, const uint8_t* ref, int batch, int h, int w, int ref_h, int ref_w, uint8_t* out,
                                          void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    const size_t need = (size_t)batch * 768 * 4 * 2 + (size_t)batch * 768;
    CODEPS_REQUIRE(img && ref && out && workspace && batch > 0 && h > 0 && w > 0 && ref_h > 0 && ref_w > 0, CODEPS_ERR_ARG, "match_histograms_u8: bad arguments");
    CODEPS_REQUIRE(workspace_bytes >= need, CODEPS_ERR_WORKSPACE, "match_histograms_u8: workspace %zu < %zu", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    unsigned* hs = (unsigned*)workspace;
    unsigned* hr = hs + (size_t)batch * 768;
    uint8_t* lut = (uint8_t*)(hr + (size_t)batch * 768);
    hipError_t e = hipMemsetAsync(workspace, 0, (size_t)batch * 768 * 4 * 2, st);
    CODEPS_REQUIRE(e == hipSuccess, CODEPS_ERR_LAUNCH, "match_histograms_u8: memset failed");
    hipLaunchKernelGGL(hist_u8_kernel, dim3(std::min((h * w + 255) / 256, 512), batch), dim3(256), 0, st, img, h * w, hs);
    hipLaunchKernelGGL(hist_u8_kernel, dim3(std::min((ref_h * ref_w + 255) / 256, 512), batch), dim3(256), 0, st, ref, ref_h * ref_w, hr);
    hipLaunchKernelGGL(match_lut_kernel, dim3(cdiv(batch * 3, 64)), dim3(64), 0, st, (const unsigned*)hs, (const unsigned*)hr, h * w, ref_h * ref_w,
                       batch, lut);
    hipLaunchKernelGGL(apply_lut_kernel, dim3(std::min((h * w + 255) / 256, 1024), batch), dim3(256), 0, st, img, h * w, (const uint8_t*)lut, out);
    CODEPS_CHECK_LAUNCH("match_histograms_u8 kernels");
    return CODEPS_OK;
}

// Pillow's Image.convert("HSV") / convert("RGB") on packed uint8 pixels (Convert.c rgb2hsv_row / hsv2rgb_row: the functions the
// hue jitter uses), for transfer_histogram_style's "hsv" / "saturation" modes (datasets/preprocessing.py:64-71)
__global__ __launch_bounds__(256) void hsv_convert_kernel(const uint8_t* __restrict__ src, long long n, int to_hsv, uint8_t* __restrict__ dst) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int a = src[i * 3], b = src[i * 3 + 1], c = src[i * 3 + 2];
        int x, y, z;
        if (to_hsv) rgb2hsv(a, b, c, x, y, z); else hsv2rgb(a, b, c, x, y, z);
        dst[i * 3] = (uint8_t)x; dst[i * 3 + 1] = (uint8_t)y; dst[i * 3 + 2] = (uint8_t)z;
    }
}

extern "C" int codeps_rgb_hsv_u8(const uint8_t* src, long long pixels, int to_hsv, uint8_t* dst, codeps_stream_t stream) {
    CODEPS_REQUIRE(src && dst && pixels > 0, CODEPS_ERR_ARG, "rgb_hsv_u8: bad arguments");
    hipLaunchKernelGGL(hsv_convert_kernel, dim3(grid1(pixels)), dim3(256), 0, (hipStream_t)stream, src, pixels, to_hsv ? 1 : 0, dst);
    CODEPS_CHECK_LAUNCH("hsv_convert_kernel");
    return CODEPS_OK;
}

extern "C" size_t codeps_match_histograms_workspace_bytes(int batch) { return batch > 0 ? (size_t)batch * 768 * 9 : 0; }

extern "C" int codeps_u8_to_tensor(const uint8_t* img, int batch, int h, int w, const float* mean, const float* std_, unsigned flip_mask,
                                   float* out, codeps_stream_t stream) {
    CODEPS_REQUIRE(img && out && batch > 0 && batch <= 32 && h > 0 && w > 0 && ((mean == nullptr) == (std_ == nullptr)), CODEPS_ERR_ARG,
                   "u8_to_tensor: bad arguments (batch <= 32)");
    NormP P;
    P.normalize = mean ? 1 : 0; P.flip_mask = flip_mask;
    for (int c = 0; c < 3; ++c) { P.mean[c] = mean ? mean[c] : 0.f; P.std[c] = std_ ? std_[c] : 1.f; }
    hipLaunchKernelGGL(to_tensor_kernel, dim3(grid1((long long)batch * h * w)), dim3(256), 0, (hipStream_t)stream, img, batch, h, w, P, out);
    CODEPS_CHECK_LAUNCH("to_tensor_kernel");
    return CODEPS_OK;
}

extern "C" size_t codeps_cityscapes_labels_workspace_bytes(void) { return (size_t)ISLOTS * 2 * sizeof(int); }

static int label_maps(int flavor, const uint8_t* semantic_ids, const uint16_t* instance_ids, int h, int w, int out_h, int out_w,
                      const uint8_t* label_map256, unsigned long long thing_class_mask, float small_instance_area, int small_instance_weight,
                      uint8_t* semantic, int* instance, uint8_t* thing_mask, uint8_t* semantic_weights, uint8_t* center_weights,
                      uint8_t* offset_weights, uint8_t* semantic_eval, void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(semantic_ids && instance_ids && label_map256 && semantic && instance && thing_mask && semantic_weights && center_weights &&
                       offset_weights && workspace && h > 0 && w > 0 && out_h > 0 && out_w > 0,
                   CODEPS_ERR_ARG, "label maps: bad arguments");
    CODEPS_REQUIRE(workspace_bytes >= codeps_cityscapes_labels_workspace_bytes(), CODEPS_ERR_WORKSPACE, "label maps: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    LabelP P;
    memcpy(P.map, label_map256, 256);
    P.thing_mask = thing_class_mask;
    hipError_t e = hipMemsetAsync(workspace, 0, codeps_cityscapes_labels_workspace_bytes(), st);
    CODEPS_REQUIRE(e == hipSuccess, CODEPS_ERR_LAUNCH, "label maps: memset failed");
    const int total = out_h * out_w;
    hipLaunchKernelGGL(city_labels_kernel, dim3(grid1(total)), dim3(256), 0, st, semantic_ids, instance_ids, h, w, out_h, out_w, P, flavor, semantic,
                       instance, thing_mask, semantic_weights, center_weights, offset_weights, semantic_eval, (int*)workspace);
    hipLaunchKernelGGL(city_small_kernel, dim3(grid1(total)), dim3(256), 0, st, (const int*)instance, total, (const int*)workspace, small_instance_area,
                       small_instance_weight, semantic_weights, semantic_eval);
    CODEPS_CHECK_LAUNCH("label map kernels");
    return CODEPS_OK;
}

extern "C" int codeps_cityscapes_labels(const uint8_t* semantic_ids, const uint16_t* instance_ids, int h, int w, int out_h, int out_w,
                                        const uint8_t* label_map256, unsigned long long thing_class_mask, float small_instance_area,
                                        int small_instance_weight, uint8_t* semantic, int* instance, uint8_t* thing_mask, uint8_t* semantic_weights,
                                        uint8_t* center_weights, uint8_t* offset_weights, void* workspace, size_t workspace_bytes,
                                        codeps_stream_t stream) {
    return label_maps(0, semantic_ids, instance_ids, h, w, out_h, out_w, label_map256, thing_class_mask, small_instance_area, small_instance_weight,
                      semantic, instance, thing_mask, semantic_weights, center_weights, offset_weights, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int codeps_kitti360_labels(const uint8_t* semantic_ids, const uint16_t* instance_ids, int h, int w, int out_h, int out_w,
                                      const uint8_t* label_map256, unsigned long long thing_class_mask, float small_instance_area,
                                      int small_instance_weight, uint8_t* semantic, int* instance, uint8_t* thing_mask, uint8_t* semantic_weights,
                                      uint8_t* center_weights, uint8_t* offset_weights, uint8_t* semantic_eval, void* workspace,
                                      size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(semantic_eval, CODEPS_ERR_ARG, "kitti360_labels: semantic_eval is required");
    return label_maps(1, semantic_ids, instance_ids, h, w, out_h, out_w, label_map256, thing_class_mask, small_instance_area, small_instance_weight,
                      semantic, instance, thing_mask, semantic_weights, center_weights, offset_weights, semantic_eval, workspace, workspace_bytes,
                      stream);
}

extern "C" size_t codeps_lidar_depth_workspace_bytes(int h, int w) { return h > 0 && w > 0 ? (size_t)h * w * sizeof(unsigned long long) : 0; }

// points: device [n][4] fp32 (x, y, z, reflectance); lidar_to_rect: HOST double [3][4]; camera_matrix: HOST double [3][3].
extern "C" int codeps_lidar_depth(const float* points, int n, const double* lidar_to_rect, const double* camera_matrix, int h, int w, int out_h,
                                  int out_w, float* depth, void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(points && lidar_to_rect && camera_matrix && depth && workspace && n >= 0 && h > 0 && w > 0 && out_h > 0 && out_w > 0, CODEPS_ERR_ARG,
                   "lidar_depth: bad arguments");
    CODEPS_REQUIRE(workspace_bytes >= codeps_lidar_depth_workspace_bytes(h, w), CODEPS_ERR_WORKSPACE, "lidar_depth: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    LidarP P;
    for (int i = 0; i < 12; ++i) P.M[i] = lidar_to_rect[i];
    for (int i = 0; i < 9; ++i) P.K[i] = camera_matrix[i];
    hipError_t e = hipMemsetAsync(workspace, 0, codeps_lidar_depth_workspace_bytes(h, w), st);
    CODEPS_REQUIRE(e == hipSuccess, CODEPS_ERR_LAUNCH, "lidar_depth: memset failed");
    if (n > 0) hipLaunchKernelGGL(lidar_scatter_kernel, dim3(grid1(n)), dim3(256), 0, st, points, n, P, w, h, (unsigned long long*)workspace);
    hipLaunchKernelGGL(lidar_gather_kernel, dim3(grid1((long long)out_h * out_w)), dim3(256), 0, st, (const unsigned long long*)workspace, h, w, out_h,
                       out_w, depth);
    CODEPS_CHECK_LAUNCH("lidar_depth kernels");
    return CODEPS_OK;
}

extern "C" int codeps_disparity_to_depth(const uint16_t* disparity, int h, int w, int out_h, int out_w, float baseline_times_fx, float* depth,
                                         codeps_stream_t stream) {
    CODEPS_REQUIRE(disparity && depth && h > 0 && w > 0 && out_h > 0 && out_w > 0, CODEPS_ERR_ARG, "disparity_to_depth: bad arguments");
    hipLaunchKernelGGL(disparity_depth_kernel, dim3(grid1((long long)out_h * out_w)), dim3(256), 0, (hipStream_t)stream, disparity, h, w, out_h, out_w,
                       baseline_times_fx, depth);
    CODEPS_CHECK_LAUNCH("disparity_depth_kernel");
    return CODEPS_OK;
}
