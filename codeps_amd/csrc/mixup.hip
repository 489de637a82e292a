// Pseudo-label mix-up on the device: the reference builds the `*mixup` items inside
// CodepsNet.adapt (codeps/online_adap.py:331-335 -> datasets/mixup.py:406-489) with float64 grid_sample calls on
// full images, boolean-mask scatters and a CPU numpy pass for the centre / offset ground truth (mixup.py:243-257 ->
// datasets/dataset.py:126-177).  Here:
//   K15 warp_c2c      one thread per output pixel, float64 coordinates and weights in the reference's operation order
//   K16 cut_mixup     the three warps of do_mixup + the three masked copies of cut_mixup in ONE pass: only pixels inside
//                     the cut rectangles are warped, nothing is materialised in float64
//   K17 offset_center per-instance (count, sum y, sum x) through a small hash table with wave-aggregated atomics,
//                     then the Gaussian splat (atomic max) and the per-pixel offsets; no host round trip
// All HBM-bound byte / index work: one read of each input, one write of each output.
#include <type_traits>

#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ camera-to-camera
struct Intr { float fx, fy, cx, cy; };

// Normalised grid coordinate of output pixel (u, v): viewing ray of the OUTPUT camera (CameraModel.get_viewing_ray,
// misc/camera_model.py:52-70, float32) -> point at depth `depth` (_ImageToPointcloud, misc/image_warper.py:83-85,
// float64) -> pixel of the INPUT camera (get_image_point, camera_model.py:43-50) -> [-1, 1] (mixup.py:61-62) ->
// unnormalised with align_corners=True (ATen grid_sampler).  No FMA contraction: every product / sum rounds like the
// reference's separate tensor ops do.
__device__ __forceinline__ void c2c_coords(int u, int v, Intr io, Intr ii, double depth, int ws, int hs, double& ix, double& iy) {
#pragma clang fp contract(off)
    float rx = ((float)u - io.cx) / io.fx;
    float ry = ((float)v - io.cy) / io.fy;
    const float norm = sqrtf(rx * rx + ry * ry + 1.0f);
    rx = rx / norm;
    ry = ry / norm;
    const float rz = 1.0f / norm;
    const double s = depth / fabs((double)rz);
    const double x3 = s * (double)rx, y3 = s * (double)ry;
    double z3 = s * (double)rz;
    z3 = z3 < 1e-5 ? 1e-5 : z3;
    const double u2 = (x3 / z3) * (double)ii.fx + (double)ii.cx;
    const double v2 = (y3 / z3) * (double)ii.fy + (double)ii.cy;
    const double un = (u2 / (double)(ws - 1) - 0.5) * 2.0;
    const double vn = (v2 / (double)(hs - 1) - 0.5) * 2.0;
    ix = ((un + 1.0) / 2.0) * (double)(ws - 1);
    iy = ((vn + 1.0) / 2.0) * (double)(hs - 1);
}

__device__ __forceinline__ double clipd(double v, double hi) { return fmin(fmax(v, 0.0), hi); }

// grid_sample(mode="bilinear", align_corners=True) on one fp32 plane, float64 weights, result rounded to fp32
__device__ __forceinline__ float sample_bilinear(const float* __restrict__ plane, int hs, int ws, double ix, double iy, bool border) {
#pragma clang fp contract(off)
    if (border) { ix = clipd(ix, (double)(ws - 1)); iy = clipd(iy, (double)(hs - 1)); }
    const double xw = floor(ix), yn = floor(iy);
    const double w = ix - xw, e = 1.0 - w, n = iy - yn, s = 1.0 - n;
    const double nw = s * e, ne = s * w, sw = n * e, se = n * w;
    // out-of-range corners contribute 0; the range test in double keeps huge coordinates out of the int conversion
    const bool x0ok = xw >= 0.0 && xw <= (double)(ws - 1), x1ok = xw + 1.0 >= 0.0 && xw + 1.0 <= (double)(ws - 1);
    const bool y0ok = yn >= 0.0 && yn <= (double)(hs - 1), y1ok = yn + 1.0 >= 0.0 && yn + 1.0 <= (double)(hs - 1);
    const int x0 = x0ok ? (int)xw : 0, x1 = x1ok ? (int)xw + 1 : 0, y0 = y0ok ? (int)yn : 0, y1 = y1ok ? (int)yn + 1 : 0;
    const double vnw = (x0ok && y0ok) ? (double)plane[(size_t)y0 * ws + x0] : 0.0;
    const double vne = (x1ok && y0ok) ? (double)plane[(size_t)y0 * ws + x1] : 0.0;
    const double vsw = (x0ok && y1ok) ? (double)plane[(size_t)y1 * ws + x0] : 0.0;
    const double vse = (x1ok && y1ok) ? (double)plane[(size_t)y1 * ws + x1] : 0.0;
    return (float)(vnw * nw + vne * ne + vsw * sw + vse * se);
}

// grid_sample(mode="nearest"): round half to even, then zeros / border
__device__ __forceinline__ bool nearest_index(int hs, int ws, double ix, double iy, bool border, size_t& idx) {
    if (border) { ix = clipd(ix, (double)(ws - 1)); iy = clipd(iy, (double)(hs - 1)); }
    const double xr = rint(ix), yr = rint(iy);
    if (!(xr >= 0.0 && xr <= (double)(ws - 1) && yr >= 0.0 && yr <= (double)(hs - 1))) return false;
    idx = (size_t)(int)yr * ws + (int)xr;
    return true;
}

template <typename T>
__global__ __launch_bounds__(256) void warp_c2c_kernel(const T* __restrict__ in, const float* __restrict__ intr_in,
                                                       const float* __restrict__ intr_out, int c, int hs, int ws, int ho, int wo,
                                                       int nearest, int border, double depth, T* __restrict__ out) {
    const int b = blockIdx.z, v = blockIdx.y, u = blockIdx.x * 256 + threadIdx.x;
    if (u >= wo) return;
    const Intr ii = {intr_in[b * 4 + 0], intr_in[b * 4 + 1], intr_in[b * 4 + 2], intr_in[b * 4 + 3]};
    const Intr io = {intr_out[b * 4 + 0], intr_out[b * 4 + 1], intr_out[b * 4 + 2], intr_out[b * 4 + 3]};
    double ix, iy;
    c2c_coords(u, v, io, ii, depth, ws, hs, ix, iy);
    const size_t plane = (size_t)hs * ws, oplane = (size_t)ho * wo;
    const T* src = in + (size_t)b * c * plane;
    T* dst = out + (size_t)b * c * oplane + (size_t)v * wo + u;
    if (nearest) {
        size_t idx;
        const bool ok = nearest_index(hs, ws, ix, iy, border != 0, idx);
        for (int k = 0; k < c; ++k) dst[k * oplane] = ok ? src[k * plane + idx] : (T)0;
    } else {
        if constexpr (std::is_same<T, float>::value) {          // bilinear exists for the fp32 instantiation only
            for (int k = 0; k < c; ++k) dst[k * oplane] = sample_bilinear(src + k * plane, hs, ws, ix, iy, border != 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------ cut mix-up
struct CutP {
    int h, w, ht, wt, nseg, geom;
    double depth;
};

// One thread per output pixel of the source-sized mix: inside a cut rectangle the target image (bilinear, zeros), the
// teacher's instance map (nearest, zeros) and semantic map (nearest, border) are sampled through the camera-to-camera
// warp (mixup.py:431-441) -- or read in place when geom == 0 -- otherwise the source pixel / labels pass through.
__global__ __launch_bounds__(256) void cut_mixup_kernel(CutP p, const float* __restrict__ img_src, const float* __restrict__ img_tgt,
                                                        const int* __restrict__ sem_src, const int* __restrict__ sem_tgt,
                                                        const int* __restrict__ inst_src, const int* __restrict__ inst_tgt,
                                                        const float* __restrict__ intr_src, const float* __restrict__ intr_tgt,
                                                        const int* __restrict__ rects, float* __restrict__ rgb_mix,
                                                        int* __restrict__ sem_mix, int* __restrict__ inst_mix) {
    const int b = blockIdx.z, v = blockIdx.y, u = blockIdx.x * 256 + threadIdx.x;
    if (u >= p.w) return;
    bool inside = false;
    for (int s = 0; s < p.nseg; ++s) {                         // rect = (y0, y1, x0, x1), end-exclusive like the slices
        const int* r = rects + ((size_t)b * p.nseg + s) * 4;
        inside |= (v >= r[0] && v < r[1] && u >= r[2] && u < r[3]);
    }
    const size_t plane = (size_t)p.h * p.w, pix = (size_t)v * p.w + u, tplane = (size_t)p.ht * p.wt;
    float rgb[3];
    int sem, inst;
    if (!inside) {
#pragma unroll
        for (int k = 0; k < 3; ++k) rgb[k] = img_src[((size_t)b * 3 + k) * plane + pix];
        sem = sem_src[b * plane + pix];
        inst = inst_src[b * plane + pix];
    } else if (!p.geom) {
#pragma unroll
        for (int k = 0; k < 3; ++k) rgb[k] = img_tgt[((size_t)b * 3 + k) * tplane + pix];
        sem = sem_tgt[b * tplane + pix];
        inst = inst_tgt[b * tplane + pix];
    } else {
        const Intr is = {intr_src[b * 4 + 0], intr_src[b * 4 + 1], intr_src[b * 4 + 2], intr_src[b * 4 + 3]};
        const Intr it = {intr_tgt[b * 4 + 0], intr_tgt[b * 4 + 1], intr_tgt[b * 4 + 2], intr_tgt[b * 4 + 3]};
        double ix, iy;
        c2c_coords(u, v, is, it, p.depth, p.wt, p.ht, ix, iy);
#pragma unroll
        for (int k = 0; k < 3; ++k) rgb[k] = sample_bilinear(img_tgt + ((size_t)b * 3 + k) * tplane, p.ht, p.wt, ix, iy, false);
        size_t idx;
        inst = nearest_index(p.ht, p.wt, ix, iy, false, idx) ? inst_tgt[b * tplane + idx] : 0;
        sem = nearest_index(p.ht, p.wt, ix, iy, true, idx) ? sem_tgt[b * tplane + idx] : 0;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb_mix[((size_t)b * 3 + k) * plane + pix] = rgb[k];
    sem_mix[b * plane + pix] = sem;
    inst_mix[b * plane + pix] = inst;
}

// ------------------------------------------------------------------------------------------------ offset / centre
constexpr int OC_CAP = 4096;                 // hash slots per image (distinct instance ids; open addressing, key 0 = empty)

struct OcTable {                             // one per image, zero-initialised
    int key[OC_CAP];
    unsigned cnt[OC_CAP];
    unsigned long long sy[OC_CAP], sx[OC_CAP];
    double cy[OC_CAP], cx[OC_CAP];
};

__device__ __forceinline__ unsigned oc_hash(int id) { return ((unsigned)id * 2654435761u) >> 20; }   // 12 bits

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Pass 1: per-instance pixel count and coordinate sums.  Lanes of a wave that hold the same id are combined first, so a
// wave issues one atomic triple per DISTINCT id (1-3 in practice) instead of 64 colliding ones.  Also clears `center`.
__global__ __launch_bounds__(256) void oc_accumulate_kernel(const int* __restrict__ inst, int h, int w, OcTable* __restrict__ tabs,
                                                            float* __restrict__ center, int* __restrict__ status) {
    const int b = blockIdx.y;
    const size_t n = (size_t)h * w, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    const int id = live ? inst[b * n + i] : 0;
    if (live) center[b * n + i] = 0.f;
    const int y = live ? (int)(i / w) : 0, x = live ? (int)(i % w) : 0;
    OcTable& t = tabs[b];
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(id != 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lid = __shfl(id, leader, 64);
        const bool mine = id == lid;
        const unsigned long long same = __ballot(mine);
        const int c = __popcll(same), sy = wave_sum_i(mine ? y : 0), sx = wave_sum_i(mine ? x : 0);
        if (lane == leader) {
            unsigned slot = oc_hash(lid) & (OC_CAP - 1);
            int probes = 0;
            for (; probes < OC_CAP; ++probes) {
                const int prev = atomicCAS(&t.key[slot], 0, lid);
                if (prev == 0 || prev == lid) break;
                slot = (slot + 1) & (OC_CAP - 1);
            }
            if (probes == OC_CAP) {
                atomicExch(&status[b], -1);                    // more distinct ids than slots: reported, never silently wrong
            } else {
                atomicAdd(&t.cnt[slot], (unsigned)c);
                atomicAdd(&t.sy[slot], (unsigned long long)sy);
                atomicAdd(&t.sx[slot], (unsigned long long)sx);
            }
        }
        todo &= ~same;
    }
}

// Pass 2: one block per occupied slot -> centre of mass (float64 like np.mean), Gaussian splat with an atomic max
// (dataset.py:150-169; the window is (6 sigma + 3)^2, clipped to the image).
__global__ __launch_bounds__(256) void oc_splat_kernel(int h, int w, double sigma, OcTable* __restrict__ tabs, float* __restrict__ center,
                                                       int* __restrict__ status) {
    const int b = blockIdx.y, slot = blockIdx.x;
    OcTable& t = tabs[b];
    if (t.key[slot] == 0) return;
    const double cy = (double)t.sy[slot] / (double)t.cnt[slot], cx = (double)t.sx[slot] / (double)t.cnt[slot];
    if (threadIdx.x == 0) {
        t.cy[slot] = cy;
        t.cx[slot] = cx;
        if (status[b] >= 0) atomicAdd(&status[b], 1);
    }
    const int y = (int)rint(cy), x = (int)rint(cx);
    if (x < 0 || x >= w || y < 0 || y >= h) return;
    const int ulx = (int)rint((double)x - 3.0 * sigma - 1.0), uly = (int)rint((double)y - 3.0 * sigma - 1.0);
    const int brx = (int)rint((double)x + 3.0 * sigma + 2.0), bry = (int)rint((double)y + 3.0 * sigma + 2.0);
    const int x_lo = max(0, ulx), x_hi = min(brx, w), y_lo = max(0, uly), y_hi = min(bry, h);
    const int ww = x_hi - x_lo, hh = y_hi - y_lo;
    const double c0 = 3.0 * sigma + 1.0, den = 2.0 * sigma * sigma;
    int* ctr = reinterpret_cast<int*>(center + (size_t)b * h * w);
    for (int k = threadIdx.x; k < ww * hh; k += 256) {
        const int py = y_lo + k / ww, px = x_lo + k % ww;
        const double gx = (double)(px - ulx) - c0, gy = (double)(py - uly) - c0;
        const float g = (float)exp(-(gx * gx + gy * gy) / den);
        atomicMax(&ctr[(size_t)py * w + px], __float_as_int(g));          // g >= 0: integer order == float order
    }
}

// Pass 3: offsets (dy, dx) to the instance's centre of mass for every instance pixel, 0 elsewhere (dataset.py:171-175).
__global__ __launch_bounds__(256) void oc_offsets_kernel(const int* __restrict__ inst, int h, int w, const OcTable* __restrict__ tabs,
                                                         float* __restrict__ offset) {
    const int b = blockIdx.y;
    const size_t n = (size_t)h * w, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int id = inst[b * n + i];
    float oy = 0.f, ox = 0.f;
    if (id != 0) {
        const OcTable& t = tabs[b];
        unsigned slot = oc_hash(id) & (OC_CAP - 1);
        int probes = 0;
        while (probes < OC_CAP && t.key[slot] != id) { slot = (slot + 1) & (OC_CAP - 1); ++probes; }
        if (probes < OC_CAP) {
            const int y = (int)(i / w), x = (int)(i % w);
            const int cyi = (int)rint(t.cy[slot]), cxi = (int)rint(t.cx[slot]);
            if (cxi >= 0 && cxi < w && cyi >= 0 && cyi < h) {                 // the reference skips instances centred outside
                oy = (float)(t.cy[slot] - (double)(float)y);
                ox = (float)(t.cx[slot] - (double)(float)x);
            }
        }
    }
    offset[((size_t)b * 2 + 0) * n + i] = oy;
    offset[((size_t)b * 2 + 1) * n + i] = ox;
}

}  // namespace

extern "C" int codeps_warp_c2c(const void* in, int dtype, const float* intr_in, const float* intr_out, int batch, int c, int hs, int ws,
                               int ho, int wo, int nearest, int border, double depth_val, void* out, codeps_stream_t stream) {
    CODEPS_REQUIRE(in && intr_in && intr_out && out, CODEPS_ERR_ARG, "codeps_warp_c2c: null pointer");
    CODEPS_REQUIRE(batch > 0 && c > 0 && hs > 1 && ws > 1 && ho > 0 && wo > 0 && batch <= 65535 && ho <= 65535, CODEPS_ERR_ARG,
                   "codeps_warp_c2c: bad shape [%d,%d,%d,%d] -> [%d,%d]", batch, c, hs, ws, ho, wo);
    CODEPS_REQUIRE(dtype >= 0 && dtype <= 2, CODEPS_ERR_ARG, "codeps_warp_c2c: dtype %d (0 = fp32, 1 = uint8, 2 = int32)", dtype);
    CODEPS_REQUIRE(nearest || dtype == 0, CODEPS_ERR_ARG, "codeps_warp_c2c: bilinear sampling needs fp32 input");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(cdiv(wo, 256), ho, batch);
    if (dtype == 0)
        hipLaunchKernelGGL(warp_c2c_kernel<float>, grid, dim3(256), 0, st, (const float*)in, intr_in, intr_out, c, hs, ws, ho, wo, nearest, border,
                           depth_val, (float*)out);
    else if (dtype == 1)
        hipLaunchKernelGGL(warp_c2c_kernel<uint8_t>, grid, dim3(256), 0, st, (const uint8_t*)in, intr_in, intr_out, c, hs, ws, ho, wo, 1, border,
                           depth_val, (uint8_t*)out);
    else
        hipLaunchKernelGGL(warp_c2c_kernel<int>, grid, dim3(256), 0, st, (const int*)in, intr_in, intr_out, c, hs, ws, ho, wo, 1, border,
                           depth_val, (int*)out);
    CODEPS_CHECK_LAUNCH("codeps_warp_c2c");
    return CODEPS_OK;
}

extern "C" int codeps_cut_mixup(const float* img_src, const float* img_tgt, const int* sem_src, const int* sem_tgt, const int* inst_src,
                                const int* inst_tgt, const float* intr_src, const float* intr_tgt, const int* rects, int nseg, int batch,
                                int h, int w, int ht, int wt, double depth_val, float* rgb_mix, int* sem_mix, int* inst_mix,
                                codeps_stream_t stream) {
    CODEPS_REQUIRE(img_src && img_tgt && sem_src && sem_tgt && inst_src && inst_tgt && rects && rgb_mix && sem_mix && inst_mix,
                   CODEPS_ERR_ARG, "codeps_cut_mixup: null pointer");
    CODEPS_REQUIRE(batch > 0 && h > 0 && w > 0 && ht > 1 && wt > 1 && nseg >= 0 && batch <= 65535 && h <= 65535, CODEPS_ERR_ARG,
                   "codeps_cut_mixup: bad shape B=%d %dx%d target %dx%d nseg=%d", batch, h, w, ht, wt, nseg);
    const int geom = (intr_src != nullptr && intr_tgt != nullptr);
    CODEPS_REQUIRE(geom || (h == ht && w == wt), CODEPS_ERR_ARG,
                   "codeps_cut_mixup: without camera models the target must have the source's size (datasets/mixup.py:91,150)");
    CutP p = {h, w, ht, wt, nseg, geom, depth_val};
    hipLaunchKernelGGL(cut_mixup_kernel, dim3(cdiv(w, 256), h, batch), dim3(256), 0, (hipStream_t)stream, p, img_src, img_tgt, sem_src, sem_tgt,
                       inst_src, inst_tgt, intr_src, intr_tgt, rects, rgb_mix, sem_mix, inst_mix);
    CODEPS_CHECK_LAUNCH("codeps_cut_mixup");
    return CODEPS_OK;
}

extern "C" size_t codeps_offset_center_workspace_bytes(int batch) { return batch > 0 ? (size_t)batch * sizeof(OcTable) : 0; }

extern "C" int codeps_offset_center(const int* instance, int batch, int h, int w, double sigma, float* center, float* offset, int* status,
                                    void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(instance && center && offset && status && workspace, CODEPS_ERR_ARG, "codeps_offset_center: null pointer");
    CODEPS_REQUIRE(batch > 0 && h > 0 && w > 0 && batch <= 65535 && sigma > 0, CODEPS_ERR_ARG, "codeps_offset_center: bad shape B=%d %dx%d sigma=%g",
                   batch, h, w, sigma);
    CODEPS_REQUIRE(workspace_bytes >= codeps_offset_center_workspace_bytes(batch), CODEPS_ERR_WORKSPACE,
                   "codeps_offset_center: workspace of %zu bytes, %zu needed", workspace_bytes, codeps_offset_center_workspace_bytes(batch));
    hipStream_t st = (hipStream_t)stream;
    OcTable* tabs = static_cast<OcTable*>(workspace);
    if (hipMemsetAsync(tabs, 0, (size_t)batch * sizeof(OcTable), st) != hipSuccess || hipMemsetAsync(status, 0, (size_t)batch * sizeof(int), st) != hipSuccess) {
        codeps_set_error("codeps_offset_center: hipMemsetAsync failed");
        return CODEPS_ERR_LAUNCH;
    }
    const size_t n = (size_t)h * w;
    const dim3 pix((unsigned)((n + 255) / 256), batch);
    hipLaunchKernelGGL(oc_accumulate_kernel, pix, dim3(256), 0, st, instance, h, w, tabs, center, status);
    hipLaunchKernelGGL(oc_splat_kernel, dim3(OC_CAP, batch), dim3(256), 0, st, h, w, sigma, tabs, center, status);
    hipLaunchKernelGGL(oc_offsets_kernel, pix, dim3(256), 0, st, instance, h, w, tabs, offset);
    CODEPS_CHECK_LAUNCH("codeps_offset_center");
    return CODEPS_OK;
}
