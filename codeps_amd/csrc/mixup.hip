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
#include <algorithm>
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


// ------------------------------------------------------------------------------------------------ conf_instance_mixup
// Mixup.get_conf_mask (datasets/mixup.py:542-652) places the teacher's instances one after the other: each goes to the column of
// its centre row that is farthest -- kornia.contrib.distance_transform -- from every thing already in the image.  The
// reference runs the transform's cascade (max(H, W) 3x3 convolutions over the whole image) per instance and reads ONE row of it.
// The cascade has a closed form: a pixel is reached in iteration i = D - 1, D its Chebyshev distance to the nearest thing
// pixel, and its value is  (3 i) / 2 - h log(sum of exp(-|d| / h) over the neighbours (replicate border) that were reached
// earlier or are things).  So the row costs three column scans, three W x W min-max passes and one block -- no iteration.
// kornia is not vendored by the reference (requirements.txt:16, unpinned): this follows oracle/mixup.py's restatement of
// its published algorithm, nine products summed in row-major kernel order.
struct ConfStats { long long n, sy, sx, min_x, max_x; };            // per instance id (256 per image)
constexpr int CONF_IDS = 256;
constexpr int CONF_FAR = 1 << 28;

__global__ void conf_stats_init_kernel(ConfStats* tab) {
    const int i = threadIdx.x;
    tab[i].n = tab[i].sy = tab[i].sx = 0;
    tab[i].min_x = CONF_FAR;
    tab[i].max_x = -1;
}

__global__ __launch_bounds__(256) void conf_stats_kernel(const int* __restrict__ inst, int h, int w, ConfStats* __restrict__ tab) {
    const long long n = (long long)h * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int id = inst[i];
        if (id <= 0 || id >= CONF_IDS) continue;
        const int y = (int)(i / w), x = (int)(i % w);
        atomicAdd(reinterpret_cast<unsigned long long*>(&tab[id].n), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long*>(&tab[id].sy), (unsigned long long)y);
        atomicAdd(reinterpret_cast<unsigned long long*>(&tab[id].sx), (unsigned long long)x);
        atomicMin(&tab[id].min_x, (long long)x);
        atomicMax(&tab[id].max_x, (long long)x);
    }
}

// v[j][c] = vertical distance from row r_j (rows row - 1, row, row + 1 clamped into the image) to the nearest thing of column c
__global__ __launch_bounds__(256) void conf_rows_kernel(const uint8_t* __restrict__ things, int h, int w, int row, int* __restrict__ v) {
    const int c = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (c >= w) return;
    const int r = min(max(row + j - 1, 0), h - 1);
    int best = CONF_FAR;
    for (int y = 0; y < h; ++y)
        if (things[(size_t)y * w + c]) best = min(best, abs(y - r));
    v[j * w + c] = best;
}

// d[j][x] = Chebyshev distance of pixel (r_j, x) to the nearest thing = min over columns c of max(|c - x|, v[j][c])
__global__ __launch_bounds__(256) void conf_cheb_kernel(const int* __restrict__ v, int w, int* __restrict__ d) {
    const int x = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (x >= w) return;
    int best = CONF_FAR;
    for (int c = 0; c < w; ++c) best = min(best, max(abs(c - x), v[j * w + c]));
    d[j * w + x] = best;
}

// the transform along the row and the first column of its maximum (torch.argmax): one block
__global__ __launch_bounds__(256) void conf_row_dt_kernel(const uint8_t* __restrict__ things, int h, int w, int row, const int* __restrict__ d,
                                                          float* __restrict__ dist_row, int* __restrict__ argmax) {
#pragma clang fp contract(off)
    // exp(-1 / 0.35), exp(-sqrt(2) / 0.35) of the float32 arguments, rounded to float32 (numpy, torch and the correctly
    // rounded double agree on these bits); centre tap = 1 (it multiplies the pixel's own, still empty, boundary value)
    const float k_edge = __uint_as_float(1030438518u), k_diag = __uint_as_float(1016074767u), hval = 0.35f;
    __shared__ float best_v[256];
    __shared__ int best_i[256];
    float bv = -1.f;
    int bi = 0x7fffffff;
    for (int x = threadIdx.x; x < w; x += 256) {
        float val = 0.f;
        const int d0 = d[w + x];
        if (!things[(size_t)row * w + x] && d0 < CONF_FAR) {
            float s = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int r = min(max(row + ky - 1, 0), h - 1);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int qx = min(max(x + kx - 1, 0), w - 1);
                    const bool in = things[(size_t)r * w + qx] != 0 || d[ky * w + qx] < d0;
                    const float kv = (ky == 1 && kx == 1) ? 1.f : ((ky == 1 || kx == 1) ? k_edge : k_diag);
                    s = s + kv * (in ? 1.f : 0.f);
                }
            }
            const float cdt = -hval * logf(s);
            val = (float)(((d0 - 1) * 3) / 2) + cdt;
        }
        if (dist_row) dist_row[x] = val;
        if (val > bv) { bv = val; bi = x; }                  // x ascends within a thread: the first maximum stays
    }
    best_v[threadIdx.x] = bv;
    best_i[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float ov = best_v[threadIdx.x + o];
            const int oi = best_i[threadIdx.x + o];
            if (ov > best_v[threadIdx.x] || (ov == best_v[threadIdx.x] && oi < best_i[threadIdx.x])) { best_v[threadIdx.x] = ov; best_i[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *argmax = best_i[0];
}

// horizontal shift of instance `cid`: to the farthest column, then back inside the image (mixup.py:586-601)
__global__ void conf_shift_kernel(const ConfStats* __restrict__ tab, int cid, int cx, const int* __restrict__ argmax, int w, int* __restrict__ shift) {
    int dx = *argmax - cx;
    const int lo = (int)tab[cid].min_x + dx, hi = (int)tab[cid].max_x + dx;
    if (lo < 0) dx -= lo;
    if (hi > w - 1) dx -= hi - (w - 1);
    shift[cid] = dx;
}

// paste bookkeeping of one instance (mixup.py:606-645 on masks): pixels the shifted instance covers become things; those no
// earlier instance holds are this instance's (a later instance is cropped where it overlaps an earlier one)
__global__ __launch_bounds__(256) void conf_apply_kernel(const int* __restrict__ inst, int h, int w, int cid, const int* __restrict__ shift,
                                                         uint8_t* __restrict__ dst, uint8_t* __restrict__ things) {
    const int dx = shift[cid];
    const long long n = (long long)h * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i % w), xs = x - dx;
        if (xs < 0 || xs >= w || inst[(size_t)y * w + xs] != cid) continue;
        things[i] = 1;
        if (dst[i] == 0) dst[i] = (uint8_t)cid;
    }
}

// the mixed item (mixup.py:366-381): a pixel held by instance c shows the target's pixel `shift[c]` columns to its left
__global__ __launch_bounds__(256) void conf_paste_kernel(const float* __restrict__ img_src, const float* __restrict__ img_tgt,
                                                         const int* __restrict__ sem_src, const int* __restrict__ sem_tgt,
                                                         const int* __restrict__ inst_src, const int* __restrict__ inst_conf,
                                                         const uint8_t* __restrict__ dst, const int* __restrict__ shift, int c, int h, int w,
                                                         float* __restrict__ rgb, int* __restrict__ sem, int* __restrict__ inst) {
    const long long n = (long long)h * w;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int cid = dst[i];
        if (cid == 0) {
            for (int k = 0; k < c; ++k) rgb[k * n + i] = img_src[k * n + i];
            sem[i] = sem_src[i];
            inst[i] = inst_src[i];
        } else {
            const long long j = i - shift[cid];
            for (int k = 0; k < c; ++k) rgb[k * n + i] = img_tgt[k * n + j];
            sem[i] = sem_tgt[j];
            inst[i] = inst_conf[j];
        }
    }
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


extern "C" size_t codeps_conf_stats_bytes(void) { return (size_t)CONF_IDS * sizeof(ConfStats); }

extern "C" int codeps_conf_instance_stats(const int* instance, int h, int w, void* table, codeps_stream_t stream) {
    CODEPS_REQUIRE(instance && table && h > 0 && w > 0, CODEPS_ERR_ARG, "codeps_conf_instance_stats: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conf_stats_init_kernel, dim3(1), dim3(CONF_IDS), 0, st, static_cast<ConfStats*>(table));
    hipLaunchKernelGGL(conf_stats_kernel, dim3((unsigned)std::min<long long>(((long long)h * w + 255) / 256, 2048)), dim3(256), 0, st, instance, h, w,
                       static_cast<ConfStats*>(table));
    CODEPS_CHECK_LAUNCH("conf_stats_kernel");
    return CODEPS_OK;
}

extern "C" size_t codeps_conf_place_workspace_bytes(int w) { return w > 0 ? (size_t)6 * w * sizeof(int) : 0; }

extern "C" int codeps_conf_place(const uint8_t* things, int h, int w, int row, float* dist_row, int* argmax, void* workspace,
                                 size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(things && argmax && workspace && h > 0 && w > 0 && row >= 0 && row < h, CODEPS_ERR_ARG, "codeps_conf_place: bad arguments");
    CODEPS_REQUIRE(workspace_bytes >= codeps_conf_place_workspace_bytes(w), CODEPS_ERR_WORKSPACE, "codeps_conf_place: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int* v = static_cast<int*>(workspace);
    int* d = v + 3 * (size_t)w;
    const dim3 grid((unsigned)((w + 255) / 256), 3);
    hipLaunchKernelGGL(conf_rows_kernel, grid, dim3(256), 0, st, things, h, w, row, v);
    hipLaunchKernelGGL(conf_cheb_kernel, grid, dim3(256), 0, st, v, w, d);
    hipLaunchKernelGGL(conf_row_dt_kernel, dim3(1), dim3(256), 0, st, things, h, w, row, d, dist_row, argmax);
    CODEPS_CHECK_LAUNCH("codeps_conf_place");
    return CODEPS_OK;
}

extern "C" int codeps_conf_apply(const int* instance, int h, int w, int cid, const void* table, int cx, const int* argmax, int* shift,
                                 uint8_t* dst, uint8_t* things, codeps_stream_t stream) {
    CODEPS_REQUIRE(instance && table && argmax && shift && dst && things && h > 0 && w > 0 && cid > 0 && cid < CONF_IDS, CODEPS_ERR_ARG,
                   "codeps_conf_apply: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conf_shift_kernel, dim3(1), dim3(1), 0, st, static_cast<const ConfStats*>(table), cid, cx, argmax, w, shift);
    hipLaunchKernelGGL(conf_apply_kernel, dim3((unsigned)std::min<long long>(((long long)h * w + 255) / 256, 2048)), dim3(256), 0, st, instance, h, w,
                       cid, shift, dst, things);
    CODEPS_CHECK_LAUNCH("codeps_conf_apply");
    return CODEPS_OK;
}

extern "C" int codeps_conf_paste(const float* img_src, const float* img_tgt, const int* sem_src, const int* sem_tgt, const int* inst_src,
                                 const int* inst_conf, const uint8_t* dst, const int* shift, int c, int h, int w, float* rgb_mix,
                                 int* sem_mix, int* inst_mix, codeps_stream_t stream) {
    CODEPS_REQUIRE(img_src && img_tgt && sem_src && sem_tgt && inst_src && inst_conf && dst && shift && rgb_mix && sem_mix && inst_mix &&
                       c > 0 && h > 0 && w > 0, CODEPS_ERR_ARG, "codeps_conf_paste: bad arguments");
    hipLaunchKernelGGL(conf_paste_kernel, dim3((unsigned)std::min<long long>(((long long)h * w + 255) / 256, 2048)), dim3(256), 0,
                       (hipStream_t)stream, img_src, img_tgt, sem_src, sem_tgt, inst_src, inst_conf, dst, shift, c, h, w, rgb_mix, sem_mix, inst_mix);
    CODEPS_CHECK_LAUNCH("conf_paste_kernel");
    return CODEPS_OK;
}
