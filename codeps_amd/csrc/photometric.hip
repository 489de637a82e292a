// Fused multi-scale photometric reconstruction loss (K7-K10) for gfx950.
//
// Replaces ReconstructionLoss.__call__ (algos/depth.py:239-326) and everything it calls: bilinear
// down-scaling (depth.py:158-173), camera scaling / back-projection / pose transform / projection
// (misc/camera_model.py:37-71, misc/image_warper.py:20-144), grid_sample(border, align_corners=True)
// (image_warper.py:178-182), 3x3 reflect-padded SSIM (depth.py:110-155), L1 mix (depth.py:221-237),
// identity losses + tie-break noise + min over 4 (depth.py:305-323), per-scale mean / 2^s, / S.
//
// Launch structure (all scales in ONE launch each, block -> (scale, batch item, tile)):
//   fwd:  pyramid_kernel      full-res (I_t, I_t-1, I_t+1, depth) -> levels 1..S-1 (10 planes each)
//         photo_fwd_kernel    per tile: warp + 4-tap gather + SSIM/L1 for 4 candidates, min, block sum,
//                             writes the selected branch per pixel (u8) for the backward
//         loss_reduce_kernel  fixed-order sum of per-block partials -> loss[0]
//   bwd:  photo_bwd_kernel    per tile: the forward's SSIM adjoint coefficients of the selected branch (halo 1) and
//                             its warped samples are READ BACK (round 3: the forward stores them, 60 B/px, instead of
//                             the backward recomputing warp + sample on a halo-2 tile and the window sums on halo 1:
//                             2.7x fewer VALU operations per pixel), gathered at each pixel, back through the
//                             bilinear sample, projection, pose and back-projection -> d depth_s, per-block d T partials
//         pose_reduce_kernel  fixed-order sum of d T partials per batch item
//         depth_combine_kernel  transposed bilinear pyramid: d depth = g_0 + sum_s down_s^T(g_s)
// HBM-bound: algorithmic traffic 40 B/px forward, 44 B/px backward + 60 B/px written and ~72 B/px read back
// for the stored samples / coefficients; planes are NCHW so
// every plane read is a unit-stride 256-B/wave stream; intermediates live in LDS tiles only.
// No atomics anywhere -> results are bitwise reproducible run to run.
#include "common.h"

namespace {

constexpr int MAXS = 8;
constexpr float C1 = 0.0001f, C2 = 0.0009f;

struct Levels {
    int num_scales, batch, H, W;
    int h[MAXS], w[MAXS];
    float su[MAXS], sv[MAXS];          // intrinsics scale  w_s / W, h_s / H
    float rh[MAXS], rw[MAXS];          // resize ratio (float)H / h_s, (float)W / w_s  (ATen area_pixel_compute_scale)
    float pixw[MAXS];                  // 1 / (S * 2^s * B*h_s*w_s)
    long long pyr_off[MAXS];           // float offset of level s (>=1) in the pyramid buffer
    long long map_off[MAXS];           // pixel offset of level s in per-level maps (B*h_s*w_s each)
    int ftx[MAXS], fty[MAXS], fblk[MAXS + 1];   // forward tiling
    int btx[MAXS], bty[MAXS], bblk[MAXS + 1];   // backward tiling
    long long map_total, pyr_total;
};

constexpr int FTH = 8, FTW = 64;       // forward tile
constexpr int BTH = 8, BTW = 32;       // backward tile

struct WsLayout {
    size_t pyr, sel, gds, loss_part, pose_part, pred, coef, total;
};

static Levels make_levels(const codeps_photometric_desc* d) {
    Levels L;
    memset(&L, 0, sizeof(L));
    L.num_scales = d->num_scales; L.batch = d->batch; L.H = d->height; L.W = d->width;
    long long pyr = 0, map = 0;
    L.fblk[0] = 0; L.bblk[0] = 0;
    for (int s = 0; s < d->num_scales; ++s) {
        const int h = d->height >> s, w = d->width >> s;
        L.h[s] = h; L.w[s] = w;
        L.su[s] = (float)((double)w / (double)d->width);
        L.sv[s] = (float)((double)h / (double)d->height);
        L.rh[s] = (float)d->height / (float)h;
        L.rw[s] = (float)d->width / (float)w;
        L.pixw[s] = (float)(1.0 / ((double)d->num_scales * (double)(1 << s) * (double)d->batch * h * w));
        L.pyr_off[s] = pyr;
        if (s > 0) pyr += (long long)d->batch * 10 * h * w;
        L.map_off[s] = map;
        map += (long long)d->batch * h * w;
        L.ftx[s] = cdiv(w, FTW); L.fty[s] = cdiv(h, FTH);
        L.fblk[s + 1] = L.fblk[s] + d->batch * L.ftx[s] * L.fty[s];
        L.btx[s] = cdiv(w, BTW); L.bty[s] = cdiv(h, BTH);
        L.bblk[s + 1] = L.bblk[s] + d->batch * L.btx[s] * L.bty[s];
    }
    L.map_total = map; L.pyr_total = pyr;
    return L;
}

static WsLayout make_layout(const Levels& L) {
    WsLayout o;
    size_t off = 0;
    o.pyr = off;       off = align_up(off + (size_t)L.pyr_total * 4, 256);
    o.sel = off;       off = align_up(off + (size_t)L.map_total, 256);
    o.gds = off;       off = align_up(off + (size_t)L.map_total * 4, 256);
    o.loss_part = off; off = align_up(off + (size_t)L.fblk[L.num_scales] * 4, 256);
    o.pose_part = off; off = align_up(off + (size_t)L.bblk[L.num_scales] * 32 * 4, 256);
    // forward results kept for the backward, per level [b][plane][h_s * w_s]: the warped samples of both source frames
    // (6 planes) and the SSIM adjoint coefficients A, B, C per channel of the selected branch (9 planes, unscaled)
    o.pred = off;      off = align_up(off + (size_t)L.map_total * 6 * 4, 256);
    o.coef = off;      off = align_up(off + (size_t)L.map_total * 9 * 4, 256);
    o.total = off;
    return o;
}

// ---------------------------------------------------------------------------------- device helpers
// Blocks are dispatched round-robin over the 8 XCDs (MI355X_MICROARCH.md, workgroup dispatch), each with its own L2.  Tiles that
// are neighbours in the image share halo rows / columns and the gathered source pixels: give every XCD one CONTIGUOUS run of
// tile ids so that a halo is fetched from HBM once per XCD boundary instead of once per tile (bijective for any grid size).
__device__ __forceinline__ int xcd_contiguous(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ATen upsample_bilinear2d (align_corners=False) source index / lambdas for one axis.
__device__ __forceinline__ void src_index(float ratio, int dst, int in_size, int out_size, int& i0, int& i1,
                                          float& l0, float& l1) {
    if (in_size == out_size) { i0 = i1 = dst; l0 = 1.f; l1 = 0.f; return; }
    float r = ratio * ((float)dst + 0.5f) - 0.5f;
    r = r < 0.f ? 0.f : r;
    i0 = min((int)floorf(r), in_size - 1);
    l1 = fminf(fmaxf(r - (float)i0, 0.f), 1.f);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l0 = 1.f - l1;
}

__device__ __forceinline__ int reflect1(int i, int n) {       // ReflectionPad2d(1) index, then clamp
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * (n - 1) - i : i;
    return min(max(i, 0), n - 1);
}

struct Cam { float fx, fy, cx, cy; };

struct Warp {                  // forward quantities needed by the backward too
    float px, py, pz;          // back-projected point P
    float rx, ry, rz;          // dP/d depth
    float X, Y, Z, Wh;         // T * [P;1]
    float xe, ye, ze, zc;      // after / Wh, clamp
    float ix, iy;              // clipped pixel coordinates in the source image
    float cgx, cgy;            // clip gradient masks (0/1)
};

__device__ __forceinline__ Warp warp_point(const Cam k, const float* __restrict__ T, int gx, int gy, float depth,
                                           int ws, int hs) {
    Warp o;
    float rx = ((float)gx - k.cx) / k.fx;
    float ry = ((float)gy - k.cy) / k.fy;
    const float norm = sqrtf(rx * rx + ry * ry + 1.0f);
    rx = rx / norm; ry = ry / norm;
    const float rz = 1.0f / norm;
    const float dz = depth / fabsf(rz);
    o.rx = rx / fabsf(rz); o.ry = ry / fabsf(rz); o.rz = rz / fabsf(rz);
    o.px = dz * rx; o.py = dz * ry; o.pz = dz * rz;
    o.X = T[0] * o.px + T[1] * o.py + T[2] * o.pz + T[3];
    o.Y = T[4] * o.px + T[5] * o.py + T[6] * o.pz + T[7];
    o.Z = T[8] * o.px + T[9] * o.py + T[10] * o.pz + T[11];
    o.Wh = T[12] * o.px + T[13] * o.py + T[14] * o.pz + T[15];
    o.xe = o.X / o.Wh; o.ye = o.Y / o.Wh; o.ze = o.Z / o.Wh;
    o.zc = fmaxf(o.ze, 1e-5f);
    const float u2 = (o.xe / o.zc) * k.fx + k.cx;
    const float v2 = (o.ye / o.zc) * k.fy + k.cy;
    const float un = (u2 / (float)(ws - 1) - 0.5f) * 2.f;
    const float vn = (v2 / (float)(hs - 1) - 0.5f) * 2.f;
    float ix = ((un + 1.f) / 2.f) * (float)(ws - 1);
    float iy = ((vn + 1.f) / 2.f) * (float)(hs - 1);
    const float mx = (float)(ws - 1), my = (float)(hs - 1);
    o.cgx = (ix > 0.f && ix < mx) ? 1.f : 0.f;
    o.cgy = (iy > 0.f && iy < my) ? 1.f : 0.f;
    // NaN-safe border clip (grid_sample padding_mode="border")
    ix = !(ix > 0.f) ? 0.f : (ix >= mx ? mx : ix);
    iy = !(iy > 0.f) ? 0.f : (iy >= my ? my : iy);
    o.ix = ix; o.iy = iy;
    return o;
}

struct Taps { int o_nw, o_ne, o_sw, o_se; float tx, ty; bool e_ok, s_ok; };

__device__ __forceinline__ Taps make_taps(float ix, float iy, int ws, int hs) {
    Taps t;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0;
    t.tx = ix - fx0; t.ty = iy - fy0;
    t.e_ok = x0 + 1 < ws; t.s_ok = y0 + 1 < hs;
    const int x1 = t.e_ok ? x0 + 1 : x0, y1 = t.s_ok ? y0 + 1 : y0;
    t.o_nw = y0 * ws + x0; t.o_ne = y0 * ws + x1; t.o_sw = y1 * ws + x0; t.o_se = y1 * ws + x1;
    return t;
}

__device__ __forceinline__ float sample(const float* __restrict__ p, const Taps& t) {
    const float nw = p[t.o_nw], ne = t.e_ok ? p[t.o_ne] : 0.f;
    const float sw = t.s_ok ? p[t.o_sw] : 0.f, se = (t.e_ok && t.s_ok) ? p[t.o_se] : 0.f;
    const float wx1 = t.tx, wx0 = 1.f - t.tx, wy1 = t.ty, wy0 = 1.f - t.ty;
    return nw * (wx0 * wy0) + ne * (wx1 * wy0) + sw * (wx0 * wy1) + se * (wx1 * wy1);
}

// counter-based N(0,1): splitmix64 hash -> Box-Muller
__device__ __forceinline__ float hash_normal(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const float u1 = ((float)(uint32_t)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(uint32_t)((z >> 8) & 0xFFFFFF) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.f * __logf(u1)) * __cosf(6.28318530718f * u2);
}

struct LevelPtrs { const float *tg, *im, *ip, *dp; int hw; };

__device__ __forceinline__ LevelPtrs level_ptrs(const Levels& L, int s, int b, const float* img_t, const float* img_m,
                                                const float* img_p, const float* depth, const float* pyr) {
    LevelPtrs p;
    p.hw = L.h[s] * L.w[s];
    if (s == 0) {
        const size_t o = (size_t)b * 3 * p.hw;
        p.tg = img_t + o; p.im = img_m + o; p.ip = img_p + o; p.dp = depth + (size_t)b * p.hw;
    } else {
        const float* base = pyr + L.pyr_off[s] + (size_t)b * 10 * p.hw;
        p.tg = base; p.im = base + 3 * (size_t)p.hw; p.ip = base + 6 * (size_t)p.hw; p.dp = base + 9 * (size_t)p.hw;
    }
    return p;
}

__device__ __forceinline__ Cam level_cam(const Levels& L, int s, const float* intr, int b) {
    Cam k;
    k.fx = intr[b * 4 + 0] * L.su[s]; k.fy = intr[b * 4 + 1] * L.sv[s];
    k.cx = intr[b * 4 + 2] * L.su[s]; k.cy = intr[b * 4 + 3] * L.sv[s];
    return k;
}

// ---------------------------------------------------------------------------------- pyramid
__global__ __launch_bounds__(256) void pyramid_kernel(Levels L, const float* __restrict__ img_t,
                                                      const float* __restrict__ img_m, const float* __restrict__ img_p,
                                                      const float* __restrict__ depth, float* __restrict__ pyr) {
    const long long total = L.pyr_total;
    const int HW = L.H * L.W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int s = 1;
        while (s + 1 < L.num_scales && i >= L.pyr_off[s + 1]) ++s;
        long long r = i - L.pyr_off[s];
        const int w = L.w[s], h = L.h[s];
        const int x = (int)(r % w); r /= w;
        const int y = (int)(r % h); r /= h;
        const int c = (int)(r % 10);
        const int b = (int)(r / 10);
        const float* src;
        if (c < 3) src = img_t + ((size_t)b * 3 + c) * HW;
        else if (c < 6) src = img_m + ((size_t)b * 3 + c - 3) * HW;
        else if (c < 9) src = img_p + ((size_t)b * 3 + c - 6) * HW;
        else src = depth + (size_t)b * HW;
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        src_index(L.rh[s], y, L.H, h, y0, y1, ly0, ly1);
        src_index(L.rw[s], x, L.W, w, x0, x1, lx0, lx1);
        pyr[i] = ly0 * (lx0 * src[y0 * L.W + x0] + lx1 * src[y0 * L.W + x1]) +
                 ly1 * (lx0 * src[y1 * L.W + x0] + lx1 * src[y1 * L.W + x1]);
    }
}

// ---------------------------------------------------------------------------------- SSIM helpers
struct Stats { float sx, sxx, sxy; };

__device__ __forceinline__ float ssim_val(float sx, float sy, float sxx, float syy, float sxy) {
    const float mux = sx / 9.f, muy = sy / 9.f;
    const float sigx = sxx / 9.f - mux * mux, sigy = syy / 9.f - muy * muy, sigxy = sxy / 9.f - mux * muy;
    const float n = (2.f * mux * muy + C1) * (2.f * sigxy + C2);
    const float d = (mux * mux + muy * muy + C1) * (sigx + sigy + C2);
    return fminf(fmaxf((1.f - n / d) / 2.f, 0.f), 1.f);
}

// d clamp((1-n/d)/2) / d x_j = A + B x_j + C y_j   (x_j, y_j = window entries)
__device__ __forceinline__ void ssim_grad(float sx, float sy, float sxx, float syy, float sxy, float& A, float& B,
                                          float& C) {
    const float mux = sx / 9.f, muy = sy / 9.f;
    const float sigx = sxx / 9.f - mux * mux, sigy = syy / 9.f - muy * muy, sigxy = sxy / 9.f - mux * muy;
    const float a1 = 2.f * mux * muy + C1, a2 = 2.f * sigxy + C2;
    const float b1 = mux * mux + muy * muy + C1, b2 = sigx + sigy + C2;
    const float n = a1 * a2, d = b1 * b2;
    const float v = (1.f - n / d) / 2.f;
    if (!(v >= 0.f && v <= 1.f)) { A = B = C = 0.f; return; }
    const float inv_d2 = 1.f / (d * d);
    const float dn_dmux = 2.f * muy * (a2 - a1), dd_dmux = 2.f * mux * (b2 - b1);
    const float df_dmux = -0.5f * (dn_dmux * d - n * dd_dmux) * inv_d2;
    const float df_dexx = 0.5f * (n * b1) * inv_d2;            // dn/dExx = 0, dd/dExx = b1
    const float df_dexy = -0.5f * (2.f * a1 * d) * inv_d2;     // dn/dExy = 2 a1, dd/dExy = 0
    A = df_dmux / 9.f; B = 2.f * df_dexx / 9.f; C = df_dexy / 9.f;
}

// ---------------------------------------------------------------------------------- forward
constexpr int FPH = FTH + 2, FPW = FTW + 2, FPN = FPH * FPW;     // halo-1 tile: 10 x 66

#ifndef CODEPS_PHOTO_WAVES
#define CODEPS_PHOTO_WAVES 3                  // waves per SIMD the register allocation leaves room for: 3 = 168 registers + 36 bytes of scratch, measured
                                              // 120 us per launch against 129 at 2 (177 registers) and 136 at 4 (128 registers, 200 bytes of scratch)
#endif
__global__ __launch_bounds__(256, CODEPS_PHOTO_WAVES) void photo_fwd_kernel(Levels L, const float* __restrict__ img_t,
                                                        const float* __restrict__ img_m,
                                                        const float* __restrict__ img_p,
                                                        const float* __restrict__ depth,
                                                        const float* __restrict__ pose_m,
                                                        const float* __restrict__ pose_p,
                                                        const float* __restrict__ intr, const float* __restrict__ noise,
                                                        uint64_t seed, const uint64_t* __restrict__ seed_ctr,
                                                        float alpha, float one_minus_alpha,
                                                        const float* __restrict__ pyr, uint8_t* __restrict__ sel,
                                                        float* __restrict__ loss_part, float* __restrict__ pred_out,
                                                        float* __restrict__ coef_out) {
    __shared__ float sm[15][FPN];         // tgt[3], im[3], ip[3], pred_m[3], pred_p[3]
    __shared__ float red[4];
    // device-side step counter (bumped once per step by codeps_step_counter_add): a replayed HIP graph draws fresh noise
    if (seed_ctr) seed += *seed_ctr * 0xD1B54A32D192ED03ull;
    const int bid = xcd_contiguous(blockIdx.x, gridDim.x);
    int s = 0;
    while (bid >= L.fblk[s + 1]) ++s;
    const int local = bid - L.fblk[s];
    const int tiles = L.ftx[s] * L.fty[s];
    const int b = local / tiles, t = local % tiles;
    const int ty0 = (t / L.ftx[s]) * FTH, tx0 = (t % L.ftx[s]) * FTW;
    const int hs = L.h[s], ws = L.w[s];
    const LevelPtrs P = level_ptrs(L, s, b, img_t, img_m, img_p, depth, pyr);
    const Cam k = level_cam(L, s, intr, b);
    const float* Tm = pose_m + b * 16;
    const float* Tp = pose_p + b * 16;

    for (int i = threadIdx.x; i < FPN; i += 256) {
        const int py = i / FPW, px = i % FPW;
        const int gy = reflect1(ty0 + py - 1, hs), gx = reflect1(tx0 + px - 1, ws);
        const int o = gy * ws + gx;
        const float d = P.dp[o];
        const Warp wm = warp_point(k, Tm, gx, gy, d, ws, hs);
        const Warp wp = warp_point(k, Tp, gx, gy, d, ws, hs);
        const Taps tm = make_taps(wm.ix, wm.iy, ws, hs), tp = make_taps(wp.ix, wp.iy, ws, hs);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            sm[c][i] = P.tg[c * P.hw + o];
            sm[3 + c][i] = P.im[c * P.hw + o];
            sm[6 + c][i] = P.ip[c * P.hw + o];
            sm[9 + c][i] = sample(P.im + c * P.hw, tm);
            sm[12 + c][i] = sample(P.ip + c * P.hw, tp);
        }
    }
    __syncthreads();

    float acc = 0.f;
    for (int j = threadIdx.x; j < FTH * FTW; j += 256) {
        const int oy = j / FTW, ox = j % FTW;
        const int gy = ty0 + oy, gx = tx0 + ox;
        if (gy >= hs || gx >= ws) continue;
        float ssim[4] = {0.f, 0.f, 0.f, 0.f}, l1[4] = {0.f, 0.f, 0.f, 0.f};
        float wsum[3][8];                 // per channel: sy, syy and (sx, sxx, sxy) of the two WARPED candidates, for the adjoint
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float sy = 0.f, syy = 0.f;
            float sx[4] = {0.f, 0.f, 0.f, 0.f}, sxx[4] = {0.f, 0.f, 0.f, 0.f}, sxy[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int q = (oy + dy) * FPW + ox + dx;
                    const float y = sm[c][q];
                    sy += y; syy += y * y;
                    const float xs[4] = {sm[9 + c][q], sm[12 + c][q], sm[3 + c][q], sm[6 + c][q]};
#pragma unroll
                    for (int m = 0; m < 4; ++m) { sx[m] += xs[m]; sxx[m] += xs[m] * xs[m]; sxy[m] += xs[m] * y; }
                }
            const int qc = (oy + 1) * FPW + ox + 1;
            const float yc = sm[c][qc];
            const float xc[4] = {sm[9 + c][qc], sm[12 + c][qc], sm[3 + c][qc], sm[6 + c][qc]};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                ssim[m] += ssim_val(sx[m], sy, sxx[m], syy, sxy[m]);
                l1[m] += fabsf(xc[m] - yc);
            }
            wsum[c][0] = sy; wsum[c][1] = syy;
            wsum[c][2] = sx[0]; wsum[c][3] = sxx[0]; wsum[c][4] = sxy[0];
            wsum[c][5] = sx[1]; wsum[c][6] = sxx[1]; wsum[c][7] = sxy[1];
        }
        float cand[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) cand[m] = alpha * (ssim[m] / 3.f) + one_minus_alpha * (l1[m] / 3.f);
        const long long nbase = 2 * L.map_off[s] + ((long long)(b * 2) * hs + gy) * ws + gx;
        float n0, n1;
        if (noise) { n0 = noise[nbase]; n1 = noise[nbase + (long long)hs * ws]; }
        else { n0 = hash_normal(seed, (uint64_t)nbase) * 1e-5f; n1 = hash_normal(seed, (uint64_t)(nbase + (long long)hs * ws)) * 1e-5f; }
        cand[2] += n0; cand[3] += n1;
        float best = cand[0]; int bi = 0;
#pragma unroll
        for (int m = 1; m < 4; ++m) if (cand[m] < best) { best = cand[m]; bi = m; }
        sel[L.map_off[s] + ((long long)b * hs + gy) * ws + gx] = (uint8_t)bi;
        acc += best;
        // kept for the backward: both warped samples at this pixel, and d SSIM / d window entry = A + B x + C y of the selected
        // WARPED branch (identity branches carry no gradient: zeros)
        {
            const size_t hw = (size_t)hs * ws, o = (size_t)gy * ws + gx;
            float* pr = pred_out + 6 * L.map_off[s] + (size_t)b * 6 * hw + o;
            float* cf = coef_out + 9 * L.map_off[s] + (size_t)b * 9 * hw + o;
            const int qc = (oy + 1) * FPW + ox + 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pr[(size_t)c * hw] = sm[9 + c][qc];
                pr[(size_t)(3 + c) * hw] = sm[12 + c][qc];
                float A = 0.f, B = 0.f, C = 0.f;
                if (bi < 2) {
                    const bool p1 = bi == 1;
                    ssim_grad(p1 ? wsum[c][5] : wsum[c][2], wsum[c][0], p1 ? wsum[c][6] : wsum[c][3], wsum[c][1], p1 ? wsum[c][7] : wsum[c][4], A, B, C);
                }
                cf[(size_t)c * hw] = A; cf[(size_t)(3 + c) * hw] = B; cf[(size_t)(6 + c) * hw] = C;
            }
        }
    }
    const float tot = block_sum<256>(acc, red);
    if (threadIdx.x == 0) loss_part[bid] = tot * L.pixw[s];
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ part, int n, float* __restrict__ loss) {
    __shared__ float red[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += part[i];
    const float tot = block_sum<256>(a, red);
    if (threadIdx.x == 0) loss[0] = tot;
}

// ---------------------------------------------------------------------------------- backward
constexpr int BP1H = BTH + 2, BP1W = BTW + 2, BP1N = BP1H * BP1W;     // halo-1 tile: 10 x 34

__global__ __launch_bounds__(256) void photo_bwd_kernel(Levels L, const float* __restrict__ img_t,
                                                        const float* __restrict__ img_m,
                                                        const float* __restrict__ img_p,
                                                        const float* __restrict__ depth,
                                                        const float* __restrict__ pose_m,
                                                        const float* __restrict__ pose_p,
                                                        const float* __restrict__ intr, float alpha,
                                                        float one_minus_alpha, const float* __restrict__ grad_loss,
                                                        const float* __restrict__ pyr, const uint8_t* __restrict__ sel,
                                                        const float* __restrict__ pred_in, const float* __restrict__ coef_in,
                                                        float* __restrict__ gd0, float* __restrict__ gds,
                                                        float* __restrict__ pose_part) {
    __shared__ float cf[9][BP1N];        // A[3], B[3], C[3] of the selected branch on the halo-1 tile (scaled by the loss weight)
    __shared__ int sk[BP1N];             // selected branch (0,1 = warped; >= 2 or -1 = no gradient)
    __shared__ float red[4][32];
    const int bid = xcd_contiguous(blockIdx.x, gridDim.x);
    int s = 0;
    while (bid >= L.bblk[s + 1]) ++s;
    const int local = bid - L.bblk[s];
    const int tiles = L.btx[s] * L.bty[s];
    const int b = local / tiles, t = local % tiles;
    const int ty0 = (t / L.btx[s]) * BTH, tx0 = (t % L.btx[s]) * BTW;
    const int hs = L.h[s], ws = L.w[s];
    const LevelPtrs P = level_ptrs(L, s, b, img_t, img_m, img_p, depth, pyr);
    const Cam k = level_cam(L, s, intr, b);
    const float* Tm = pose_m + b * 16;
    const float* Tp = pose_p + b * 16;
    const float gout = grad_loss[0] * L.pixw[s];
    const float w_ssim = gout * alpha / 3.f, w_l1 = gout * one_minus_alpha / 3.f;
    const size_t hw = (size_t)hs * ws;
    const float* prd = pred_in + 6 * L.map_off[s] + (size_t)b * 6 * hw;
    const float* cfi = coef_in + 9 * L.map_off[s] + (size_t)b * 9 * hw;

    // phase 1: the forward's adjoint coefficients of the selected branch at every p of the halo-1 tile (unit-stride plane reads)
    for (int i = threadIdx.x; i < BP1N; i += 256) {
        const int py = i / BP1W, px = i % BP1W;
        const int gy = ty0 + py - 1, gx = tx0 + px - 1;
        int kk = -1;
        if (gy >= 0 && gy < hs && gx >= 0 && gx < ws) kk = sel[L.map_off[s] + ((long long)b * hs + gy) * ws + gx];
        sk[i] = kk;
        if (kk == 0 || kk == 1) {
            const size_t o = (size_t)gy * ws + gx;
#pragma unroll
            for (int c = 0; c < 9; ++c) cf[c][i] = cfi[(size_t)c * hw + o] * w_ssim;
        }
    }
    __syncthreads();

    // phase 3: one pixel q per thread
    float dT[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) dT[i] = 0.f;
    {
        const int oy = threadIdx.x / BTW, ox = threadIdx.x % BTW;
        const int gy = ty0 + oy, gx = tx0 + ox;
        if (gy < hs && gx < ws) {
            const int o = gy * ws + gx;
            float xq[2][3], yq[3];               // the forward's warped samples of both source frames and the target at q
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                xq[0][c] = prd[(size_t)c * hw + o];
                xq[1][c] = prd[(size_t)(3 + c) * hw + o];
                yq[c] = P.tg[c * P.hw + o];
            }
            float g[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
            bool any[2] = {false, false};
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int pyy = gy + dy, pxx = gx + dx;
                    if (pyy < 0 || pyy >= hs || pxx < 0 || pxx >= ws) continue;
                    const int pi = (oy + 1 + dy) * BP1W + ox + 1 + dx;
                    const int kk = sk[pi];
                    if (kk != 0 && kk != 1) continue;
                    // multiplicity of q inside p's reflect-padded window
                    const float my = 1.f + ((pyy == 0 && gy == 1) ? 1.f : 0.f) + ((pyy == hs - 1 && gy == hs - 2) ? 1.f : 0.f);
                    const float mx = 1.f + ((pxx == 0 && gx == 1) ? 1.f : 0.f) + ((pxx == ws - 1 && gx == ws - 2) ? 1.f : 0.f);
                    const float m = my * mx;
                    // (compile-time indices only: a runtime-indexed register array would live in scratch memory)
                    if (kk == 0) any[0] = true; else any[1] = true;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v = m * (cf[c][pi] + cf[3 + c][pi] * (kk == 0 ? xq[0][c] : xq[1][c]) + cf[6 + c][pi] * yq[c]);
                        if (kk == 0) g[0][c] += v; else g[1][c] += v;
                    }
                }
            const int ks = sk[(oy + 1) * BP1W + ox + 1];
            if (ks == 0 || ks == 1) {
                if (ks == 0) any[0] = true; else any[1] = true;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float df = (ks == 0 ? xq[0][c] : xq[1][c]) - yq[c];
                    if (ks == 0) g[0][c] += w_l1 * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
                    else g[1][c] += w_l1 * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
                }
            }
            float gdepth = 0.f;
            const float d = P.dp[o];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if (!any[kk]) continue;
                const float* T = kk == 0 ? Tm : Tp;
                const float* src = kk == 0 ? P.im : P.ip;
                const Warp w = warp_point(k, T, gx, gy, d, ws, hs);
                const Taps tp = make_taps(w.ix, w.iy, ws, hs);
                float gix = 0.f, giy = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float* p = src + c * P.hw;
                    const float nw = p[tp.o_nw], ne = tp.e_ok ? p[tp.o_ne] : 0.f;
                    const float sw = tp.s_ok ? p[tp.o_sw] : 0.f, se = (tp.e_ok && tp.s_ok) ? p[tp.o_se] : 0.f;
                    gix += g[kk][c] * ((ne - nw) * (1.f - tp.ty) + (se - sw) * tp.ty);
                    giy += g[kk][c] * ((sw - nw) * (1.f - tp.tx) + (se - ne) * tp.tx);
                }
                const float du = gix * w.cgx, dv = giy * w.cgy;
                const float dxe = du * k.fx / w.zc, dye = dv * k.fy / w.zc;
                const float dzc = -(du * k.fx * w.xe + dv * k.fy * w.ye) / (w.zc * w.zc);
                const float dze = w.ze >= 1e-5f ? dzc : 0.f;
                const float dX = dxe / w.Wh, dY = dye / w.Wh, dZ = dze / w.Wh;
                const float dW = -(dxe * w.X + dye * w.Y + dze * w.Z) / (w.Wh * w.Wh);
                const float dP[4] = {dX, dY, dZ, dW};
                const float ph[4] = {w.px, w.py, w.pz, 1.f};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dT[kk * 16 + i * 4 + j] = dP[i] * ph[j];
                const float dpx = T[0] * dX + T[4] * dY + T[8] * dZ + T[12] * dW;
                const float dpy = T[1] * dX + T[5] * dY + T[9] * dZ + T[13] * dW;
                const float dpz = T[2] * dX + T[6] * dY + T[10] * dZ + T[14] * dW;
                gdepth += dpx * w.rx + dpy * w.ry + dpz * w.rz;
            }
            if (s == 0) gd0[(size_t)b * hs * ws + o] = gdepth;
            else gds[L.map_off[s] + (long long)b * hs * ws + o] = gdepth;
        }
    }
    // block reduction of the 2 x 16 pose gradient entries (fixed order).  Transposing butterfly: at step s the lanes whose bit s
    // is set keep the upper half of the remaining entries and hand the lower half to their partner (and vice versa), so the 32
    // entries cost 16 + 8 + 4 + 2 + 1 + 1 = 32 cross-lane moves per wave instead of 32 x 6 (one full butterfly per entry)
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
    int idx = 0;
#define CODEPS_BFLY(HALF, BIT)                                                   \
    {                                                                            \
        const bool up = (ln & (BIT)) != 0;                                       \
        idx += up ? (HALF) : 0;                                                  \
        _Pragma("unroll") for (int i = 0; i < (HALF); ++i) {                     \
            const float keep = up ? dT[i + (HALF)] : dT[i];                      \
            const float give = up ? dT[i] : dT[i + (HALF)];                      \
            dT[i] = keep + __shfl_xor(give, (BIT), 64);                          \
        }                                                                        \
    }
    CODEPS_BFLY(16, 1) CODEPS_BFLY(8, 2) CODEPS_BFLY(4, 4) CODEPS_BFLY(2, 8) CODEPS_BFLY(1, 16)
#undef CODEPS_BFLY
    dT[0] += __shfl_xor(dT[0], 32, 64);
    if (ln < 32) red[wv][idx] = dT[0];
    __syncthreads();
    if (threadIdx.x < 32) pose_part[(size_t)bid * 32 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(1024) void pose_reduce_kernel(Levels L, const float* __restrict__ part,
                                                           float* __restrict__ gpm, float* __restrict__ gpp) {
    // one block of 1024 threads per batch item; thread (e = tid % 32) sums entry e over a strided set of tiles with 4
    // independent chains (fixed assignment and order -> deterministic)
    __shared__ float acc[32][32];
    const int b = blockIdx.x, e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int s = 0; s < L.num_scales; ++s) {
        const int tiles = L.btx[s] * L.bty[s];
        const float* p0 = part + (size_t)(L.bblk[s] + b * tiles) * 32 + e;
        int i = grp;
        for (; i + 96 < tiles; i += 128) {
            a0 += p0[(size_t)i * 32]; a1 += p0[(size_t)(i + 32) * 32]; a2 += p0[(size_t)(i + 64) * 32]; a3 += p0[(size_t)(i + 96) * 32];
        }
        for (; i < tiles; i += 32) a0 += p0[(size_t)i * 32];
    }
    acc[grp][e] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < 32) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) v += acc[i][threadIdx.x];
        if (threadIdx.x < 16) gpm[b * 16 + threadIdx.x] = v;
        else gpp[b * 16 + threadIdx.x - 16] = v;
    }
}

// d depth(Y,X) += sum over levels >= 1 of the transposed bilinear down-scale
__global__ __launch_bounds__(256) void depth_combine_kernel(Levels L, const float* __restrict__ gds,
                                                            float* __restrict__ gdepth) {
    const long long total = (long long)L.batch * L.H * L.W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int X = (int)(i % L.W);
        const int Y = (int)((i / L.W) % L.H);
        const int b = (int)(i / ((long long)L.W * L.H));
        float a = gdepth[i];
        for (int s = 1; s < L.num_scales; ++s) {
            const int hs = L.h[s], ws = L.w[s];
            const float* g = gds + L.map_off[s] + (long long)b * hs * ws;
            const int yc = (int)(((float)Y + 0.5f) / L.rh[s]), xc = (int)(((float)X + 0.5f) / L.rw[s]);
            for (int y = max(yc - 1, 0); y <= min(yc + 1, hs - 1); ++y) {
                int y0, y1; float ly0, ly1;
                src_index(L.rh[s], y, L.H, hs, y0, y1, ly0, ly1);
                const float wy = (y0 == Y ? ly0 : 0.f) + (y1 == Y ? ly1 : 0.f);
                if (wy == 0.f) continue;
                for (int x = max(xc - 1, 0); x <= min(xc + 1, ws - 1); ++x) {
                    int x0, x1; float lx0, lx1;
                    src_index(L.rw[s], x, L.W, ws, x0, x1, lx0, lx1);
                    const float wx = (x0 == X ? lx0 : 0.f) + (x1 == X ? lx1 : 0.f);
                    if (wx != 0.f) a += wy * wx * g[y * ws + x];
                }
            }
        }
        gdepth[i] = a;
    }
}

static int check_desc(const codeps_photometric_desc* d) {
    CODEPS_REQUIRE(d != nullptr, CODEPS_ERR_ARG, "photometric: null descriptor");
    CODEPS_REQUIRE(d->batch > 0 && d->height > 0 && d->width > 0, CODEPS_ERR_ARG, "photometric: bad shape %dx%dx%d",
                   d->batch, d->height, d->width);
    CODEPS_REQUIRE(d->num_scales >= 1 && d->num_scales <= MAXS, CODEPS_ERR_ARG, "photometric: num_scales %d not in [1,%d]",
                   d->num_scales, MAXS);
    CODEPS_REQUIRE((d->height >> (d->num_scales - 1)) >= 2 && (d->width >> (d->num_scales - 1)) >= 2, CODEPS_ERR_ARG,
                   "photometric: coarsest scale of %dx%d with %d scales is smaller than 2x2 (reflection pad needs >= 2)",
                   d->height, d->width, d->num_scales);
    return CODEPS_OK;
}

}  // namespace

extern "C" size_t codeps_photometric_workspace_bytes(const codeps_photometric_desc* d) {
    if (check_desc(d) != CODEPS_OK) return 0;
    return make_layout(make_levels(d)).total;
}

extern "C" int codeps_photometric_fwd(const codeps_photometric_desc* d, const float* img_t, const float* img_m,
                                      const float* img_p, const float* depth, const float* pose_m, const float* pose_p,
                                      const float* intr, const float* noise, const uint64_t* noise_counter, float* loss,
                                      void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    int rc = check_desc(d);
    if (rc != CODEPS_OK) return rc;
    CODEPS_REQUIRE(img_t && img_m && img_p && depth && pose_m && pose_p && intr && loss && workspace, CODEPS_ERR_ARG,
                   "photometric_fwd: null pointer argument");
    const Levels L = make_levels(d);
    const WsLayout W = make_layout(L);
    CODEPS_REQUIRE(workspace_bytes >= W.total, CODEPS_ERR_WORKSPACE, "photometric_fwd: workspace %zu < required %zu",
                   workspace_bytes, W.total);
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    if (L.pyr_total > 0) {
        const int grid = (int)std::min<long long>((L.pyr_total + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(pyramid_kernel, dim3(grid), dim3(256), 0, st, L, img_t, img_m, img_p, depth, (float*)(ws + W.pyr));
        CODEPS_CHECK_LAUNCH("pyramid_kernel");
    }
    const int nblk = L.fblk[L.num_scales];
    hipLaunchKernelGGL(photo_fwd_kernel, dim3(nblk), dim3(256), 0, st, L, img_t, img_m, img_p, depth, pose_m, pose_p, intr,
                       noise, (uint64_t)d->noise_seed, noise_counter, d->alpha, (float)(1.0 - (double)d->alpha),
                       (const float*)(ws + W.pyr), (uint8_t*)(ws + W.sel), (float*)(ws + W.loss_part), (float*)(ws + W.pred),
                       (float*)(ws + W.coef));
    CODEPS_CHECK_LAUNCH("photo_fwd_kernel");
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, (const float*)(ws + W.loss_part), nblk, loss);
    CODEPS_CHECK_LAUNCH("loss_reduce_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_photometric_bwd(const codeps_photometric_desc* d, const float* img_t, const float* img_m,
                                      const float* img_p, const float* depth, const float* pose_m, const float* pose_p,
                                      const float* intr, const float* grad_loss, float* grad_depth, float* grad_pose_m,
                                      float* grad_pose_p, void* workspace, size_t workspace_bytes,
                                      codeps_stream_t stream) {
    int rc = check_desc(d);
    if (rc != CODEPS_OK) return rc;
    CODEPS_REQUIRE(img_t && img_m && img_p && depth && pose_m && pose_p && intr && grad_loss && grad_depth &&
                       grad_pose_m && grad_pose_p && workspace,
                   CODEPS_ERR_ARG, "photometric_bwd: null pointer argument");
    const Levels L = make_levels(d);
    const WsLayout W = make_layout(L);
    CODEPS_REQUIRE(workspace_bytes >= W.total, CODEPS_ERR_WORKSPACE, "photometric_bwd: workspace %zu < required %zu",
                   workspace_bytes, W.total);
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = L.bblk[L.num_scales];
    hipLaunchKernelGGL(photo_bwd_kernel, dim3(nblk), dim3(256), 0, st, L, img_t, img_m, img_p, depth, pose_m, pose_p, intr,
                       d->alpha, (float)(1.0 - (double)d->alpha), grad_loss, (const float*)(ws + W.pyr),
                       (const uint8_t*)(ws + W.sel), (const float*)(ws + W.pred), (const float*)(ws + W.coef), grad_depth,
                       (float*)(ws + W.gds), (float*)(ws + W.pose_part));
    CODEPS_CHECK_LAUNCH("photo_bwd_kernel");
    hipLaunchKernelGGL(pose_reduce_kernel, dim3(L.batch), dim3(1024), 0, st, L, (const float*)(ws + W.pose_part), grad_pose_m,
                       grad_pose_p);
    CODEPS_CHECK_LAUNCH("pose_reduce_kernel");
    if (L.num_scales > 1) {
        const long long total = (long long)L.batch * L.H * L.W;
        const int grid = (int)std::min<long long>((total + 255) / 256, 256 * 16);
        hipLaunchKernelGGL(depth_combine_kernel, dim3(grid), dim3(256), 0, st, L, (const float*)(ws + W.gds), grad_depth);
        CODEPS_CHECK_LAUNCH("depth_combine_kernel");
    }
    return CODEPS_OK;
}
