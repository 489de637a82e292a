// Per-pixel losses of the hot path (K11-K13), pose matrix (K6 tail), reflect-pad gradient fold and the
// fused Adam + EMA update (K15).  All HBM-bound; every reduction is two-stage and fixed-order, the
// top-k threshold comes from an integer radix select -> bitwise reproducible results.
//
// Reference: EdgeAwareSmoothnessLoss (algos/depth.py:58-107), SemanticLoss (algos/semantic_seg.py:15-67),
// CenterLoss / OffsetLoss / BinaryMaskLoss (algos/instance_seg.py:12-51), PoseHead tail + Rodrigues
// (models/pose_head.py:46-137), torch.optim.Adam (misc/train_utils.py:194-201), EMA update
// (codeps/online_adap.py:279-289).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int RB = 1024;                  // max partial blocks of the generic reductions

static inline int blocks_for(long long n) { return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, RB)); }

// final fixed-order sum of `n` partials (x scale) into out[0]
__global__ __launch_bounds__(256) void finish_sum_kernel(const float* __restrict__ part, int n, float scale, float* __restrict__ out,
                                                         int accumulate) {
    __shared__ float red[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += part[i];
    const float t = block_sum<256>(a, red);
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + t * scale;
}

// ================================================================================ smoothness
// per-item mean of disp: part[b][blk]
__global__ __launch_bounds__(256) void item_sum_kernel(const float* __restrict__ x, const float* __restrict__ y, int HW, int nblk,
                                                       float* __restrict__ part) {
    // sum over one item of x (or x*y when y != null); grid = (nblk, B)
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float* px = x + (size_t)b * HW;
    const float* py = y ? y + (size_t)b * HW : nullptr;
    float a = 0.f;
    if ((HW & 3) == 0) {                                        // 16-byte loads (the planes start at multiples of HW floats)
        const f32x4* p4 = reinterpret_cast<const f32x4*>(px);
        const f32x4* q4 = reinterpret_cast<const f32x4*>(py);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW / 4; i += nblk * 256) {
            const f32x4 v = q4 ? p4[i] * q4[i] : p4[i];
            a += (v[0] + v[1]) + (v[2] + v[3]);
        }
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += nblk * 256) a += py ? px[i] * py[i] : px[i];
    }
    const float t = block_sum<256>(a, red);
    if (threadIdx.x == 0) part[b * nblk + blockIdx.x] = t;
}

__global__ __launch_bounds__(64) void item_finish_kernel(const float* __restrict__ part, int nblk, float scale, float* __restrict__ out) {
    // one wave per item: lane l adds partials l, l + 64, ... in order, the lanes meet in a fixed-order xor tree
    const int b = blockIdx.x;
    float a = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 64) a += part[b * nblk + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (threadIdx.x == 0) out[b] = a * scale;
}

__device__ __forceinline__ float img_grad_w(const float* __restrict__ img, int HW, int i, int j) {
    // exp(-mean_c |I_i - I_j|)
    const float g = fabsf(img[i] - img[j]) + fabsf(img[HW + i] - img[HW + j]) + fabsf(img[2 * HW + i] - img[2 * HW + j]);
    return expf(-(g / 3.f));
}

__global__ __launch_bounds__(256) void smooth_fwd_kernel(const float* __restrict__ img, const float* __restrict__ disp,
                                                         const float* __restrict__ mean, int B, int H, int W, float wx, float wy,
                                                         float* __restrict__ part) {
    __shared__ float red[4];
    const int HW = H * W;
    const long long total = (long long)B * HW;
    float a = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / HW), p = (int)(i % HW), x = p % W, y = p / W;
        const float inv = 1.f / (mean[b] + 1e-7f);
        const float* im = img + (size_t)b * 3 * HW;
        const float* dp = disp + (size_t)b * HW;
        const float d0 = dp[p] * inv;
        if (x + 1 < W) a += wx * fabsf(d0 - dp[p + 1] * inv) * img_grad_w(im, HW, p, p + 1);
        if (y + 1 < H) a += wy * fabsf(d0 - dp[p + W] * inv) * img_grad_w(im, HW, p, p + W);
    }
    const float t = block_sum<256>(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// g[p] = d loss / d dhat[p]   (dhat = disp / (mean + eps))
__global__ __launch_bounds__(256) void smooth_bwd_g_kernel(const float* __restrict__ img, const float* __restrict__ disp,
                                                           const float* __restrict__ mean, int B, int H, int W, float wx, float wy,
                                                           float* __restrict__ g) {
    const int HW = H * W;
    const long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / HW), p = (int)(i % HW), x = p % W, y = p / W;
        const float* im = img + (size_t)b * 3 * HW;
        const float* dp = disp + (size_t)b * HW;
        const float d0 = dp[p];
        float a = 0.f;
        auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
        if (x + 1 < W) a += wx * sgn(d0 - dp[p + 1]) * img_grad_w(im, HW, p, p + 1);
        if (x > 0) a -= wx * sgn(dp[p - 1] - d0) * img_grad_w(im, HW, p - 1, p);
        if (y + 1 < H) a += wy * sgn(d0 - dp[p + W]) * img_grad_w(im, HW, p, p + W);
        if (y > 0) a -= wy * sgn(dp[p - W] - d0) * img_grad_w(im, HW, p - W, p);
        g[i] = a;      // sign(dhat_i - dhat_j) == sign(disp_i - disp_j) since mean + eps > 0
    }
}

// ddisp = gout * ( g/(m+eps) - (sum_q g_q disp_q) / ((m+eps)^2 HW) )
__global__ __launch_bounds__(256) void smooth_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ mean,
                                                               const float* __restrict__ gdot, const float* __restrict__ gout, int B,
                                                               int HW, float* __restrict__ ddisp) {
    const long long total = (long long)B * HW;
    const float go = gout[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / HW);
        const float m = mean[b] + 1e-7f;
        ddisp[i] = go * (g[i] / m - gdot[b] / (m * m * (float)HW));
    }
}

// ================================================================================ semantic CE + top-k
// CrossEntropyLoss applied to PROBABILITIES (the reference's quirk): loss = -w_t * (p_t - logsumexp(p)) * pixel_weight
__device__ __forceinline__ float ce_of_probs(const float* __restrict__ p, int C, int t, float& lse) {
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, p[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(p[c] - mx);
    lse = mx + logf(s);
    return lse - p[t];
}
// The same over the channels NOT in `drop` (SemanticLoss with ignore_labels, algos/semantic_seg.py:50-55: the reference slices
// the preserved channels out of the probabilities and keeps indexing them with the ORIGINAL label values, so label t selects
// the t-th preserved channel).  ct <- that channel, or -1 when t is past the preserved set (the reference raises there).
__device__ __forceinline__ float ce_of_probs_kept(const float* __restrict__ p, int C, unsigned long long drop, int t, float& lse, int& ct) {
    float mx = -INFINITY;
    int r = 0;
    ct = -1;
    for (int c = 0; c < C; ++c) {
        if ((drop >> c) & 1ull) continue;
        mx = fmaxf(mx, p[c]);
        if (r == t) ct = c;
        ++r;
    }
    float s = 0.f;
    for (int c = 0; c < C; ++c) if (!((drop >> c) & 1ull)) s += expf(p[c] - mx);
    lse = mx + logf(s);
    return ct >= 0 ? lse - p[ct] : 0.f;
}

__global__ __launch_bounds__(256) void sem_px_kernel(const float* __restrict__ probs, int C, int Cp, const uint8_t* __restrict__ target,
                                                     const uint8_t* __restrict__ pw, const float* __restrict__ cw, int ignore,
                                                     unsigned long long drop, long long N, float* __restrict__ loss_px,
                                                     uint32_t* __restrict__ hist0) {
    // hist0 != null: the first pass of the top-k radix select (the keys' high byte; no prefix yet) rides here -- one launch and one pass
    // over the loss map less
    __shared__ uint32_t h[256];
    if (hist0) { h[threadIdx.x] = 0; __syncthreads(); }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        const int t = target[i];
        float l = 0.f;
        if (t != ignore && t < C && !(drop && ((drop >> t) & 1ull))) {
            float lse;
            int ct = t;
            if (drop) l = ce_of_probs_kept(probs + (size_t)i * Cp, C, drop, t, lse, ct);
            else l = ce_of_probs(probs + (size_t)i * Cp, C, t, lse);
            // with ignore_labels the class-weight vector holds the preserved classes only: a label past them (ct < 0; the
            // reference raises) contributes 0 and must not index it
            if (cw && ct >= 0) l *= cw[t];
            if (pw) l *= (float)pw[i];
        }
        loss_px[i] = l;
        if (hist0) atomicAdd(&h[(l > 0.f ? __float_as_uint(l) : 0u) >> 24], 1u);
    }
    if (hist0) {
        __syncthreads();
        if (h[threadIdx.x]) atomicAdd(&hist0[threadIdx.x], h[threadIdx.x]);
    }
}

// sem_px_kernel through an LDS tile for Cp <= 32 (see sem_bwd_tile_kernel): the block's 256 rows arrive by 16-byte loads in memory order
__global__ __launch_bounds__(256) void sem_px_tile_kernel(const float* __restrict__ probs, int C, int Cp, const uint8_t* __restrict__ target,
                                                          const uint8_t* __restrict__ pw, const float* __restrict__ cw, int ignore,
                                                          unsigned long long drop, long long N, float* __restrict__ loss_px,
                                                          uint32_t* __restrict__ hist0) {
    __shared__ float tile[256 * 33];
    __shared__ uint32_t h[256];
    if (hist0) h[threadIdx.x] = 0;
    const int pitch = Cp + 1, q4 = Cp / 4;
    const long long chunks = (N + 255) / 256;
    for (long long ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
        const long long i0 = ch * 256;
        const int npx = (int)min((long long)256, N - i0);
        __syncthreads();                                       // the previous chunk's rows have been read (first trip: h is zero)
        for (int e = threadIdx.x; e < npx * q4; e += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(probs + (size_t)i0 * Cp + (size_t)e * 4);
            float* dst = tile + (e / q4) * pitch + (e % q4) * 4;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        __syncthreads();
        if ((int)threadIdx.x < npx) {
            const long long i = i0 + threadIdx.x;
            const float* row = tile + threadIdx.x * pitch;
            const int t = target[i];
            float l = 0.f;
            if (t != ignore && t < C && !(drop && ((drop >> t) & 1ull))) {
                float lse;
                int ct = t;
                if (drop) l = ce_of_probs_kept(row, C, drop, t, lse, ct);
                else l = ce_of_probs(row, C, t, lse);
                if (cw && ct >= 0) l *= cw[t];
                if (pw) l *= (float)pw[i];
            }
            loss_px[i] = l;
            if (hist0) atomicAdd(&h[(l > 0.f ? __float_as_uint(l) : 0u) >> 24], 1u);
        }
    }
    if (hist0) {
        __syncthreads();
        if (h[threadIdx.x]) atomicAdd(&hist0[threadIdx.x], h[threadIdx.x]);
    }
}

// radix-select state: [0] prefix (key bits found so far), [1] k remaining, [2] n_gt (elements strictly above the final bin), [3] the
// threshold's multiplicity.  Keys = float bits of non-negative losses (monotone as uint32); four passes of 8 bits, one 256-bin histogram
// each (hist[pass][256], zeroed by radix_init_kernel).  Round 5: the single-block "pick" launches between the passes are gone -- EVERY
// block of pass j + 1 (and of the top-k sum after the last pass) derives the state after pass j from hist[j] and the state before it
// (a 256-thread suffix scan: ~1 us instead of a 12 us launch with a serial walk), block 0 records it for the pass after (and, at the
// end, as the final state the backward reads).  Same integers, same threshold.
struct RadixState { uint32_t prefix, k, n_gt, mult; };

// all 256 threads of the block; `scan` = 256 words of LDS, `bin_s` = one word
__device__ __forceinline__ RadixState radix_pick(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ prev, int shift, uint32_t* scan,
                                                  int* bin_s) {
    const int t = threadIdx.x;
    const uint32_t mine = hist[t];
    scan[t] = mine;
    if (t == 0) *bin_s = 0;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {                  // inclusive SUFFIX sums: scan[t] = sum_{i >= t} hist[i]
        const uint32_t add = t + off < 256 ? scan[t + off] : 0u;
        __syncthreads();
        scan[t] += add;
        __syncthreads();
    }
    RadixState st{prev[0], prev[1], prev[2], prev[3]};
    // the reference walk: from bin 255 down, stop at the first bin whose suffix sum reaches k; bin 0 if none does
    if (t >= 1 && scan[t] >= st.k) atomicMax(bin_s, t);
    __syncthreads();
    const int bin = *bin_s;
    const uint32_t above = bin < 255 ? scan[bin + 1] : 0u;
    const uint32_t cnt = scan[bin] - above;
    __syncthreads();                                           // (scan / bin_s may be reused by the caller)
    st.prefix |= ((uint32_t)bin) << shift;
    st.k -= above;
    st.n_gt += above;
    if (shift == 0) st.mult = cnt;
    return st;
}

// pass `pass` (1 .. 3; shift = 24 - 8 pass): state before it = pick(hist[pass - 1], states[pass - 1]); block 0 records it in states[pass]
__global__ __launch_bounds__(256) void radix_hist_kernel(const float* __restrict__ v, long long N, int pass, uint32_t* __restrict__ states,
                                                         uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[256];
    __shared__ int bin_s;
    const int shift = 24 - 8 * pass;
    const RadixState st = radix_pick(hist + (pass - 1) * 256, states + (pass - 1) * 4, shift + 8, h, &bin_s);
    if (blockIdx.x == 0 && threadIdx.x == 0) { uint32_t* o = states + pass * 4; o[0] = st.prefix; o[1] = st.k; o[2] = st.n_gt; o[3] = st.mult; }
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t himask = 0xFFFFFFFFu << (shift + 8);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        const float vi = v[i];
        const uint32_t key = vi > 0.f ? __float_as_uint(vi) : 0u;
        if ((key & himask) == (st.prefix & himask)) atomicAdd(&h[(key >> shift) & 0xFF], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[pass * 256 + threadIdx.x], h[threadIdx.x]);
}

// sum of the top-k: sum_{v > tau} v + (k - n_gt) * tau     -> partials
__global__ __launch_bounds__(256) void topk_sum_kernel(const float* __restrict__ v, long long N, uint32_t* __restrict__ state,
                                                       const uint32_t* __restrict__ states, const uint32_t* __restrict__ hist,
                                                       float* __restrict__ part) {
    __shared__ float red[4];
    __shared__ uint32_t scan[256];
    __shared__ int bin_s;
    float tau;
    if (hist) {                                                // top-k: the state after the last pass, from hist[3] (block 0 records it: `state`)
        const RadixState st = radix_pick(hist + 3 * 256, states + 3 * 4, 0, scan, &bin_s);
        if (blockIdx.x == 0 && threadIdx.x == 0) { state[0] = st.prefix; state[1] = st.k; state[2] = st.n_gt; state[3] = st.mult; }
        tau = __uint_as_float(st.prefix);
    } else {
        tau = __uint_as_float(state[0]);
    }
    float a = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        const float x = fmaxf(v[i], 0.f);
        if (x > tau) a += x;
    }
    const float t = block_sum<256>(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void topk_finish_kernel(const float* __restrict__ part, int n, const uint32_t* __restrict__ state, uint32_t k,
                                                          float* __restrict__ out) {
    __shared__ float red[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += part[i];
    const float t = block_sum<256>(a, red);
    if (threadIdx.x == 0) {
        const float tau = __uint_as_float(state[0]);
        const uint32_t n_gt = state[2];
        out[0] = (t + (float)(k - n_gt) * tau) / (float)k;
    }
}

// d probs: selected pixels get gout/k * w_t * pw * (softmax(p) - onehot); ties at tau share (k-n_gt)/n_eq.
// One pixel's row: p -> d (may be the same memory: d[c] depends on p[c] and the row's constants only)
__device__ __forceinline__ void sem_bwd_row(const float* p, float* d, int C, int Cp, int t, float l, float tau, float frac, float go,
                                            const uint8_t* __restrict__ pw, const float* __restrict__ cw, int ignore, unsigned long long drop,
                                            long long i) {
    float sel = l > tau ? 1.f : (l == tau ? frac : 0.f);
    if (t == ignore || t >= C || (drop && ((drop >> t) & 1ull))) sel = 0.f;
    float lse;
    int ct = t;
    if (sel != 0.f) {
        if (drop) ce_of_probs_kept(p, C, drop, t, lse, ct); else ce_of_probs(p, C, t, lse);
        if (ct < 0) sel = 0.f;
    }
    if (sel == 0.f) {
        for (int c = 0; c < Cp; ++c) d[c] = 0.f;
        return;
    }
    float coef = go * sel;
    if (cw) coef *= cw[t];
    if (pw) coef *= (float)pw[i];
    for (int c = 0; c < C; ++c) d[c] = ((drop >> c) & 1ull) ? 0.f : coef * (expf(p[c] - lse) - (c == ct ? 1.f : 0.f));
    for (int c = C; c < Cp; ++c) d[c] = 0.f;
}

__global__ __launch_bounds__(256) void sem_bwd_kernel(const float* __restrict__ probs, int C, int Cp, const uint8_t* __restrict__ target,
                                                      const uint8_t* __restrict__ pw, const float* __restrict__ cw, int ignore,
                                                      unsigned long long drop, long long N,
                                                      const float* __restrict__ loss_px, const uint32_t* __restrict__ state, uint32_t k,
                                                      const float* __restrict__ gout, float* __restrict__ dprobs) {
    const float tau = state ? __uint_as_float(state[0]) : -1.f;
    const float frac = state ? (float)(k - state[2]) / (float)max(state[3], 1u) : 1.f;
    const float go = gout[0] / (float)k;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256)
        sem_bwd_row(probs + (size_t)i * Cp, dprobs + (size_t)i * Cp, C, Cp, target[i], fmaxf(loss_px[i], 0.f), tau, frac, go, pw, cw, ignore, drop, i);
}

// The same through LDS for Cp <= 32 (round 5): a thread per pixel walking its Cp floats reads and writes 4 bytes per lane at a stride of
// Cp floats -- 64 lanes touch 40 cache lines per instruction (2.2 TB/s at Cp = 20).  Here a block moves its 256 pixels x Cp floats with
// 16-byte accesses in memory order, each thread works on its row in LDS (odd pitch: conflict-free), in place.  Same arithmetic per row.
__global__ __launch_bounds__(256) void sem_bwd_tile_kernel(const float* __restrict__ probs, int C, int Cp, const uint8_t* __restrict__ target,
                                                           const uint8_t* __restrict__ pw, const float* __restrict__ cw, int ignore,
                                                           unsigned long long drop, long long N,
                                                           const float* __restrict__ loss_px, const uint32_t* __restrict__ state, uint32_t k,
                                                           const float* __restrict__ gout, float* __restrict__ dprobs) {
    __shared__ float tile[256 * 33];
    const float tau = state ? __uint_as_float(state[0]) : -1.f;
    const float frac = state ? (float)(k - state[2]) / (float)max(state[3], 1u) : 1.f;
    const float go = gout[0] / (float)k;
    const int pitch = Cp + 1, q4 = Cp / 4;
    const long long chunks = (N + 255) / 256;
    for (long long ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
        const long long i0 = ch * 256;
        const int npx = (int)min((long long)256, N - i0);
        __syncthreads();                                       // the previous chunk's stores have read the tile
        for (int e = threadIdx.x; e < npx * q4; e += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(probs + (size_t)i0 * Cp + (size_t)e * 4);
            float* dst = tile + (e / q4) * pitch + (e % q4) * 4;
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
        __syncthreads();
        if ((int)threadIdx.x < npx) {
            const long long i = i0 + threadIdx.x;
            float* row = tile + threadIdx.x * pitch;
            sem_bwd_row(row, row, C, Cp, target[i], fmaxf(loss_px[i], 0.f), tau, frac, go, pw, cw, ignore, drop, i);
        }
        __syncthreads();
        for (int e = threadIdx.x; e < npx * q4; e += 256) {
            const float* src = tile + (e / q4) * pitch + (e % q4) * 4;
            *reinterpret_cast<f32x4*>(dprobs + (size_t)i0 * Cp + (size_t)e * 4) = f32x4{src[0], src[1], src[2], src[3]};
        }
    }
}

// ================================================================================ instance losses
// kind 0: squared error (center), 1: absolute error (offset), 2: CE on probabilities with u8 target (thing mask)
// pred NHWC [N][Cp] (first C channels), target planar NCHW ([B][C][HW] float) or u8 [N] for kind 2
__global__ __launch_bounds__(256) void inst_fwd_kernel(int kind, const float* __restrict__ pred, int C, int Cp, const float* __restrict__ tf,
                                                       const uint8_t* __restrict__ tu, int ignore, int B, int HW, float* __restrict__ part) {
    __shared__ float red[4];
    const long long N = (long long)B * HW;
    float a = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        const float* p = pred + (size_t)i * Cp;
        if (kind == 2) {
            const int t = tu[i];
            if (t != ignore && t < C) { float lse; a += ce_of_probs(p, C, t, lse); }
        } else {
            const int b = (int)(i / HW), hw = (int)(i % HW);
            for (int c = 0; c < C; ++c) {
                const float d = p[c] - tf[((size_t)b * C + c) * HW + hw];
                a += kind == 0 ? d * d : fabsf(d);
            }
        }
    }
    const float t = block_sum<256>(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void inst_bwd_kernel(int kind, const float* __restrict__ pred, int C, int Cp, const float* __restrict__ tf,
                                                       const uint8_t* __restrict__ tu, int ignore, int B, int HW, float scale,
                                                       const float* __restrict__ gout, float* __restrict__ dpred) {
    const long long N = (long long)B * HW;
    const float go = gout[0] * scale;
    if (C <= 4 && (Cp % 4) == 0) {
        // (round 5) the heads' 1 / 2 real channels sit in the first quad of a 16-channel row: one thread per (pixel, quad), 16-byte accesses
        // in memory order -- quad 0 computes, the others write zeros (a thread per pixel issued Cp scalar stores at a Cp-float stride)
        const int q4 = Cp / 4;
        for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < N * q4; j += (long long)gridDim.x * 256) {
            const long long i = j / q4;
            f32x4 dv = {0.f, 0.f, 0.f, 0.f};
            if (j % q4 == 0) {
                const f32x4 pv = *reinterpret_cast<const f32x4*>(pred + (size_t)i * Cp);
                const float pr[4] = {pv[0], pv[1], pv[2], pv[3]};
                if (kind == 2) {
                    const int t = tu[i];
                    if (t != ignore && t < C) {
                        float lse;
                        ce_of_probs(pr, C, t, lse);
                        for (int c = 0; c < C; ++c) dv[c] = go * (expf(pr[c] - lse) - (c == t ? 1.f : 0.f));
                    }
                } else {
                    const int b = (int)(i / HW), hw = (int)(i % HW);
                    for (int c = 0; c < C; ++c) {
                        const float df = pr[c] - tf[((size_t)b * C + c) * HW + hw];
                        dv[c] = kind == 0 ? go * 2.f * df : go * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
                    }
                }
            }
            *reinterpret_cast<f32x4*>(dpred + j * 4) = dv;
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long long)gridDim.x * 256) {
        const float* p = pred + (size_t)i * Cp;
        float* d = dpred + (size_t)i * Cp;
        for (int c = C; c < Cp; ++c) d[c] = 0.f;
        if (kind == 2) {
            const int t = tu[i];
            if (t == ignore || t >= C) { for (int c = 0; c < C; ++c) d[c] = 0.f; continue; }
            float lse;
            ce_of_probs(p, C, t, lse);
            for (int c = 0; c < C; ++c) d[c] = go * (expf(p[c] - lse) - (c == t ? 1.f : 0.f));
        } else {
            const int b = (int)(i / HW), hw = (int)(i % HW);
            for (int c = 0; c < C; ++c) {
                const float df = p[c] - tf[((size_t)b * C + c) * HW + hw];
                d[c] = kind == 0 ? go * 2.f * df : go * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
            }
        }
    }
}

// ================================================================================ pose matrix
// feat [B][HW][Cp] -> vec = 0.01 * mean over HW of channels 0..5 -> T [B][16]
__global__ __launch_bounds__(256) void pose_fwd_kernel(const float* __restrict__ feat, int HW, int Cp, unsigned long long invert_mask,
                                                       float* __restrict__ vec, float* __restrict__ T) {
    __shared__ float red[4];
    __shared__ float v[6];
    const int b = blockIdx.x;
    const bool invert = (invert_mask >> (b & 63)) & 1ull;
    for (int c = 0; c < 6; ++c) {
        float a = 0.f;
        for (int i = threadIdx.x; i < HW; i += 256) a += feat[((size_t)b * HW + i) * Cp + c];
        const float t = block_sum<256>(a, red);
        if (threadIdx.x == 0) v[c] = 0.01f * (t / (float)HW);
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    for (int c = 0; c < 6; ++c) vec[b * 6 + c] = v[c];
    const float ang = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const float x = v[0] / (ang + 1e-7f), y = v[1] / (ang + 1e-7f), z = v[2] / (ang + 1e-7f);
    const float ca = cosf(ang), sa = sinf(ang), Cc = 1.f - ca;
    float R[9] = {x * (x * Cc) + ca, x * (y * Cc) - z * sa, z * (x * Cc) + y * sa,
                  x * (y * Cc) + z * sa, y * (y * Cc) + ca, y * (z * Cc) - x * sa,
                  z * (x * Cc) - y * sa, y * (z * Cc) + x * sa, z * (z * Cc) + ca};
    float* M = T + b * 16;
    if (!invert) {
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[i * 3 + j]; M[i * 4 + 3] = v[3 + i]; }
    } else {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[j * 3 + i];
            M[i * 4 + 3] = -(R[0 * 3 + i] * v[3] + R[1 * 3 + i] * v[4] + R[2 * 3 + i] * v[5]);
        }
    }
    M[12] = 0.f; M[13] = 0.f; M[14] = 0.f; M[15] = 1.f;
}

__global__ __launch_bounds__(256) void pose_bwd_kernel(const float* __restrict__ dT, const float* __restrict__ vec, int HW, int Cp,
                                                       unsigned long long invert_mask, float* __restrict__ dfeat) {
    __shared__ float dv[6];
    const int b = blockIdx.x;
    const bool invert = (invert_mask >> (b & 63)) & 1ull;
    if (threadIdx.x == 0) {
        const float* v = vec + b * 6;
        const float* dM = dT + b * 16;
        const float ang = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        const float den = ang + 1e-7f;
        const float x = v[0] / den, y = v[1] / den, z = v[2] / den;
        const float ca = cosf(ang), sa = sinf(ang), Cc = 1.f - ca;
        float dR[9], dt[3];
        if (!invert) {
            for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) dR[i * 3 + j] = dM[i * 4 + j]; dt[i] = dM[i * 4 + 3]; }
        } else {
            const float R[9] = {x * (x * Cc) + ca, x * (y * Cc) - z * sa, z * (x * Cc) + y * sa,
                                x * (y * Cc) + z * sa, y * (y * Cc) + ca, y * (z * Cc) - x * sa,
                                z * (x * Cc) - y * sa, y * (z * Cc) + x * sa, z * (z * Cc) + ca};
            // M33 = R^T, m = -R^T t
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) dR[j * 3 + i] = dM[i * 4 + j] - dM[i * 4 + 3] * v[3 + j];
            for (int j = 0; j < 3; ++j) dt[j] = -(R[j * 3 + 0] * dM[0 * 4 + 3] + R[j * 3 + 1] * dM[1 * 4 + 3] + R[j * 3 + 2] * dM[2 * 4 + 3]);
        }
        const float s01 = dR[1] + dR[3], s02 = dR[2] + dR[6], s12 = dR[5] + dR[7];
        const float a01 = dR[3] - dR[1], a02 = dR[2] - dR[6], a12 = dR[7] - dR[5];
        const float dx = dR[0] * 2.f * x * Cc + s01 * y * Cc + s02 * z * Cc + a12 * sa;
        const float dy = dR[4] * 2.f * y * Cc + s01 * x * Cc + s12 * z * Cc + a02 * sa;
        const float dz = dR[8] * 2.f * z * Cc + s02 * x * Cc + s12 * y * Cc + a01 * sa;
        const float dca = dR[0] + dR[4] + dR[8];
        const float dC = dR[0] * x * x + dR[4] * y * y + dR[8] * z * z + s01 * x * y + s02 * z * x + s12 * y * z;
        const float dsa = a01 * z + a02 * y + a12 * x;
        float dang = -dca * sa + dC * sa + dsa * ca;
        dang += -(dx * v[0] + dy * v[1] + dz * v[2]) / (den * den);
        const float inv_ang = ang > 0.f ? 1.f / ang : 0.f;
        dv[0] = dx / den + dang * v[0] * inv_ang;
        dv[1] = dy / den + dang * v[1] * inv_ang;
        dv[2] = dz / den + dang * v[2] * inv_ang;
        dv[3] = dt[0]; dv[4] = dt[1]; dv[5] = dt[2];
    }
    __syncthreads();
    const float k = 0.01f / (float)HW;
    for (int i = threadIdx.x; i < HW * Cp; i += 256) {
        const int c = i % Cp;
        dfeat[(size_t)b * HW * Cp + i] = c < 6 ? dv[c] * k : 0.f;
    }
}

// ================================================================================ reflect-pad gradient fold
// dxp [B][H+2p][W+2p][C] (gradient w.r.t. the reflect-padded tensor) -> dx [B][H][W][C]
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* __restrict__ dxp, int B, int H, int W, int C, int p,
                                                           float* __restrict__ dx) {
    const int G = C / 4, Hp = H + 2 * p, Wp = W + 2 * p;
    const long long total = (long long)B * H * W * G;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int g = (int)(i % G);
        long long pix = i / G;
        const int x = (int)(pix % W); pix /= W;
        const int y = (int)(pix % H);
        const int b = (int)(pix / H);
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + p;
        if (y >= 1 && y <= p) ys[ny++] = p - y;
        if (y <= H - 2 && H - 1 - y <= p) ys[ny++] = 2 * (H - 1) - y + p;
        xs[nx++] = x + p;
        if (x >= 1 && x <= p) xs[nx++] = p - x;
        if (x <= W - 2 && W - 1 - x <= p) xs[nx++] = 2 * (W - 1) - x + p;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int iy = 0; iy < ny; ++iy)
            for (int ix = 0; ix < nx; ++ix)
                a += *reinterpret_cast<const f32x4*>(dxp + (((size_t)b * Hp + ys[iy]) * Wp + xs[ix]) * C + g * 4);
        *reinterpret_cast<f32x4*>(dx + (((size_t)b * H + y) * W + x) * C + g * 4) = a;
    }
}

// ================================================================================ Adam (+ EMA), flat buffers
__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, float* __restrict__ ema, long long n, float lr, float b1,
                                                       float b2, float eps, float bc1, float bc2_sqrt, float grad_scale, float ema_alpha) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * grad_scale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        const float pi = p[i] - (lr / bc1) * (mi / denom);
        p[i] = pi;
        if (ema) ema[i] = ema_alpha * ema[i] + (1.f - ema_alpha) * pi;
    }
}

}  // namespace

// =============================================================================================== C ABI
extern "C" size_t codeps_loss_workspace_bytes(long long pixels) {
    // per-pixel float map + partials + radix state; generous and shape-only
    return align_up((size_t)pixels * 4, 256) + (size_t)RB * 64 * 4 + 4096;
}

extern "C" int codeps_smoothness_fwd(const float* img, const float* disp, int batch, int h, int w, float* mean_disp, float* loss,
                                     void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(img && disp && mean_disp && loss && workspace && batch > 0 && h > 1 && w > 1, CODEPS_ERR_ARG, "smoothness_fwd: bad arguments");
    CODEPS_REQUIRE(workspace_bytes >= codeps_loss_workspace_bytes((long long)batch * h * w), CODEPS_ERR_WORKSPACE, "smoothness_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    const int HW = h * w, nb = std::min(256, cdiv(HW, 1024));
    hipLaunchKernelGGL(item_sum_kernel, dim3(nb, batch), dim3(256), 0, st, disp, (const float*)nullptr, HW, nb, part);
    hipLaunchKernelGGL(item_finish_kernel, dim3(batch), dim3(64), 0, st, (const float*)part, nb, 1.f / (float)HW, mean_disp);
    CODEPS_CHECK_LAUNCH("item_sum_kernel");
    const int nblk = blocks_for((long long)batch * HW);
    float* part2 = part + (size_t)RB * 32;
    const float wx = (float)(1.0 / ((double)batch * h * (w - 1))), wy = (float)(1.0 / ((double)batch * (h - 1) * w));
    hipLaunchKernelGGL(smooth_fwd_kernel, dim3(nblk), dim3(256), 0, st, img, disp, (const float*)mean_disp, batch, h, w, wx, wy, part2);
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)part2, nblk, 1.f, loss, 0);
    CODEPS_CHECK_LAUNCH("smooth_fwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_smoothness_bwd(const float* img, const float* disp, const float* mean_disp, const float* grad_loss, int batch, int h,
                                     int w, float* grad_disp, void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(img && disp && mean_disp && grad_loss && grad_disp && workspace && batch > 0 && h > 1 && w > 1, CODEPS_ERR_ARG,
                   "smoothness_bwd: bad arguments");
    const long long N = (long long)batch * h * w;
    CODEPS_REQUIRE(workspace_bytes >= codeps_loss_workspace_bytes(N), CODEPS_ERR_WORKSPACE, "smoothness_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* g = (float*)workspace;
    float* part = (float*)((char*)workspace + align_up((size_t)N * 4, 256));
    float* gdot = part + (size_t)RB * 32;
    const int HW = h * w, nb = std::min(256, cdiv(HW, 1024));
    const float wx = (float)(1.0 / ((double)batch * h * (w - 1))), wy = (float)(1.0 / ((double)batch * (h - 1) * w));
    hipLaunchKernelGGL(smooth_bwd_g_kernel, dim3(blocks_for(N)), dim3(256), 0, st, img, disp, mean_disp, batch, h, w, wx, wy, g);
    hipLaunchKernelGGL(item_sum_kernel, dim3(nb, batch), dim3(256), 0, st, (const float*)g, disp, HW, nb, part);
    hipLaunchKernelGGL(item_finish_kernel, dim3(batch), dim3(64), 0, st, (const float*)part, nb, 1.f, gdot);
    hipLaunchKernelGGL(smooth_bwd_apply_kernel, dim3(blocks_for(N)), dim3(256), 0, st, (const float*)g, mean_disp, (const float*)gdot, grad_loss,
                       batch, HW, grad_disp);
    CODEPS_CHECK_LAUNCH("smooth_bwd kernels");
    return CODEPS_OK;
}

// state: [0..3] final state (what the backward reads), [4..19] states[pass][4] = the state BEFORE pass `pass`'s pick ... hist: [4][256]
__global__ void radix_init_kernel(uint32_t* state, uint32_t* hist, uint32_t k) {
    const int i = threadIdx.x;
    for (int j = i; j < 4 * 256; j += 256) hist[j] = 0;
    if (i < 20) state[i] = i == 5 ? k : 0;                     // states[0] = (prefix 0, k, n_gt 0, mult 0)
}

extern "C" int codeps_semantic_loss_fwd(const float* probs, int c, int c_pad, const uint8_t* target, const uint8_t* pixel_weights,
                                        const float* class_weights, int ignore_index, unsigned long long dropped_channels, long long pixels,
                                        double top_k, float* loss, void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(probs && target && loss && workspace && pixels > 0 && c > 0 && c <= c_pad && c <= 255 && top_k > 0.0 && top_k <= 1.0,
                   CODEPS_ERR_ARG, "semantic_loss_fwd: bad arguments");
    CODEPS_REQUIRE(dropped_channels == 0 || c <= 64, CODEPS_ERR_ARG, "semantic_loss_fwd: ignore labels need <= 64 classes");
    const unsigned long long drop = dropped_channels;
    CODEPS_REQUIRE(workspace_bytes >= codeps_loss_workspace_bytes(pixels), CODEPS_ERR_WORKSPACE, "semantic_loss_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* lpx = (float*)workspace;
    float* part = (float*)((char*)workspace + align_up((size_t)pixels * 4, 256));
    uint32_t* state = (uint32_t*)(part + (size_t)RB * 32);
    uint32_t* hist = state + 32;
    const int nblk = blocks_for(pixels);
    static const int tile_on = getenv("CODEPS_SEM_TILE") ? atoi(getenv("CODEPS_SEM_TILE")) : 1;          // A/B switch
    auto kpx = (tile_on && c_pad <= 32 && (c_pad % 4) == 0) ? sem_px_tile_kernel : sem_px_kernel;
    if (top_k < 1.0) {
        const uint32_t k = (uint32_t)((double)top_k * (double)pixels);     // int(top_k * numel), algos/semantic_seg.py:62
        CODEPS_REQUIRE(k >= 1, CODEPS_ERR_ARG, "semantic_loss_fwd: top-k selects zero pixels");
        // 7 launches (round 4: 12): init | per-pixel loss + pass 0 | passes 1-3, each picking the previous one's bin itself | sum (+ last pick) | finish
        hipLaunchKernelGGL(radix_init_kernel, dim3(1), dim3(256), 0, st, state, hist, k);
        hipLaunchKernelGGL(kpx, dim3(blocks_for(pixels)), dim3(256), 0, st, probs, c, c_pad, target, pixel_weights, class_weights,
                           ignore_index, drop, pixels, lpx, hist);
        CODEPS_CHECK_LAUNCH("sem_px_kernel");
        for (int pass = 1; pass < 4; ++pass)
            hipLaunchKernelGGL(radix_hist_kernel, dim3(blocks_for(pixels)), dim3(256), 0, st, (const float*)lpx, pixels, pass, state + 4, hist);
        hipLaunchKernelGGL(topk_sum_kernel, dim3(nblk), dim3(256), 0, st, (const float*)lpx, pixels, state, (const uint32_t*)(state + 4),
                           (const uint32_t*)hist, part);
        hipLaunchKernelGGL(topk_finish_kernel, dim3(1), dim3(256), 0, st, (const float*)part, nblk, (const uint32_t*)state, k, loss);
    } else {
        hipLaunchKernelGGL(kpx, dim3(blocks_for(pixels)), dim3(256), 0, st, probs, c, c_pad, target, pixel_weights, class_weights,
                           ignore_index, drop, pixels, lpx, (uint32_t*)nullptr);
        CODEPS_CHECK_LAUNCH("sem_px_kernel");
        hipLaunchKernelGGL(radix_init_kernel, dim3(1), dim3(256), 0, st, state, hist, 0u);     // tau = 0, n_gt = 0: unused
        // plain mean over all pixels (tau = +0: every positive loss is summed)
        hipLaunchKernelGGL(topk_sum_kernel, dim3(nblk), dim3(256), 0, st, (const float*)lpx, pixels, state, (const uint32_t*)nullptr,
                           (const uint32_t*)nullptr, part);
        hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)part, nblk, (float)(1.0 / (double)pixels), loss, 0);
    }
    CODEPS_CHECK_LAUNCH("semantic_loss_fwd kernels");
    return CODEPS_OK;
}

extern "C" int codeps_semantic_loss_bwd(const float* probs, int c, int c_pad, const uint8_t* target, const uint8_t* pixel_weights,
                                        const float* class_weights, int ignore_index, unsigned long long dropped_channels, long long pixels,
                                        double top_k, const float* grad_loss, float* grad_probs, void* workspace, size_t workspace_bytes,
                                        codeps_stream_t stream) {
    CODEPS_REQUIRE(probs && target && grad_loss && grad_probs && workspace && pixels > 0 && c > 0 && c <= c_pad, CODEPS_ERR_ARG,
                   "semantic_loss_bwd: bad arguments");
    CODEPS_REQUIRE(workspace_bytes >= codeps_loss_workspace_bytes(pixels), CODEPS_ERR_WORKSPACE, "semantic_loss_bwd: workspace too small");
    const float* lpx = (const float*)workspace;
    const float* part = (const float*)((const char*)workspace + align_up((size_t)pixels * 4, 256));
    const uint32_t* state = (const uint32_t*)(part + (size_t)RB * 32);
    const bool topk = top_k < 1.0;
    const uint32_t k = topk ? (uint32_t)((double)top_k * (double)pixels) : (uint32_t)pixels;
    static const int tile_on = getenv("CODEPS_SEM_TILE") ? atoi(getenv("CODEPS_SEM_TILE")) : 1;          // A/B switch
    auto kb = (tile_on && c_pad <= 32 && (c_pad % 4) == 0) ? sem_bwd_tile_kernel : sem_bwd_kernel;
    hipLaunchKernelGGL(kb, dim3(blocks_for(pixels)), dim3(256), 0, (hipStream_t)stream, probs, c, c_pad, target, pixel_weights,
                       class_weights, ignore_index, dropped_channels, pixels, lpx, topk ? state : (const uint32_t*)nullptr, k, grad_loss, grad_probs);
    CODEPS_CHECK_LAUNCH("sem_bwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_instance_loss_fwd(int kind, const float* pred, int c, int c_pad, const float* target_f, const uint8_t* target_u8,
                                        int ignore_index, int batch, int hw, float* loss, void* workspace, size_t workspace_bytes,
                                        codeps_stream_t stream) {
    CODEPS_REQUIRE(kind >= 0 && kind <= 2 && pred && loss && workspace && c > 0 && c <= c_pad && batch > 0 && hw > 0 &&
                       (kind == 2 ? target_u8 != nullptr : target_f != nullptr),
                   CODEPS_ERR_ARG, "instance_loss_fwd: bad arguments");
    CODEPS_REQUIRE(workspace_bytes >= (size_t)RB * 4, CODEPS_ERR_WORKSPACE, "instance_loss_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const long long N = (long long)batch * hw;
    const int nblk = blocks_for(N);
    hipLaunchKernelGGL(inst_fwd_kernel, dim3(nblk), dim3(256), 0, st, kind, pred, c, c_pad, target_f, target_u8, ignore_index, batch, hw,
                       (float*)workspace);
    const double denom = kind == 2 ? (double)N : (double)N * c;
    hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, nblk, (float)(1.0 / denom), loss, 0);
    CODEPS_CHECK_LAUNCH("inst_fwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_instance_loss_bwd(int kind, const float* pred, int c, int c_pad, const float* target_f, const uint8_t* target_u8,
                                        int ignore_index, int batch, int hw, const float* grad_loss, float* grad_pred,
                                        codeps_stream_t stream) {
    CODEPS_REQUIRE(kind >= 0 && kind <= 2 && pred && grad_loss && grad_pred && c > 0 && c <= c_pad && batch > 0 && hw > 0 &&
                       (kind == 2 ? target_u8 != nullptr : target_f != nullptr),
                   CODEPS_ERR_ARG, "instance_loss_bwd: bad arguments");
    const long long N = (long long)batch * hw;
    const double denom = kind == 2 ? (double)N : (double)N * c;
    hipLaunchKernelGGL(inst_bwd_kernel, dim3(blocks_for(N)), dim3(256), 0, (hipStream_t)stream, kind, pred, c, c_pad, target_f, target_u8,
                       ignore_index, batch, hw, (float)(1.0 / denom), grad_loss, grad_pred);
    CODEPS_CHECK_LAUNCH("inst_bwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_pose_fwd(const float* feat, int batch, int hw, int c_pad, unsigned long long invert_mask, float* vec6, float* T,
                               codeps_stream_t stream) {
    CODEPS_REQUIRE(feat && vec6 && T && batch > 0 && hw > 0 && c_pad >= 6, CODEPS_ERR_ARG, "pose_fwd: bad arguments");
    CODEPS_REQUIRE(batch <= 64 || invert_mask == 0 || invert_mask == ~0ull, CODEPS_ERR_ARG, "pose_fwd: a per-item invert mask covers 64 items");
    hipLaunchKernelGGL(pose_fwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, feat, hw, c_pad, invert_mask, vec6, T);
    CODEPS_CHECK_LAUNCH("pose_fwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_pose_bwd(const float* grad_T, const float* vec6, int batch, int hw, int c_pad, unsigned long long invert_mask,
                               float* grad_feat, codeps_stream_t stream) {
    CODEPS_REQUIRE(grad_T && vec6 && grad_feat && batch > 0 && hw > 0 && c_pad >= 6, CODEPS_ERR_ARG, "pose_bwd: bad arguments");
    CODEPS_REQUIRE(batch <= 64 || invert_mask == 0 || invert_mask == ~0ull, CODEPS_ERR_ARG, "pose_bwd: a per-item invert mask covers 64 items");
    hipLaunchKernelGGL(pose_bwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, grad_T, vec6, hw, c_pad, invert_mask, grad_feat);
    CODEPS_CHECK_LAUNCH("pose_bwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_reflect_pad_fold(const float* grad_padded, int batch, int h, int w, int c, int pad, float* grad_x, codeps_stream_t stream) {
    CODEPS_REQUIRE(grad_padded && grad_x && batch > 0 && h > pad && w > pad && pad >= 1 && (c % 4) == 0, CODEPS_ERR_ARG, "reflect_pad_fold: bad arguments");
    const long long items = (long long)batch * h * w * (c / 4);
    hipLaunchKernelGGL(reflect_fold_kernel, dim3((int)std::min<long long>((items + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, grad_padded,
                       batch, h, w, c, pad, grad_x);
    CODEPS_CHECK_LAUNCH("reflect_fold_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_adam_ema_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema_params, long long n,
                                    int step, float lr, float beta1, float beta2, float eps, float grad_scale, float ema_alpha,
                                    codeps_stream_t stream) {
    CODEPS_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, CODEPS_ERR_ARG, "adam_ema_step: bad arguments");
    const float bc1 = (float)(1.0 - pow((double)beta1, step)), bc2s = (float)sqrt(1.0 - pow((double)beta2, step));
    hipLaunchKernelGGL(adam_ema_kernel, dim3((int)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                       exp_avg_sq, ema_params, n, lr, beta1, beta2, eps, bc1, bc2s, grad_scale, ema_alpha);
    CODEPS_CHECK_LAUNCH("adam_ema_kernel");
    return CODEPS_OK;
}
