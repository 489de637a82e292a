// Panoptic fusion (K14) entirely on the device, bit-exact w.r.t. the reference's CPU path:
// misc/post_processing_panoptic.py:31-300 as driven by algos/instance_seg.py:207-258
// (threshold 0.1, 7x7 NMS, top_k 200 with the strict '>' rule, nearest-centre grouping with first-index
// ties, per-instance majority vote with smallest-label ties, ids = class*1000 + per-class counter in
// ascending instance order, stuff -> class*1000, void / unassigned thing -> -1).
// No host synchronisation: the number of centres stays in device memory; nothing like the reference's
// [K, H*W, 2] float tensor (0.8 GB at 512x1024) is ever materialised.  Integer atomics only.
//
//   nms_flag_kernel      thresholded 7x7 local maxima -> u8 flag per pixel
//   block_count_kernel   candidates per 1024-pixel chunk
//   select_kernel        (1 block / image) exclusive scan of chunk counts, 200-th largest score by an
//                        LDS radix select when there are >= 200 candidates, ordered list of kept centres
//   group_kernel         per thing pixel argmin_k sqrt(dy^2+dx^2) (IEEE ops, no FMA contraction),
//                        LDS-privatised per-instance class histogram
//   assign_kernel        (1 block / image) majority class + running per-class instance counters
//   paste_kernel         panoptic int16 + instance u8 maps
#include "common.h"

namespace {

constexpr int MAXC = 256;        // centre capacity (reference keeps < top_k = 200)
constexpr int NCLS = 32;         // class histogram stride (semantic ids < 32; 255 = void)
constexpr int CHUNK = 1024;
constexpr int CAND_CAP = 16384;  // compact candidate list per image (more candidates than that: the selection walks the pixels as before)

struct PanP {
    int H, W, top_k, nms, label_divisor, void_label;
    float threshold;
    unsigned long long thing_mask;   // bit c set <=> class c is a thing
    // element strides of the centre heat map (batch, pixel) and of the offset field (batch, channel, pixel): the maps may be
    // channel planes of NCHW tensors (pixel stride 1) or channels of NHWC tensors (pixel stride = padded channel count)
    long long ctr_bs, off_bs, off_cs;
    int ctr_ps, off_ps;
};

__device__ __forceinline__ float thr_val(float v, float thr) { return v > thr ? v : -1.f; }

// 7x7 (<= 7x7) window maximum as a row pass and a column pass over a 32 x 32 tile with its halo in LDS (max is associative: the same
// value as the direct 49-element maximum, 14 + 2 LDS reads per pixel instead of 49 global ones)
constexpr int NT = 32, NR = 3;
__global__ __launch_bounds__(256) void nms_flag_tile_kernel(PanP p, int tiles_x, const float* __restrict__ ctr, uint8_t* __restrict__ flag) {
    __shared__ float t[NT + 2 * NR][NT + 2 * NR + 1];
    __shared__ float rm[NT + 2 * NR][NT + 1];
    const int HW = p.H * p.W, b = blockIdx.y, r = (p.nms - 1) / 2, E = NT + 2 * r;
    const int y0 = (blockIdx.x / tiles_x) * NT, x0 = (blockIdx.x % tiles_x) * NT;
    const float* c = ctr + (size_t)b * p.ctr_bs;
    for (int i = threadIdx.x; i < E * E; i += 256) {
        const int yy = i / E, xx = i % E, y = y0 - r + yy, x = x0 - r + xx;
        t[yy][xx] = (y >= 0 && y < p.H && x >= 0 && x < p.W) ? thr_val(c[(size_t)(y * p.W + x) * p.ctr_ps], p.threshold) : -INFINITY;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < E * NT; i += 256) {
        const int yy = i / NT, xx = i % NT;
        float m = t[yy][xx];
        for (int d = 1; d <= 2 * r; ++d) m = fmaxf(m, t[yy][xx + d]);
        rm[yy][xx] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NT * NT; i += 256) {
        const int yy = i / NT, xx = i % NT, y = y0 + yy, x = x0 + xx;
        if (y >= p.H || x >= p.W) continue;
        float m = rm[yy][xx];
        for (int d = 1; d <= 2 * r; ++d) m = fmaxf(m, rm[yy + d][xx]);
        const float v = t[yy + r][xx + r];
        flag[(size_t)b * HW + y * p.W + x] = (v == m && v > 0.f) ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void nms_flag_kernel(PanP p, const float* __restrict__ ctr, uint8_t* __restrict__ flag) {
    const int HW = p.H * p.W, b = blockIdx.y, r = (p.nms - 1) / 2;
    const float* c0 = ctr + (size_t)b * p.ctr_bs;
    const int ps = p.ctr_ps;
    auto cat = [&](int i) { return c0[(size_t)i * ps]; };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const int y = i / p.W, x = i % p.W;
        const float v = thr_val(cat(i), p.threshold);
        float m = -INFINITY;
        for (int dy = -r; dy <= r; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= p.H) continue;
            for (int dx = -r; dx <= r; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= p.W) continue;
                m = fmaxf(m, thr_val(cat(yy * p.W + xx), p.threshold));
            }
        }
        flag[(size_t)b * HW + i] = (v == m && v > 0.f) ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void block_count_kernel(int HW, int nchunk, const uint8_t* __restrict__ flag, int* __restrict__ counts) {
    __shared__ float red[4];
    const int b = blockIdx.y, ch = blockIdx.x;
    const uint8_t* f = flag + (size_t)b * HW;
    int a = 0;
    for (int i = ch * CHUNK + threadIdx.x; i < min((ch + 1) * CHUNK, HW); i += 256) a += f[i];
    const float t = block_sum<256>((float)a, red);        // <= 1024: exact in fp32
    if (threadIdx.x == 0) counts[b * nchunk + ch] = (int)t;
}

// Candidates of chunk `ch` (1024 pixels) in row-major order -> cand[b][rank] = pixel index, rank = candidates in earlier chunks + rank
// inside the chunk (round 5: the selection below then walks K candidates instead of H * W pixels four times over).  One block per
// chunk that holds a candidate; a block sums the counts of the chunks in front of it (<= 512 ints).
__global__ __launch_bounds__(256) void compact_kernel(int HW, int nchunk, const uint8_t* __restrict__ flag, const int* __restrict__ counts,
                                                      int* __restrict__ cand, int cap) {
    __shared__ int sh[256];
    __shared__ int base;
    const int b = blockIdx.y, ch = blockIdx.x;
    const int* cnt = counts + b * nchunk;
    if (cnt[ch] == 0) return;                                     // block-uniform
    int a = 0;
    for (int i = threadIdx.x; i < ch; i += 256) a += cnt[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) base = sh[0];
    __syncthreads();
    // four consecutive pixels per thread: rank inside the chunk by an exclusive scan of the per-thread counts
    const uint8_t* f = flag + (size_t)b * HW;
    const int p0 = ch * CHUNK + threadIdx.x * 4;
    int k[4], n = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { k[e] = (p0 + e < HW && f[p0 + e]) ? 1 : 0; n += k[e]; }
    __syncthreads();
    sh[threadIdx.x] = n;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    int pos = base + sh[threadIdx.x] - n;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (k[e]) { if (pos < cap) cand[(size_t)b * cap + pos] = p0 + e; ++pos; }
}

// state per image: [0] K candidates, [1] Kc kept centres, [2] tau bits
__global__ __launch_bounds__(1024) void select_kernel(PanP p, int nchunk, const float* __restrict__ ctr, const uint8_t* __restrict__ flag,
                                                      int* __restrict__ counts, int* __restrict__ state, int* __restrict__ centers,
                                                      const int* __restrict__ cand, int cap) {
    __shared__ unsigned hist[256];
    __shared__ int sh_prefix, sh_rank, sh_cnt;
    __shared__ int scan[1024];
    const int b = blockIdx.x, HW = p.H * p.W, tid = threadIdx.x;
    const float* c0 = ctr + (size_t)b * p.ctr_bs;
    const int cps = p.ctr_ps;
    const uint8_t* f = flag + (size_t)b * HW;
    int* cnt = counts + b * nchunk;
    // total number of candidates
    int a = 0;
    for (int i = tid; i < nchunk; i += 1024) a += cnt[i];
    scan[tid] = a;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) { if (tid < s) scan[tid] += scan[tid + s]; __syncthreads(); }
    const int K = scan[0];
    __syncthreads();
    const int* cl = cand + (size_t)b * cap;
    const bool listed = K <= cap;                                // the compact list holds every candidate (else: the pixel walk below)
    // 200-th largest candidate score (keys: float bits of positive scores, monotone as uint32)
    float tau = -INFINITY;
    if (K >= p.top_k && listed) {
        if (tid == 0) { sh_prefix = 0; sh_rank = p.top_k; }
        __syncthreads();
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = (unsigned)sh_prefix;
            const unsigned himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
            for (int j = tid; j < K; j += 1024) {
                const unsigned key = __float_as_uint(c0[(size_t)cl[j] * cps]);
                if ((key & himask) == (prefix & himask)) atomicAdd(&hist[(key >> shift) & 0xFF], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned k = (unsigned)sh_rank, above = 0;
                int bin = 255;
                for (; bin > 0; --bin) { if (above + hist[bin] >= k) break; above += hist[bin]; }
                sh_prefix = (int)(prefix | ((unsigned)bin << shift));
                sh_rank = (int)(k - above);
            }
            __syncthreads();
        }
        tau = __uint_as_float((unsigned)sh_prefix);
    } else if (K >= p.top_k) {
        if (tid == 0) { sh_prefix = 0; sh_rank = p.top_k; }
        __syncthreads();
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = (unsigned)sh_prefix;
            const unsigned himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
            for (int i = tid; i < HW; i += 1024) {
                if (!f[i]) continue;
                const unsigned key = __float_as_uint(c0[(size_t)i * cps]);
                if ((key & himask) == (prefix & himask)) atomicAdd(&hist[(key >> shift) & 0xFF], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned k = (unsigned)sh_rank, above = 0;
                int bin = 255;
                for (; bin > 0; --bin) { if (above + hist[bin] >= k) break; above += hist[bin]; }
                sh_prefix = (int)(prefix | ((unsigned)bin << shift));
                sh_rank = (int)(k - above);
            }
            __syncthreads();
        }
        tau = __uint_as_float((unsigned)sh_prefix);
    }
    // ordered list of kept centres (row-major order), at most MAXC
    if (tid == 0) sh_cnt = 0;
    __syncthreads();
    if (listed) {                                                 // the candidates in order, 1024 at a time
        for (int base = 0; base < K; base += 1024) {
            const int j = base + tid;
            const int i = j < K ? cl[j] : 0;
            const int keep = (j < K && c0[(size_t)i * cps] > tau) ? 1 : 0;
            scan[tid] = keep;
            __syncthreads();
            for (int off = 1; off < 1024; off <<= 1) {
                const int v = tid >= off ? scan[tid - off] : 0;
                __syncthreads();
                scan[tid] += v;
                __syncthreads();
            }
            const int pos = sh_cnt + scan[tid] - keep;
            if (keep && pos < MAXC) { centers[(b * MAXC + pos) * 2] = i / p.W; centers[(b * MAXC + pos) * 2 + 1] = i % p.W; }
            __syncthreads();
            if (tid == 1023) sh_cnt += scan[1023];
            __syncthreads();
        }
    } else
    for (int base = 0; base < HW; base += 1024) {
        if (cnt[base / CHUNK] == 0) continue;               // block-uniform: most chunks hold no candidate
        const int i = base + tid;
        const int keep = (i < HW && f[i] && c0[(size_t)i * cps] > tau) ? 1 : 0;
        scan[tid] = keep;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {          // inclusive Hillis-Steele scan
            const int v = tid >= off ? scan[tid - off] : 0;
            __syncthreads();
            scan[tid] += v;
            __syncthreads();
        }
        const int pos = sh_cnt + scan[tid] - keep;
        if (keep && pos < MAXC) { centers[(b * MAXC + pos) * 2] = i / p.W; centers[(b * MAXC + pos) * 2 + 1] = i % p.W; }
        __syncthreads();
        if (tid == 1023) sh_cnt += scan[1023];
        __syncthreads();
    }
    if (tid == 0) { state[b * 4] = K; state[b * 4 + 1] = min(sh_cnt, MAXC); state[b * 4 + 2] = (int)__float_as_uint(tau); state[b * 4 + 3] = sh_cnt; }
}

__global__ __launch_bounds__(256) void group_kernel(PanP p, const uint8_t* __restrict__ sem, const float* __restrict__ offs,
                                                    const int* __restrict__ state, const int* __restrict__ centers, uint8_t* __restrict__ ins,
                                                    int* __restrict__ hist) {
    __shared__ float cy[MAXC], cx[MAXC];
    __shared__ int lh[MAXC * NCLS];
    const int b = blockIdx.y, HW = p.H * p.W;
    const int Kc = state[b * 4 + 1];
    for (int i = threadIdx.x; i < MAXC; i += 256) {
        cy[i] = i < Kc ? (float)centers[(b * MAXC + i) * 2] : 0.f;
        cx[i] = i < Kc ? (float)centers[(b * MAXC + i) * 2 + 1] : 0.f;
    }
    for (int i = threadIdx.x; i < MAXC * NCLS; i += 256) lh[i] = 0;
    __syncthreads();
    const uint8_t* s = sem + (size_t)b * HW;
    const float* oy = offs + (size_t)b * p.off_bs;
    const float* ox = oy + p.off_cs;
    const int ops = p.off_ps;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const int cls = s[i];
        int id = 0;
        if (Kc > 0 && cls < 64 && ((p.thing_mask >> cls) & 1ull)) {
            const float yy = __fadd_rn((float)(i / p.W), oy[(size_t)i * ops]), xx = __fadd_rn((float)(i % p.W), ox[(size_t)i * ops]);
            // argmin_k sqrt(d2_k), first index on ties (the reference compares the ROOTS).  The correctly rounded root is monotone, so
            // d2_k >= best d2 cannot give a strictly smaller root: only a smaller d2 takes the root (a handful of times per pixel)
            float best = INFINITY, best2 = INFINITY;
            int bi = 0;
            for (int k = 0; k < Kc; ++k) {
                const float dy = __fsub_rn(cy[k], yy), dx = __fsub_rn(cx[k], xx);
                const float d2 = __fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dx, dx));
                if (d2 < best2) {
                    const float d = __fsqrt_rn(d2);
                    if (d < best) { best = d; bi = k; best2 = d2; }
                }
            }
            id = bi + 1;
            if (cls < NCLS) atomicAdd(&lh[bi * NCLS + cls], 1);
        }
        ins[(size_t)b * HW + i] = (uint8_t)id;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MAXC * NCLS; i += 256)
        if (lh[i]) atomicAdd(&hist[(size_t)b * MAXC * NCLS + i], lh[i]);
}

__global__ __launch_bounds__(256) void assign_kernel(PanP p, const int* __restrict__ state, const int* __restrict__ hist, int* __restrict__ idmap) {
    // majority class per instance in parallel (thread k = instance k; round 4 walked 256 x 32 bins in ONE thread: 140 us of serial loads),
    // then the running per-class counters in ascending instance order (the reference's loop order) by one thread over LDS
    __shared__ int best[MAXC];
    const int b = blockIdx.x, k = threadIdx.x;
    const int Kc = state[b * 4 + 1];
    int bestc = -1;
    if (k < Kc) {
        const int* h = hist + ((size_t)b * MAXC + k) * NCLS;
        int bestn = 0;
        for (int c = 0; c < NCLS; ++c)
            if (h[c] > bestn) { bestn = h[c]; bestc = c; }              // first max = smallest label (torch.mode)
    }
    best[k] = bestc;
    __syncthreads();
    if (k != 0) return;
    int counter[NCLS];
    for (int c = 0; c < NCLS; ++c) counter[c] = 0;
    for (int i = 0; i < MAXC; ++i) {
        const int c = best[i];
        int val = -1;
        if (c >= 0) { counter[c] += 1; val = c * p.label_divisor + counter[c]; }
        idmap[b * MAXC + i] = val;
    }
}

__global__ __launch_bounds__(256) void paste_kernel(PanP p, const uint8_t* __restrict__ sem, const uint8_t* __restrict__ ins,
                                                    const int* __restrict__ idmap, int16_t* __restrict__ pan, long long* __restrict__ ins64) {
    const int b = blockIdx.y, HW = p.H * p.W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const size_t o = (size_t)b * HW + i;
        const int id = ins[o], cls = sem[o];
        int v;
        if (id > 0) v = idmap[b * MAXC + id - 1];
        else if (cls == p.void_label || (cls < 64 && ((p.thing_mask >> cls) & 1ull))) v = -1;
        else v = cls * p.label_divisor;
        pan[o] = (int16_t)v;
        if (ins64) ins64[o] = id;
    }
}

__global__ void zero_int_kernel(int* p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0;
}

}  // namespace

extern "C" size_t codeps_panoptic_workspace_bytes(int batch, int h, int w) {
    const size_t HW = (size_t)h * w, nchunk = (HW + CHUNK - 1) / CHUNK;
    return align_up((size_t)batch * HW, 256)                      // flags
           + align_up((size_t)batch * nchunk * 4, 256)            // chunk counts
           + align_up((size_t)batch * MAXC * NCLS * 4, 256)       // histograms
           + align_up((size_t)batch * MAXC * 4, 256)              // id map
           + align_up((size_t)batch * CAND_CAP * 4, 256);         // compact candidate list
}

extern "C" int codeps_panoptic_fusion(const uint8_t* semantic, const float* center, const float* offset, int batch, int h, int w,
                                      unsigned long long thing_class_mask, int label_divisor, int void_label, float threshold,
                                      int nms_kernel, int top_k, int16_t* panoptic, uint8_t* instance, int* centers, int* state,
                                      void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    return codeps_panoptic_fusion_strided(semantic, center, (long long)h * w, 1, offset, 2LL * h * w, (long long)h * w, 1, batch, h, w, thing_class_mask,
                                          label_divisor, void_label, threshold, nms_kernel, top_k, panoptic, instance, nullptr, centers, state, workspace,
                                          workspace_bytes, stream);
}

extern "C" int codeps_panoptic_fusion_strided(const uint8_t* semantic, const float* center, long long center_batch_stride, int center_pixel_stride,
                                              const float* offset, long long offset_batch_stride, long long offset_channel_stride,
                                              int offset_pixel_stride, int batch, int h, int w, unsigned long long thing_class_mask, int label_divisor,
                                              int void_label, float threshold, int nms_kernel, int top_k, int16_t* panoptic, uint8_t* instance,
                                              long long* instance_i64, int* centers, int* state, void* workspace, size_t workspace_bytes,
                                              codeps_stream_t stream) {
    CODEPS_REQUIRE(semantic && center && offset && panoptic && instance && centers && state && workspace && batch > 0 && h > 0 && w > 0,
                   CODEPS_ERR_ARG, "panoptic_fusion: null pointer / bad shape");
    CODEPS_REQUIRE(top_k >= 1 && top_k <= MAXC && (nms_kernel % 2) == 1 && label_divisor > 0, CODEPS_ERR_ARG,
                   "panoptic_fusion: top_k must be in [1,%d], nms_kernel odd", MAXC);
    CODEPS_REQUIRE(workspace_bytes >= codeps_panoptic_workspace_bytes(batch, h, w), CODEPS_ERR_WORKSPACE, "panoptic_fusion: workspace too small");
    PanP p;
    p.H = h; p.W = w; p.top_k = top_k; p.nms = nms_kernel; p.label_divisor = label_divisor; p.void_label = void_label;
    p.threshold = threshold; p.thing_mask = thing_class_mask;
    CODEPS_REQUIRE(center_pixel_stride >= 1 && offset_pixel_stride >= 1, CODEPS_ERR_ARG, "panoptic_fusion: strides must be positive");
    p.ctr_bs = center_batch_stride; p.ctr_ps = center_pixel_stride;
    p.off_bs = offset_batch_stride; p.off_cs = offset_channel_stride; p.off_ps = offset_pixel_stride;
    const int HW = h * w, nchunk = cdiv(HW, CHUNK);
    char* ws = (char*)workspace;
    uint8_t* flag = (uint8_t*)ws; ws += align_up((size_t)batch * HW, 256);
    int* counts = (int*)ws;       ws += align_up((size_t)batch * nchunk * 4, 256);
    int* hist = (int*)ws;         ws += align_up((size_t)batch * MAXC * NCLS * 4, 256);
    int* idmap = (int*)ws;        ws += align_up((size_t)batch * MAXC * 4, 256);
    int* cand = (int*)ws;
    hipStream_t st = (hipStream_t)stream;
    const int gx = std::min(cdiv(HW, 256), 1024);
    // centre list: zero beyond the kept entries (callers read the whole [B,256,2] array)
    CODEPS_REQUIRE(hipMemsetAsync(centers, 0, (size_t)batch * MAXC * 2 * sizeof(int), st) == hipSuccess, CODEPS_ERR_LAUNCH, "panoptic_fusion: memset failed");
    hipLaunchKernelGGL(zero_int_kernel, dim3(64), dim3(256), 0, st, hist, (long long)batch * MAXC * NCLS);
    if (nms_kernel <= 2 * NR + 1) {
        const int tiles_x = cdiv(w, NT), tiles_y = cdiv(h, NT);
        hipLaunchKernelGGL(nms_flag_tile_kernel, dim3(tiles_x * tiles_y, batch), dim3(256), 0, st, p, tiles_x, center, flag);
    } else {
        hipLaunchKernelGGL(nms_flag_kernel, dim3(gx, batch), dim3(256), 0, st, p, center, flag);
    }
    hipLaunchKernelGGL(block_count_kernel, dim3(nchunk, batch), dim3(256), 0, st, HW, nchunk, (const uint8_t*)flag, counts);
    hipLaunchKernelGGL(compact_kernel, dim3(nchunk, batch), dim3(256), 0, st, HW, nchunk, (const uint8_t*)flag, (const int*)counts, cand, CAND_CAP);
    hipLaunchKernelGGL(select_kernel, dim3(batch), dim3(1024), 0, st, p, nchunk, center, (const uint8_t*)flag, counts, state, centers, (const int*)cand,
                       CAND_CAP);
    // (four blocks per CU -- what the 34 KB of LDS per block allow: the centre loop is a dependent compare chain, one wave per SIMD leaves it exposed)
    hipLaunchKernelGGL(group_kernel, dim3(std::min(gx, 1024), batch), dim3(256), 0, st, p, semantic, offset, (const int*)state, (const int*)centers,
                       instance, hist);
    hipLaunchKernelGGL(assign_kernel, dim3(batch), dim3(MAXC), 0, st, p, (const int*)state, (const int*)hist, idmap);
    hipLaunchKernelGGL(paste_kernel, dim3(gx, batch), dim3(256), 0, st, p, semantic, (const uint8_t*)instance, (const int*)idmap, panoptic, instance_i64);
    CODEPS_CHECK_LAUNCH("panoptic kernels");
    return CODEPS_OK;
}
