// On-device evaluation metrics: the reference computes them once per incoming frame inside the
// teacher's "eval" pass (scripts/adapt_codeps.py:166-172 -> codeps/online_adap.py:197-243) with boolean-mask gathers, two
// full sorts per sample for the medians, torch.unique + Python dictionaries for the panoptic matching, and host syncs
// throughout.  Here:
//   K19 depth_metrics     eval/depth.py:21-68,111-131: medians by a 3-pass radix select on the float bit patterns (no sort, no
//                         gather), then one pass for the seven error sums
//   K20 confusion_matrix  eval/semantic.py:52-65: LDS histogram per block, one atomic per non-empty bin
//   K21 panoptic_stats    eval/panoptic.py:26-45,134-209: (gt id, pred id) pair counts through a hash table with
//                         wave-aggregated atomics, then the IoU matching / FN / FP rules as three tiny kernels over table slots
// All HBM-bound integer / index work: each input is read once per pass.
#include <algorithm>

#include "common.h"

namespace {

// ================================================================================================ depth metrics
constexpr int DM_BINS = 2048;

struct DmSel {                      // per (sample, array): radix-select state
    unsigned prefix;                // decided high bits of the median's key
    unsigned k;                     // rank still to find inside the current prefix
    unsigned n;                     // number of valid elements
    unsigned pad;
};

struct DmWork {                     // per sample
    unsigned hist[2][DM_BINS];
    DmSel sel[2];
    double sums[7];
    double pad;
};

__device__ __forceinline__ unsigned f2key(float v) {               // order-preserving map float -> uint
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

struct DmP {
    int h, w, use_class, class_id, crop_y0, crop_y1, crop_x0, crop_x1;
};

__device__ __forceinline__ bool dm_valid(const DmP& p, float gt, const uint8_t* cls, size_t i) {
    if (!(gt > 0.f)) return false;                                 // mask = depth_gt > 0 (eval/depth.py:37)
    if (p.use_class && cls[i] != p.class_id) return false;
    const int y = (int)(i / p.w), x = (int)(i % p.w);
    return y >= p.crop_y0 && y < p.crop_y1 && x >= p.crop_x0 && x < p.crop_x1;
}

// pass = 0, 1, 2: histogram of key bits [31:21], [20:10], [9:0] of the elements whose higher bits equal the prefix
__global__ __launch_bounds__(256) void dm_hist_kernel(DmP p, const float* __restrict__ gt, const float* __restrict__ pred,
                                                      const uint8_t* __restrict__ cls, DmWork* __restrict__ work, int pass) {
    __shared__ unsigned lh[2][DM_BINS];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * DM_BINS; i += 256) (&lh[0][0])[i] = 0;
    __syncthreads();
    const size_t n = (size_t)p.h * p.w;
    const float* g = gt + b * n;
    const float* q = pred + b * n;
    const uint8_t* c = cls ? cls + b * n : nullptr;
    const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
    const unsigned bins_mask = pass == 2 ? 1023u : 2047u;
    const unsigned hi_mask = pass == 0 ? 0u : (pass == 1 ? 0xFFE00000u : 0xFFFFFC00u);
    const unsigned pg = work[b].sel[0].prefix, pq = work[b].sel[1].prefix;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gv = g[i];
        if (!dm_valid(p, gv, c, i)) continue;
        const unsigned kg = f2key(gv), kq = f2key(q[i]);
        if ((kg & hi_mask) == pg) atomicAdd(&lh[0][(kg >> shift) & bins_mask], 1u);
        if ((kq & hi_mask) == pq) atomicAdd(&lh[1][(kq >> shift) & bins_mask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * DM_BINS; i += 256) {
        const unsigned v = (&lh[0][0])[i];
        if (v) atomicAdd(&(&work[b].hist[0][0])[i], v);
    }
}

// one block per (sample, array): locate the bin that holds rank k, extend the prefix, clear the histogram for the next pass
__global__ __launch_bounds__(256) void dm_select_kernel(DmWork* __restrict__ work, int pass) {
    __shared__ unsigned part[256];
    const int b = blockIdx.y, a = blockIdx.x, tid = threadIdx.x;
    unsigned* hist = work[b].hist[a];
    DmSel& s = work[b].sel[a];
    const int bins = pass == 2 ? 1024 : DM_BINS, per = DM_BINS / 256;
    unsigned local[DM_BINS / 256], sum = 0;
    for (int i = 0; i < per; ++i) { local[i] = hist[tid * per + i]; sum += local[i]; hist[tid * per + i] = 0; }
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned total = 0;
        for (int i = 0; i < 256; ++i) total += part[i];
        if (pass == 0) { s.n = total; s.k = total ? (total - 1) / 2 : 0; s.prefix = 0; }     // torch.median: lower median
        unsigned k = s.k, acc = 0;
        int t = 0;
        while (t < 255 && acc + part[t] <= k) { acc += part[t]; ++t; }
        part[0] = (unsigned)t;                       // thread that owns the bin
        s.k = k - acc;
    }
    __syncthreads();
    if (tid == (int)part[0]) {
        unsigned k = s.k, acc = 0;
        int i = 0;
        while (i < per - 1 && acc + local[i] <= k) { acc += local[i]; ++i; }
        const unsigned bin = (unsigned)(tid * per + i);
        const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
        s.prefix |= (bin < (unsigned)bins ? bin : (unsigned)bins - 1) << shift;
        s.k = k - acc;
    }
}

__global__ __launch_bounds__(256) void dm_stats_kernel(DmP p, const float* __restrict__ gt, const float* __restrict__ pred,
                                                       const uint8_t* __restrict__ cls, DmWork* __restrict__ work, int use_gt_scale,
                                                       float min_depth, float max_depth) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const size_t n = (size_t)p.h * p.w;
    const float* g = gt + b * n;
    const float* q = pred + b * n;
    const uint8_t* c = cls ? cls + b * n : nullptr;
    const float ratio = use_gt_scale ? key2f(work[b].sel[0].prefix) / key2f(work[b].sel[1].prefix) : 1.f;   // gt.median() / pred.median()
    float s[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float gv = g[i];
        if (!dm_valid(p, gv, c, i)) continue;
        float pv = q[i] * ratio;
        gv = fminf(fmaxf(gv, min_depth), max_depth);
        pv = fminf(fmaxf(pv, min_depth), max_depth);
        const float th = fmaxf(gv / pv, pv / gv), d = gv - pv, dl = logf(gv) - logf(pv);
        s[0] += th < 1.25f ? 1.f : 0.f;
        s[1] += th < 1.5625f ? 1.f : 0.f;
        s[2] += th < 1.953125f ? 1.f : 0.f;
        s[3] += d * d;
        s[4] += dl * dl;
        s[5] += fabsf(d) / gv;
        s[6] += d * d / gv;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const float t = block_sum<256>(s[k], red);
        if (threadIdx.x == 0 && t != 0.f) atomicAdd(&work[b].sums[k], (double)t);
    }
}

// stats [B][8] = d_a1, d_a2, d_a3, d_rmse, d_rmse_log, d_abs_rel, d_sq_rel (eval/depth.py:111-131), number of valid pixels
__global__ void dm_finish_kernel(const DmWork* __restrict__ work, float* __restrict__ stats, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double n = (double)work[b].sel[0].n;
    const double* s = work[b].sums;
    float* o = stats + b * 8;
    o[0] = (float)(s[0] / n); o[1] = (float)(s[1] / n); o[2] = (float)(s[2] / n);
    o[3] = sqrtf((float)(s[3] / n)); o[4] = sqrtf((float)(s[4] / n));
    o[5] = (float)(s[5] / n); o[6] = (float)(s[6] / n);
    o[7] = (float)n;
}

// ================================================================================================ confusion matrix
__global__ __launch_bounds__(256) void confusion_kernel(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ target, long long n,
                                                        int nc, int ignore_index, unsigned long long ignore_mask, int* __restrict__ conf,
                                                        int* __restrict__ bad) {
    extern __shared__ unsigned lconf[];
    for (int i = threadIdx.x; i < nc * nc; i += 256) lconf[i] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int t = target[i], p = pred[i];
        if (t == ignore_index || (t < 64 && ((ignore_mask >> t) & 1ull))) continue;
        if (t >= nc || p >= nc) { atomicExch(bad, 1); continue; }          // index_add_ would write outside its row
        atomicAdd(&lconf[t * nc + p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nc * nc; i += 256)
        if (lconf[i]) atomicAdd(&conf[i], (int)lconf[i]);
}

// ================================================================================================ panoptic quality
constexpr int PQ_PAIRS = 8192, PQ_IDS = 2048;

struct PqTable {                     // one per image, zero-initialised
    unsigned long long pair_key[PQ_PAIRS];      // ((enc(gt) << 32) | enc(pred)) + 1, 0 = empty
    unsigned pair_cnt[PQ_PAIRS];
    unsigned id_key[2][PQ_IDS];                 // [0] ground truth, [1] prediction: enc(id) + 1, 0 = empty
    unsigned id_area[2][PQ_IDS];
    unsigned id_void[2][PQ_IDS];                // intersection with the other image's void label ([1] is what the rules use)
    unsigned id_matched[2][PQ_IDS];
};

__device__ __forceinline__ unsigned pq_enc(int id, int void_label) { return id == void_label ? 0u : (unsigned)id + 1u; }
__device__ __forceinline__ unsigned pq_hash64(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33;
    return (unsigned)k;
}

__device__ __forceinline__ int pq_id_slot(unsigned* keys, unsigned key, bool insert) {
    unsigned slot = (key * 2654435761u) >> 21;                      // 11 bits
    for (int probes = 0; probes < PQ_IDS; ++probes) {
        const unsigned cur = insert ? atomicCAS(&keys[slot], 0u, key) : keys[slot];
        if (cur == key || (insert && cur == 0u)) return (int)slot;
        if (!insert && cur == 0u) return -1;
        slot = (slot + 1) & (PQ_IDS - 1);
    }
    return -1;
}

__global__ __launch_bounds__(256) void pq_pairs_kernel(const int* __restrict__ gt, const int* __restrict__ pred, int h, int w, int void_label,
                                                       PqTable* __restrict__ tabs, int* __restrict__ status) {
    const int b = blockIdx.y;
    const size_t n = (size_t)h * w, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    const int gi = live ? gt[b * n + i] : 0, pi = live ? pred[b * n + i] : 0;
    if (live && ((gi < 0 && gi != void_label) || (pi < 0 && pi != void_label))) atomicExch(&status[b], -2);   // negative non-void id
    const unsigned long long key = live ? (((unsigned long long)pq_enc(gi, void_label) << 32) | pq_enc(pi, void_label)) + 1ull : 0ull;
    PqTable& t = tabs[b];
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(live);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned long long lk = __shfl(key, leader, 64);
        const unsigned long long same = __ballot(key == lk) & todo;
        if (lane == leader) {
            unsigned slot = pq_hash64(lk) & (PQ_PAIRS - 1);
            int probes = 0;
            for (; probes < PQ_PAIRS; ++probes) {
                const unsigned long long prev = atomicCAS(&t.pair_key[slot], 0ull, lk);
                if (prev == 0ull || prev == lk) break;
                slot = (slot + 1) & (PQ_PAIRS - 1);
            }
            if (probes == PQ_PAIRS) atomicExch(&status[b], -1);
            else atomicAdd(&t.pair_cnt[slot], (unsigned)__popcll(same));
        }
        todo &= ~same;
    }
}

// per pair slot: segment areas (= what _get_meta_from_panoptic counts, eval/panoptic.py:26-40) and void intersections
__global__ __launch_bounds__(256) void pq_areas_kernel(PqTable* __restrict__ tabs, int* __restrict__ status) {
    const int b = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    PqTable& t = tabs[b];
    if (t.pair_key[s] == 0ull) return;
    const unsigned long long k = t.pair_key[s] - 1ull;
    const unsigned eg = (unsigned)(k >> 32), ep = (unsigned)k, cnt = t.pair_cnt[s];
    const int sg = pq_id_slot(t.id_key[0], eg + 1u, true), sp = pq_id_slot(t.id_key[1], ep + 1u, true);
    if (sg < 0 || sp < 0) { atomicExch(&status[b], -1); return; }
    atomicAdd(&t.id_area[0][sg], cnt);
    atomicAdd(&t.id_area[1][sp], cnt);
    if (eg == 0u) atomicAdd(&t.id_void[1][sp], cnt);
    if (ep == 0u) atomicAdd(&t.id_void[0][sg], cnt);
}

// per pair slot: true positives (eval/panoptic.py:163-183)
__global__ __launch_bounds__(256) void pq_match_kernel(PqTable* __restrict__ tabs, int label_divisor, int ncat, double* __restrict__ iou,
                                                       int* __restrict__ tpfpfn, int* __restrict__ status) {
    const int b = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    PqTable& t = tabs[b];
    if (t.pair_key[s] == 0ull) return;
    const unsigned long long k = t.pair_key[s] - 1ull;
    const unsigned eg = (unsigned)(k >> 32), ep = (unsigned)k;
    if (eg == 0u || ep == 0u) return;                               // void in the pair
    const int cg = (int)((eg - 1u) / (unsigned)label_divisor), cp = (int)((ep - 1u) / (unsigned)label_divisor);
    if (cg != cp) return;
    if (cg >= ncat) { atomicExch(&status[b], -3); return; }
    const int sg = pq_id_slot(t.id_key[0], eg + 1u, false), sp = pq_id_slot(t.id_key[1], ep + 1u, false);
    const long long inter = t.pair_cnt[s];
    const long long uni = (long long)t.id_area[1][sp] + (long long)t.id_area[0][sg] - inter - (long long)t.id_void[1][sp];
    const double v = (double)inter / (double)uni;
    if (v > 0.5) {
        atomicAdd(&tpfpfn[cg * 3 + 0], 1);
        atomicAdd(&iou[cg], v);
        t.id_matched[0][sg] = 1u;                                   // iou > 0.5 makes the match unique: no race on these flags
        t.id_matched[1][sp] = 1u;
    }
}

// per id slot: false negatives and false positives (eval/panoptic.py:185-199)
__global__ __launch_bounds__(256) void pq_unmatched_kernel(PqTable* __restrict__ tabs, int label_divisor, int ncat, int* __restrict__ tpfpfn,
                                                           int* __restrict__ status) {
    const int b = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    PqTable& t = tabs[b];
    const int which = s / PQ_IDS, slot = s % PQ_IDS;               // grid covers 2 * PQ_IDS
    const unsigned key = t.id_key[which][slot];
    if (key <= 1u) return;                                          // empty or the void label
    if (t.id_matched[which][slot]) return;
    const int cat = (int)((key - 2u) / (unsigned)label_divisor);
    if (cat >= ncat) { atomicExch(&status[b], -3); return; }
    if (which == 0) {
        atomicAdd(&tpfpfn[cat * 3 + 2], 1);
    } else {
        if ((double)t.id_void[1][slot] / (double)t.id_area[1][slot] > 0.5) return;
        atomicAdd(&tpfpfn[cat * 3 + 1], 1);
    }
}

}  // namespace

extern "C" size_t codeps_depth_metrics_workspace_bytes(int batch) { return batch > 0 ? (size_t)batch * sizeof(DmWork) : 0; }

extern "C" int codeps_depth_metrics(const float* depth_gt, const float* depth_pred, const uint8_t* class_map, int class_id, int batch, int h,
                                    int w, int use_gt_scale, float min_depth, float max_depth, int garg_crop, float* stats, void* workspace,
                                    size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(depth_gt && depth_pred && stats && workspace, CODEPS_ERR_ARG, "codeps_depth_metrics: null pointer");
    CODEPS_REQUIRE(batch > 0 && batch <= 65535 && h > 0 && w > 0, CODEPS_ERR_ARG, "codeps_depth_metrics: bad shape B=%d %dx%d", batch, h, w);
    CODEPS_REQUIRE(workspace_bytes >= codeps_depth_metrics_workspace_bytes(batch), CODEPS_ERR_WORKSPACE, "codeps_depth_metrics: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    DmWork* work = static_cast<DmWork*>(workspace);
    if (hipMemsetAsync(work, 0, (size_t)batch * sizeof(DmWork), st) != hipSuccess) {
        codeps_set_error("codeps_depth_metrics: hipMemsetAsync failed");
        return CODEPS_ERR_LAUNCH;
    }
    DmP p = {h, w, class_map != nullptr, class_id, 0, h, 0, w};
    if (garg_crop) {                                                   // eval/depth.py:40-46
        p.crop_y0 = (int)(0.4080 * h); p.crop_y1 = (int)(0.9891 * h);
        p.crop_x0 = (int)(0.0354 * w); p.crop_x1 = (int)(0.9638 * w);
    }
    const size_t n = (size_t)h * w;
    const dim3 grid((unsigned)std::min<size_t>((n + 255) / 256, 512), batch);
    if (use_gt_scale)
        for (int pass = 0; pass < 3; ++pass) {
            hipLaunchKernelGGL(dm_hist_kernel, grid, dim3(256), 0, st, p, depth_gt, depth_pred, class_map, work, pass);
            hipLaunchKernelGGL(dm_select_kernel, dim3(2, batch), dim3(256), 0, st, work, pass);
        }
    else {                                                             // only the count of valid pixels is needed
        hipLaunchKernelGGL(dm_hist_kernel, grid, dim3(256), 0, st, p, depth_gt, depth_pred, class_map, work, 0);
        hipLaunchKernelGGL(dm_select_kernel, dim3(2, batch), dim3(256), 0, st, work, 0);
    }
    hipLaunchKernelGGL(dm_stats_kernel, grid, dim3(256), 0, st, p, depth_gt, depth_pred, class_map, work, use_gt_scale, min_depth, max_depth);
    hipLaunchKernelGGL(dm_finish_kernel, dim3(cdiv(batch, 64)), dim3(64), 0, st, work, stats, batch);
    CODEPS_CHECK_LAUNCH("codeps_depth_metrics");
    return CODEPS_OK;
}

extern "C" int codeps_confusion_matrix(const uint8_t* pred, const uint8_t* target, long long n, int num_classes, int ignore_index,
                                       unsigned long long ignore_class_mask, int* conf, int* bad, codeps_stream_t stream) {
    CODEPS_REQUIRE(pred && target && conf && bad, CODEPS_ERR_ARG, "codeps_confusion_matrix: null pointer");
    CODEPS_REQUIRE(n >= 0 && num_classes > 0 && num_classes <= 64, CODEPS_ERR_ARG, "codeps_confusion_matrix: n=%lld classes=%d (<= 64)", n, num_classes);
    if (n == 0) return CODEPS_OK;
    const int blocks = (int)std::min<long long>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(confusion_kernel, dim3(blocks), dim3(256), (size_t)num_classes * num_classes * 4, (hipStream_t)stream, pred, target, n,
                       num_classes, ignore_index, ignore_class_mask, conf, bad);
    CODEPS_CHECK_LAUNCH("codeps_confusion_matrix");
    return CODEPS_OK;
}

extern "C" size_t codeps_panoptic_stats_workspace_bytes(int batch) { return batch > 0 ? (size_t)batch * sizeof(PqTable) : 0; }

extern "C" int codeps_panoptic_stats(const int* pan_gt, const int* pan_pred, int batch, int h, int w, int label_divisor, int void_label,
                                     int num_categories, double* iou, int* tp_fp_fn, int* status, void* workspace, size_t workspace_bytes,
                                     codeps_stream_t stream) {
    CODEPS_REQUIRE(pan_gt && pan_pred && iou && tp_fp_fn && status && workspace, CODEPS_ERR_ARG, "codeps_panoptic_stats: null pointer");
    CODEPS_REQUIRE(batch > 0 && batch <= 65535 && h > 0 && w > 0 && label_divisor > 0 && num_categories > 0 && void_label < 0, CODEPS_ERR_ARG,
                   "codeps_panoptic_stats: bad arguments B=%d %dx%d divisor=%d categories=%d void=%d", batch, h, w, label_divisor, num_categories,
                   void_label);
    CODEPS_REQUIRE(workspace_bytes >= codeps_panoptic_stats_workspace_bytes(batch), CODEPS_ERR_WORKSPACE, "codeps_panoptic_stats: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    PqTable* tabs = static_cast<PqTable*>(workspace);
    if (hipMemsetAsync(tabs, 0, (size_t)batch * sizeof(PqTable), st) != hipSuccess || hipMemsetAsync(status, 0, (size_t)batch * 4, st) != hipSuccess) {
        codeps_set_error("codeps_panoptic_stats: hipMemsetAsync failed");
        return CODEPS_ERR_LAUNCH;
    }
    const size_t n = (size_t)h * w;
    hipLaunchKernelGGL(pq_pairs_kernel, dim3((unsigned)((n + 255) / 256), batch), dim3(256), 0, st, pan_gt, pan_pred, h, w, void_label, tabs, status);
    hipLaunchKernelGGL(pq_areas_kernel, dim3(PQ_PAIRS / 256, batch), dim3(256), 0, st, tabs, status);
    hipLaunchKernelGGL(pq_match_kernel, dim3(PQ_PAIRS / 256, batch), dim3(256), 0, st, tabs, label_divisor, num_categories, iou, tp_fp_fn, status);
    hipLaunchKernelGGL(pq_unmatched_kernel, dim3(2 * PQ_IDS / 256, batch), dim3(256), 0, st, tabs, label_divisor, num_categories, tp_fp_fn, status);
    CODEPS_CHECK_LAUNCH("codeps_panoptic_stats");
    return CODEPS_OK;
}
