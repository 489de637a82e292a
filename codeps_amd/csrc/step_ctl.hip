// Step control on the device: everything the reference's adaptation step decides on the host per frame, kept on the
// stream so that a whole step can be captured into ONE HIP graph and replayed without freezing any decision:
//   * codeps_step_counter_add   -- the per-step counter the photometric tie-break noise is mixed with
//   * codeps_depth_key_weights  -- the rolling "car not moving" rule of DepthAlgo.adaptation (algos/depth.py:537-558)
//                                  + the image-count weights of the per-key depth losses (depth.py:562-577)
//   * codeps_weighted_sum_*     -- sum_i w_i f_i loss_i / D over scalar losses (scripts/adapt_codeps.py:274-288)
// All single-block scalar kernels: their cost is the launch.
#include "common.h"

namespace {

constexpr int MAXK = 8;       // keys of one adaptation step (source, target, target_replay)
constexpr int MAXL = 16;      // scalar losses of one weighted sum

struct KeyCounts { float n[MAXK]; };
struct SumArgs {
    const float* loss[MAXL];
    const float* factor[MAXL];
    float* grad[MAXL];
    float w[MAXL];
};

__global__ void counter_add_kernel(unsigned long long* ctr, unsigned long long inc) { *ctr += inc; }

// state: [0] = entries in the ring (int), [1] = ring head (int), [2] = 1 when the last frame was skipped (int), [3] = last
// distance (float), [4 .. 4 + length] = ring of length + 1 distances.  The reference keeps a Python list: pop(0) when it
// holds MORE than `length` entries, skip when it holds exactly `length` and their mean exceeds 10 x the frame's distance,
// append otherwise -- so the list alternates between `length` and `length + 1` entries once it is full.
__global__ void depth_key_weights_kernel(const float* __restrict__ pose_prev, const float* __restrict__ pose_next,
                                         float* __restrict__ state, int length, KeyCounts kc, int nk, int target,
                                         float global_total, float global_target, int world, float* __restrict__ keep_io, int mode,
                                         float* __restrict__ coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float keep = 1.f;
    if (mode == 2) {
        if (keep_io) keep = *keep_io;
    } else if (target >= 0 && state != nullptr) {
        const float a0 = pose_prev[3], a1 = pose_prev[7], a2 = pose_prev[11];
        const float b0 = pose_next[3], b1 = pose_next[7], b2 = pose_next[11];
        const float na = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(a0, a0), __fmul_rn(a1, a1)), __fmul_rn(a2, a2)));
        const float nb = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(b0, b0), __fmul_rn(b1, b1)), __fmul_rn(b2, b2)));
        const float dist = __fdiv_rn(__fadd_rn(na, nb), 2.f);
        int* si = reinterpret_cast<int*>(state);
        float* ring = state + 4;
        const int cap = length + 1;
        int len = si[0], head = si[1];
        if (len > length) { head = (head + 1) % cap; --len; }
        bool skip = false;
        if (len == length) {
            float s = 0.f;                                  // Python sum(): left to right, oldest first
            for (int i = 0; i < len; ++i) s = __fadd_rn(s, ring[(head + i) % cap]);
            skip = __fdiv_rn(s, (float)length) > __fmul_rn(10.f, dist);
        }
        if (!skip) { ring[(head + len) % cap] = dist; ++len; }
        si[0] = len; si[1] = head; si[2] = skip ? 1 : 0;
        state[3] = dist;
        keep = skip ? 0.f : 1.f;
    }
    if (keep_io && mode != 2) *keep_io = keep;
    if (mode == 1) return;
    float total = 0.f;
    for (int k = 0; k < nk; ++k) {
        const float c = kc.n[k] * (k == target ? keep : 1.f);
        coef[k] = c;
        total += c;
    }
    // the denominator; 0 when every key was skipped.  Sharded: the global image count / world, so that DDP's mean over ranks
    // of (sum_k n_k^r loss_k^r) / denominator is the global count-weighted mean
    coef[nk] = global_total > 0.f ? (global_total - (1.f - keep) * global_target) / (float)world : total;
}

__global__ void weighted_sum_fwd_kernel(SumArgs a, int n, const float* __restrict__ denom, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = 0.f;
    for (int i = 0; i < n; ++i) {
        const float c = a.factor[i] ? __fmul_rn(a.w[i], *a.factor[i]) : a.w[i];
        if (c != 0.f) s = __fadd_rn(s, __fmul_rn(*a.loss[i], c));     // a skipped key contributes nothing (not 0 x NaN)
    }
    if (denom) s = (*denom > 0.f) ? __fdiv_rn(s, *denom) : 0.f;
    *out = s;
}

__global__ void weighted_sum_bwd_kernel(SumArgs a, int n, const float* __restrict__ denom, const float* __restrict__ g) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float gg = *g;
    if (denom) gg = (*denom > 0.f) ? __fdiv_rn(gg, *denom) : 0.f;
    for (int i = 0; i < n; ++i) {
        if (!a.grad[i]) continue;
        const float c = a.factor[i] ? __fmul_rn(a.w[i], *a.factor[i]) : a.w[i];
        *a.grad[i] = __fmul_rn(gg, c);
    }
}

// up to 64 int64 counters bumped by one launch: the pointers travel BY VALUE in the kernel arguments (no device table to fill)
struct CounterSet { long long* p[64]; int inc[64]; };
__global__ void counters_add_kernel(CounterSet s, int n) {
    const int i = threadIdx.x;
    if (i < n) *s.p[i] += s.inc[i];
}

}  // namespace

extern "C" int codeps_counters_add(long long* const* counters, const int* increments, int n, codeps_stream_t stream) {
    CODEPS_REQUIRE(counters && increments && n >= 0, CODEPS_ERR_ARG, "counters_add: bad arguments");
    for (int base = 0; base < n; base += 64) {
        CounterSet s;
        const int m = n - base < 64 ? n - base : 64;
        for (int i = 0; i < m; ++i) {
            CODEPS_REQUIRE(counters[base + i] != nullptr, CODEPS_ERR_ARG, "counters_add: null counter");
            s.p[i] = counters[base + i]; s.inc[i] = increments[base + i];
        }
        hipLaunchKernelGGL(counters_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, s, m);
    }
    CODEPS_CHECK_LAUNCH("counters_add_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_step_counter_add(uint64_t* counter, uint64_t increment, codeps_stream_t stream) {
    CODEPS_REQUIRE(counter != nullptr, CODEPS_ERR_ARG, "step_counter_add: null counter");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)counter,
                       (unsigned long long)increment);
    CODEPS_CHECK_LAUNCH("counter_add_kernel");
    return CODEPS_OK;
}

extern "C" size_t codeps_depth_rule_state_floats(int length) { return length > 0 ? (size_t)length + 5 : 0; }

extern "C" int codeps_depth_key_weights(const float* pose_prev, const float* pose_next, float* rule_state, int length, int n_keys,
                                        const float* key_counts, int target_index, float global_total, float global_target, int world,
                                        float* keep_io, int mode, float* coef, codeps_stream_t stream) {
    CODEPS_REQUIRE(n_keys >= 0 && n_keys <= MAXK && (n_keys == 0 || key_counts) && mode >= 0 && mode <= 2 && world >= 1, CODEPS_ERR_ARG,
                   "depth_key_weights: up to %d keys, mode 0..2, world >= 1", MAXK);
    CODEPS_REQUIRE(mode == 1 || (coef && n_keys >= 1), CODEPS_ERR_ARG, "depth_key_weights: the weights need >= 1 key and the coef buffer");
    CODEPS_REQUIRE(target_index < std::max(n_keys, 1), CODEPS_ERR_ARG, "depth_key_weights: target index %d of %d keys", target_index, n_keys);
    CODEPS_REQUIRE(mode == 2 || target_index < 0 || (pose_prev && pose_next && rule_state && length > 0), CODEPS_ERR_ARG,
                   "depth_key_weights: the rule needs both target poses, its state buffer and a window length");
    CODEPS_REQUIRE(mode != 1 || keep_io, CODEPS_ERR_ARG, "depth_key_weights: mode 1 writes the keep flag to keep_io");
    KeyCounts kc;
    for (int k = 0; k < MAXK; ++k) kc.n[k] = k < n_keys ? key_counts[k] : 0.f;
    hipLaunchKernelGGL(depth_key_weights_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, pose_prev, pose_next, rule_state, length, kc,
                       n_keys, target_index, global_total, global_target, world, keep_io, mode, coef);
    CODEPS_CHECK_LAUNCH("depth_key_weights_kernel");
    return CODEPS_OK;
}

static int fill_sum_args(SumArgs& a, const float* const* losses, const float* weights, const float* const* factors,
                         float* const* grads, int n) {
    CODEPS_REQUIRE(n >= 1 && n <= MAXL && weights, CODEPS_ERR_ARG, "weighted_sum: 1..%d losses and a weight per loss", MAXL);
    for (int i = 0; i < MAXL; ++i) {
        a.loss[i] = (i < n && losses) ? losses[i] : nullptr;
        a.factor[i] = (i < n && factors) ? factors[i] : nullptr;
        a.grad[i] = (i < n && grads) ? grads[i] : nullptr;
        a.w[i] = i < n ? weights[i] : 0.f;
    }
    return CODEPS_OK;
}

extern "C" int codeps_weighted_sum_fwd(const float* const* losses, const float* weights, const float* const* factors, int n,
                                       const float* denom, float* out, codeps_stream_t stream) {
    SumArgs a;
    int rc = fill_sum_args(a, losses, weights, factors, nullptr, n);
    if (rc != CODEPS_OK) return rc;
    CODEPS_REQUIRE(losses && out, CODEPS_ERR_ARG, "weighted_sum_fwd: null pointer argument");
    for (int i = 0; i < n; ++i) CODEPS_REQUIRE(losses[i], CODEPS_ERR_ARG, "weighted_sum_fwd: loss %d is null", i);
    hipLaunchKernelGGL(weighted_sum_fwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a, n, denom, out);
    CODEPS_CHECK_LAUNCH("weighted_sum_fwd_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_weighted_sum_bwd(const float* grad_out, const float* weights, const float* const* factors, int n,
                                       const float* denom, float* const* grads, codeps_stream_t stream) {
    SumArgs a;
    int rc = fill_sum_args(a, nullptr, weights, factors, grads, n);
    if (rc != CODEPS_OK) return rc;
    CODEPS_REQUIRE(grad_out && grads, CODEPS_ERR_ARG, "weighted_sum_bwd: null pointer argument");
    hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a, n, denom, grad_out);
    CODEPS_CHECK_LAUNCH("weighted_sum_bwd_kernel");
    return CODEPS_OK;
}
