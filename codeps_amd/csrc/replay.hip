// Replay-buffer diversity sampler on the device: ReplayBuffer.add with maximize_diversity
// (datasets/replay_buffer.py:387-455), called once per frame with the shared encoder's last feature map
// (scripts/adapt_codeps.py:307).  The reference pools the [1,2048,h,w] map on the GPU, copies it to the host, L2-normalises
// it and asks a faiss inner-product index (faiss-gpu==1.7.2, requirements.txt) for the most similar stored vector; a frame
// that is different enough is stored, and once the buffer overflows the entry with the largest summed cosine similarity
// to all others is evicted (pairwise matrix kept on the host).  Here the stored vectors, the (cap+1)^2 similarity matrix and
// the ids live in ONE caller-owned device buffer and a frame costs three small launches, no host round trip:
//   replay_pool    mean over w, then over h, of the NHWC feature map            (4 MB read at config 3)
//   replay_scores  one block per slot: cosine similarity of the normalised query with the stored vector  (2.4 MB read)
//   replay_decide  one block: threshold test, insertion (vector + matrix row / column), eviction by arg-max of the
//                  column sums without the diagonal
// The matrix is maintained incrementally from the first insertion; the reference builds it at the first overflow and
// updates one row / column per insertion afterwards -- the same inner products either way.
#include "common.h"

namespace {

struct ReplayState {                // header of the state buffer; arrays follow (see layout())
    int count;                      // vectors in the index
    int error;                      // sticky: 1 = no free slot
    int pad[2];
};

struct Layout {
    long long* ids;                 // [slots]   sample index, -1 = free
    float* feats;                   // [slots][dim]
    float* sim;                     // [slots][slots]
    float* pooled;                  // [dim]  scratch: pooled query
    float* qn;                      // [dim]  scratch: normalised query
    float* sims;                    // [slots] scratch: similarity of the query to every slot
    float* self;                    // [1]    scratch: <qn, qn>
};

__host__ __device__ inline size_t up16(size_t v) { return (v + 15) & ~(size_t)15; }

__host__ __device__ inline size_t state_bytes(int capacity, int dim) {
    const size_t slots = (size_t)capacity + 1;
    return up16(sizeof(ReplayState)) + up16(slots * 8) + up16(slots * dim * 4) + up16(slots * slots * 4) + 2 * up16((size_t)dim * 4) +
           up16(slots * 4) + 16;
}

__host__ __device__ inline Layout layout(void* state, int capacity, int dim) {
    const size_t slots = (size_t)capacity + 1;
    char* p = static_cast<char*>(state) + up16(sizeof(ReplayState));
    Layout l;
    l.ids = reinterpret_cast<long long*>(p);    p += up16(slots * 8);
    l.feats = reinterpret_cast<float*>(p);      p += up16(slots * dim * 4);
    l.sim = reinterpret_cast<float*>(p);        p += up16(slots * slots * 4);
    l.pooled = reinterpret_cast<float*>(p);     p += up16((size_t)dim * 4);
    l.qn = reinterpret_cast<float*>(p);         p += up16((size_t)dim * 4);
    l.sims = reinterpret_cast<float*>(p);       p += up16(slots * 4);
    l.self = reinterpret_cast<float*>(p);
    return l;
}

// image_features.mean(-1).mean(-1) (replay_buffer.py:392) on an NHWC map: rows of `stride` floats per pixel.
// Block = 64 channels x 4 row groups; per image row the sum over w is divided by w, the row means are summed and divided by h.
__global__ __launch_bounds__(256) void replay_pool_kernel(const float* __restrict__ f, int h, int w, int stride, int dim, float* __restrict__ pooled) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < dim)
        for (int y = g; y < h; y += 4) {
            float row = 0.f;
            const float* p = f + ((size_t)y * w) * stride + c;
            for (int x = 0; x < w; ++x) row += p[(size_t)x * stride];
            acc += row / (float)w;
        }
    part[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (g == 0 && c < dim) pooled[c] = (part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]) / (float)h;
}

// One block per slot: inverse norm of the query (faiss::fvec_renorm_L2: x *= 1 / sqrtf(<x, x>) when the norm is positive),
// similarity to the slot's vector; block 0 also stores the normalised query and its self-similarity.
__global__ __launch_bounds__(256) void replay_scores_kernel(void* state, int capacity, int dim, int prenormalised) {
    __shared__ float red[4];
    Layout l = layout(state, capacity, dim);
    const int slot = blockIdx.x;
    float nr = 0.f;
    for (int i = threadIdx.x; i < dim; i += 256) nr += l.pooled[i] * l.pooled[i];
    nr = block_sum<256>(nr, red);
    const float inv = (prenormalised || !(nr > 0.f)) ? 1.f : 1.f / sqrtf(nr);
    const bool valid = l.ids[slot] >= 0;
    float dot = 0.f, self = 0.f;
    const float* v = l.feats + (size_t)slot * dim;
    for (int i = threadIdx.x; i < dim; i += 256) {
        const float q = l.pooled[i] * inv;
        if (slot == 0) { l.qn[i] = q; self += q * q; }
        if (valid) dot += v[i] * q;
    }
    dot = block_sum<256>(dot, red);
    if (slot == 0) {
        self = block_sum<256>(self, red);
        if (threadIdx.x == 0) *l.self = self;
    }
    if (threadIdx.x == 0) l.sims[slot] = dot;
}

// value / index arg-max over a block of 1024 threads, first index wins ties (np.argmax)
__device__ __forceinline__ void block_argmax(float& v, int& idx, float* sv, int* si) {
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sv[wv] = v; si[wv] = idx; }
    __syncthreads();
    v = sv[0]; idx = si[0];
    for (int i = 1; i < 16; ++i)
        if (sv[i] > v || (sv[i] == v && si[i] < idx)) { v = sv[i]; idx = si[i]; }
}

__global__ __launch_bounds__(1024) void replay_decide_kernel(void* state, int capacity, int dim, long long index, float threshold, int force,
                                                             long long* __restrict__ result) {
    __shared__ float sv[16];
    __shared__ int si[16];
    ReplayState* hdr = static_cast<ReplayState*>(state);
    Layout l = layout(state, capacity, dim);
    const int slots = capacity + 1, tid = threadIdx.x;
    const int count = hdr->count;
    // most similar stored vector (IndexFlatIP.search(q, 1)); an empty index counts as similarity 0 (replay_buffer.py:404-405)
    float best = -INFINITY;
    int bi = 0x7fffffff, free_slot = 0x7fffffff;
    for (int j = tid; j < slots; j += 1024) {
        if (l.ids[j] >= 0) { if (l.sims[j] > best) { best = l.sims[j]; bi = j; } }
        else free_slot = min(free_slot, j);
    }
    block_argmax(best, bi, sv, si);
    if (count == 0) best = 0.f;
    float nf = -(float)free_slot;                       // lowest free slot = arg-max of the negated index
    int fi = free_slot;
    block_argmax(nf, fi, sv, si);
    free_slot = fi;
    const bool add = force || best < threshold;
    long long removed = -1;
    int added = add ? 1 : 0;
    if (add && free_slot > capacity) {                  // cannot happen while count <= capacity on entry; reported, not ignored
        added = -1;
        if (tid == 0) hdr->error = 1;
    } else if (add) {
        const int f = free_slot;
        for (int i = tid; i < dim; i += 1024) l.feats[(size_t)f * dim + i] = l.qn[i];
        for (int j = tid; j < slots; j += 1024) {
            const float s = (j == f) ? *l.self : (l.ids[j] >= 0 ? l.sims[j] : -1.f);
            l.sim[(size_t)f * slots + j] = s;
            l.sim[(size_t)j * slots + f] = s;
        }
        __syncthreads();
        if (tid == 0) l.ids[f] = index;
        __syncthreads();
        if (!force && count + 1 > capacity) {           // evict the entry most similar to everything else (replay_buffer.py:439-445)
            float bs = -INFINITY;
            int bj = 0x7fffffff;
            for (int j = tid; j < slots; j += 1024) {
                if (l.ids[j] < 0) continue;
                float s = 0.f;
                for (int i = 0; i < slots; ++i)
                    if (l.ids[i] >= 0) s += l.sim[(size_t)i * slots + j];
                s -= l.sim[(size_t)j * slots + j];
                if (s > bs) { bs = s; bj = j; }
            }
            block_argmax(bs, bj, sv, si);
            if (tid == 0) {
                removed = l.ids[bj];
                l.ids[bj] = -1;
                hdr->count = count;                     // +1 -1
            }
        } else if (tid == 0) {
            hdr->count = count + 1;
        }
    }
    if (tid == 0) {
        result[0] = added;
        result[1] = removed;
        result[2] = (long long)__float_as_uint(best);
    }
}

}  // namespace

extern "C" size_t codeps_replay_state_bytes(int capacity, int dim) { return (capacity > 0 && dim > 0) ? state_bytes(capacity, dim) : 0; }

extern "C" int codeps_replay_reset(void* state, size_t bytes, int capacity, int dim, codeps_stream_t stream) {
    CODEPS_REQUIRE(state && capacity > 0 && dim > 0, CODEPS_ERR_ARG, "codeps_replay_reset: bad arguments");
    CODEPS_REQUIRE(bytes >= state_bytes(capacity, dim), CODEPS_ERR_WORKSPACE, "codeps_replay_reset: state of %zu bytes, %zu needed", bytes,
                   state_bytes(capacity, dim));
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(state, 0, state_bytes(capacity, dim), st) != hipSuccess) {
        codeps_set_error("codeps_replay_reset: hipMemsetAsync failed");
        return CODEPS_ERR_LAUNCH;
    }
    Layout l = layout(state, capacity, dim);
    if (hipMemsetAsync(l.ids, 0xFF, ((size_t)capacity + 1) * 8, st) != hipSuccess) {       // all ids = -1
        codeps_set_error("codeps_replay_reset: hipMemsetAsync failed");
        return CODEPS_ERR_LAUNCH;
    }
    return CODEPS_OK;
}

extern "C" int codeps_replay_add(void* state, size_t bytes, int capacity, int dim, const float* features, int h, int w, int stride,
                                 long long index, float similarity_threshold, int force, long long* result, codeps_stream_t stream) {
    CODEPS_REQUIRE(state && features && result, CODEPS_ERR_ARG, "codeps_replay_add: null pointer");
    CODEPS_REQUIRE(capacity > 0 && dim > 0 && h > 0 && w > 0 && stride >= dim && index >= 0, CODEPS_ERR_ARG,
                   "codeps_replay_add: bad arguments capacity=%d dim=%d map %dx%d stride=%d index=%lld", capacity, dim, h, w, stride, index);
    CODEPS_REQUIRE(bytes >= state_bytes(capacity, dim), CODEPS_ERR_WORKSPACE, "codeps_replay_add: state of %zu bytes, %zu needed", bytes,
                   state_bytes(capacity, dim));
    hipStream_t st = (hipStream_t)stream;
    Layout l = layout(state, capacity, dim);
    hipLaunchKernelGGL(replay_pool_kernel, dim3(cdiv(dim, 64)), dim3(256), 0, st, features, h, w, stride, dim, l.pooled);
    hipLaunchKernelGGL(replay_scores_kernel, dim3(capacity + 1), dim3(256), 0, st, state, capacity, dim, force);
    hipLaunchKernelGGL(replay_decide_kernel, dim3(1), dim3(1024), 0, st, state, capacity, dim, index, similarity_threshold, force, result);
    CODEPS_CHECK_LAUNCH("codeps_replay_add");
    return CODEPS_OK;
}

// Read-out of the index for save_state (replay_buffer.py:462-473): ids int64 [capacity+1] (-1 = free slot) and the stored
// (normalised) vectors fp32 [capacity+1][dim], device to device.
extern "C" int codeps_replay_export(const void* state, size_t bytes, int capacity, int dim, long long* ids, float* feats, codeps_stream_t stream) {
    CODEPS_REQUIRE(state && ids && feats && capacity > 0 && dim > 0, CODEPS_ERR_ARG, "codeps_replay_export: bad arguments");
    CODEPS_REQUIRE(bytes >= state_bytes(capacity, dim), CODEPS_ERR_WORKSPACE, "codeps_replay_export: state too small");
    Layout l = layout(const_cast<void*>(state), capacity, dim);
    hipStream_t st = (hipStream_t)stream;
    const size_t slots = (size_t)capacity + 1;
    if (hipMemcpyAsync(ids, l.ids, slots * 8, hipMemcpyDeviceToDevice, st) != hipSuccess ||
        hipMemcpyAsync(feats, l.feats, slots * dim * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        codeps_set_error("codeps_replay_export: hipMemcpyAsync failed");
        return CODEPS_ERR_LAUNCH;
    }
    return CODEPS_OK;
}
