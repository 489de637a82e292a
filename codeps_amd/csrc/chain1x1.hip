// Register-chained pair of 1x1 convolutions of the FROZEN ResNet bottleneck stack (inference only):
//
//     out[m][n2] = relu(scale2[n2] * sum_c  W2[n2][c] * y1[m][c]  + shift2[n2] + res[m][n2])        expand   (K = 256 -> N2)
//     y0n[m][c'] = relu(scale3[c'] * sum_n2 W3[c'][n2] * out[m][n2] + shift3[c'])                    next block's reduce (N2 -> 256)
//
// i.e. conv3 + bn3 + residual + ReLU of bottleneck i and conv1 + bn1 + ReLU of bottleneck i + 1 (models/resnet_encoder.py:115-126
// -> torchvision Bottleneck.forward) in ONE launch: in the unfused form the 22 expand / reduce pairs of ResNet-101's layer3
// are 2.8 ms of the adaptation step at 133 / 147 TFLOP/s -- the expand is all prologue / epilogue (8 K-steps per tile), the
// reduce re-reads, re-splits and re-stages the 67 MB tensor the expand has just written.
//
// Same arithmetic as the conv engine (csrc/conv.hip): every fp32 operand is the exact sum of three bf16 pieces, six piece
// products of order <= 2 on v_mfma_f32_16x16x32_bf16, fp32 accumulation.  What is different is where operands live:
//   * both GEMMs are computed TRANSPOSED, D[channel][pixel]: the accumulator of a 16 x 16 tile then holds, per lane, FOUR
//     CONSECUTIVE CHANNELS of one pixel (rows 4 kg + r, column l16) -- which is (a) a 16-byte NHWC store / residual load and
//     (b) after the in-register epilogue and bf16 split, directly the B operand of the next MFMA that contracts over
//     channels (guide: "an accumulator tile as the next MFMA's operand"): `out` never goes through LDS or back from HBM.
//     The k order inside a 32-deep step is permuted (element j of lane group kg = channel 16 (j >> 2) + 4 kg + (j & 3)); the
//     reduce weights are stored in that order (host side: codeps_amd.functional.chain_weights).
//   * the activations of the expand (y1: 16 pixels x 256 channels per wave) are loaded ONCE, straight from HBM in B-fragment
//     layout, split in registers and kept there (96 VGPRs) for all N2 / 32 stages.
//   * LDS holds nothing but weights: a ring of three 52 KB tiles filled by `buffer_load_dwordx4 ... lds` (no VGPR round trip,
//     no ds_write), consumed alternately as E_s (32 expand rows x K 256) and R_s (256 reduce rows x 32 k) -- both are images
//     of 256 rows x 208 bytes (192 data + 16 pad, conflict-free ds_read_b128 fragment reads).  The pad of an E row carries the
//     row's folded BatchNorm (scale, shift): the affine of the expand reaches the epilogue through the same DMA.
//   * every vector-memory operation of the main loop is issued through inline assembly and its completion is counted by hand
//     (vmcnt: loads, stores and LDS DMA retire in issue order), so a tile has a full stage (two MFMA phases) to land.
// One block = 128 pixels x one K slice: eight waves x 16 pixels (two per SIMD), and the N2 / 32 stages are cut into `ksplit`
// slices over blockIdx.y (M = 16384 at config 3: 128 pixel tiles x 2 slices = one block per CU).  The first version ran 64
// pixels x all stages per CU: every CU then streams ALL 3 MB of weights for 64 pixels of work -- 52 KB per 1536 MFMA cycles
// = the 66-73 GB/s a CU takes from L2 (MI355X_MICROARCH.md), measured 133 us, no faster than the separate layers.  With 128
// pixels per tile a weight tile feeds twice the MFMAs (26 KB per 1536 cycles per SIMD pair) and the slices halve what a CU
// streams; the price is the partial sums of the reduce (raw fp32 slabs + a fixed-order finish pass with the affine and ReLU).
#include <algorithm>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int KMID = 256;                 // channels of y1 (K of the expand) = output channels of the reduce
constexpr int ROWB = 208;                 // bytes per LDS row: [piece][32 bf16] + 16 pad
constexpr int TILE_B = 256 * ROWB;        // one ring slot
constexpr int W2ROW = KMID * 6 + 16;      // bytes per expand-weight row in HBM: 8 groups x [3 pieces][32 bf16] + (scale, shift, 0, 0)
constexpr int NWAVE = 8, BPIX = 16 * NWAVE;  // waves and pixels per block
constexpr int NDMA = 13;                  // DMA instructions per LOADER wave and tile (4 waves x 13 x 64 lanes x 16 B = 256 x 208 B)

struct ChainP {
    long long M;                          // pixels (multiple of 128)
    int ksplit;                           // K slices of the reduce = slices of the expand's output channels (gridDim.y)
    float* part;                          // [ksplit][M][256] raw partial sums of the reduce (ksplit > 1)
    int N2;                               // expand output channels (multiple of 32)
    const float* y1;                      // [M][256]
    const float* res;                     // [M][N2]
    const void* w2;                       // [N2][W2ROW]   pre-split expand weights + affine record
    const void* w3;                       // [256][N2 / 32][3][32] bf16, k order permuted per 32-group
    const float* scale3;                  // [256]
    const float* shift3;
    float* out;                           // [M][N2]
    float* y0n;                           // [M][256]
};

// x = p0 + p1 + p2 exactly (round to nearest each time); element e of the three pieces
__device__ __forceinline__ void split3(float x, unsigned short (&p)[3]) {
    float r = x;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const unsigned short h = __builtin_bit_cast(unsigned short, (__bf16)r);
        p[q] = h;
        r -= __builtin_bit_cast(float, (unsigned)h << 16);
    }
}

// eight fp32 values -> three bf16x8 fragments
__device__ __forceinline__ void split8(const f32x4 lo, const f32x4 hi, bf16x8 (&f)[3]) {
    unsigned short h[8][3];
#pragma unroll
    for (int e = 0; e < 4; ++e) { split3(lo[e], h[e]); split3(hi[e], h[4 + e]); }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        i32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (int)((unsigned)h[2 * i][q] | ((unsigned)h[2 * i + 1][q] << 16));
        f[q] = __builtin_bit_cast(bf16x8, v);
    }
}

#define CHAIN_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

CODEPS_STAMP_DECL(chain)

__global__ __launch_bounds__(512) void chain_expand_reduce_kernel(ChainP p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    CODEPS_STAMP_BEGIN();
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0 = (long long)blockIdx.x * BPIX;
    const long long m = m0 + wave * 16 + l16;                  // this lane's pixel
    const int NS = p.N2 / 32 / p.ksplit;                       // stages of this block (even), global stages [sbase, sbase + NS)
    const int sbase = (int)blockIdx.y * NS;

    // ---- buffer resources (wave-uniform)
    auto rsrc_of = [](const void* base, size_t bytes) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(base);
        i32x4 r;
        r[0] = (int)__builtin_amdgcn_readfirstlane((unsigned)a);
        r[1] = (int)(__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xFFFFu);
        r[2] = (int)__builtin_amdgcn_readfirstlane((unsigned)(bytes < 0xFFFF0000ull ? bytes : 0xFFFF0000ull));
        r[3] = 0x00020000;
        return r;
    };
    // the block's 128 rows of res / out: offsets inside the block's slab stay far below 4 GB whatever M is
    const i32x4 rsW2 = rsrc_of(p.w2, (size_t)p.N2 * W2ROW);
    const i32x4 rsW3 = rsrc_of(p.w3, (size_t)KMID * p.N2 * 6);
    const i32x4 rsRes = rsrc_of(p.res + (size_t)m0 * p.N2, (size_t)BPIX * p.N2 * 4);
    const i32x4 rsOut = rsrc_of(p.out + (size_t)m0 * p.N2, (size_t)BPIX * p.N2 * 4);
    const unsigned row_off = (unsigned)((wave * 16 + l16) * p.N2 + 4 * kg) * 4u;      // byte offset of (pixel, channel 4 kg) in those slabs

    // ---- DMA roles.  Only waves 4-7 (the second wave of every SIMD) issue the weight tiles: issuing an LDS-DMA instruction costs
    //      its wave 60-185 cycles (MI355X_MICROARCH.md), and with all eight waves issuing right behind the barrier the matrix
    //      pipes idled for that long twice per stage.  Now a SIMD's first wave starts its MFMAs at once while its partner
    //      issues; the partner's MFMAs (and its epilogue VALU work) then run beside the first wave's next phase -- the two
    //      waves of a SIMD stay half a phase apart.  Instruction i of loader wave w covers linear 16-byte segments
    //      L = (w - 4 + 4 i) * 64 + lane of the tile image; segment L -> (row L / 13, seg L % 13), seg 12 = the pad
    unsigned offE[NDMA], offR[NDMA], ldsdst[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int ins = (wave & 3) + 4 * i, L = ins * 64 + lane, prow = L / 13, seg = L % 13;
        ldsdst[i] = (unsigned)ins * 1024u;
        const int ks = prow >> 5, r = prow & 31;               // E tile: pseudo-row = (k-step, expand row within the stage)
        // fragment reads (lane = (row l16, k group kg)) are serviced in the lane groups {0-3, 12-15, 20-27}, ...: at a 208-byte pitch
        // the kg = 0 and kg = 1 rows of a group share 16-byte slots (8 LDS cycles per read instead of 4) unless the k groups of rows
        // 4 .. 11 (mod 16) are stored pairwise swapped -- the DMA picks the source segment per lane, the reader flips bit 0 of kg
        const int swz = ((prow >> 2) ^ (prow >> 3)) & 1;
        offE[i] = (unsigned)(r * W2ROW + (seg < 12 ? ks * 192 + (seg ^ swz) * 16 : KMID * 6));
        offR[i] = (unsigned)((size_t)prow * p.N2 * 6 + (seg < 12 ? (seg ^ swz) : 0) * 16);
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const bool loader = wave >= 4;                             // wave-uniform (wave is an SGPR)
    auto dma_tile = [&](int t) {                               // local tile t: even = E, odd = R of local stage t / 2; slot t % 3
        if (!loader) return;
        const int s = sbase + (t >> 1);
        const unsigned slot = lds_base + (unsigned)(t % 3) * TILE_B;
        if (t & 1) {
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(s * 192);
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(slot + ldsdst[i]);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(dst), "v"(offR[i]), "s"(rsW3), "s"(soff) : "memory");
            }
        } else {
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(s * 32 * W2ROW);
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(slot + ldsdst[i]);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(dst), "v"(offE[i]), "s"(rsW2), "s"(soff) : "memory");
            }
        }
    };
    auto res_load = [&](int sl, f32x4 (&r)[2]) {               // residual of local stage sl: channels 32 s + 16 nt + 4 kg .. + 3 of this lane's pixel
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((sbase + sl) * 128);
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r[0]) : "v"(row_off), "s"(rsRes), "s"(soff) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:64" : "=v"(r[1]) : "v"(row_off), "s"(rsRes), "s"(soff) : "memory");
    };
    auto out_store = [&](int sl, const f32x4 (&v)[2]) {
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((sbase + sl) * 128);
        asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(v[0]), "v"(row_off), "s"(rsOut), "s"(soff) : "memory");
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:64\n\ts_nop 1" :: "v"(v[1]), "v"(row_off), "s"(rsOut), "s"(soff) : "memory");
    };

    // ---- prologue: three weight tiles in flight, then this wave's 16 x 256 activations straight into B-fragment registers
    // residual registers ping-pong between two NAMED sets (stages are unrolled in pairs).  Their loads are inline assembly: the
    // compiler takes the destination for written at once, so a copy it makes before the data lands copies garbage.  Every
    // residual load is therefore issued BEFORE the weight tile of the same point of the loop -- the counted wait for that tile
    // (operations retire in order) then covers it -- and the set is handed to the compiler ("claimed" by an empty asm behind a
    // sched_barrier) only after that wait.
    f32x4 rA[2], rB[2];
    res_load(0, rA);
    dma_tile(0);
    dma_tile(1);
    if (NS > 1) dma_tile(2);
    bf16x8 yf[8][3];
    {
        const float* yrow = p.y1 + (size_t)m * KMID + kg * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(yrow + ks * 32), hi = *reinterpret_cast<const f32x4*>(yrow + ks * 32 + 4);
            split8(lo, hi, yf[ks]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rA[0]), "+v"(rA[1]) :: "memory");          // everything issued so far has landed
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc3[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc3[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* arow = lds + l16 * ROWB + (kg ^ (((l16 >> 2) ^ (l16 >> 3)) & 1)) * 16;      // fragment address of row l16 of a tile, piece 0 (swizzle: see offE)

    auto stage = [&](const int s, f32x4 (&rcur)[2], f32x4 (&rnext)[2]) {
        const bool last = s + 1 == NS;
        // ================= expand stage s: tile 2 s
        // vector-memory queue of this wave, oldest first, at this point (steady state):
        //   loader waves:  R(s)x2 D(2s)x13 | S(s-1)x2  D(2s+1)x13   -> residual of this stage + tile 2 s landed at <= 15 younger
        //   the others:    R(s)x2          | S(s-1)x2                                                             at <= 2
        if (s > 0) { if (loader) CHAIN_WAIT(15); else CHAIN_WAIT(2); }
        __syncthreads();                                       // every wave's part of tile 2 s is in LDS; reduce s - 1 has been read
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(rcur[0]), "+v"(rcur[1]));       // claim: from here on the compiler may touch the residual registers
        res_load(s + 1, rnext);                                // (one past the last stage: columns of the next slice or the next row, or past the slab -> zeros; never used)
        if (s > 0 && 2 * s + 2 < 2 * NS) dma_tile(2 * s + 2);  // into the slot of tile 2 s - 1
        const char* et = arow + ((2 * s) % 3) * TILE_B;
        CODEPS_STAMP_PHASE_START();
        f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const char* a = et + (ks * 32 + nt * 16) * ROWB;
                bf16x8 wf[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) wf[q] = *reinterpret_cast<const bf16x8*>(a + q * 64);
#pragma unroll
                for (int ord = 2; ord >= 0; --ord)
#pragma unroll
                    for (int qa = 0; qa <= ord; ++qa)
                        acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[qa], yf[ks][ord - qa], acc2[nt], 0, 0, 0);
            }
        CODEPS_STAMP_PHASE_A();                                // A: fragment reads + the 96 MFMAs of the expand stage
        // epilogue of the expand in registers: lane = pixel l16, channels 32 s + 16 nt + 4 kg + r; the affine record sits in the
        // pad of row (k-step 0, channel) of the E tile
        f32x4 v[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* aff = reinterpret_cast<const float*>(lds + ((2 * s) % 3) * TILE_B + (nt * 16 + 4 * kg + r) * ROWB + 192);
                v[nt][r] = acc2[nt][r] * aff[0] + aff[1];
            }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = v[nt][r] + rcur[nt][r];
                v[nt][r] = t > 0.f ? t : 0.f;
            }
        out_store(s, v);
        bf16x8 of[3];                                          // B fragment of reduce k-step s: element j = channel 16 (j >> 2) + 4 kg + (j & 3)
        split8(v[0], v[1], of);
        // ================= reduce k-step s: tile 2 s + 1
        //   loader waves: D(2s+1)x13 | R(s+1)x2 D(2s+2)x13 S(s)x2   (last stage: no D(2s+2): wait for everything)
        if (loader) { if (last) CHAIN_WAIT(0); else CHAIN_WAIT(17); }
        __syncthreads();                                       // tile 2 s + 1 is in LDS; every wave has read tile 2 s
        if (2 * s + 3 < 2 * NS) dma_tile(2 * s + 3);           // into the slot of tile 2 s
        const char* rt = arow + ((2 * s + 1) % 3) * TILE_B;
        CODEPS_STAMP_PHASE_START();
#pragma unroll
        for (int jt = 0; jt < 16; ++jt) {
            const char* a = rt + jt * 16 * ROWB;
            bf16x8 wf[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) wf[q] = *reinterpret_cast<const bf16x8*>(a + q * 64);
#pragma unroll
            for (int ord = 2; ord >= 0; --ord)
#pragma unroll
                for (int qa = 0; qa <= ord; ++qa)
                    acc3[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[qa], of[ord - qa], acc3[jt], 0, 0, 0);
        }
        CODEPS_STAMP_PHASE_B();                                // B: fragment reads + the 96 MFMAs of the reduce k-step
    };
    for (int s = 0; s < NS; s += 2) {                          // NS is even (n2 % 64 == 0)
        stage(s, rA, rB);
        stage(s + 1, rB, rA);
    }
    CHAIN_WAIT(0);

    // ---- epilogue of the reduce: y0n[m][16 jt + 4 kg + r] = relu(acc * scale3 + shift3), 16-byte stores; with K slices the raw
    //      partial sums go to part[slice][m][256] and chain_finish_kernel adds them in slice order
    if (p.ksplit > 1) {
        float* po = p.part + ((size_t)blockIdx.y * p.M + m) * KMID + 4 * kg;
#pragma unroll
        for (int jt = 0; jt < 16; ++jt) *reinterpret_cast<f32x4*>(po + jt * 16) = acc3[jt];
        CODEPS_STAMP_END(chain);
        return;
    }
    float* yo = p.y0n + (size_t)m * KMID + 4 * kg;
#pragma unroll
    for (int jt = 0; jt < 16; ++jt) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale3 + jt * 16 + 4 * kg), sh = *reinterpret_cast<const f32x4*>(p.shift3 + jt * 16 + 4 * kg);
        f32x4 o = acc3[jt] * sc + sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = o[r] > 0.f ? o[r] : 0.f;
        *reinterpret_cast<f32x4*>(yo + jt * 16) = o;
    }
    CODEPS_STAMP_END(chain);
}

// ---------------------------------------------------------------------------------- the same kernel on two fp16 pieces (round 6)
// Arithmetic 16 of the conv engine (csrc/conv.hip, split_f16): x 2^k = hi + lo in fp16, three products hi hi + hi lo + lo hi on
// v_mfma_f32_16x16x32_f16 -- 48 instead of 96 matrix instructions per phase, weight tiles of 256 x 144 bytes (128 data + 16 pad; ring 110 KB).
// Scales: the weights' per-tensor 2^kb comes from the pack and its inverse is folded into the affines on the host (scale2, scale3);
// the activations are scaled PER PIXEL: in the transposed GEMMs a pixel is a column of D, all accumulators of a lane belong to ONE pixel,
// and the four lanes that share it agree on its largest exponent with two lane swaps (v_permlane16_swap / v_permlane32_swap) -- y1 once in
// the prologue, the expand's output per stage (running maximum; when it grows the lane's 64 reduce sums are multiplied by the exact ratio).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int ROWH = 144;                 // bytes per LDS row: [piece][32 fp16] + 16 pad
constexpr int W2ROWH = KMID * 4 + 16;     // bytes per expand-weight row in HBM at mid width 256: 8 groups x [2 pieces][32 fp16] + (scale 2^-kb, shift, 0, 0)
constexpr int H_TARGET = 14, H_EMIN = 27; // |x 2^k| < 2^15; magnitudes below 2^-100 count as 2^-100 (conv.hip: F16_TARGET, F16_EMIN)

__device__ __forceinline__ float h_scale_of(int e) { return __builtin_bit_cast(float, (unsigned)(254 + H_TARGET - e) << 23); }
__device__ __forceinline__ float h_inv_scale_of(int e) { return __builtin_bit_cast(float, (unsigned)(e - H_TARGET) << 23); }
// biased exponent of the largest of the values m (>= 0) held by the four lanes l16 + 16 kg of one pixel, floored at H_EMIN
__device__ __forceinline__ int pixel_exponent(float m) {
    unsigned u = __builtin_bit_cast(unsigned, m);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    u = max(a[0], a[1]);                                       // (non-negative floats order like their bit patterns)
    auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    u = max(b[0], b[1]);
    return max((int)(u >> 23), H_EMIN);
}
// eight fp32 values x s -> (hi, lo) fp16x8 fragments
__device__ __forceinline__ void split8h(const f32x4 lo4, const f32x4 hi4, const float s, f16x8 (&f)[2]) {
    _Float16 h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? lo4[e] : hi4[e - 4];
        h[e] = (_Float16)(x * s);
        l[e] = (_Float16)__builtin_fmaf(x, s, -(float)h[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { f[0][e] = h[e]; f[1][e] = l[e]; }
}

// KM = the bottleneck's mid width: 256 (layer3), 128 (layer2), 64 (layer1) -- the tile image is KM rows x 144 bytes for both GEMMs (expand: KM / 32
// k-steps x 32 rows; reduce: KM output rows x 32 k), a pixel's y1 row is KM / 32 fragment pairs, the reduce keeps KM / 16 accumulator tiles.
template <int KM>
__global__ __launch_bounds__(512, KM == 256 ? 2 : 4) void chain_expand_reduce_h_kernel(ChainP p) {
    constexpr int KS = KM / 32, JT = KM / 16;
    constexpr int TILE = KM * ROWH, W2R = KM * 4 + 16;
    constexpr int BINS = KM * 9 / 64, ND = (BINS + 3) / 4;       // DMA instructions per tile / per loader wave (waves past the last one repeat their previous instruction)
    static_assert((KM * 9) % 64 == 0 && KM % 32 == 0, "tile image: whole 1 KiB DMA pieces");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    CODEPS_STAMP_BEGIN();
    const int tid = threadIdx.x, lane = tid & 63, l16 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0 = (long long)blockIdx.x * BPIX;
    const long long m = m0 + wave * 16 + l16;                  // this lane's pixel
    const int NS = p.N2 / 32 / p.ksplit;                       // stages of this block (even), global stages [sbase, sbase + NS)
    const int sbase = (int)blockIdx.y * NS;
    auto rsrc_of = [](const void* base, size_t bytes) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(base);
        i32x4 r;
        r[0] = (int)__builtin_amdgcn_readfirstlane((unsigned)a);
        r[1] = (int)(__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xFFFFu);
        r[2] = (int)__builtin_amdgcn_readfirstlane((unsigned)(bytes < 0xFFFF0000ull ? bytes : 0xFFFF0000ull));
        r[3] = 0x00020000;
        return r;
    };
    const i32x4 rsW2 = rsrc_of(p.w2, (size_t)p.N2 * W2R);
    const i32x4 rsW3 = rsrc_of(p.w3, (size_t)KM * p.N2 * 4);
    const i32x4 rsRes = rsrc_of(p.res + (size_t)m0 * p.N2, (size_t)BPIX * p.N2 * 4);
    const i32x4 rsOut = rsrc_of(p.out + (size_t)m0 * p.N2, (size_t)BPIX * p.N2 * 4);
    const unsigned row_off = (unsigned)((wave * 16 + l16) * p.N2 + 4 * kg) * 4u;

    // DMA roles as in chain_expand_reduce_kernel; a tile row is 9 segments of 16 bytes (8 data + pad).  At mid width 256 the per-lane source
    // offsets live in registers; the narrower forms (two blocks per CU: 128 registers) recompute them per tile -- a dozen integer operations
    // beside a tile's worth of matrix work.
    constexpr bool KEEP = KM == 256;
    auto off_of = [&](const int i, const int ln, const bool reduce) {
        const int ins = (wave & 3) + 4 * i < BINS ? (wave & 3) + 4 * i : (wave & 3) + 4 * (i - 1), L = ins * 64 + ln, prow = L / 9, seg = L % 9;
        const int ks = prow >> 5, r = prow & 31;
        const int swz = ((prow >> 2) ^ (prow >> 3)) & 1;       // odd slot pitch: k groups of rows 4 .. 11 (mod 16) swapped pairwise
        return reduce ? (unsigned)((size_t)prow * p.N2 * 4 + (seg < 8 ? (seg ^ swz) : 0) * 16)
                      : (unsigned)(r * W2R + (seg < 8 ? ks * 128 + (seg ^ swz) * 16 : KM * 4));
    };
    auto dst_of = [&](const int i) { return (unsigned)((wave & 3) + 4 * i < BINS ? (wave & 3) + 4 * i : (wave & 3) + 4 * (i - 1)) * 1024u; };
    unsigned offE[KEEP ? ND : 1], offR[KEEP ? ND : 1];
    if constexpr (KEEP) {
#pragma unroll
        for (int i = 0; i < ND; ++i) { offE[i] = off_of(i, lane, false); offR[i] = off_of(i, lane, true); }
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const bool loader = wave >= 4;
    auto dma_tile = [&](int t) {
        if (!loader) return;
        const int s = sbase + (t >> 1);
        const unsigned slot = lds_base + (unsigned)(t % 3) * TILE;
        int ln = lane;
        if constexpr (!KEEP) asm volatile("" : "+v"(ln));      // (keeps the offsets out of the loop-invariant set)
        if (t & 1) {
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(s * 128);
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(slot + dst_of(i));
                const unsigned off = KEEP ? offR[KEEP ? i : 0] : off_of(i, ln, true);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(dst), "v"(off), "s"(rsW3), "s"(soff) : "memory");
            }
        } else {
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(s * 32 * W2R);
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(slot + dst_of(i));
                const unsigned off = KEEP ? offE[KEEP ? i : 0] : off_of(i, ln, false);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(dst), "v"(off), "s"(rsW2), "s"(soff) : "memory");
            }
        }
    };
    auto res_load = [&](int sl, f32x4 (&r)[2]) {
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((sbase + sl) * 128);
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r[0]) : "v"(row_off), "s"(rsRes), "s"(soff) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:64" : "=v"(r[1]) : "v"(row_off), "s"(rsRes), "s"(soff) : "memory");
    };
    auto out_store = [&](int sl, const f32x4 (&v)[2]) {
        const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((sbase + sl) * 128);
        asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(v[0]), "v"(row_off), "s"(rsOut), "s"(soff) : "memory");
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:64\n\ts_nop 1" :: "v"(v[1]), "v"(row_off), "s"(rsOut), "s"(soff) : "memory");
    };

    f32x4 rA[2], rB[2];
    res_load(0, rA);
    dma_tile(0);
    dma_tile(1);
    if (NS > 1) dma_tile(2);
    f16x8 yf[KS][2];
    float inv_y;
    {
        const float* yrow = p.y1 + (size_t)m * KM + kg * 8;
        f32x4 ylo[KS], yhi[KS];
        float ym = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            ylo[ks] = *reinterpret_cast<const f32x4*>(yrow + ks * 32); yhi[ks] = *reinterpret_cast<const f32x4*>(yrow + ks * 32 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) ym = fmaxf(ym, fmaxf(fabsf(ylo[ks][e]), fabsf(yhi[ks][e])));
        }
        const int ey = pixel_exponent(ym);                     // the pixel's 256 channels lie in its four lanes
        const float sy = h_scale_of(ey);
        inv_y = h_inv_scale_of(ey);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) split8h(ylo[ks], yhi[ks], sy, yf[ks]);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rA[0]), "+v"(rA[1]) :: "memory");
    __builtin_amdgcn_sched_barrier(0);

    f32x4 acc3[JT];
#pragma unroll
    for (int j = 0; j < JT; ++j) acc3[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int e_run = H_EMIN;                                        // largest exponent of this pixel's expand outputs so far

    const char* arow = lds + l16 * ROWH + (kg ^ (((l16 >> 2) ^ (l16 >> 3)) & 1)) * 16;

    auto stage = [&](const int s, f32x4 (&rcur)[2], f32x4 (&rnext)[2]) {
        const bool last = s + 1 == NS;
        // vector-memory queue as in chain_expand_reduce_kernel with 9 DMA instructions per tile: 9 + 2 younger operations here ...
        if (s > 0) { if (loader) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ND + 2) : "memory"); else CHAIN_WAIT(2); }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(rcur[0]), "+v"(rcur[1]));
        res_load(s + 1, rnext);
        if (s > 0 && 2 * s + 2 < 2 * NS) dma_tile(2 * s + 2);
        const char* et = arow + ((2 * s) % 3) * TILE;
        CODEPS_STAMP_PHASE_START();
        f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const char* a = et + (ks * 32 + nt * 16) * ROWH;
                f16x8 wf[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) wf[q] = *reinterpret_cast<const f16x8*>(a + q * 64);
                acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], yf[ks][1], acc2[nt], 0, 0, 0);
                acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1], yf[ks][0], acc2[nt], 0, 0, 0);
                acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], yf[ks][0], acc2[nt], 0, 0, 0);
            }
        CODEPS_STAMP_PHASE_A();
        f32x4 v[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* aff = reinterpret_cast<const float*>(lds + ((2 * s) % 3) * TILE + (nt * 16 + 4 * kg + r) * ROWH + 128);
                v[nt][r] = acc2[nt][r] * (aff[0] * inv_y) + aff[1];          // aff[0] = scale2 2^-kb (host), inv_y = the pixel's 2^-ky
            }
        float vm = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = v[nt][r] + rcur[nt][r];
                v[nt][r] = t > 0.f ? t : 0.f;
                vm = fmaxf(vm, v[nt][r]);
            }
        out_store(s, v);
        // the pixel's running scale: when this stage's outputs outgrow it, the lane's reduce sums move to the new scale (exact)
        const int e_new = max(e_run, pixel_exponent(vm));
        if (__builtin_amdgcn_ballot_w64(e_new != e_run)) {
            const int d = e_run - e_new;
            const float rb = d < -126 ? 0.f : __builtin_bit_cast(float, (unsigned)(127 + d) << 23);
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) acc3[jt] *= rb;
            e_run = e_new;
        }
        f16x8 of[2];                                           // B fragment of reduce k-step s: element j = channel 16 (j >> 2) + 4 kg + (j & 3)
        split8h(v[0], v[1], h_scale_of(e_run), of);
        // ... and 2 + 9 + 2 here
        if (loader) { if (last) CHAIN_WAIT(0); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(ND + 4) : "memory"); }
        __syncthreads();
        if (2 * s + 3 < 2 * NS) dma_tile(2 * s + 3);
        const char* rt = arow + ((2 * s + 1) % 3) * TILE;
        CODEPS_STAMP_PHASE_START();
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const char* a = rt + jt * 16 * ROWH;
            f16x8 wf[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) wf[q] = *reinterpret_cast<const f16x8*>(a + q * 64);
            acc3[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], of[1], acc3[jt], 0, 0, 0);
            acc3[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1], of[0], acc3[jt], 0, 0, 0);
            acc3[jt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], of[0], acc3[jt], 0, 0, 0);
        }
        CODEPS_STAMP_PHASE_B();
    };
    for (int s = 0; s < NS; s += 2) {
        stage(s, rA, rB);
        stage(s + 1, rB, rA);
    }
    CHAIN_WAIT(0);

    const float inv_o = h_inv_scale_of(e_run);                 // (scale3 carries the reduce weights' 2^-kb: host)
    if (p.ksplit > 1) {
        float* po = p.part + ((size_t)blockIdx.y * p.M + m) * KM + 4 * kg;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) *reinterpret_cast<f32x4*>(po + jt * 16) = acc3[jt] * inv_o;
        CODEPS_STAMP_END(chain);
        return;
    }
    float* yo = p.y0n + (size_t)m * KM + 4 * kg;
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale3 + jt * 16 + 4 * kg), sh = *reinterpret_cast<const f32x4*>(p.shift3 + jt * 16 + 4 * kg);
        f32x4 o = (acc3[jt] * inv_o) * sc + sh;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = o[r] > 0.f ? o[r] : 0.f;
        *reinterpret_cast<f32x4*>(yo + jt * 16) = o;
    }
    CODEPS_STAMP_END(chain);
}

// y0n = relu((sum_z part[z]) * scale3 + shift3), slices added in fixed order
__global__ __launch_bounds__(256) void chain_finish_kernel(const float* __restrict__ part, int ksplit, long long MN, int kmid, const float* __restrict__ scale3,
                                                           const float* __restrict__ shift3, float* __restrict__ y0n) {
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < MN; i += (long long)gridDim.x * 1024) {
        f32x4 v = *reinterpret_cast<const f32x4*>(part + i);
        for (int z = 1; z < ksplit; ++z) v += *reinterpret_cast<const f32x4*>(part + (size_t)z * MN + i);
        const int c = (int)(i % kmid);
        v = v * *reinterpret_cast<const f32x4*>(scale3 + c) + *reinterpret_cast<const f32x4*>(shift3 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        *reinterpret_cast<f32x4*>(y0n + i) = v;
    }
}

// K slices: as many as fill the 256 CUs with one block each, while every slice keeps an even number (>= 4) of stages
static int chain_slices(long long rows, int n2) {
    const long long tiles = rows / BPIX;
    int ks = 1;
    while (ks < 8 && tiles * ks * 2 <= 256 && (n2 / 32) % (ks * 4) == 0 && (n2 / 32) / (ks * 2) >= 4) ks *= 2;
    return ks;
}

}  // namespace

CODEPS_STAMP_SETTER(chain)


extern "C" size_t codeps_chain1x1_w2_row_bytes(void) { return W2ROW; }

extern "C" size_t codeps_chain1x1_workspace_bytes(long long rows, int n2) {
    if (rows <= 0 || rows % BPIX || n2 < 64 || n2 % 64) return 0;
    const int ks = chain_slices(rows, n2);
    return ks > 1 ? (size_t)ks * rows * KMID * sizeof(float) : 0;
}

static int chain_launch(bool h16, int kmid, const float* y1, const float* residual, const void* w2_rows, const void* w3_split,
                        const float* scale3, const float* shift3, long long rows, int n2, float* out, float* y0_next,
                        void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    CODEPS_REQUIRE(y1 && residual && w2_rows && w3_split && scale3 && shift3 && out && y0_next, CODEPS_ERR_ARG, "chain1x1: null pointer argument");
    CODEPS_REQUIRE(rows > 0 && rows % BPIX == 0 && rows / BPIX < (1ll << 31), CODEPS_ERR_ARG, "chain1x1: rows (%lld) must be a positive multiple of %d", rows, BPIX);
    CODEPS_REQUIRE(n2 >= 64 && n2 % 64 == 0 && n2 <= 4096, CODEPS_ERR_ARG, "chain1x1: expand channels (%d) must be a multiple of 64 in [64, 4096]", n2);
    CODEPS_REQUIRE(kmid == 256 || (h16 && (kmid == 128 || kmid == 64)), CODEPS_ERR_ARG, "chain1x1: mid width %d (256; 128 and 64 on the fp16-pair form only)", kmid);
    ChainP p;
    p.M = rows; p.N2 = n2; p.y1 = y1; p.res = residual; p.w2 = w2_rows; p.w3 = w3_split; p.scale3 = scale3; p.shift3 = shift3;
    p.out = out; p.y0n = y0_next;
    p.ksplit = chain_slices(rows, n2);
    p.part = nullptr;
    if (p.ksplit > 1) {
        CODEPS_REQUIRE(workspace && workspace_bytes >= (size_t)p.ksplit * rows * kmid * sizeof(float), CODEPS_ERR_WORKSPACE,
                       "chain1x1: workspace too small (codeps_chain1x1_workspace_bytes)");
        p.part = static_cast<float*>(workspace);
    }
    hipStream_t st = (hipStream_t)stream;
    if (h16) {
        auto kh = kmid == 256 ? chain_expand_reduce_h_kernel<256> : (kmid == 128 ? chain_expand_reduce_h_kernel<128> : chain_expand_reduce_h_kernel<64>);
        codeps_allow_lds((const void*)kh, 160 * 1024, "chain_expand_reduce_h_kernel");
        hipLaunchKernelGGL(kh, dim3((unsigned)(rows / BPIX), p.ksplit), dim3(64 * NWAVE), (size_t)3 * kmid * ROWH, st, p);
    } else {
        codeps_allow_lds((const void*)chain_expand_reduce_kernel, 160 * 1024, "chain_expand_reduce_kernel");
        hipLaunchKernelGGL(chain_expand_reduce_kernel, dim3((unsigned)(rows / BPIX), p.ksplit), dim3(64 * NWAVE), 3 * TILE_B, st, p);
    }
    if (p.ksplit > 1) {
        const long long MN = rows * kmid;
        hipLaunchKernelGGL(chain_finish_kernel, dim3((int)std::min<long long>((MN / 4 + 255) / 256, 2048)), dim3(256), 0, st, (const float*)p.part, p.ksplit, MN, kmid,
                           scale3, shift3, y0_next);
    }
    CODEPS_CHECK_LAUNCH("chain_expand_reduce_kernel");
    return CODEPS_OK;
}

extern "C" int codeps_chain1x1_expand_reduce(const float* y1, const float* residual, const void* w2_rows, const void* w3_split,
                                             const float* scale3, const float* shift3, long long rows, int n2, float* out, float* y0_next,
                                             void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    return chain_launch(false, KMID, y1, residual, w2_rows, w3_split, scale3, shift3, rows, n2, out, y0_next, workspace, workspace_bytes, stream);
}

extern "C" size_t codeps_chain1x1_w2_row_bytes_h(void) { return W2ROWH; }

extern "C" int codeps_chain1x1_expand_reduce_h(const float* y1, const float* residual, const void* w2_rows, const void* w3_split,
                                               const float* scale3, const float* shift3, long long rows, int n2, float* out, float* y0_next,
                                               void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    return chain_launch(true, KMID, y1, residual, w2_rows, w3_split, scale3, shift3, rows, n2, out, y0_next, workspace, workspace_bytes, stream);
}

// ... at the mid widths of the other stages (fp16-pair form only): kmid = 256 (layer3), 128 (layer2), 64 (layer1).  y1 [rows][kmid], w2 rows of
// kmid * 4 + 16 bytes, w3 [kmid][n2 / 32][2][32] fp16, scale3 / shift3 [kmid], y0_next [rows][kmid]; workspace: codeps_chain1x1_workspace_bytes_k.
extern "C" size_t codeps_chain1x1_workspace_bytes_k(long long rows, int n2, int kmid) {
    if (rows <= 0 || rows % BPIX || n2 < 64 || n2 % 64 || (kmid != 256 && kmid != 128 && kmid != 64)) return 0;
    const int ks = chain_slices(rows, n2);
    return ks > 1 ? (size_t)ks * rows * kmid * sizeof(float) : 0;
}
extern "C" int codeps_chain1x1_expand_reduce_hk(const float* y1, const float* residual, const void* w2_rows, const void* w3_split,
                                                const float* scale3, const float* shift3, long long rows, int n2, int kmid, float* out, float* y0_next,
                                                void* workspace, size_t workspace_bytes, codeps_stream_t stream) {
    return chain_launch(true, kmid, y1, residual, w2_rows, w3_split, scale3, shift3, rows, n2, out, y0_next, workspace, workspace_bytes, stream);
}
