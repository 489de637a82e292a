/* libcodeps_diag.so -- measurement tools and measured-and-rejected prototypes, kept OUT of the product library
 * (libcodeps_hip.so, include/codeps_hip.h): nothing on the hot path calls them.  Built by the same Makefile from
 * csrc/diag.hip (micro-benchmarks behind tools/peaks.py, tools/feed_rate.py and bench.py's `roofline.measured_peak`) and
 * csrc/conv1x1s.hip (the pre-split-activation 1x1 prototype of round 2, tools/conv1x1s_bench.py).  Same conventions as the
 * product ABI: extern "C", plain pointers, a void* HIP stream, 0 on success, codeps_diag_last_error() for the message. */
#ifndef CODEPS_DIAG_H
#define CODEPS_DIAG_H
#include "codeps_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

int codeps_diag_version(void);
const char* codeps_diag_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Diagnostics (csrc/diag.hip): the two rooflines of SURVEY.md 8d as THIS box delivers them -- bench.py reports them beside the
 * vendor figures.  Both time `reps` launches with HIP events on `stream` and return milliseconds per launch.
 * codeps_diag_mfma_rate: register-resident MFMA loop, no memory traffic (mode 0 v_mfma_f32_32x32x16_bf16, 1 v_mfma_f32_32x32x2_f32,
 *   2 v_mfma_f32_16x16x32_bf16; random_data 0: zero operands, 1: operand fragments that change between consecutive MFMAs);
 *   blocks of 256 threads, 16 * iters MFMAs per wave; scratch float [blocks * 256]; *flops_per_launch = the FLOPs of one launch.
 * codeps_diag_stream_triad: a[i] = b[i] + 1.5 c[i], n floats (n % 4 == 0): 12 n bytes of HBM traffic per launch. */
int codeps_diag_mfma_rate(int mode, int random_data, int blocks, int iters, int reps, float* scratch, double* flops_per_launch,
                          float* ms_per_launch, codeps_stream_t stream);
int codeps_diag_stream_triad(float* a, const float* b, const float* c, long long n, int reps, float* ms_per_launch,
                             codeps_stream_t stream);
/* codeps_diag_stream_copy: dst[i] = src[i], n floats (n % 4 == 0), 16-byte nontemporal vectors: 8 n bytes of HBM traffic per launch -- the form the
 *   hardware guide quotes its measured HBM roof on (6.29 TB/s); the triad (two read streams + one write stream) reaches 4.4-4.9 TB/s on these boxes. */
int codeps_diag_stream_copy(float* dst, const float* src, long long n, int reps, float* ms_per_launch, codeps_stream_t stream);
/* codeps_diag_mfma_valu: vector-ALU work next to a matrix-core stream on one SIMD.  Waves 0-3 of every block issue 16 * iters
 *   v_mfma_f32_32x32x16_bf16 on changing register operands; `valu_per_mfma` (0, 2, 4, 8 or 16) v_fma_f32 per MFMA are issued by
 *   the same wave between the MFMAs (other_wave 0) or by a second wave of the same SIMD that does nothing else (other_wave 1:
 *   512-thread blocks).  scratch: blocks * 512 floats.  Returns the time per launch. */
int codeps_diag_mfma_valu(int valu_per_mfma, int other_wave, int blocks, int iters, int reps, float* scratch, float* ms_per_launch,
                          codeps_stream_t stream);
/* The same two loops with IN-KERNEL CLOCK STAMPS (MI355X_MICROARCH.md "DVFS give-back" item 6): every MFMA wave stamps s_memtime (shader
 *   cycles) and s_memrealtime (100 MHz) before and after its loop into stamps[wave][4] = t0, r0, t1, r1 (uint64; blocks * 4 waves; a
 *   buffer nothing else reads).  cycles per MFMA = (t1 - t0) / (16 * iters); clock held = (t1 - t0) / (r1 - r0) x 100 MHz.  No assumed
 *   frequency enters.  Run >= 2 s of back-to-back launches on random data before reading the stamps of the last one (tools/clock_cycles.py). */
int codeps_diag_mfma_rate_clock(int mode, int random_data, int blocks, int iters, int reps, float* scratch, unsigned long long* stamps,
                                float* ms_per_launch, codeps_stream_t stream);
int codeps_diag_mfma_valu_clock(int valu_per_mfma, int other_wave, int blocks, int iters, int reps, float* scratch,
                                unsigned long long* stamps, float* ms_per_launch, codeps_stream_t stream);
/* codeps_diag_mfma_once: ONE matrix instruction on caller-supplied operands (mode 0 v_mfma_f32_16x16x32_f16, 1 v_mfma_f32_16x16x32_bf16):
 *   a [16][32], b [16][32] raw 16-bit patterns (k contiguous), c / d [16][16] fp32, d[m][n] = c[m][n] + sum_k a[m][k] b[n][k].  Answers whether
 *   subnormal inputs are honoured and how exactly the 32 products are formed and summed (tools/f16_mfma_probe.py). */
int codeps_diag_mfma_once(int mode, const void* a, const void* b, const float* c, float* d, codeps_stream_t stream);
/* Pre-split activations and their first consumer (csrc/conv1x1s.hip; DESIGN.md 8: the measurement behind "activations kept in the
 * matrix cores' operand form between the frozen encoder's layers").
 * codeps_split_pack_rows: x fp32 [rows][row_stride] (first `channels` of every row, channels % 32 == 0) ->
 *      out [rows][channels / 32][piece 0..2][32] bf16 (rows * channels * 6 bytes), piece sums reproduce x exactly: the layout of
 *      the pre-split weight pack behind codeps_conv2d_pack_weights' fp32 part (1x1: row = output channel).
 * codeps_conv1x1_split_fwd: y[rows][out_c] = act(scale * (A . W^T) + shift + residual); a_split / w_split in that layout
 *      (K = in_c, in_c % 32 == 0, >= 64; out_c % 64 == 0); scale / shift / residual may be NULL; act 0 none, 1 ReLU, 2 ELU.
 *      Both operands reach LDS by DMA; same six-product arithmetic as codeps_conv2d_fwd. */
int codeps_split_pack_rows(const float* x, long long rows, int channels, int row_stride, void* out, codeps_stream_t stream);
int codeps_conv1x1_split_fwd(const void* a_split, long long rows, int in_c, int out_c, const void* w_split, const float* scale,
                             const float* shift, const float* residual, int act, float* y, codeps_stream_t stream);

/* codeps_diag_feed_rate: how many bytes per clock a CU takes in next to a full-rate MFMA stream.  One block of four waves per
 *      CU; per iteration each wave issues 96 v_mfma_f32_16x16x32_bf16 on register operands and the block brings `kb` KiB
 *      (multiple of 4, <= 64) from `window` (device memory, 1 MiB <= window_bytes < 2 GiB; keep it L2-sized) into LDS.
 *      mode 0: no loads; 1: buffer loads into VGPRs + ds_write_b128 (how the conv kernels stage); 2: buffer_load ... lds
 *      (LDS DMA); + 4: every wave also reads 24 ds_read_b128 fragments per iteration from the tile (one per four MFMAs, the
 *      conv kernels' fragment traffic; kb >= 8).  scratch: blocks * 256 floats.  Returns the time per launch. */
int codeps_diag_feed_rate(int mode, int kb, int blocks, int iters, int reps, const void* window, long long window_bytes,
                          float* scratch, float* ms_per_launch, codeps_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CODEPS_DIAG_H */
