/* libcodeps_hip.so -- C ABI of the MI355X-native CoDEPS dense-vision hot path.
 *
 * Plain C types only (no torch, no HIP types): pointers are DEVICE pointers unless stated, sizes
 * are element counts unless suffixed _bytes, `stream` is a hipStream_t passed as void*.
 * Every entry point only ENQUEUES work on `stream` (no device sync, no allocation); the caller owns
 * every buffer including the workspace.  Return 0 = OK, negative = error (see codeps_last_error()).
 * Thread-safe and re-entrant; the error string is thread-local.
 *
 * Each family cites the reference interface (file:line, relative to the CoDEPS repo root) it
 * replaces; INTEGRATION.md shows the ctypes binding the reference side would add.
 */
#ifndef CODEPS_HIP_H
#define CODEPS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CODEPS_OK 0
#define CODEPS_ERR_ARG (-1)      /* bad shape / null pointer / unsupported configuration */
#define CODEPS_ERR_WORKSPACE (-2)/* workspace too small */
#define CODEPS_ERR_LAUNCH (-3)   /* hipGetLastError() after a launch */

typedef void* codeps_stream_t;

int codeps_version(void);                 /* ABI version (10), bumped on any signature / buffer-contract change */
const char* codeps_last_error(void);      /* thread-local, valid until the next failing call */

/* ------------------------------------------------------------------------------------------------
 * Photometric reconstruction loss (K7-K10), replaces ReconstructionLoss.__call__
 * (algos/depth.py:239-326) incl. Interpolate (depth.py:158-173), ImageWarper/CoordinateWarper
 * (misc/image_warper.py:100-184), CameraModel scaling/ray/projection (misc/camera_model.py:37-71),
 * SSIMLoss (depth.py:110-155), _compute_loss (depth.py:221-237) and the auto-mask min.
 *   img_*   [B,3,H,W] fp32 NCHW planar (t, t-1, t+1);   depth [B,1,H,W];
 *   pose_*  [B,16] row-major 4x4 (t->t-1, t->t+1);      intr [B,4] = fx,fy,cx,cy at (H,W).
 *   noise   optional [sum_s B*2*Hs*Ws] tie-break noise ALREADY scaled by 1e-5 (scale-major, then
 *           b, k, y, x); NULL => in-kernel counter-based N(0,1)*1e-5 from `noise_seed`, mixed with
 *           *noise_counter when that optional DEVICE counter is given (codeps_step_counter_add bumps it once
 *           per step ON THE STREAM, so a replayed HIP graph draws new noise every step; the reference
 *           draws fresh randn per call, algos/depth.py:316-318).
 *   loss    [1].   Scale s has size (H >> s, W >> s) (floor division), s < num_scales <= 8.
 * The workspace written by _fwd (image/depth pyramids + per-pixel selected-branch map) must be
 * handed unchanged to _bwd.
 */
typedef struct {
    int32_t batch, height, width, num_scales;
    float alpha;                 /* SSIM weight, reference default 0.85 */
    uint64_t noise_seed;
} codeps_photometric_desc;

size_t codeps_photometric_workspace_bytes(const codeps_photometric_desc* d);
int codeps_photometric_fwd(const codeps_photometric_desc* d, const float* img_t, const float* img_m,
                           const float* img_p, const float* depth, const float* pose_m,
                           const float* pose_p, const float* intr, const float* noise,
                           const uint64_t* noise_counter, float* loss,
                           void* workspace, size_t workspace_bytes, codeps_stream_t stream);
/* grad_loss [1] (device scalar: d total / d loss); grad_depth [B,1,H,W]; grad_pose_* [B,16]. */
int codeps_photometric_bwd(const codeps_photometric_desc* d, const float* img_t, const float* img_m,
                           const float* img_p, const float* depth, const float* pose_m,
                           const float* pose_p, const float* intr, const float* grad_loss,
                           float* grad_depth, float* grad_pose_m, float* grad_pose_p,
                           void* workspace, size_t workspace_bytes, codeps_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * conv2d: NHWC fp32 implicit-GEMM convolution on the f32 MFMA.  Replaces the ATen/cuDNN convs behind
 * torchvision ResNet (models/resnet_encoder.py:115-126), the Depth/Semantic decoders
 * (models/depth_head.py:60-71, models/semantic_head.py:50-72), InstanceHead 1x1 convs
 * (models/instance_head.py:24-57,99-111) and PoseHead (models/pose_head.py:36-44).
 * Activations are NHWC with channel counts padded to multiples of 16 (padding channels hold zeros);
 * nn.Parameter weights stay OIHW and are repacked by codeps_conv2d_pack_weights.
 *   input  = concat(source0 [B, in_h(/2), in_w(/2), c0] (nearest-upsampled x2 on load if up0),
 *                   source1 [B, in_h, in_w, c1] (c1 may be 0))
 *   y[B,out_h,out_w,out_c] = act( conv(input, W) * scale[n] + shift[n] + residual )
 *   scale/shift/residual nullable (bias-only conv: scale NULL, shift = bias; folded eval BN: both).
 */
typedef struct {
    int32_t batch, in_h, in_w;   /* logical input size (after the optional x2 upsample of source 0) */
    int32_t c0, c1;              /* channels of source 0 / source 1 (multiples of 16) */
    int32_t up0;                 /* 1: source 0 is stored at half resolution, nearest x2 on load */
    int32_t out_c, out_h, out_w;
    int32_t kh, kw, stride, pad; /* vertical stride (1 or 2) / padding */
    int32_t pad_mode;            /* 0 zeros, 1 reflect (ReflectionPad2d) */
    int32_t act;                 /* 0 none, 1 ReLU, 2 ELU(alpha=1) */
    int32_t stride_w, pad_w;     /* horizontal stride / padding (= stride / pad for every nn.Conv2d of the reference; they
                                  * differ only for the double-pixel stem form, see codeps_nchw_to_nhwc_xpad) */
    int32_t pix_pitch, row_pitch;/* forward only, 0 = dense: source 0 as OVERLAPPING pixels -- pixel (y, x) starts
                                  * (y * row_pitch + x) * pix_pitch floats into the image and spans c0 >= pix_pitch floats
                                  * (c1 = 0, no upsample, kw = 1, pad_w = 0).  The 3-channel frozen stem uses it: rows of
                                  * 4-channel pixels, a "pixel" = the 8 neighbouring ones a 7-wide / stride-2 window reads
                                  * (c0 = 32, pix_pitch = 8): conv1 as a 7 x 1 convolution with K = 224 instead of 448 */
} codeps_conv_desc;

/* OIHW [out_c][in_c][kh][kw] -> w_fwd [out_c_pad][kh*kw][in_c_pad] and/or w_dgrad [in_c_pad][kh*kw][out_c_pad] (fp32), each
 * followed -- when its K extent (in_c_pad resp. out_c_pad) is a multiple of 32 channels -- by the same values pre-split into
 * three bf16 pieces ([row][K group of 32][piece][32], 6 bytes per element) from which the split-operand kernel stages its
 * weight operand.  Buffers hold codeps_conv2d_packed_weight_floats(..., dgrad = 0 / 1) floats; codeps_conv2d_fwd / _dgrad
 * expect packs produced by this function. */
size_t codeps_conv2d_packed_weight_floats(int out_c_pad, int in_c_pad, int kh, int kw, int dgrad);
/* All weight packs of a step in ONE launch.  A table of codeps_pack_item_bytes()-sized entries, one per (parameter, pack
 * kind), is filled on the HOST by codeps_pack_item_fill (first_group = running sum of the returned work-group counts; `out`
 * = a buffer of codeps_conv2d_packed_weight_floats floats; dgrad 0 = forward pack, 1 = data-gradient pack), copied to the
 * device once by the caller, and replayed every step by codeps_conv2d_pack_weights_multi: same bytes as
 * codeps_conv2d_pack_weights writes, one kernel instead of up to three per parameter and kind. */
size_t codeps_pack_item_bytes(void);
long long codeps_pack_item_fill(void* item, const float* w_oihw, float* out, int out_c, int in_c, int kh, int kw,
                                int out_c_pad, int in_c_pad, int dgrad, long long first_group);
int codeps_conv2d_pack_weights_multi(const void* items_device, int n_items, long long total_groups,
                                     codeps_stream_t stream);
int codeps_conv2d_pack_weights(int out_c, int in_c, int kh, int kw, int out_c_pad, int in_c_pad,
                               const float* w_oihw, float* w_fwd, float* w_dgrad, codeps_stream_t stream);
/* Arithmetic of a launch of this shape (mode 0 forward, 1 data gradient w.r.t. `source`): 3 = every fp32 operand as the exact
 * sum of three bf16 pieces, six piece products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (product error ~6e-9
 * relative, below fp32 rounding); 0 = v_mfma_f32_*_f32 (16-column kernels, 16-channel K chunks, all weight gradients).
 * Reporting only (bench.py prices each launch against the matching peak). */
int codeps_conv2d_arith(const codeps_conv_desc* d, int mode, int source);
/* Name of the kernel variant the last conv2d_{fwd,dgrad,wgrad} call of this PROCESS launched (diagnostics / tests; not
 * synchronised: meaningful when one thread at a time drives the library, e.g. forward, then backward, then the query). */
const char* codeps_conv2d_last_kernel(void);
/* Name of the finish pass of the last conv2d_{fwd,dgrad} launch that was cut into K slices: "splitk_finish_kernel", or
 * "splitk_finish_ex_kernel" when the epilogue extras (codeps_conv_epilogue) ran there ("" before any such launch). */
const char* codeps_conv2d_last_finish_kernel(void);
/* Process-wide operand arithmetic of the conv engine (CODEPS_SPLIT): 16 (default since ABI 10) = fp32 operands as TWO fp16 pieces of x 2^k
 * (power-of-two scales: per weight tensor from the pack, per block / row tile / pixel for activations and gradients inside the kernels), three
 * products per multiply on the f16 MFMA; kernels without that form run the next one.  3 = three exact bf16 pieces, six products on the bf16
 * MFMA.  0 = v_mfma_f32_32x32x2_f32.  Returns the previous value; any other argument only queries.  A weight pack carries the pre-split pieces
 * of the arithmetic it was made under: re-pack after a switch (codeps_amd.functional stamps its packs with the arithmetic). */
int codeps_conv2d_set_arith(int pieces);
/* Process-wide form of the 3x3 / stride-1 halo-tile ring kernel: 0 (default) = the 256-thread kernel in which every wave stages and
 * multiplies, 1 (CODEPS_HALO_WS=1) = 384 threads, four consumer waves (fragment reads + MFMAs) + two producer waves (weight tiles by
 * LDS DMA, activation tiles).  Same arithmetic and summation order: bit-identical results; measured within 1 % of each other (DESIGN.md
 * 6c).  Returns the previous value; other arguments only query. */
int codeps_conv2d_set_halo_ws(int on);
/* workspace (optional, may be NULL): scratch for split-K partial sums; codeps_conv2d_fwd_workspace_bytes(d) is the size
 * that enables it for this shape (0 = never splits).  Small-M / deep-K launches (B = 1 inference, 1/32-resolution
 * layers) are cut into K slices + a fixed-order finish pass so that they still fill the 256 CUs. */
size_t codeps_conv2d_fwd_workspace_bytes(const codeps_conv_desc* d);
int codeps_conv2d_fwd(const codeps_conv_desc* d, const float* x0, const float* x1, const float* w_fwd,
                      const float* scale, const float* shift, const float* residual, float* y,
                      void* workspace, size_t workspace_bytes, codeps_stream_t stream);
/* dz = gradient w.r.t. the pre-activation conv output [B,out_h,out_w,out_c]; dx = gradient of `source`
 * at ITS stored resolution (the x2-upsample transpose, a 2x2 sum, is fused for source 0 when up0). */
/* workspace (optional, may be NULL): split-K scratch as in the forward; codeps_conv2d_dgrad_workspace_bytes(d, source) sizes it. */
size_t codeps_conv2d_dgrad_workspace_bytes(const codeps_conv_desc* d, int source);
int codeps_conv2d_dgrad(const codeps_conv_desc* d, int source, const float* dz, const float* w_dgrad,
                        float* dx, void* workspace, size_t workspace_bytes, codeps_stream_t stream);
/* The same with a second gradient of the same tensor added in the kernel's epilogue: dx = data gradient + add (add: dx's shape, may
 * be NULL = codeps_conv2d_dgrad).  What autograd does with a tensor that has two consumers (a residual connection: the block's first
 * convolution and the identity path, models/resnet_encoder.py via torchvision's BasicBlock / Bottleneck) without a separate
 * full-size sum afterwards.  Stride-2 layers: pixels no tap reaches receive `add` alone. */
int codeps_conv2d_dgrad_add(const codeps_conv_desc* d, int source, const float* dz, const float* w_dgrad, const float* add,
                            float* dx, void* workspace, size_t workspace_bytes, codeps_stream_t stream);
/* Epilogue extras of a forward / data-gradient launch (round 5): work that used to be separate full-size passes right after the
 * convolution, done on the value the epilogue is about to store (after the residual / second-gradient add).
 *   gate_act 1 / 2: v *= ReLU' / ELU'(gate)  -- `gate` = the OUTPUT of the activation that produced the tensor whose gradient this
 *                   launch computes (its own input x0): the producer's activation backward (models/depth_head.py:63-71, the decoders'
 *                   conv + ELU chain; models/pose_head.py:36-44) needs no pass of its own.
 *   gate_act 3:     train-mode BatchNorm (+ ReLU) backward, first half (torchvision BasicBlock / Bottleneck behind
 *                   models/resnet_encoder.py:115-126): xhat = (bn_x - mean[g]) * invstd[g], g = statistic group of the row's image;
 *                   v is masked with (gate > 0) when `gate` (the block output, a residual fed the ReLU) is given, else with
 *                   (xhat * gamma + beta > 0) when bn_relu; stat_mode 3 sums (v, v * xhat) per tile: BatchNorm's backward reduce pass.
 *   stat_mode 1:    per M tile (shift = the tile's first row, sum (v - shift), sum (v - shift)^2) of a FORWARD launch: the batch
 *                   statistics of the train-mode BatchNorm that follows, without reading the tensor again.
 *   stat_mode 2:    per M tile column sums (the bias gradient of the gated tensor's producer).
 * stat_part: [tiles][3 | 1 | 2][cols] floats, codeps_conv2d_epilogue_part_bytes(GEMM rows of the launch, cols, stat_mode) bytes.
 * Out fields: `applied` bit 0 = the gate was applied, bit 1 = the statistics were written.  Statistics and the BatchNorm gate need
 * every M tile inside one image and a launch without K slices; when the tile plan of the shape cannot give that, they are dropped
 * TOGETHER (bits clear, result ungated) and the caller runs the separate kernels.  Tiles of segment k (the stride-2 data gradient is
 * four launches = up to four segments) are seg_tile0[k] .. + seg_tiles[k], seg_tile_rows[k] rows each, seg_tiles_per_image[k] per image. */
typedef struct {
    const float* gate; int32_t gate_act; int32_t stat_mode;
    float* stat_part; size_t stat_part_bytes;
    const float* bn_x; const float* bn_mean; const float* bn_invstd; const float* bn_gamma; const float* bn_beta;
    int32_t bn_relu, bn_groups; int32_t bn_group_b0[8];
    int32_t applied, n_seg; int32_t seg_tile0[4], seg_tiles_per_image[4], seg_tile_rows[4], seg_tiles[4];
} codeps_conv_epilogue;
size_t codeps_conv2d_epilogue_part_bytes(long long gemm_rows, int cols, int stat_mode);
/* codeps_conv2d_fwd / codeps_conv2d_dgrad_add with epilogue extras (epilogue may be NULL = the plain call). */
int codeps_conv2d_fwd_ex(const codeps_conv_desc* d, const float* x0, const float* x1, const float* w_fwd,
                         const float* scale, const float* shift, const float* residual, float* y,
                         void* workspace, size_t workspace_bytes, codeps_conv_epilogue* epilogue, codeps_stream_t stream);
int codeps_conv2d_dgrad_ex(const codeps_conv_desc* d, int source, const float* dz, const float* w_dgrad, const float* add,
                           float* dx, void* workspace, size_t workspace_bytes, codeps_conv_epilogue* epilogue,
                           codeps_stream_t stream);
/* Train-mode BatchNorm + ReLU applied on the CONSUMER's operand load (round 6; models/resnet_encoder.py:115-126 via torchvision's Bottleneck:
 * conv -> bn -> relu -> 1x1 conv; models/instance_head.py:38-46: depthwise 5x5 -> bn -> relu -> 1x1 conv).  x0 is the BatchNorm's INPUT,
 * in_affine [batch][mean | invstd | gamma | beta][c0] (codeps_bn_in_affine): the kernels compute relu((x - mean) invstd gamma + beta) in
 * registers where they stage their operand -- bit for bit what codeps_bn_fwd_mg's apply pass writes -- and the normalised tensor never
 * exists.  Served shapes (codeps_conv2d_in_affine_ok; which = 0 forward, 1 weight gradient): 1x1 / stride 1 / no padding, one dense source;
 * forward: > 16 output columns and out_h * out_w % 256 == 0; weight gradient: the shapes of the split-operand 1x1 kernel (>= 48 channels on
 * both sides) and of the <= 32-output-channel kernel.  The data
 * gradient needs nothing new: it never reads x0 (the BatchNorm gate of codeps_conv_epilogue recomputes the ReLU mask from bn_x). */
int codeps_conv2d_in_affine_ok(const codeps_conv_desc* d, int which);
int codeps_conv2d_fwd_in(const codeps_conv_desc* d, const float* x0, const float* in_affine, const float* w_fwd,
                         const float* scale, const float* shift, const float* residual, float* y,
                         void* workspace, size_t workspace_bytes, codeps_conv_epilogue* epilogue, codeps_stream_t stream);
int codeps_conv2d_wgrad_in(const codeps_conv_desc* d, const float* x0, const float* in_affine, const float* dz,
                           float* dw_oihw, int out_c_real, int in_c_real, int accumulate, void* workspace,
                           size_t workspace_bytes, codeps_stream_t stream);
/* Two FROZEN 1x1 layers of consecutive ResNet bottlenecks in one launch (inference only; csrc/chain1x1.hip; replaces
 * conv3 + bn3 + residual + ReLU of bottleneck i and conv1 + bn1 + ReLU of bottleneck i + 1, models/resnet_encoder.py:115-126):
 *     out[m][n2]  = relu(scale2 * (y1[m][:] . W2[n2][:]) + shift2 + residual[m][n2])     y1 [rows][256], out / residual [rows][n2]
 *     y0_next[m][c] = relu(scale3[c] * (out[m][:] . W3[c][:]) + shift3[c])               [rows][256]
 * `out` is handed from the first product to the second in registers (MFMA accumulator -> next MFMA operand).
 * n2 % 64 == 0.  w2_rows: n2 rows of codeps_chain1x1_w2_row_bytes() bytes = the pre-split pack row of W2 (256 x 6 bytes, the
 * layout behind codeps_conv2d_pack_weights' fp32 part) + (scale2, shift2, 0, 0) as four floats.  w3_split: the pre-split pack
 * of W3 [256][n2] whose k order inside every 32-channel group is p = 8 g + j -> channel 16 (j >> 2) + 4 g + (j & 3). */
size_t codeps_chain1x1_w2_row_bytes(void);
/* rows % 128 == 0.  Pixel tiles of 128 rows x K slices of the expand's channels fill the chip; with more than one slice the
 * reduce's partial sums pass through `workspace` (codeps_chain1x1_workspace_bytes, 0 = not needed) and a fixed-order finish pass. */
size_t codeps_chain1x1_workspace_bytes(long long rows, int n2);
int codeps_chain1x1_expand_reduce(const float* y1, const float* residual, const void* w2_rows, const void* w3_split,
                                  const float* scale3, const float* shift3, long long rows, int n2, float* out, float* y0_next,
                                  void* workspace, size_t workspace_bytes, codeps_stream_t stream);
/* The same launch on the two-fp16-piece arithmetic (codeps_conv2d_set_arith(16)): w2_rows = n2 rows of codeps_chain1x1_w2_row_bytes_h() bytes =
 * the fp16-pair pack row of W2 (256 x 4 bytes) + (scale2 * 2^-kb, shift2, 0, 0) with 2^-kb the pack's inverted per-tensor scale; w3_split = the
 * fp16-pair pack of the k-permuted reduce weights; scale3 must carry THAT pack's 2^-kb as well (codeps_amd.functional.chain_weights builds all
 * three).  Activations are scaled per pixel inside the kernel. */
size_t codeps_chain1x1_w2_row_bytes_h(void);
int codeps_chain1x1_expand_reduce_h(const float* y1, const float* residual, const void* w2_rows, const void* w3_split,
                                  const float* scale3, const float* shift3, long long rows, int n2, float* out, float* y0_next,
                                  void* workspace, size_t workspace_bytes, codeps_stream_t stream);
/* ... at the mid widths of the other bottleneck stages (fp16-pair arithmetic only): kmid = 256 (ResNet-50/101 layer3), 128 (layer2), 64 (layer1).
 * y1 / y0_next [rows][kmid]; w2_rows: n2 rows of kmid * 4 + 16 bytes; w3_split: fp16-pair pack of W3 [kmid][n2] (k-permuted as above);
 * scale3 / shift3 [kmid].  codeps_chain1x1_workspace_bytes_k returns 0 for an unsupported shape as well as for "not needed": the launch
 * validates the shape itself. */
size_t codeps_chain1x1_workspace_bytes_k(long long rows, int n2, int kmid);
int codeps_chain1x1_expand_reduce_hk(const float* y1, const float* residual, const void* w2_rows, const void* w3_split,
                                     const float* scale3, const float* shift3, long long rows, int n2, int kmid, float* out, float* y0_next,
                                     void* workspace, size_t workspace_bytes, codeps_stream_t stream);
size_t codeps_conv2d_wgrad_workspace_bytes(const codeps_conv_desc* d);
/* dw_oihw [out_c_real][in_c_real][kh][kw] (= or += when accumulate); split-K slabs + fixed-order reduce. */
int codeps_conv2d_wgrad(const codeps_conv_desc* d, const float* x0, const float* x1, const float* dz,
                        float* dw_oihw, int out_c_real, int in_c_real, int accumulate, void* workspace,
                        size_t workspace_bytes, codeps_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * NHWC building blocks (HBM-bound; rows = B*H*W pixels, c = padded channel count, multiple of 4).
 * Per-channel reductions are two-stage and fixed-order (deterministic); their scratch is
 * codeps_colsum_workspace_bytes(rows, c, k) with k = number of simultaneously reduced quantities.
 */
/* NCHW planar -> NHWC with channels [c_off, c_off+c) of a c_pad-wide pixel; zero_rest=1 also zero-fills
 * the other channels (image -> padded stem input; two calls build the 6-channel pose-net input). */
int codeps_nchw_to_nhwc(const float* x, int batch, int c, int h, int w, float* y, int c_pad, int c_off,
                        int zero_rest, codeps_stream_t stream);
/* Same, into rows that carry pad_left / pad_right extra zero pixels: y[batch, h, pad_left + w + pad_right, c_pad].
 * Double-pixel stem form (ResNet conv1 = 7x7 / 2 / 3 on <= 8 input channels, models/resnet_encoder.py:115): with
 * c_pad = 8, pad_left = 4, pad_right = 2 two neighbouring pixels form one 16-float "pixel" of a
 * [batch, h, (w + 6) / 2, 16] tensor, and conv1 becomes a 7 x 4 convolution of stride (2, 1), padding (3, 0) over it
 * whose K dimension is 7*4*16 = 448 instead of 7*7*16 = 784 (weights W'[o][p*8 + c][ky][j] = W[o][c][ky][2j + p - 1]). */
int codeps_nchw_to_nhwc_xpad(const float* x, int batch, int c, int h, int w, float* y, int c_pad, int c_off,
                             int zero_rest, int pad_left, int pad_right, codeps_stream_t stream);
/* Several NCHW frame batches into ONE such tensor in one launch: slab s = slab_batch[s] consecutive batch items of y whose pixels
 * take the c channels of frames_a[s] at channel 0 and (frames_b && frames_b[s]) of frames_b[s] at channel c, zeros elsewhere -- the pose
 * net's frame pairs concatenated along the batch (algos/depth.py:408-413) or the encoder's image keys (frames_b = NULL); 1..16 slabs. */
int codeps_frames_to_nhwc(const float* const* frames_a, const float* const* frames_b, const int* slab_batch, int n_slabs, int c,
                          int h, int w, float* y, int c_pad, int pad_left, int pad_right, codeps_stream_t stream);
int codeps_nhwc_to_nchw(const float* x, int batch, int c, int h, int w, int c_pad, float* y,
                        codeps_stream_t stream);
size_t codeps_colsum_workspace_bytes(long long rows, int c, int k);
/* dz = dy * act'(y) * scale[c];  dbias[c_real] (= / +=) sum_rows dy * act'(y).  act: 0 none 1 ReLU 2 ELU
 * (derivative from the OUTPUT y).  dz or dbias may be NULL.  Replaces ELU/ReLU backward + bias grad. */
int codeps_act_bwd(const float* dy, const float* y, const float* scale, int act, long long rows, int c,
                   float* dz, float* dbias, int c_real, int accumulate, void* workspace,
                   size_t workspace_bytes, codeps_stream_t stream);
/* out[c_real] (= / +=) sum over tiles of part[tiles][c]: the bias gradient from the column sums a data-gradient launch left per M tile
 * (codeps_conv_epilogue stat_mode 2), fixed order.  With the gate of that launch it replaces codeps_act_bwd for a conv + activation
 * whose output feeds one convolution (models/depth_head.py:63-71). */
int codeps_colsum_finish(const float* part, int tiles, int c, float* out, int c_real, int accumulate, codeps_stream_t stream);
/* n (1..32) such finishes in ONE launch, each accumulating into its output (outs[k][0 .. c_real[k]) += column sums of parts[k]
 * [tiles[k]][c[k]]): the bias gradients a backward pass leaves pending (codeps_amd.functional.flush_colsums). */
int codeps_colsum_finish_multi(const float* const* parts, const int* tiles, const int* c, float* const* outs, const int* c_real,
                               int n, codeps_stream_t stream);
/* Train-mode BatchNorm (nn.BatchNorm2d / SyncBatchNorm, misc/train_utils.py:50-51): statistics leave
 * the library as raw sums so the caller can all-reduce them over ranks before normalising.
 *   bn_stats:       sums[0][c] = sum (x - s_c), sums[1][c] = sum (x - s_c)^2 with the shift s_c = x[0][c]
 *                   (mean = s + sums[0]/n, var = sums[1]/n - (sums[0]/n)^2: no cancellation)
 *   bn_apply:       y = act((x - mean) * invstd * gamma + beta + residual)        act 0/1
 *   bn_bwd_reduce:  g = dy * act'(y);  sums[0] = sum g, sums[1] = sum g * xhat
 *   bn_bwd_apply:   dx = gamma*invstd*(g - sums[0]*inv_count - xhat*sums[1]*inv_count); dres = g (nullable) */
int codeps_bn_stats(const float* x, long long rows, int c, float* sums, void* workspace,
                    size_t workspace_bytes, codeps_stream_t stream);
int codeps_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, const float* residual, int act, long long rows, int c, float* y,
                    codeps_stream_t stream);
int codeps_bn_bwd_reduce(const float* dy, const float* y, const float* x, const float* mean,
                         const float* invstd, int act, long long rows, int c, float* sums, void* workspace,
                         size_t workspace_bytes, codeps_stream_t stream);
int codeps_bn_bwd_apply(const float* dy, const float* y, const float* x, const float* mean,
                        const float* invstd, const float* gamma, const float* sums, float inv_count, int act,
                        long long rows, int c, float* dx, float* dres, codeps_stream_t stream);
/* Multi-group train-mode BatchNorm: n_groups (<= 8) contiguous row ranges [row0[g], row0[g]+rows[g]) of one
 * [rows][c] tensor, each with its OWN batch statistics (several reference-style calls batched into one forward);
 * host arrays row0 / rows.  _fwd: statistics (shifted sums), mean / biased var / invstd [n_groups][c], running-stat
 * momentum updates applied group after group, then y = act((x-mean_g)*invstd_g*gamma+beta+residual).
 * stats_only / apply_only split the call for SyncBN (cross-rank combination of mean/var in between).
 * _bwd: sums [n_groups][2][c] = (sum g, sum g*xhat), dgamma/dbeta [c_real] (=, or += when accumulate) summed over
 * groups, dx and dres; inv_count = HOST array [n_groups] (1 / global element count of each group). */
size_t codeps_bn_mg_workspace_bytes(int n_groups, int c);
int codeps_bn_fwd_mg(const float* x, int n_groups, const long long* row0, const long long* rows, int c, int c_real,
                     const float* gamma, const float* beta, const float* residual, int act, float eps,
                     float momentum, float* running_mean, float* running_var, float* mean, float* var,
                     float* invstd, float* y, int stats_only, int apply_only, void* workspace,
                     size_t workspace_bytes, codeps_stream_t stream);
/* The table the kernels above read, from per-group batch statistics (mean / var / invstd [n_groups][c]; group g = images group_b0[g] ..
 * group_b1[g] - 1), + the running-statistics momentum updates that otherwise ride in the apply pass (running_mean may be NULL: SyncBN's
 * combine kernel already did them).  codeps_in_affine_apply: y = relu(BatchNorm(x)) from the table, for a consumer that cannot normalise
 * on its load. */
int codeps_bn_in_affine(const float* mean, const float* var, const float* invstd, const float* gamma, const float* beta, int n_groups,
                        const int* group_b0, const int* group_b1, int batch, int rows_per_image, int c, int c_real, float momentum,
                        float* running_mean, float* running_var, float* table, codeps_stream_t stream);
int codeps_in_affine_apply(const float* x, const float* table, int batch, int rows_per_image, int c, float* y, codeps_stream_t stream);
/* codeps_bn_finalize_tiles + codeps_bn_in_affine in one launch (running statistics: n_groups == 1 only, else NULL + codeps_bn_in_affine). */
int codeps_bn_finalize_tiles_table(const float* part, int tile0, int tiles_per_image, int tile_rows, int n_groups, const int* group_b0,
                                   const int* group_b1, int c, int c_real, float eps, float momentum, const float* gamma, const float* beta,
                                   float* mean, float* var, float* invstd, float* running_mean, float* running_var, float* table,
                                   codeps_stream_t stream);
/* SyncBatchNorm (misc/train_utils.py:50-51, torch.nn.SyncBatchNorm): after the all-gather of every rank's per-group
 * (mean, biased var) -- gathered [world][2][n_groups][c] -- merge them (parallel-variance formula), compute invstd and
 * apply the running-stat momentum updates group by group with the global counts (host array [n_groups]).
 * rank_counts: NULL = equal element counts on all ranks; else a HOST array [world][n_groups] of element counts (0 = the
 * rank holds nothing of that group, its slot of `gathered` is ignored) -- the unequal shards of BASELINE configs[3]. */
int codeps_bn_sync_combine(const float* gathered, int world, int n_groups, const float* global_counts,
                           const float* rank_counts, int c, int c_real, float eps, float momentum,
                           float* running_mean, float* running_var, float* mean, float* var, float* invstd,
                           codeps_stream_t stream);
int codeps_bn_bwd_mg(const float* dy, const float* y, const float* x, int n_groups, const long long* row0,
                     const long long* rows, int c, int c_real, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, const float* inv_count, int act, float* sums,
                     float* dgamma, float* dbeta, int accumulate, float* dx, float* dres, int reduce_only,
                     int apply_only, void* workspace, size_t workspace_bytes, codeps_stream_t stream);
/* Train-mode BatchNorm statistics from convolution epilogues (codeps_conv_epilogue, round 5) instead of passes over the tensor
 * (torchvision BatchNorm2d behind models/resnet_encoder.py:115-126; SyncBatchNorm: misc/train_utils.py:50-51).  Group g = images
 * [group_b0[g], group_b1[g]).
 * codeps_bn_finalize_tiles: stat_mode-1 partials of the producing forward launch -> mean / biased var / invstd [groups][c]; then
 *   codeps_bn_fwd_mg(apply_only = 1), which -- given running_mean / running_var there -- also applies the momentum updates group after
 *   group (SyncBatchNorm: mean / var go into the all-gather packet instead, codeps_bn_sync_combine updates the running statistics).
 * codeps_bn_bwd_tiles: stat_mode-3 partials of the data-gradient launch that produced AND masked BatchNorm's incoming gradient g.
 *   phase 1: partials -> local_sums [groups][2][c];  phase 2: dx = gamma * invstd * (g - sum g / n - xhat * sum (g xhat) / n) with `sums`
 *   (NULL = local_sums; SyncBatchNorm: the all-reduced copy) and dgamma / dbeta (= / +=) from local_sums;  phase 0: both. */
int codeps_bn_finalize_tiles(const float* part, int tile0, int tiles_per_image, int tile_rows, int n_groups, const int* group_b0,
                             const int* group_b1, int c, float eps, float* mean, float* var, float* invstd, codeps_stream_t stream);
int codeps_bn_bwd_tiles(const float* part, int n_seg, const int* seg_tile0, const int* seg_tiles_per_image, const int* seg_tile_rows,
                        int n_groups, const long long* row0, const long long* rows, const int* group_b0, const int* group_b1,
                        const float* g, const float* x, int c, int c_real, const float* mean, const float* invstd,
                        const float* gamma, const float* inv_count, float* local_sums, const float* sums, float* dgamma,
                        float* dbeta, int accumulate, float* dx, int phase, codeps_stream_t stream);
/* codeps_bn_bwd_mg: y may be NULL when act == 1 and the forward had no residual: the ReLU mask is then recomputed from
 * x, mean, invstd, gamma, beta (one tensor less to keep and to read per backward pass). */
/* The stem in train mode, fused (models/resnet_encoder.py:118-121: bn1 -> relu -> maxpool; torchvision ResNet._forward_impl):
 * fwd = relu(batch-norm(z)) max-pooled (3 x 3 / 2 / 1) in ONE pass over z -- the full-resolution activation is never written;
 * statistics (mean, invstd per group) come from codeps_bn_fwd_mg(stats_only = 1).  bwd = BatchNorm's two passes (sums, then dz)
 * reading the POOLED gradient + arg-max bytes: no full-resolution gradient of the ReLU output either.  Groups are whole images
 * (row0 / rows in pixels, multiples of h * w).  Bit-identical to codeps_bn_fwd_mg + codeps_maxpool3x3s2_fwd resp.
 * codeps_maxpool3x3s2_bwd + codeps_bn_bwd_mg (same expressions, same orders of summation).  workspace: codeps_bn_mg_workspace_bytes.
 * running_mean != NULL (round 5): the batch statistics came from codeps_bn_finalize_tiles (partials of the stem convolution's
 * epilogue) -- the momentum updates of running_mean / running_var [c_real] from mean / batch_var [n_groups, c], group after group,
 * ride in this launch. */
int codeps_bn_relu_pool_fwd(const float* z, int n_groups, const long long* row0, const long long* rows, int batch, int h,
                            int w, int c, const float* mean, const float* invstd, const float* gamma, const float* beta,
                            float* pooled, uint8_t* argmax, float* running_mean, float* running_var, const float* batch_var,
                            float momentum, int c_real, codeps_stream_t stream);
int codeps_bn_relu_pool_bwd(const float* dpooled, const uint8_t* argmax, const float* z, int n_groups, const long long* row0,
                            const long long* rows, int batch, int h, int w, int c, int c_real, const float* mean,
                            const float* invstd, const float* gamma, const float* beta, const float* inv_count, float* sums,
                            float* dgamma, float* dbeta, int accumulate, float* dz, void* workspace, size_t workspace_bytes,
                            int phase, codeps_stream_t stream);
/* (phase 0: everything; 1: sums + dgamma / dbeta only; 2: dz from `sums` only -- SyncBatchNorm all-reduces the sums in between) */
/* nn.MaxPool2d(3, 2, 1) (models/resnet_encoder.py:31); argmax = window tap 0..8 per element (u8). */
int codeps_maxpool3x3s2_fwd(const float* x, int batch, int h, int w, int c, float* y, uint8_t* argmax,
                            codeps_stream_t stream);
int codeps_maxpool3x3s2_bwd(const float* dy, const uint8_t* argmax, int batch, int h, int w, int c, float* dx,
                            codeps_stream_t stream);
/* 5x5 depthwise conv, stride 1, zero pad 2 (models/instance_head.py:38-46,90-98); weights [25][c];
 * y = act(conv * scale + shift); flip=1 gives the data gradient. */
int codeps_dwconv5x5(const float* x, const float* w25c, const float* scale, const float* shift, int act,
                     int batch, int h, int w, int c, int flip, float* y, codeps_stream_t stream);
int codeps_dwconv5x5_wgrad(const float* x, const float* dz, int batch, int h, int w, int c, float* dw25c,
                           void* workspace, size_t workspace_bytes, codeps_stream_t stream);
/* F.interpolate(mode="bilinear", align_corners=...) on NHWC, output written into channels
 * [y_c_off, y_c_off+c) of a y_c-wide pixel (concat-by-pointer); _bwd is the deterministic transpose. */
int codeps_bilinear_fwd(const float* x, int batch, int h, int w, int c, int out_h, int out_w,
                        int align_corners, float* y, int y_c, int y_c_off, codeps_stream_t stream);
int codeps_bilinear_bwd(const float* dy, int batch, int h, int w, int c, int out_h, int out_w,
                        int align_corners, int y_c, int y_c_off, float* dx, codeps_stream_t stream);
/* GUDA fusion (models/semantic_head.py:64-72) without the 240-channel concat: conv3x3(bilinear_up(d)) is evaluated as
 * sum_tap bilinear_up(W_tap . d)(p + delta_tap) -- the channel contraction runs at the decoder level's own (low)
 * resolution on the conv engine, this op interpolates the 9 x 16 tap planes at full resolution:
 *   out[B,h,w,16] = base[B,h,w,16] + sum_k sum_tap up(y_k[..., tap*16:(tap+1)*16])(p + delta_tap)   (zero outside the image)
 *   y = host array of n_branches (<= 3) device pointers, y_k = [B, bh[k], bw[k], 144] NHWC, align_corners=False.
 * _bwd: grad_y_k = transpose of the above for ONE branch (grad_out [B,h,w,16] -> grad_y [B,bh,bw,144]), as two
 * separable 1-D passes through a [B,h,bw,3,16] workspace. */
int codeps_guda_gather_fwd(const float* base, int batch, int h, int w, int n_branches, const float* const* y,
                           const int* bh, const int* bw, float* out, codeps_stream_t stream);
size_t codeps_guda_gather_bwd_workspace_bytes(int batch, int h, int bw);
int codeps_guda_gather_bwd(const float* grad_out, int batch, int h, int w, int bh, int bw, float* grad_y,
                           void* workspace, size_t workspace_bytes, codeps_stream_t stream);
/* nn.Softmax(dim=1) over the first c of c_pad channels (padding written as 0) and its backward. */
int codeps_softmax_fwd(const float* x, long long rows, int c, int c_pad, float* y, codeps_stream_t stream);
int codeps_softmax_bwd(const float* dy, const float* y, long long rows, int c, int c_pad, float* dx,
                       codeps_stream_t stream);
/* sigmoid + DepthHead.disp_to_depth (models/depth_head.py:48-54,71-74): channel 0 of logits[rows][c_pad]
 * -> planar disp[rows], depth[rows]; _bwd writes d logits (channel 0, rest zero). */
int codeps_disp_depth_fwd(const float* logits, long long rows, int c_pad, float min_depth, float max_depth,
                          float* disp, float* depth, codeps_stream_t stream);
int codeps_disp_depth_bwd(const float* ddisp, const float* ddepth, const float* disp, long long rows,
                          int c_pad, float min_depth, float max_depth, float* dlogits,
                          codeps_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Per-pixel losses (K11-K13).  Scalars (loss, grad_loss) are 1-element DEVICE buffers (no host sync).
 * Scratch for all of them: codeps_loss_workspace_bytes(pixels); the buffer written by a _fwd must be
 * handed unchanged to the matching _bwd (it holds the per-pixel loss map and the top-k threshold).
 */
size_t codeps_loss_workspace_bytes(long long pixels);
/* EdgeAwareSmoothnessLoss.__call__ (algos/depth.py:58-107): img [B,3,H,W], disp [B,1,H,W] planar;
 * mean_disp [B] is an output of _fwd and an input of _bwd. */
int codeps_smoothness_fwd(const float* img, const float* disp, int batch, int h, int w, float* mean_disp,
                          float* loss, void* workspace, size_t workspace_bytes, codeps_stream_t stream);
int codeps_smoothness_bwd(const float* img, const float* disp, const float* mean_disp,
                          const float* grad_loss, int batch, int h, int w, float* grad_disp, void* workspace,
                          size_t workspace_bytes, codeps_stream_t stream);
/* SemanticLoss.__call__ (algos/semantic_seg.py:45-67): CE applied to softmax PROBABILITIES
 * (probs NHWC [pixels][c_pad], first c real), target / pixel_weights u8 [pixels] (pixel_weights and
 * class_weights nullable), ignore_index 255, hard-pixel mining over the whole flattened batch with
 * k = int(top_k * pixels) via an integer radix select (top_k = 1.0: plain mean).
 * dropped_channels: bit c set = label c is in SemanticLoss.ignore_labels (semantic_seg.py:50-55): such
 * targets are ignored and the CE runs over the preserved channels only, which the reference keeps
 * indexing with the ORIGINAL label values (label t -> t-th preserved channel; class_weights hold one
 * entry per preserved channel).  0 = no ignore labels. */
int codeps_semantic_loss_fwd(const float* probs, int c, int c_pad, const uint8_t* target,
                             const uint8_t* pixel_weights, const float* class_weights, int ignore_index,
                             unsigned long long dropped_channels, long long pixels, double top_k, float* loss,
                             void* workspace, size_t workspace_bytes, codeps_stream_t stream);
int codeps_semantic_loss_bwd(const float* probs, int c, int c_pad, const uint8_t* target,
                             const uint8_t* pixel_weights, const float* class_weights, int ignore_index,
                             unsigned long long dropped_channels, long long pixels, double top_k,
                             const float* grad_loss, float* grad_probs, void* workspace, size_t workspace_bytes,
                             codeps_stream_t stream);
/* CenterLoss (kind 0, MSE), OffsetLoss (kind 1, L1), BinaryMaskLoss (kind 2, CE on probabilities,
 * ignore 255) -- algos/instance_seg.py:12-51.  pred NHWC [batch*hw][c_pad]; target planar NCHW fp32
 * [batch][c][hw] (kinds 0/1) or u8 [batch*hw] (kind 2).  Plain mean; pixel weights are ignored like
 * the reference does. */
int codeps_instance_loss_fwd(int kind, const float* pred, int c, int c_pad, const float* target_f,
                             const uint8_t* target_u8, int ignore_index, int batch, int hw, float* loss,
                             void* workspace, size_t workspace_bytes, codeps_stream_t stream);
int codeps_instance_loss_bwd(int kind, const float* pred, int c, int c_pad, const float* target_f,
                             const uint8_t* target_u8, int ignore_index, int batch, int hw,
                             const float* grad_loss, float* grad_pred, codeps_stream_t stream);
/* PoseHead tail (models/pose_head.py:46-137): feat NHWC [batch][hw][c_pad] -> vec6 = 0.01 * spatial mean
 * of channels 0..5 (frame 0: axis-angle, translation) -> T [batch][16] = T*R, or R^T*T(-t) for the items whose bit of
 * invert_mask is set (bit b & 63: 0 = none, ~0 = all; per-item masks need batch <= 64). */
int codeps_pose_fwd(const float* feat, int batch, int hw, int c_pad, unsigned long long invert_mask, float* vec6,
                    float* T, codeps_stream_t stream);
int codeps_pose_bwd(const float* grad_T, const float* vec6, int batch, int hw, int c_pad,
                    unsigned long long invert_mask, float* grad_feat, codeps_stream_t stream);
/* Weight re-layouts as single launches (csrc/weight_ops.hip).  stem: conv1 weight [out_c][in_c <= 8][7][7] <-> its
 * double-pixel form [out_c][16][7][4] (W'[o][p*8+c][ky][j] = W[o][c][ky][2j+p-1]).  guda: semconv_guda weight
 * [n_classes][in_c][3][3] (models/semantic_head.py:64-72) <-> n_coarse tap matrices [9*16][channels[k]] (row = tap * 16 +
 * class) followed by the fine branch's [n_classes][channels[n_coarse]][3][3] block; channels / outs are HOST arrays of
 * n_coarse + 1 entries (outs: device pointers; grad_outs entries may be NULL = zero).  accumulate: grad_w += . */
int codeps_stem_weight_fwd(const float* w, int out_c, int in_c, float* w_double_pixel, codeps_stream_t stream);
int codeps_stem_weight_bwd(const float* grad_double_pixel, int out_c, int in_c, int accumulate, float* grad_w,
                           codeps_stream_t stream);
int codeps_guda_weight_fwd(const float* w, int n_classes, int in_c, int n_coarse, const int* channels,
                           float* const* outs, codeps_stream_t stream);
int codeps_guda_weight_bwd(float* const* grad_outs, int n_classes, int in_c, int n_coarse, const int* channels,
                           int accumulate, float* grad_w, codeps_stream_t stream);
/* The data gradient of "nearest x2 -> conv3x3(padding 1)" w.r.t. the half-resolution source (the decoders' up-convolutions,
 * models/depth_head.py:60-71) is a 4 x 4 / stride-2 / padding-1 forward convolution of dz: w [out_c][in_c_total][3][3] -> wt
 * [c0][out_c][4][4], the OIHW weight of that convolution with the coinciding taps pre-summed (16 taps instead of 36 (sub-pixel,
 * tap) pairs); c0 = channels of the up-sampled source (input channels 0 .. c0 - 1). */
int codeps_up_dgrad_weight(const float* w, int out_c, int in_c_total, int c0, float* wt, codeps_stream_t stream);
/* Transpose of nn.ReflectionPad2d(pad) on NHWC: grad_padded [B][h+2p][w+2p][c] -> grad_x [B][h][w][c]. */
int codeps_reflect_pad_fold(const float* grad_padded, int batch, int h, int w, int c, int pad, float* grad_x,
                            codeps_stream_t stream);
/* One fused multi-tensor step on FLAT buffers: torch.optim.Adam (lr, betas, eps; no weight decay;
 * misc/train_utils.py:194-201) with grads pre-scaled by grad_scale (1/world_size after a SUM
 * all-reduce), optionally followed by the EMA teacher update ema = a*ema + (1-a)*p
 * (codeps/online_adap.py:279-289) when ema_params != NULL. */
int codeps_adam_ema_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                         float* ema_params, long long n, int step, float lr, float beta1, float beta2,
                         float eps, float grad_scale, float ema_alpha, codeps_stream_t stream);

/* SSIMLoss.__call__ (algos/depth.py:128-155) on its own (the training path evaluates it inside
 * codeps_photometric_*): x, y, out [planes][h][w] (planes = B * C of NCHW tensors); ReflectionPad2d(1),
 * 3x3 mean pooling, out = clamp((1 - n / d) / 2, 0, 1).  _bwd: grad_out [planes][h][w] -> grad_x,
 * grad_y (either nullable); workspace = codeps_ssim_bwd_workspace_bytes. */
int codeps_ssim_fwd(const float* x, const float* y, int planes, int h, int w, float c1, float c2, float* out,
                    codeps_stream_t stream);
size_t codeps_ssim_bwd_workspace_bytes(int planes, int h, int w);
int codeps_ssim_bwd(const float* x, const float* y, const float* grad_out, int planes, int h, int w, float c1,
                    float c2, float* grad_x, float* grad_y, void* workspace, size_t workspace_bytes,
                    codeps_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Image-file data path on the device (csrc/preprocess.hip): datasets/dataset.py:66-67 (Resize LANCZOS),
 * datasets/preprocessing.py:12-105 (colour jitter, flip, ToTensor + Normalize, histogram style transfer), datasets/
 * cityscapes.py:166-262 (label path, disparity -> depth).  Images are uint8 [batch][h][w][3] as the PNG decoder (host)
 * leaves them.
 * codeps_lanczos_coeffs (HOST function): Pillow's 8.22 fixed-point Lanczos tables for in_size -> out_size; returns ksize
 *   (bounds / coeffs NULL: size query); bounds int [out_size][2], coeffs int [out_size][ksize].  The caller uploads them.
 * codeps_resample_u8: one separable pass (axis 1 = along x, 0 = along y); PIL resizes x first, then y, 8-bit in between.
 * codeps_color_jitter_u8: n_ops / ops / factors are HOST arrays ([batch], [batch][4], [batch][4]; op 0 brightness, 1 contrast,
 *   2 saturation, 3 hue; application order = array order, preprocessing.py:102-104); tmp = second image buffer, grey_sums =
 *   192 bytes of device scratch; batch <= 16.
 * codeps_match_histograms_u8: transfer_histogram_style(mode="rgb") = skimage match_histograms per channel.
 * codeps_u8_to_tensor: flip (bit b of flip_mask), ToTensor, Normalize (mean / std HOST arrays [3] or both NULL) -> fp32 NCHW.
 * codeps_cityscapes_labels: labelIds u8 / instanceIds u16 [h][w] -> network-resolution maps (cv2.INTER_NEAREST resize,
 *   label_map256 = HOST table labelId -> train id (255 = ignore), thing ids as a bit mask, small-instance weights by area).
 * codeps_disparity_to_depth: Cityscapes 16-bit disparity -> depth [out_h][out_w] (baseline_times_fx = baseline * fx). */
int codeps_lanczos_coeffs(int in_size, int out_size, int* bounds, int* coeffs);
/* Pillow's Image.convert("HSV") (to_hsv != 0) / convert("RGB") of HSV pixels (to_hsv == 0) on packed uint8 [pixels][3]
 * (Convert.c rgb2hsv_row / hsv2rgb_row, operation by operation): the "hsv" / "saturation" modes of transfer_histogram_style
 * (datasets/preprocessing.py:64-71) are convert -> codeps_match_histograms_u8 -> convert back. */
int codeps_rgb_hsv_u8(const uint8_t* src, long long pixels, int to_hsv, uint8_t* dst, codeps_stream_t stream);
int codeps_resample_u8(const uint8_t* src, int batch, int h, int w, int axis, int out_size, const int* bounds,
                       const int* coeffs, int ksize, uint8_t* dst, codeps_stream_t stream);
int codeps_color_jitter_u8(const uint8_t* src, int batch, int h, int w, const int* n_ops, const int* ops,
                           const float* factors, uint8_t* dst, uint8_t* tmp, unsigned long long* grey_sums,
                           codeps_stream_t stream);
size_t codeps_match_histograms_workspace_bytes(int batch);
int codeps_match_histograms_u8(const uint8_t* img, const uint8_t* ref, int batch, int h, int w, int ref_h, int ref_w,
                               uint8_t* out, void* workspace, size_t workspace_bytes, codeps_stream_t stream);
int codeps_u8_to_tensor(const uint8_t* img, int batch, int h, int w, const float* mean, const float* std_,
                        unsigned flip_mask, float* out, codeps_stream_t stream);
size_t codeps_cityscapes_labels_workspace_bytes(void);
int codeps_cityscapes_labels(const uint8_t* semantic_ids, const uint16_t* instance_ids, int h, int w, int out_h,
                             int out_w, const uint8_t* label_map256, unsigned long long thing_class_mask,
                             float small_instance_area, int small_instance_weight, uint8_t* semantic, int* instance,
                             uint8_t* thing_mask, uint8_t* semantic_weights, uint8_t* center_weights,
                             uint8_t* offset_weights, void* workspace, size_t workspace_bytes, codeps_stream_t stream);
int codeps_disparity_to_depth(const uint16_t* disparity, int h, int w, int out_h, int out_w, float baseline_times_fx,
                              float* depth, codeps_stream_t stream);
/* KITTI-360 flavour of the label path (datasets/kitti_360.py:236-300): every thing pixel is an instance pixel, centre and offset
 * weights = thing mask, plus semantic_eval (instances below a tenth of the small-instance area -> 255).  codeps_lidar_depth:
 * the Velodyne projection of kitti_360.py:303-330 (points [n][4] fp32 on the device; lidar_to_rect [3][4] and camera_matrix [3][3]
 * HOST doubles; the last point in file order wins a pixel; nearest resize to [out_h][out_w]). */
int codeps_kitti360_labels(const uint8_t* semantic_ids, const uint16_t* instance_ids, int h, int w, int out_h, int out_w,
                           const uint8_t* label_map256, unsigned long long thing_class_mask, float small_instance_area,
                           int small_instance_weight, uint8_t* semantic, int* instance, uint8_t* thing_mask,
                           uint8_t* semantic_weights, uint8_t* center_weights, uint8_t* offset_weights,
                           uint8_t* semantic_eval, void* workspace, size_t workspace_bytes, codeps_stream_t stream);
size_t codeps_lidar_depth_workspace_bytes(int h, int w);
int codeps_lidar_depth(const float* points, int n, const double* lidar_to_rect, const double* camera_matrix, int h, int w,
                       int out_h, int out_w, float* depth, void* workspace, size_t workspace_bytes, codeps_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Step control on the stream (csrc/step_ctl.hip): the per-frame decisions the reference takes on the
 * host, kept on the device so that a captured HIP graph replays a REAL adaptation step.
 * codeps_step_counter_add: *counter += increment (the counter codeps_photometric_fwd mixes into its
 *   tie-break noise seed; one launch at the start of every step).
 * codeps_depth_key_weights: DepthAlgo.adaptation's image-count weights (algos/depth.py:562-577) with
 *   the rolling "car not moving" rule (depth.py:537-558) evaluated on the device: key_counts is a HOST
 *   array of n_keys image counts (order of the losses: source, target, target_replay); target_index
 *   = position of the "target" key or -1 (rule off); pose_prev / pose_next = the target key's [.,16]
 *   transformations (item 0 is read); rule_state = codeps_depth_rule_state_floats(length) zero-
 *   initialised floats kept by the caller across frames (length = 300 in the reference).
 *   coef [n_keys + 1] <- n_k * keep_k, then the denominator (their sum; 0 when every key is skipped).
 *   Batch-sharded step (world > 1): key_counts are this rank's image counts, global_total / global_target the image counts
 *   of all ranks (all depth keys / the target key); the denominator becomes (global_total - (1 - keep) * global_target) /
 *   world so that the mean over ranks of the local losses is the global count-weighted mean.  The rule is evaluated by ONE
 *   rank (mode 1: rule only, keep flag -> *keep_io), broadcast by the caller, and consumed by all (mode 2: keep flag <-
 *   *keep_io, no rule).  mode 0 = rule + weights in one launch (single rank; keep_io may be NULL).
 * codeps_weighted_sum_{fwd,bwd}: out = sum_i w_i f_i loss_i / D  (scripts/adapt_codeps.py:274-288 and
 *   depth.py:562-577): losses / factors / grads are HOST arrays of n DEVICE scalar pointers (factors
 *   may be NULL or hold NULL entries = 1; grads entries may be NULL), weights a HOST array, denom an
 *   optional DEVICE scalar (D <= 0 gives out = 0 and zero gradients).  n <= 16. */
int codeps_step_counter_add(uint64_t* counter, uint64_t increment, codeps_stream_t stream);
/* n int64 device counters bumped by increments[i] in ONE launch (host arrays of n pointers / n increments; the pointers are passed by
 * value in the kernel arguments): BatchNorm's num_batches_tracked of every layer a forward ran in train mode
 * (torch.nn.BatchNorm2d semantics, models/resnet_encoder.py via torchvision). */
int codeps_counters_add(long long* const* counters, const int* increments, int n, codeps_stream_t stream);
size_t codeps_depth_rule_state_floats(int length);
int codeps_depth_key_weights(const float* pose_prev, const float* pose_next, float* rule_state, int length,
                             int n_keys, const float* key_counts, int target_index, float global_total,
                             float global_target, int world, float* keep_io, int mode, float* coef,
                             codeps_stream_t stream);
int codeps_weighted_sum_fwd(const float* const* losses, const float* weights, const float* const* factors,
                            int n, const float* denom, float* out, codeps_stream_t stream);
int codeps_weighted_sum_bwd(const float* grad_out, const float* weights, const float* const* factors, int n,
                            const float* denom, float* const* grads, codeps_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Panoptic fusion (K14): InstanceSegAlgo.panoptic_fusion (algos/instance_seg.py:207-258) ->
 * get_panoptic_segmentation (misc/post_processing_panoptic.py:223-300), bit-exact, no host sync.
 *   semantic u8 [B,H,W] (255 = void), center fp32 [B,1,H,W], offset fp32 [B,2,H,W] (dy,dx) NCHW planar,
 *   thing_class_mask: bit c set <=> class c is a thing (c < 64).
 *   panoptic int16 [B,H,W] (class*label_divisor + id, stuff class*label_divisor, -1 unassigned),
 *   instance u8 [B,H,W] (1-based nearest-centre id inside the thing region, 0 elsewhere),
 *   centers int32 [B,256,2] (y,x; first state[b][1] rows valid), state int32 [B,4] =
 *   {#NMS candidates, #kept centres, bits of the top-k threshold, #kept before capping}. */
size_t codeps_panoptic_workspace_bytes(int batch, int h, int w);
int codeps_panoptic_fusion(const uint8_t* semantic, const float* center, const float* offset, int batch, int h,
                           int w, unsigned long long thing_class_mask, int label_divisor, int void_label,
                           float threshold, int nms_kernel, int top_k, int16_t* panoptic, uint8_t* instance,
                           int* centers, int* state, void* workspace, size_t workspace_bytes,
                           codeps_stream_t stream);
/* The same on maps that are channels of larger tensors (no repacking): element strides of the centre heat map (batch, pixel) and of
 * the offset field (batch, channel dy -> dx, pixel) -- NCHW planes: (H W, 1) and (2 H W, H W, 1); channels of an NHWC tensor with Cp
 * channels: (H W Cp, Cp) and (H W Cp, 1, Cp).  instance_i64 (nullable): the instance map once more as int64 (the reference returns it
 * in the semantic map's dtype, algos/instance_seg.py:226-240).  centers is zero-filled beyond the kept entries by the call. */
int codeps_panoptic_fusion_strided(const uint8_t* semantic, const float* center, long long center_batch_stride, int center_pixel_stride,
                                   const float* offset, long long offset_batch_stride, long long offset_channel_stride,
                                   int offset_pixel_stride, int batch, int h, int w, unsigned long long thing_class_mask,
                                   int label_divisor, int void_label, float threshold, int nms_kernel, int top_k, int16_t* panoptic,
                                   uint8_t* instance, long long* instance_i64, int* centers, int* state, void* workspace,
                                   size_t workspace_bytes, codeps_stream_t stream);

/* dst = sources[0] ++ ... ++ sources[n - 1] (1 <= n <= 8 contiguous fp32 arrays of counts[i] elements; host arrays): concatenation along
 * the leading dimension and plain copies in one launch. */
int codeps_concat_f32(const float* const* sources, const long long* counts, int n, float* dst, codeps_stream_t stream);

/* torch.argmax(x, dim = channel) on NHWC rows x [rows][c_pad] (first c channels; first index of the maximum): the semantic label map
 * of a probability tensor (algos/semantic_seg.py:70-86) as int64 and / or uint8 (either may be NULL) in one pass. */
int codeps_channel_argmax(const float* x, long long rows, int c, int c_pad, long long* out_i64, unsigned char* out_u8,
                          codeps_stream_t stream);

/* ImageWarper / CoordinateWarper (misc/image_warper.py:92-184), the reference's public warping API
 * (the training path's warp is fused into codeps_photometric_*): src [B,c,H,W] NCHW, depth [B,1,H,W],
 * pose [B,16], intr [B,4], motion = optional object_motion_map [B,3,H,W] added to the transformed
 * points (image_warper.py:133-134);
 * out [B,c,H,W] (nullable) = grid_sample(border, align_corners=True; nearest=1 for interp_mode
 * "nearest"); grid [B,H,W,2] (nullable) = normalised sampling coordinates.
 * _bwd: transpose of both outputs.  grad_out [B,c,H,W] and / or grad_grid [B,H,W,2] ->
 * grad_src [B,c,H,W] (ZERO-INITIALISED by the caller; accumulated with float atomics, the only ones in
 * this library), grad_depth [B,1,H,W], grad_pose [B,16], grad_motion [B,3,H,W] -- all nullable. */
int codeps_image_warp(const float* src, const float* depth, const float* pose, const float* intr,
                      const float* motion, int batch, int c, int h, int w, int nearest, float* out,
                      float* grid, codeps_stream_t stream);
size_t codeps_image_warp_bwd_workspace_bytes(int batch);
int codeps_image_warp_bwd(const float* src, const float* depth, const float* pose, const float* intr,
                          const float* motion, const float* grad_out, const float* grad_grid, int batch, int c,
                          int h, int w, int nearest, float* grad_src, float* grad_depth, float* grad_pose,
                          float* grad_motion, void* workspace, size_t workspace_bytes, codeps_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Pseudo-label mix-up on the device (Mixup.do_mixup, datasets/mixup.py:406-489, called from
 * CodepsNet.adapt, codeps/online_adap.py:331-335).  Camera intrinsics are [B,4] fp32 = (fx, fy, cx, cy)
 * (CameraModel.to_tensor, misc/camera_model.py:22-23); images / label maps are NCHW planar like the reference's.
 *
 * K15  Mixup.warp_c2c (mixup.py:211-229): out [B,c,ho,wo] = grid_sample(in.double(), grid.double(),
 *      align_corners=True) where the grid is the projection of the OUTPUT camera's viewing rays (intr_out, ho x wo,
 *      depth depth_val) into the INPUT camera (intr_in, hs x ws).  float64 coordinates / weights in the reference's
 *      operation order; the result is rounded once to the storage type.  dtype: 0 = fp32, 1 = uint8, 2 = int32;
 *      nearest = 1 -> mode "nearest" (the only mode for integer maps); border = 1 -> padding_mode "border", else "zeros". */
int codeps_warp_c2c(const void* in, int dtype, const float* intr_in, const float* intr_out, int batch, int c, int hs, int ws,
                    int ho, int wo, int nearest, int border, double depth_val, void* out, codeps_stream_t stream);
/* K16  Mixup.cut_mixup (mixup.py:314-344) fused with the three warp_c2c calls that precede it in do_mixup
 *      (mixup.py:431-441: target image bilinear / zeros, teacher instance map nearest / zeros, teacher semantic map
 *      nearest / border): inside the rectangles rects [B][nseg][4] = (y0, y1, x0, x1), end-exclusive exactly like the
 *      reference's slices (mixup.py:138-144 -- note its "-1"), the outputs take the (warped) target values, elsewhere the
 *      source's.  intr_src / intr_tgt NULL <=> cfg.general.geom_augment False (then ht x wt must equal h x w).
 *      img_* [B,3,.,.] fp32, label maps [B,.,.] int32. */
int codeps_cut_mixup(const float* img_src, const float* img_tgt, const int* sem_src, const int* sem_tgt, const int* inst_src,
                     const int* inst_tgt, const float* intr_src, const float* intr_tgt, const int* rects, int nseg, int batch,
                     int h, int w, int ht, int wt, double depth_val, float* rgb_mix, int* sem_mix, int* inst_mix,
                     codeps_stream_t stream);
/* K17  Mixup.get_off_cen (mixup.py:243-257) -> Dataset.get_offset_center (datasets/dataset.py:126-177) without the
 *      device->host->device round trip: instance int32 [B,H,W] (0 = stuff / unlabeled) -> center fp32 [B,1,H,W]
 *      (max of the per-instance Gaussians of width sigma around the rounded centre of mass), offset fp32 [B,2,H,W]
 *      ((dy, dx) to the float64 centre of mass), status int32 [B] = number of instances, or -1 when an image holds
 *      more than 4096 distinct ids (outputs then invalid; the host mirror raises). */
size_t codeps_offset_center_workspace_bytes(int batch);
int codeps_offset_center(const int* instance, int batch, int h, int w, double sigma, float* center, float* offset, int* status,
                         void* workspace, size_t workspace_bytes, codeps_stream_t stream);
/* Mixup.conf_instance_mixup / get_conf_mask (mixup.py:347-404, 542-652), one image at a time; label maps int32 [h][w]:
 *  codeps_conf_instance_stats: per instance id 1..255 of `instance` (the teacher's panoptic fusion) five int64
 *      (pixels, sum y, sum x, min x, max x; min x = 2^28 / max x = -1 when absent) into table (codeps_conf_stats_bytes).
 *  codeps_conf_place: kornia.contrib.distance_transform(things) (kernel 3, h 0.35; mixup.py:583-587) along `row` in closed
 *      form (Chebyshev ring index * 3 / 2 - h log(sum of the kernel over the neighbours reached earlier), replicate
 *      border) -> dist_row fp32 [w] (nullable) and *argmax = first column of its maximum.  things uint8 [h][w].
 *      kornia is not vendored by the reference (requirements.txt:16, unpinned): oracle/mixup.py restates its algorithm.
 *  codeps_conf_apply: shift[cid] = *argmax - cx, moved back inside the image with the instance's column range from
 *      `table` (mixup.py:586-601); the shifted instance's pixels become things and, where no earlier instance holds
 *      them, dst = cid (the later instance is cropped, mixup.py:611-628).  dst / things uint8 [h][w], shift int32 [256].
 *  codeps_conf_paste: the mixed item (mixup.py:366-381): dst == 0 -> source pixel, else the target pixel shift[dst]
 *      columns to the left (image, teacher semantic map, teacher instance map).  img_* [c][h][w] fp32. */
size_t codeps_conf_stats_bytes(void);
int codeps_conf_instance_stats(const int* instance, int h, int w, void* table, codeps_stream_t stream);
size_t codeps_conf_place_workspace_bytes(int w);
int codeps_conf_place(const uint8_t* things, int h, int w, int row, float* dist_row, int* argmax, void* workspace,
                      size_t workspace_bytes, codeps_stream_t stream);
int codeps_conf_apply(const int* instance, int h, int w, int cid, const void* table, int cx, const int* argmax, int* shift,
                      uint8_t* dst, uint8_t* things, codeps_stream_t stream);
int codeps_conf_paste(const float* img_src, const float* img_tgt, const int* sem_src, const int* sem_tgt, const int* inst_src,
                      const int* inst_conf, const uint8_t* dst, const int* shift, int c, int h, int w, float* rgb_mix,
                      int* sem_mix, int* inst_mix, codeps_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K18  Replay-buffer diversity sampler: the maximize_diversity branch of ReplayBuffer.add
 *      (datasets/replay_buffer.py:387-455, called per frame from scripts/adapt_codeps.py:307) with the faiss inner-product
 *      index (faiss-gpu==1.7.2: IndexIDMap(IndexFlatIP) + normalize_L2), the pairwise similarity matrix and the ids in
 *      ONE caller-owned device buffer `state` (codeps_replay_state_bytes; capacity = cfg.replay_buffer.target_size,
 *      dim = channels of the pooled feature).
 *      codeps_replay_add: features = the encoder's last NHWC feature map of ONE image ([h][w][stride] fp32, the first
 *      `dim` channels of every pixel used) -> mean over w then h -> L2-normalise -> cosine similarity to every stored
 *      vector; stored under id `index` when the largest similarity is < similarity_threshold (an empty index counts as
 *      similarity 0); when that makes capacity + 1 entries, the entry with the largest column sum of the similarity matrix
 *      (diagonal excluded, first maximum) is evicted.  force = 1: `features` is an already normalised vector (h = w = 1)
 *      that is stored unconditionally without eviction (ReplayBuffer.load_state, replay_buffer.py:475-488).
 *      result (device, int64 [3]) = {1 stored / 0 rejected / -1 no free slot, evicted id or -1, float bits of the
 *      largest similarity}.  Nothing synchronises. */
size_t codeps_replay_state_bytes(int capacity, int dim);
int codeps_replay_reset(void* state, size_t state_bytes, int capacity, int dim, codeps_stream_t stream);
int codeps_replay_add(void* state, size_t state_bytes, int capacity, int dim, const float* features, int h, int w, int stride,
                      long long index, float similarity_threshold, int force, long long* result, codeps_stream_t stream);
/* ids int64 [capacity+1] (-1 = free slot) and stored vectors fp32 [capacity+1][dim] for ReplayBuffer.save_state
 * (replay_buffer.py:462-473); device-to-device copies. */
int codeps_replay_export(const void* state, size_t state_bytes, int capacity, int dim, long long* ids, float* feats,
                         codeps_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * On-device evaluation metrics, run by the reference once per incoming frame in the teacher's "eval" pass
 * (scripts/adapt_codeps.py:166-172).
 *
 * K19  DepthEvaluator.compute_depth_metrics (eval/depth.py:21-68) / _compute_depth_stats (:111-131): per sample, over the
 *      pixels with depth_gt > 0 (and inside the Garg crop when garg_crop; and of class class_id when class_map != NULL):
 *      prediction scaled by median(gt) / median(pred) when use_gt_scale (torch.median = lower median), both clamped to
 *      [min_depth, max_depth]; stats fp32 [B][8] = d_a1, d_a2, d_a3, d_rmse, d_rmse_log, d_abs_rel, d_sq_rel, #valid pixels
 *      (NaN statistics for a sample without valid pixels).  depth_* fp32 [B,h,w], class_map uint8 [B,h,w]. */
size_t codeps_depth_metrics_workspace_bytes(int batch);
int codeps_depth_metrics(const float* depth_gt, const float* depth_pred, const uint8_t* class_map, int class_id, int batch, int h,
                         int w, int use_gt_scale, float min_depth, float max_depth, int garg_crop, float* stats, void* workspace,
                         size_t workspace_bytes, codeps_stream_t stream);
/* K20  SemanticEvaluator.compute_confusion_matrix (eval/semantic.py:52-65): conf int32 [C][C] (row = target, column =
 *      prediction) += counts over the n pixels whose target is neither ignore_index nor a class of ignore_class_mask (bit c).
 *      *bad is set to 1 if a valid pixel holds a class >= num_classes (index_add_ would leave its row).  C <= 64. */
int codeps_confusion_matrix(const uint8_t* pred, const uint8_t* target, long long n, int num_classes, int ignore_index,
                            unsigned long long ignore_class_mask, int* conf, int* bad, codeps_stream_t stream);
/* K21  PanopticEvaluator.update (eval/panoptic.py:26-45) -> compute_panoptic_stats (:134-209): pan_* int32 [B,h,w]
 *      (category * label_divisor + instance, void_label < 0 for unlabelled); iou fp64 [num_categories] += IoU of the matched
 *      (IoU > 0.5, same category) segment pairs; tp_fp_fn int32 [num_categories][3] += counts (FP skipped for predicted
 *      segments more than half void).  status int32 [B]: 0 ok, -1 more than 8192 distinct (gt, pred) pairs or 2048 ids in an
 *      image, -2 negative non-void id, -3 category >= num_categories. */
size_t codeps_panoptic_stats_workspace_bytes(int batch);
int codeps_panoptic_stats(const int* pan_gt, const int* pan_pred, int batch, int h, int w, int label_divisor, int void_label,
                          int num_categories, double* iou, int* tp_fp_fn, int* status, void* workspace, size_t workspace_bytes,
                          codeps_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The depth decoder's disparity output in one pass each way (csrc/disp_head.hip): dispconvs["0"] = ReflectionPad2d(1) ->
 * Conv2d(16, 1, 3), Sigmoid, DepthHead.disp_to_depth (models/depth_head.py:44-59,73-77).  x [B][h][w][16] NHWC fp32, weight
 * [1][16][3][3] (OIHW), bias [1] (nullable); disp / depth [B][1][h][w]; logit (nullable) [B][1][h][w] receives the pre-sigmoid
 * convolution output (models/depth_head.py:73: the quantity north_star's "logits within 1e-3" is stated on).  _bwd: ddisp / ddepth [B][1][h][w] (either nullable)
 * -> dx [B][h][w][16] (nullable), dweight [1][16][3][3], dbias [1] (nullable; += when accumulate); deterministic. */
int codeps_disp_head_fwd(const float* x, const float* weight, const float* bias, int batch, int h, int w, int c, float min_depth,
                         float max_depth, float* disp, float* depth, float* logit, codeps_stream_t stream);
size_t codeps_disp_head_bwd_workspace_bytes(int batch, int h, int w);
int codeps_disp_head_bwd(const float* x, const float* weight, const float* disp, const float* ddisp, const float* ddepth, int batch,
                         int h, int w, int c, float min_depth, float max_depth, float* dx, float* dweight, float* dbias,
                         int accumulate, void* workspace, size_t workspace_bytes, codeps_stream_t stream);
/* The same for an input x that is the OUTPUT of an activation read by this head alone (round 5; models/depth_head.py:63-71, the
 * decoder's last ELU): x_act 1 / 2 = ReLU / ELU: dx is multiplied by act'(x) in the kernel that produces it, and
 * x_colsum_part [codeps_disp_head_dx_blocks(batch, h, w)][16] (optional) receives the per-block column sums of the gated dx -- the
 * producer's activation backward and bias gradient without a pass over the full-resolution tensor. */
int codeps_disp_head_dx_blocks(int batch, int h, int w);
int codeps_disp_head_bwd_gated(const float* x, const float* weight, const float* disp, const float* ddisp, const float* ddepth,
                               int batch, int h, int w, int c, float min_depth, float max_depth, float* dx, float* dweight,
                               float* dbias, int accumulate, int x_act, float* x_colsum_part, void* workspace,
                               size_t workspace_bytes, codeps_stream_t stream);

/* The decoders' last up-convolution (csrc/upconv16.hip): nearest x2 -> Conv2d(16, 16, 3, padding 1) (+ bias, act 0 none / 1 ReLU /
 * 2 ELU) with no skip tensor (models/depth_head.py:60-71 at scale 0) as four 2 x 2 convolutions of the half-resolution tensor
 * (taps that read the same source pixel pre-summed), exact fp32 on the f32 MFMA.  x [B][hs][ws][16] NHWC, weight [16][16][3][3],
 * y / dz [B][2 hs][2 ws][16]; workspace (forward, _dgrad) = codeps_upconv16_workspace_bytes: the pre-summed weights.
 * _dgrad: dz -> dx [B][hs][ws][16]; _wgrad: dweight [16][16][3][3] (+= when accumulate; workspace =
 * codeps_upconv16_wgrad_workspace_bytes, deterministic block order). */
size_t codeps_upconv16_workspace_bytes(void);
int codeps_upconv16_fwd(const float* x, const float* weight, const float* bias, int batch, int hs, int ws, int act, float* y,
                        void* workspace, size_t workspace_bytes, codeps_stream_t stream);
int codeps_upconv16_dgrad(const float* dz, const float* weight, int batch, int hs, int ws, float* dx, void* workspace,
                          size_t workspace_bytes, codeps_stream_t stream);
/* ... with the gate of its source tensor (x = the activation output this layer alone reads; see codeps_disp_head_bwd_gated):
 * x_colsum_part [codeps_upconv16_dgrad_blocks(batch, hs, ws)][16]. */
int codeps_upconv16_dgrad_blocks(int batch, int hs, int ws);
int codeps_upconv16_dgrad_gated(const float* dz, const float* weight, int batch, int hs, int ws, float* dx, const float* x,
                                int x_act, float* x_colsum_part, void* workspace, size_t workspace_bytes,
                                codeps_stream_t stream);
size_t codeps_upconv16_wgrad_workspace_bytes(void);
int codeps_upconv16_wgrad(const float* x, const float* dz, int batch, int hs, int ws, float* dweight, int accumulate, void* workspace,
                          size_t workspace_bytes, codeps_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CODEPS_HIP_H */
