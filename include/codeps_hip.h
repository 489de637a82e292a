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

int codeps_version(void);                 /* ABI version, bumped on any signature change */
const char* codeps_last_error(void);      /* thread-local, valid until the next failing call */

/* ------------------------------------------------------------------------------------------------
 * Photometric reconstruction loss (K7-K10), replaces ReconstructionLoss.__call__
 * (algos/depth.py:239-326) incl. Interpolate (depth.py:158-173), ImageWarper/CoordinateWarper
 * (misc/image_warper.py:100-184), CameraModel scaling/ray/projection (misc/camera_model.py:37-71),
 * SSIMLoss (depth.py:110-155), _compute_loss (depth.py:221-237) and the auto-mask min.
 *   img_*   [B,3,H,W] fp32 NCHW planar (t, t-1, t+1);   depth [B,1,H,W];
 *   pose_*  [B,16] row-major 4x4 (t->t-1, t->t+1);      intr [B,4] = fx,fy,cx,cy at (H,W).
 *   noise   optional [sum_s B*2*Hs*Ws] tie-break noise ALREADY scaled by 1e-5 (scale-major, then
 *           b, k, y, x); NULL => in-kernel counter-based N(0,1)*1e-5 from `noise_seed`.
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
                           const float* pose_p, const float* intr, const float* noise, float* loss,
                           void* workspace, size_t workspace_bytes, codeps_stream_t stream);
/* grad_loss [1] (device scalar: d total / d loss); grad_depth [B,1,H,W]; grad_pose_* [B,16]. */
int codeps_photometric_bwd(const codeps_photometric_desc* d, const float* img_t, const float* img_m,
                           const float* img_p, const float* depth, const float* pose_m,
                           const float* pose_p, const float* intr, const float* grad_loss,
                           float* grad_depth, float* grad_pose_m, float* grad_pose_p,
                           void* workspace, size_t workspace_bytes, codeps_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CODEPS_HIP_H */
