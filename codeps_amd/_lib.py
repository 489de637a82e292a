"""ctypes binding of libcodeps_hip.so (the C ABI declared in include/codeps_hip.h).

`import torch` happens first on purpose: the library NEEDs libamdhip64.so.7 and must bind to the copy
PyTorch-ROCm already mapped (one HIP runtime per process).  There is no fallback path: if the
library is missing `lib()` raises, and every op in codeps_amd goes through `lib()`.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint64, c_void_p

import torch  # noqa: F401  (must precede loading the .so, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# CODEPS_HIP_LIB: developer override to A/B two builds of the library in one session (never set in production)
LIB_PATH = os.environ.get("CODEPS_HIP_LIB") or os.path.join(_HERE, "libcodeps_hip.so")
CSRC = os.path.join(_HERE, "csrc")
ABI_VERSION = 11
DIAG_LIB_PATH = os.path.join(_HERE, "libcodeps_diag.so")

_lib = None
_diag = None


class CodepsHipError(RuntimeError):
    pass


class PhotometricDesc(Structure):
    _fields_ = [("batch", c_int32), ("height", c_int32), ("width", c_int32), ("num_scales", c_int32),
                ("alpha", c_float), ("noise_seed", c_uint64)]


class ConvEpilogue(Structure):
    """codeps_conv_epilogue (include/codeps_hip.h): gate + per-tile statistics of a forward / data-gradient launch's epilogue."""
    _fields_ = [("gate", c_void_p), ("gate_act", c_int32), ("stat_mode", c_int32), ("stat_part", c_void_p), ("stat_part_bytes", c_size_t),
                ("bn_x", c_void_p), ("bn_mean", c_void_p), ("bn_invstd", c_void_p), ("bn_gamma", c_void_p), ("bn_beta", c_void_p),
                ("bn_relu", c_int32), ("bn_groups", c_int32), ("bn_group_b0", c_int32 * 8),
                ("applied", c_int32), ("n_seg", c_int32), ("seg_tile0", c_int32 * 4), ("seg_tiles_per_image", c_int32 * 4),
                ("seg_tile_rows", c_int32 * 4), ("seg_tiles", c_int32 * 4)]


class ConvDesc(Structure):
    _fields_ = [(n, c_int32) for n in ("batch", "in_h", "in_w", "c0", "c1", "up0", "out_c", "out_h", "out_w", "kh", "kw",
                                       "stride", "pad", "pad_mode", "act", "stride_w", "pad_w", "pix_pitch", "row_pitch")]


def build(jobs=8, verbose=False):
    """Compile every HIP source for gfx950 into the in-tree shared library (hipcc cross-compiles
    without a GPU)."""
    out = subprocess.run(["make", "-C", CSRC, f"-j{jobs}"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout[-4000:], out.stderr[-8000:])
    if out.returncode != 0:
        raise CodepsHipError("building libcodeps_hip.so failed")
    return LIB_PATH


def _sig(fn, res, *args):
    fn.restype = res
    fn.argtypes = list(args)


# name -> (restype, argtypes); this table IS the Python view of include/codeps_hip.h and is checked
# against the header by tests/test_abi.py
P = c_void_p
I = c_int
LL = ctypes.c_longlong
SIGNATURES = {
    "codeps_version": (c_int, []),
    "codeps_last_error": (c_char_p, []),
    "codeps_photometric_workspace_bytes": (c_size_t, [POINTER(PhotometricDesc)]),
    "codeps_photometric_fwd": (c_int, [POINTER(PhotometricDesc), P, P, P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "codeps_photometric_bwd": (c_int, [POINTER(PhotometricDesc), P, P, P, P, P, P, P, P, P, P, P, P, c_size_t, P]),
    "codeps_conv2d_pack_weights": (c_int, [I, I, I, I, I, I, P, P, P, P]),
    "codeps_conv2d_packed_weight_floats": (c_size_t, [I, I, I, I, I]),
    "codeps_pack_item_bytes": (c_size_t, []),
    "codeps_pack_item_fill": (LL, [P, P, P, I, I, I, I, I, I, I, LL]),
    "codeps_conv2d_pack_weights_multi": (c_int, [P, I, LL, P]),
    "codeps_conv2d_arith": (c_int, [POINTER(ConvDesc), I, I]),
    "codeps_conv2d_last_kernel": (c_char_p, []),
    "codeps_conv2d_last_finish_kernel": (c_char_p, []),
    "codeps_conv2d_set_arith": (c_int, [I]),
    "codeps_conv2d_set_halo_ws": (c_int, [I]),
    "codeps_conv2d_fwd_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "codeps_conv2d_fwd": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P, P, c_size_t, P]),
    "codeps_conv2d_dgrad_workspace_bytes": (c_size_t, [POINTER(ConvDesc), I]),
    "codeps_conv2d_dgrad": (c_int, [POINTER(ConvDesc), I, P, P, P, P, c_size_t, P]),
    "codeps_conv2d_dgrad_add": (c_int, [POINTER(ConvDesc), I, P, P, P, P, P, c_size_t, P]),
    "codeps_conv2d_epilogue_part_bytes": (c_size_t, [LL, I, I]),
    "codeps_conv2d_fwd_ex": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P, P, c_size_t, POINTER(ConvEpilogue), P]),
    "codeps_conv2d_dgrad_ex": (c_int, [POINTER(ConvDesc), I, P, P, P, P, P, c_size_t, POINTER(ConvEpilogue), P]),
    "codeps_conv2d_in_affine_ok": (c_int, [POINTER(ConvDesc), I]),
    "codeps_conv2d_fwd_in": (c_int, [POINTER(ConvDesc), P, P, P, P, P, P, P, P, c_size_t, POINTER(ConvEpilogue), P]),
    "codeps_conv2d_wgrad_in": (c_int, [POINTER(ConvDesc), P, P, P, P, I, I, I, P, c_size_t, P]),
    "codeps_conv2d_wgrad_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "codeps_conv2d_wgrad": (c_int, [POINTER(ConvDesc), P, P, P, P, I, I, I, P, c_size_t, P]),
    "codeps_chain1x1_w2_row_bytes": (c_size_t, []),
    "codeps_chain1x1_workspace_bytes": (c_size_t, [LL, I]),
    "codeps_chain1x1_expand_reduce": (c_int, [P, P, P, P, P, P, LL, I, P, P, P, c_size_t, P]),
    "codeps_chain1x1_w2_row_bytes_h": (c_size_t, []),
    "codeps_chain1x1_expand_reduce_h": (c_int, [P, P, P, P, P, P, LL, I, P, P, P, c_size_t, P]),
    "codeps_chain1x1_workspace_bytes_k": (c_size_t, [LL, I, I]),
    "codeps_chain1x1_expand_reduce_hk": (c_int, [P, P, P, P, P, P, LL, I, I, P, P, P, c_size_t, P]),
    "codeps_nchw_to_nhwc": (c_int, [P, I, I, I, I, P, I, I, I, P]),
    "codeps_nchw_to_nhwc_xpad": (c_int, [P, I, I, I, I, P, I, I, I, I, I, P]),
    "codeps_frames_to_nhwc": (c_int, [P, P, P, I, I, I, I, P, I, I, I, P]),
    "codeps_nhwc_to_nchw": (c_int, [P, I, I, I, I, I, P, P]),
    "codeps_colsum_workspace_bytes": (c_size_t, [LL, I, I]),
    "codeps_act_bwd": (c_int, [P, P, P, I, LL, I, P, P, I, I, P, c_size_t, P]),
    "codeps_bn_stats": (c_int, [P, LL, I, P, P, c_size_t, P]),
    "codeps_bn_apply": (c_int, [P, P, P, P, P, P, I, LL, I, P, P]),
    "codeps_bn_bwd_reduce": (c_int, [P, P, P, P, P, I, LL, I, P, P, c_size_t, P]),
    "codeps_bn_bwd_apply": (c_int, [P, P, P, P, P, P, P, c_float, I, LL, I, P, P, P]),
    "codeps_bn_mg_workspace_bytes": (c_size_t, [I, I]),
    "codeps_bn_fwd_mg": (c_int, [P, I, P, P, I, I, P, P, P, I, c_float, c_float, P, P, P, P, P, P, I, I, P, c_size_t, P]),
    "codeps_bn_in_affine": (c_int, [P, P, P, P, P, I, P, P, I, I, I, I, c_float, P, P, P, P]),
    "codeps_in_affine_apply": (c_int, [P, P, I, I, I, P, P]),
    "codeps_bn_finalize_tiles_table": (c_int, [P, I, I, I, I, P, P, I, I, c_float, c_float, P, P, P, P, P, P, P, P, P]),
    "codeps_bn_sync_combine": (c_int, [P, I, I, P, P, I, I, c_float, c_float, P, P, P, P, P, P]),
    "codeps_bn_bwd_mg": (c_int, [P, P, P, I, P, P, I, I, P, P, P, P, P, I, P, P, P, I, P, P, I, I, P, c_size_t, P]),
    "codeps_colsum_finish": (c_int, [P, I, I, P, I, I, P]),
    "codeps_colsum_finish_multi": (c_int, [P, P, P, P, P, I, P]),
    "codeps_bn_finalize_tiles": (c_int, [P, I, I, I, I, P, P, I, c_float, P, P, P, P]),
    "codeps_bn_bwd_tiles": (c_int, [P, I, P, P, P, I, P, P, P, P, P, P, I, I, P, P, P, P, P, P, P, P, I, P, I, P]),
    "codeps_bn_relu_pool_fwd": (c_int, [P, I, P, P, I, I, I, I, P, P, P, P, P, P, P, P, P, c_float, I, P]),
    "codeps_bn_relu_pool_bwd": (c_int, [P, P, P, I, P, P, I, I, I, I, I, P, P, P, P, P, P, P, P, I, P, P, c_size_t, I, P]),
    "codeps_maxpool3x3s2_fwd": (c_int, [P, I, I, I, I, P, P, P]),
    "codeps_maxpool3x3s2_bwd": (c_int, [P, P, I, I, I, I, P, P]),
    "codeps_dwconv5x5": (c_int, [P, P, P, P, I, I, I, I, I, I, P, P]),
    "codeps_dwconv5x5_wgrad": (c_int, [P, P, I, I, I, I, P, P, c_size_t, P]),
    "codeps_bilinear_fwd": (c_int, [P, I, I, I, I, I, I, I, P, I, I, P]),
    "codeps_bilinear_bwd": (c_int, [P, I, I, I, I, I, I, I, I, I, P, P]),
    "codeps_guda_gather_fwd": (c_int, [P, I, I, I, I, P, P, P, P, P]),
    "codeps_guda_gather_bwd_workspace_bytes": (c_size_t, [I, I, I]),
    "codeps_guda_gather_bwd": (c_int, [P, I, I, I, I, I, P, P, c_size_t, P]),
    "codeps_softmax_fwd": (c_int, [P, LL, I, I, P, P]),
    "codeps_softmax_bwd": (c_int, [P, P, LL, I, I, P, P]),
    "codeps_disp_depth_fwd": (c_int, [P, LL, I, c_float, c_float, P, P, P]),
    "codeps_disp_depth_bwd": (c_int, [P, P, P, LL, I, c_float, c_float, P, P]),
    "codeps_loss_workspace_bytes": (c_size_t, [LL]),
    "codeps_smoothness_fwd": (c_int, [P, P, I, I, I, P, P, P, c_size_t, P]),
    "codeps_smoothness_bwd": (c_int, [P, P, P, P, I, I, I, P, P, c_size_t, P]),
    "codeps_semantic_loss_fwd": (c_int, [P, I, I, P, P, P, I, ctypes.c_ulonglong, LL, c_double, P, P, c_size_t, P]),
    "codeps_semantic_loss_bwd": (c_int, [P, I, I, P, P, P, I, ctypes.c_ulonglong, LL, c_double, P, P, P, c_size_t, P]),
    "codeps_instance_loss_fwd": (c_int, [I, P, I, I, P, P, I, I, I, P, P, c_size_t, P]),
    "codeps_instance_loss_bwd": (c_int, [I, P, I, I, P, P, I, I, I, P, P, P]),
    "codeps_pose_fwd": (c_int, [P, I, I, I, ctypes.c_ulonglong, P, P, P]),
    "codeps_pose_bwd": (c_int, [P, P, I, I, I, ctypes.c_ulonglong, P, P]),
    "codeps_stem_weight_fwd": (c_int, [P, I, I, P, P]),
    "codeps_stem_weight_bwd": (c_int, [P, I, I, I, P, P]),
    "codeps_guda_weight_fwd": (c_int, [P, I, I, I, P, P, P]),
    "codeps_guda_weight_bwd": (c_int, [P, I, I, I, P, I, P, P]),
    "codeps_up_dgrad_weight": (c_int, [P, I, I, I, P, P]),
    "codeps_reflect_pad_fold": (c_int, [P, I, I, I, I, I, P, P]),
    "codeps_panoptic_workspace_bytes": (c_size_t, [I, I, I]),
    "codeps_channel_argmax": (c_int, [P, LL, I, I, P, P, P]),
    "codeps_counters_add": (c_int, [P, P, I, P]),
    "codeps_concat_f32": (c_int, [P, P, I, P, P]),
    "codeps_panoptic_fusion": (c_int, [P, P, P, I, I, I, ctypes.c_ulonglong, I, I, c_float, I, I, P, P, P, P, P, c_size_t, P]),
    "codeps_panoptic_fusion_strided": (c_int, [P, P, LL, I, P, LL, LL, I, I, I, I, ctypes.c_ulonglong, I, I, c_float, I, I, P, P, P, P, P, P,
                                               c_size_t, P]),
    "codeps_image_warp": (c_int, [P, P, P, P, P, I, I, I, I, I, P, P, P]),
    "codeps_image_warp_bwd_workspace_bytes": (c_size_t, [I]),
    "codeps_image_warp_bwd": (c_int, [P, P, P, P, P, P, P, I, I, I, I, I, P, P, P, P, P, c_size_t, P]),
    "codeps_warp_c2c": (c_int, [P, I, P, P, I, I, I, I, I, I, I, I, ctypes.c_double, P, P]),
    "codeps_cut_mixup": (c_int, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, ctypes.c_double, P, P, P, P]),
    "codeps_offset_center_workspace_bytes": (c_size_t, [I]),
    "codeps_offset_center": (c_int, [P, I, I, I, ctypes.c_double, P, P, P, P, c_size_t, P]),
    "codeps_conf_stats_bytes": (c_size_t, []),
    "codeps_conf_instance_stats": (c_int, [P, I, I, P, P]),
    "codeps_conf_place_workspace_bytes": (c_size_t, [I]),
    "codeps_conf_place": (c_int, [P, I, I, I, P, P, P, c_size_t, P]),
    "codeps_conf_apply": (c_int, [P, I, I, I, P, I, P, P, P, P, P]),
    "codeps_conf_paste": (c_int, [P, P, P, P, P, P, P, P, I, I, I, P, P, P, P]),
    "codeps_replay_state_bytes": (c_size_t, [I, I]),
    "codeps_replay_reset": (c_int, [P, c_size_t, I, I, P]),
    "codeps_replay_add": (c_int, [P, c_size_t, I, I, P, I, I, I, LL, c_float, I, P, P]),
    "codeps_replay_export": (c_int, [P, c_size_t, I, I, P, P, P]),
    "codeps_depth_metrics_workspace_bytes": (c_size_t, [I]),
    "codeps_depth_metrics": (c_int, [P, P, P, I, I, I, I, I, c_float, c_float, I, P, P, c_size_t, P]),
    "codeps_confusion_matrix": (c_int, [P, P, LL, I, I, ctypes.c_ulonglong, P, P, P]),
    "codeps_panoptic_stats_workspace_bytes": (c_size_t, [I]),
    "codeps_panoptic_stats": (c_int, [P, P, I, I, I, I, I, I, P, P, P, P, c_size_t, P]),
    "codeps_adam_ema_step": (c_int, [P, P, P, P, P, LL, I, c_float, c_float, c_float, c_float, c_float, c_float, P]),
    "codeps_ssim_fwd": (c_int, [P, P, I, I, I, c_float, c_float, P, P]),
    "codeps_ssim_bwd_workspace_bytes": (c_size_t, [I, I, I]),
    "codeps_ssim_bwd": (c_int, [P, P, P, I, I, I, c_float, c_float, P, P, P, c_size_t, P]),
    "codeps_lanczos_coeffs": (c_int, [I, I, P, P]),
    "codeps_resample_u8": (c_int, [P, I, I, I, I, I, P, P, I, P, P]),
    "codeps_color_jitter_u8": (c_int, [P, I, I, I, P, P, P, P, P, P, P]),
    "codeps_match_histograms_workspace_bytes": (c_size_t, [I]),
    "codeps_match_histograms_u8": (c_int, [P, P, I, I, I, I, I, P, P, c_size_t, P]),
    "codeps_rgb_hsv_u8": (c_int, [P, LL, I, P, P]),
    "codeps_u8_to_tensor": (c_int, [P, I, I, I, P, P, ctypes.c_uint, P, P]),
    "codeps_cityscapes_labels_workspace_bytes": (c_size_t, []),
    "codeps_cityscapes_labels": (c_int, [P, P, I, I, I, I, P, ctypes.c_ulonglong, c_float, I, P, P, P, P, P, P, P, c_size_t, P]),
    "codeps_disparity_to_depth": (c_int, [P, I, I, I, I, c_float, P, P]),
    "codeps_kitti360_labels": (c_int, [P, P, I, I, I, I, P, ctypes.c_ulonglong, c_float, I, P, P, P, P, P, P, P, P, c_size_t, P]),
    "codeps_lidar_depth_workspace_bytes": (c_size_t, [I, I]),
    "codeps_lidar_depth": (c_int, [P, I, P, P, I, I, I, I, P, P, c_size_t, P]),
    "codeps_step_counter_add": (c_int, [P, c_uint64, P]),
    "codeps_depth_rule_state_floats": (c_size_t, [I]),
    "codeps_depth_key_weights": (c_int, [P, P, P, I, I, P, I, c_float, c_float, I, P, I, P, P]),
    "codeps_weighted_sum_fwd": (c_int, [P, P, P, I, P, P, P]),
    "codeps_weighted_sum_bwd": (c_int, [P, P, P, I, P, P, P]),
    "codeps_disp_head_fwd": (c_int, [P, P, P, I, I, I, I, c_float, c_float, P, P, P, P]),
    "codeps_disp_head_bwd_workspace_bytes": (c_size_t, [I, I, I]),
    "codeps_disp_head_bwd": (c_int, [P, P, P, P, P, I, I, I, I, c_float, c_float, P, P, P, I, P, c_size_t, P]),
    "codeps_disp_head_dx_blocks": (c_int, [I, I, I]),
    "codeps_disp_head_bwd_gated": (c_int, [P, P, P, P, P, I, I, I, I, c_float, c_float, P, P, P, I, I, P, P, c_size_t, P]),
    "codeps_upconv16_workspace_bytes": (c_size_t, []),
    "codeps_upconv16_fwd": (c_int, [P, P, P, I, I, I, I, P, P, c_size_t, P]),
    "codeps_upconv16_dgrad": (c_int, [P, P, I, I, I, P, P, c_size_t, P]),
    "codeps_upconv16_dgrad_blocks": (c_int, [I, I, I]),
    "codeps_upconv16_dgrad_gated": (c_int, [P, P, I, I, I, P, P, I, P, P, c_size_t, P]),
    "codeps_upconv16_wgrad_workspace_bytes": (c_size_t, []),
    "codeps_upconv16_wgrad": (c_int, [P, P, I, I, I, P, I, P, c_size_t, P]),
}

# include/codeps_diag.h: libcodeps_diag.so (micro-benchmarks + the pre-split 1x1 prototype; not part of the product ABI)
DIAG_SIGNATURES = {
    "codeps_diag_version": (c_int, []),
    "codeps_diag_last_error": (c_char_p, []),
    "codeps_diag_mfma_rate": (c_int, [I, I, I, I, I, P, P, P, P]),
    "codeps_diag_stream_triad": (c_int, [P, P, P, LL, I, P, P]),
    "codeps_diag_stream_copy": (c_int, [P, P, LL, I, P, P]),
    "codeps_diag_mfma_valu": (c_int, [I, I, I, I, I, P, P, P]),
    "codeps_diag_feed_rate": (c_int, [I, I, I, I, I, P, LL, P, P, P]),
    "codeps_diag_mfma_rate_clock": (c_int, [I, I, I, I, I, P, P, P, P]),
    "codeps_diag_mfma_valu_clock": (c_int, [I, I, I, I, I, P, P, P, P]),
    "codeps_diag_mfma_once": (c_int, [I, P, P, P, P, P]),
    "codeps_split_pack_rows": (c_int, [P, LL, I, I, P, P]),
    "codeps_conv1x1_split_fwd": (c_int, [P, LL, I, I, P, P, P, P, I, P, P]),
}


def lib():
    """The loaded library (cached).  Raises CodepsHipError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CodepsHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C codeps_amd/csrc`).  codeps_amd has no CPU / eager fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            _sig(getattr(handle, name), res, *args)
        if handle.codeps_version() != ABI_VERSION:
            raise CodepsHipError(f"libcodeps_hip.so ABI {handle.codeps_version()} != binding {ABI_VERSION}; rebuild")
        _lib = handle
    return _lib


def diag_lib():
    """libcodeps_diag.so (tools/, bench.py's measured peaks, the diagnostics tests); never needed by the hot path."""
    global _diag
    if _diag is None:
        if not os.path.exists(DIAG_LIB_PATH):
            raise CodepsHipError(f"{DIAG_LIB_PATH} not found: build it with `make -C codeps_amd/csrc`")
        handle = ctypes.CDLL(DIAG_LIB_PATH)
        for name, (res, args) in DIAG_SIGNATURES.items():
            _sig(getattr(handle, name), res, *args)
        _diag = handle
    return _diag


def check(rc, what):
    if rc != 0:
        raise CodepsHipError(f"{what} failed ({rc}): {lib().codeps_last_error().decode()}")


def check_diag(rc, what):
    if rc != 0:
        raise CodepsHipError(f"{what} failed ({rc}): {diag_lib().codeps_diag_last_error().decode()}")


def stream_ptr():
    """Current PyTorch HIP stream as the void* the C ABI expects."""
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise CodepsHipError("codeps_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
