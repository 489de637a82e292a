"""CoordinateWarper / ImageWarper: drop-in for misc/image_warper.py:92-184, differentiable like the reference's
(gradients w.r.t. the source image, the depth map, the transformation and the object-motion map).  The warp the training
path uses lives inside the fused photometric kernel (csrc/photometric.hip); this is the public API on its own kernels
(csrc/warp.hip)."""
from typing import List

import torch
from torch import nn

from .. import _lib
from .._lib import check, lib, ptr, stream_ptr
from .camera_model import CameraModel


def _intr(batch_camera_models, device):
    if isinstance(batch_camera_models, torch.Tensor):
        return batch_camera_models.to(device=device, dtype=torch.float32).contiguous()
    return CameraModel.batch_tensor(batch_camera_models, device).contiguous()


def _f32c(t):
    return None if t is None else t.float().contiguous()


class _Warp(torch.autograd.Function):
    """(out, grid) = warp(src, depth, T, motion); either output may be switched off (C ABI: codeps_image_warp{,_bwd})."""

    @staticmethod
    def forward(ctx, src, depth, T, motion, intr, nearest, want_out, want_grid):
        _lib.require_cuda(src, depth, T, motion, intr)
        depth, T, motion, src = _f32c(depth), _f32c(T).reshape(-1, 16), _f32c(motion), _f32c(src)
        b, _, h, w = depth.shape
        c = src.shape[1] if src is not None else 1
        dev = depth.device
        out = torch.empty(b, c, h, w, dtype=torch.float32, device=dev) if want_out else None
        grid = torch.empty(b, h, w, 2, dtype=torch.float32, device=dev) if want_grid else None
        check(lib().codeps_image_warp(ptr(src), ptr(depth), ptr(T), ptr(intr), ptr(motion), b, c, h, w, int(nearest), ptr(out), ptr(grid),
                                      stream_ptr()), "codeps_image_warp")
        ctx.save_for_backward(src, depth, T, motion, intr)
        ctx.meta = (b, c, h, w, int(nearest))
        ctx.set_materialize_grads(False)
        return out, grid

    @staticmethod
    def backward(ctx, g_out, g_grid):
        src, depth, T, motion, intr = ctx.saved_tensors
        b, c, h, w, nearest = ctx.meta
        if g_out is None and g_grid is None:
            return (None,) * 8
        dev = depth.device
        need_src, need_depth, need_T, need_m = ctx.needs_input_grad[:4]
        g_src = torch.zeros_like(src) if (need_src and g_out is not None and src is not None) else None
        g_depth = torch.empty_like(depth) if need_depth else None
        g_T = torch.empty(b, 16, dtype=torch.float32, device=dev) if need_T else None
        g_m = torch.empty_like(motion) if (need_m and motion is not None) else None
        nbytes = lib().codeps_image_warp_bwd_workspace_bytes(b)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        check(lib().codeps_image_warp_bwd(ptr(src), ptr(depth), ptr(T), ptr(intr), ptr(motion), ptr(_f32c(g_out)), ptr(_f32c(g_grid)),
                                          b, c, h, w, nearest, ptr(g_src), ptr(g_depth), ptr(g_T), ptr(g_m), ptr(ws), nbytes,
                                          stream_ptr()), "codeps_image_warp_bwd")
        return g_src, g_depth, None if g_T is None else g_T.view(b, 4, 4), g_m, None, None, None, None


class CoordinateWarper(nn.Module):

    def __init__(self, img_width: int, img_height: int, device: torch.device = None):
        super().__init__()
        self.img_width, self.img_height, self.device = img_width, img_height, device

    def forward(self, batch_camera_models: List[CameraModel], batch_depth_map, T, object_motion_map=None):
        assert batch_depth_map.dim() == 4, f"The input batch of depth maps has {batch_depth_map.dim()} dimensions which is != 4"
        assert batch_depth_map.size(1) == 1, f"The input batch of depth maps has {batch_depth_map.size(1)} channels which is != 1"
        intr = _intr(batch_camera_models, batch_depth_map.device)
        return _Warp.apply(None, batch_depth_map, T, object_motion_map, intr, False, False, True)[1]


class ImageWarper(nn.Module):

    def __init__(self, img_width: int, img_height: int, device: torch.device = None):
        super().__init__()
        self.coordinate_warper = CoordinateWarper(img_width, img_height, device)

    def forward(self, batch_camera_models: List[CameraModel], batch_src_img, batch_depth_map, T, interp_mode="bilinear",
                object_motion_map=None):
        assert batch_src_img.dim() == 4, f"The input batch of source images has {batch_src_img.dim()} dimensions which is != 4"
        if interp_mode not in ("bilinear", "nearest"):
            raise ValueError(f"interp_mode {interp_mode!r}: grid_sample on 4-D inputs knows 'bilinear' and 'nearest' here")
        intr = _intr(batch_camera_models, batch_src_img.device)
        return _Warp.apply(batch_src_img, batch_depth_map, T, object_motion_map, intr, interp_mode == "nearest", True, False)[0]
