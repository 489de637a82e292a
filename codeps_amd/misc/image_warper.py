"""CoordinateWarper / ImageWarper: drop-in for misc/image_warper.py:92-184 (forward-only public API;
the differentiable warp used for training lives inside the fused photometric kernel)."""
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


class CoordinateWarper(nn.Module):

    def __init__(self, img_width: int, img_height: int, device: torch.device = None):
        super().__init__()
        self.img_width, self.img_height, self.device = img_width, img_height, device

    def forward(self, batch_camera_models: List[CameraModel], batch_depth_map, T, object_motion_map=None):
        assert batch_depth_map.dim() == 4 and batch_depth_map.size(1) == 1, "depth must be [B,1,H,W]"
        if object_motion_map is not None:
            raise NotImplementedError("object_motion_map belongs to the flow head, which is out of scope (SURVEY.md §2 #6)")
        _lib.require_cuda(batch_depth_map, T)
        b, _, h, w = batch_depth_map.shape
        depth = batch_depth_map.detach().float().contiguous()
        grid = torch.empty(b, h, w, 2, dtype=torch.float32, device=depth.device)
        check(lib().codeps_image_warp(None, ptr(depth), ptr(T.detach().float().contiguous()), ptr(_intr(batch_camera_models, depth.device)),
                                      b, 1, h, w, 0, None, ptr(grid), stream_ptr()), "codeps_image_warp")
        return grid


class ImageWarper(nn.Module):

    def __init__(self, img_width: int, img_height: int, device: torch.device = None):
        super().__init__()
        self.coordinate_warper = CoordinateWarper(img_width, img_height, device)

    def forward(self, batch_camera_models: List[CameraModel], batch_src_img, batch_depth_map, T, interp_mode="bilinear",
                object_motion_map=None):
        assert batch_src_img.dim() == 4, "source images must be [B,C,H,W]"
        if object_motion_map is not None:
            raise NotImplementedError("object_motion_map belongs to the flow head, which is out of scope (SURVEY.md §2 #6)")
        _lib.require_cuda(batch_src_img, batch_depth_map, T)
        b, c, h, w = batch_src_img.shape
        src = batch_src_img.detach().float().contiguous()
        depth = batch_depth_map.detach().float().contiguous()
        out = torch.empty_like(src)
        check(lib().codeps_image_warp(ptr(src), ptr(depth), ptr(T.detach().float().contiguous()), ptr(_intr(batch_camera_models, src.device)),
                                      b, c, h, w, 1 if interp_mode == "nearest" else 0, ptr(out), None, stream_ptr()),
              "codeps_image_warp")
        return out
