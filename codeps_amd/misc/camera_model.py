"""CameraModel: drop-in for misc/camera_model.py:9-71.

Same constructor / methods, but the intrinsics may stay a DEVICE tensor: `from_tensor` does not force
the reference's per-sample D2H sync (camera_model.py:27) -- the fused photometric kernel reads a
[B,4] intrinsics tensor directly.  Scalar accessors convert lazily (and only then synchronise)."""
from __future__ import annotations

from collections import OrderedDict

import torch
from torch import Tensor


class CameraModel:

    def __init__(self, img_width: int, img_height: int, fx, fy, cx, cy, _tensor: Tensor = None):
        assert img_width > 0, "img_width <= 0 is not allowed"
        assert img_height > 0, "img_height <= 0 is not allowed"
        self.image_size = {"width": img_width, "height": img_height}
        self._tensor = _tensor
        if _tensor is None:
            assert fx >= 0 and fy >= 0 and cx >= 0 and cy >= 0, "negative intrinsics are not allowed"
            self._intr = OrderedDict([("fx", fx), ("fy", fy), ("cx", cx), ("cy", cy)])
        else:
            self._intr = None

    @property
    def intrinsics(self) -> OrderedDict:
        if self._intr is None:          # lazy host copy (syncs) -- only when somebody asks for scalars
            v = self._tensor.detach().cpu().numpy()
            self._intr = OrderedDict([("fx", v[0]), ("fy", v[1]), ("cx", v[2]), ("cy", v[3])])
        return self._intr

    def to_tensor(self) -> Tensor:
        if self._tensor is not None:
            return self._tensor
        return Tensor([float(v) for v in self.intrinsics.values()])

    @classmethod
    def from_tensor(cls, img_width: int, img_height: int, intrinsics: Tensor) -> "CameraModel":
        return cls(img_width, img_height, None, None, None, None, _tensor=intrinsics)

    @staticmethod
    def batch_tensor(models, device=None) -> Tensor:
        """[B,4] fp32 (fx,fy,cx,cy) for a list of models, without touching the host when they wrap
        device tensors."""
        rows = [m.to_tensor().to(torch.float32) for m in models]
        out = torch.stack(rows)
        return out if device is None else out.to(device)

    def get_scaled_model(self, scale_u, scale_v) -> "CameraModel":
        i = self.intrinsics
        return CameraModel(self.image_size["width"] * scale_u, self.image_size["height"] * scale_v, i["fx"] * scale_u,
                           i["fy"] * scale_v, i["cx"] * scale_u, i["cy"] * scale_v)

    def get_scaled_model_image_size(self, width, height) -> "CameraModel":
        su, sv = width / self.image_size["width"], height / self.image_size["height"]
        i = self.intrinsics
        return CameraModel(width, height, i["fx"] * su, i["fy"] * sv, i["cx"] * su, i["cy"] * sv)

    def get_image_point(self, x3d, y3d, z3d):
        i = self.intrinsics
        return (x3d / z3d) * i["fx"] + i["cx"], (y3d / z3d) * i["fy"] + i["cy"]

    def get_viewing_ray(self, u2d, v2d):
        i = self.intrinsics
        ray_x = (u2d - i["cx"]) / i["fx"]
        ray_y = (v2d - i["cy"]) / i["fy"]
        norm = torch.sqrt(ray_x ** 2 + ray_y ** 2 + 1.0)
        return ray_x / norm, ray_y / norm, 1.0 / norm
