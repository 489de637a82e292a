"""Host-side mirror of the reference's `misc` package (hot-path parts only: misc/__init__.py)."""
from .camera_model import CameraModel
from .image_warper import CoordinateWarper, ImageWarper

__all__ = ["CameraModel", "CoordinateWarper", "ImageWarper"]
