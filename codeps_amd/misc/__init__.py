"""Host-side mirror of the hot-path parts of the reference's `misc` package (camera model, image / coordinate warper)."""
import importlib

# module -> public names (the package surface a user of the reference expects)
_EXPORTS = {'camera_model': ['CameraModel'], 'image_warper': ['ImageWarper', 'CoordinateWarper']}

for _module, _names in _EXPORTS.items():
    _loaded = importlib.import_module("." + _module, __name__)
    for _name in _names:
        globals()[_name] = getattr(_loaded, _name)
__all__ = sorted(_name for _names in _EXPORTS.values() for _name in _names)
del importlib
