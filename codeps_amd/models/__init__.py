"""Host-side mirror of the reference's `models` package: same class names, constructor / forward signatures and
state_dict keys; arithmetic in libcodeps_hip.so."""
import importlib

# module -> public names (the package surface a user of the reference expects)
_EXPORTS = {'resnet_encoder': ['ResnetEncoder'], 'depth_head': ['DepthHead'], 'pose_head': ['PoseHead'], 'semantic_head': ['SemanticHead'], 'instance_head': ['InstanceHead']}

for _module, _names in _EXPORTS.items():
    _loaded = importlib.import_module("." + _module, __name__)
    for _name in _names:
        globals()[_name] = getattr(_loaded, _name)
__all__ = sorted(_name for _names in _EXPORTS.values() for _name in _names)
del importlib
