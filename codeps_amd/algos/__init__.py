"""Host-side mirror of the hot-path members of the reference's `algos` package (losses and the three *Algo drivers)."""
import importlib

# module -> public names (the package surface a user of the reference expects)
_EXPORTS = {'depth': ['DepthAlgo', 'ReconstructionLoss', 'SSIMLoss', 'EdgeAwareSmoothnessLoss'], 'semantic_seg': ['SemanticSegAlgo', 'SemanticLoss'], 'instance_seg': ['InstanceSegAlgo', 'CenterLoss', 'OffsetLoss', 'BinaryMaskLoss']}

for _module, _names in _EXPORTS.items():
    _loaded = importlib.import_module("." + _module, __name__)
    for _name in _names:
        globals()[_name] = getattr(_loaded, _name)
__all__ = sorted(_name for _names in _EXPORTS.values() for _name in _names)
del importlib
