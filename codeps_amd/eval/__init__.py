"""Device-side mirrors of the reference's `eval` package: same class names, constructors and
methods as eval/{depth,semantic,panoptic}.py, computed by the K19-K21 kernels without host round trips."""
import importlib

# module -> public names (the package surface a user of the reference expects)
_EXPORTS = {'depth': ['DepthEvaluator'], 'semantic': ['SemanticEvaluator'], 'panoptic': ['PanopticEvaluator']}

for _module, _names in _EXPORTS.items():
    _loaded = importlib.import_module("." + _module, __name__)
    for _name in _names:
        globals()[_name] = getattr(_loaded, _name)
__all__ = sorted(_name for _names in _EXPORTS.values() for _name in _names)
del importlib
