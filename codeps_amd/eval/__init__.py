"""Device-side mirrors of the reference's `eval` package: same class names, constructors and
methods as eval/{depth,semantic,panoptic}.py, computed by the K19-K21 kernels without host round trips."""
from .depth import DepthEvaluator
from .panoptic import PanopticEvaluator
from .semantic import SemanticEvaluator

__all__ = ["DepthEvaluator", "PanopticEvaluator", "SemanticEvaluator"]
