"""Host-side mirror of the reference's `algos` package (algos/__init__.py:1-25), hot-path members."""
from .depth import DepthAlgo, EdgeAwareSmoothnessLoss, ReconstructionLoss, SSIMLoss
from .instance_seg import BinaryMaskLoss, CenterLoss, InstanceSegAlgo, OffsetLoss
from .semantic_seg import SemanticLoss, SemanticSegAlgo

__all__ = ["BinaryMaskLoss", "CenterLoss", "DepthAlgo", "EdgeAwareSmoothnessLoss", "InstanceSegAlgo", "OffsetLoss",
           "ReconstructionLoss", "SemanticLoss", "SemanticSegAlgo", "SSIMLoss"]
