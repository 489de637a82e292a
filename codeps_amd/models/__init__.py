"""Host-side mirror of the reference's `models` package (models/__init__.py:1-8): same class names,
constructor / forward signatures and state_dict keys; arithmetic in libcodeps_hip.so."""
from .depth_head import DepthHead
from .instance_head import InstanceHead
from .pose_head import PoseHead
from .resnet_encoder import ResnetEncoder
from .semantic_head import SemanticHead

__all__ = ["DepthHead", "InstanceHead", "PoseHead", "ResnetEncoder", "SemanticHead"]
