"""Import shim for the *real* CoDEPS reference (development container only).

Used ONLY by tests/golden/make_golden.py to generate fixtures.  /root/reference does not exist on
the GPU box, nothing in tests/, bench.py or __graft_entry__ imports this module at run time.
Missing third-party packages of the reference (torchvision, cv2, yacs, ...) are replaced by inert
placeholders so that the reference's own `models`, `algos`, `misc`, `codeps` packages import and run
on CPU (recipe: SURVEY.md Appendix B).
"""
import copy
import sys
import types
from unittest.mock import MagicMock

from torch import nn

REF = "/root/reference"


def install():
    if REF in sys.path:
        return
    sys.path.insert(0, REF)
    tv, tvm, tvr = (types.ModuleType(n) for n in
                    ("torchvision", "torchvision.models", "torchvision.models.resnet"))
    tv.__version__ = "0.13.1"

    class _R(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tvm.ResNet, tvr.BasicBlock, tvr.Bottleneck, tvm.resnet, tv.models = _R, object, object, tvr, tvm
    for n in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(tvm, n, None)
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm,
                        "torchvision.models.resnet": tvr,
                        "torchvision.transforms": MagicMock(),
                        "torchvision.transforms.functional": MagicMock()})
    for m in ("cv2", "kornia", "kornia.contrib", "faiss", "skimage", "skimage.exposure", "mmcv",
              "mmcv.parallel", "mmcv.parallel.data_container", "wandb", "seaborn"):
        sys.modules[m] = MagicMock()

    def _distance_transform(image, kernel_size=3, h=0.35):
        """kornia is absent: the reference's `conf_instance_mixup` runs on the oracle's restatement of
        kornia.contrib.distance_transform (oracle/mixup.py; parity unpinned for that function, pinned for everything the
        reference builds around it)."""
        import numpy as np
        import torch
        from oracle.mixup import kornia_distance_transform
        out = np.stack([np.stack([kornia_distance_transform(c.numpy(), kernel_size, h) for c in b]) for b in image])
        return torch.from_numpy(out)

    sys.modules["kornia.contrib"].distance_transform = _distance_transform

    class CfgNode(dict):
        def __getattr__(self, k):
            if k in self:
                return self[k]
            raise AttributeError(k)
        __setattr__ = dict.__setitem__

        def clone(self):
            return copy.deepcopy(self)

    y, yc = types.ModuleType("yacs"), types.ModuleType("yacs.config")
    yc.CfgNode = CfgNode
    y.config = yc
    sys.modules.update({"yacs": y, "yacs.config": yc})
