"""Instance losses and InstanceSegAlgo: drop-in for algos/instance_seg.py:12-258."""
from typing import Dict, Optional

import torch
from torch import Tensor

from .. import functional as F


class CenterLoss:
    """mean squared error; `pixel_weights` is accepted and ignored like the reference (instance_seg.py:17-25)."""

    def __call__(self, prediction: Tensor, target: Tensor, pixel_weights=None) -> Tensor:
        return F.center_loss(F.from_nchw(prediction), target)


class OffsetLoss:
    """mean absolute error; `pixel_weights` ignored like the reference (instance_seg.py:33-41)."""

    def __call__(self, prediction: Tensor, target: Tensor, pixel_weights=None) -> Tensor:
        return F.offset_loss(F.from_nchw(prediction), target)


class BinaryMaskLoss:
    def __init__(self, ignore_index: int = 255):
        self.ignore_index = ignore_index

    def __call__(self, prediction: Tensor, target: Tensor) -> Tensor:
        return F.binary_mask_loss(F.from_nchw(prediction), target, self.ignore_index)


class InstanceSegAlgo:

    def __init__(self, center_loss: CenterLoss, offset_loss: OffsetLoss, evaluator=None, binary_mask_loss: Optional[BinaryMaskLoss] = None,
                 thing_list=None):
        self.center_loss, self.offset_loss, self.binary_mask_loss = center_loss, offset_loss, binary_mask_loss
        self.evaluator = evaluator
        self._thing_list = thing_list

    @property
    def thing_list(self):
        if self._thing_list is not None:
            return self._thing_list
        return self.evaluator.thing_list

    def training(self, feats, center: Tensor, offset: Tensor, center_weights, offset_weights, thing_mask, instance_head):
        center_pred, offset_pred, thing_mask_logits = instance_head(feats)
        center_loss = self.center_loss(center_pred, center, center_weights)
        offset_loss = self.offset_loss(offset_pred, offset, offset_weights)
        if thing_mask_logits is not None:
            thing_mask_loss = self.binary_mask_loss(thing_mask_logits, thing_mask.squeeze(1))
            thing_mask_pred = F.channel_argmax(thing_mask_logits.detach(), torch.uint8)
        else:
            thing_mask_loss, thing_mask_pred = None, None
        return center_loss, offset_loss, center_pred, offset_pred, thing_mask_loss, thing_mask_pred

    def inference(self, feats, instance_head):
        center_pred, offset_pred, thing_mask_logits = instance_head(feats)
        thing_mask_pred = None if thing_mask_logits is None else F.channel_argmax(thing_mask_logits, torch.uint8)
        return center_pred, offset_pred, thing_mask_pred

    def evaluation(self):
        raise NotImplementedError

    def adaptation(self, feats: Dict[str, list], center, offset, center_weights, offset_weights, thing_mask, instance_head,
                   center_ema: Optional[Tensor] = None, offset_ema: Optional[Tensor] = None):
        if center_ema is not None or offset_ema is not None:
            raise NotImplementedError("EMA consistency branch is dormant in the reference (no caller sets *_ema)")
        center_pred = offset_pred = thing_mask_pred = None
        if "target" in feats:           # (a rank of a batch-sharded step may hold no target item)
            center_pred, offset_pred, thing_mask_pred = self.inference(feats["target"][0], instance_head)
        sc = so = sm = None
        if "source" in feats:
            sc, so, _, _, sm, _ = self.training(feats["source"][0], center["source"], offset["source"], center_weights["source"],
                                                offset_weights["source"], thing_mask["source"], instance_head)
        center_losses, offset_losses = {"source": sc, "ema": None}, {"source": so, "ema": None}
        if instance_head.use_thing_mask:
            return center_losses, offset_losses, center_pred, offset_pred, {"source": sm}, thing_mask_pred
        return center_losses, offset_losses, center_pred, offset_pred

    def panoptic_fusion(self, semantic: Tensor, center: Tensor, offset: Tensor, return_center: bool = False,
                        threshold_center: float = None, thing_mask: Optional[Tensor] = None, do_merge_semantic_and_instance=True):
        """algos/instance_seg.py:207-258 on the device, no host sync (label_divisor 1000, void 255,
        threshold 0.1, 7x7 NMS, top-200)."""
        if thing_mask is not None:
            raise NotImplementedError("an explicit foreground mask is never passed by the reference's callers")
        threshold = .1 if threshold_center is None else threshold_center
        pan, inst, centers, state = F.panoptic_fusion(semantic, center, offset, self.thing_list, 1000, 255, threshold, 7, 200)
        if not do_merge_semantic_and_instance:
            pan = None
        if return_center:
            counts = state[:, 1].tolist()          # explicit request for python lists -> one sync
            return pan, inst, [centers[i, :k].long().unsqueeze(0) for i, k in enumerate(counts)]
        return pan, inst
