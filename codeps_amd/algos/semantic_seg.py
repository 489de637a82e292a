"""SemanticLoss / SemanticSegAlgo: drop-in for algos/semantic_seg.py:15-67,150-249
(SemanticConsistencyLoss is dead code in the reference, SURVEY.md §2 #10, and is not mirrored)."""
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor

from .. import functional as F


class SemanticLoss:
    """Hard-pixel-mined CE applied to softmax probabilities (algos/semantic_seg.py:15-67)."""

    def __init__(self, device=None, ignore_index: int = 255, ignore_labels: Optional[List] = None,
                 top_k_percent_pixels: float = 1.0, class_weights: Optional[Tuple[float, ...]] = None):
        if not 0. < top_k_percent_pixels <= 1.0:
            raise ValueError('top_k_percent_pixels must be within (0, 1]')
        self.device, self.ignore_labels = device, ignore_labels
        self.top_k_percent_pixels, self.ignore_index = top_k_percent_pixels, ignore_index
        self.class_weights = None
        if class_weights is not None:
            kept = [w for label, w in enumerate(class_weights) if not ignore_labels or label not in ignore_labels]
            self.class_weights = torch.tensor(kept, dtype=torch.float32)
        self._cw_dev = None

    def __call__(self, prediction_softmax: Tensor, target: Tensor, pixel_weights: Optional[Tensor], return_per_pixel=False):
        if return_per_pixel:
            raise NotImplementedError("return_per_pixel is only used by the reference's dead SemanticConsistencyLoss")
        if self.ignore_labels:
            # the reference remaps ignore_labels to 255 IN PLACE on the caller's target and slices those channels away
            # (algos/semantic_seg.py:50-55); the kernel does the slicing through a channel mask
            for ignore_label in self.ignore_labels:
                target.masked_fill_(target == ignore_label, self.ignore_index)
        probs = F.from_nchw(prediction_softmax)
        c = prediction_softmax.shape[1]
        cw = None
        if self.class_weights is not None:
            if self._cw_dev is None or self._cw_dev.device != probs.device:
                self._cw_dev = self.class_weights.to(probs.device)
            cw = self._cw_dev
        return F.semantic_loss(probs, c, target, pixel_weights, cw, self.top_k_percent_pixels, self.ignore_index,
                               self.ignore_labels or None)


class SemanticSegAlgo:
    """algos/semantic_seg.py:150-249."""

    def __init__(self, semantic_loss: SemanticLoss, evaluator=None, semantic_consistency_loss=None):
        self.semantic_loss, self.evaluator = semantic_loss, evaluator
        self.semantic_consistency_loss = semantic_consistency_loss

    def _confusion(self, pred, gt):
        if self.evaluator is None or gt is None:
            return None
        return self.evaluator.compute_confusion_matrix(pred, gt)

    def training(self, feats, semantic_head, semantic_gt: Tensor, semantic_weights: Tensor, ignore_classes: Optional[List] = None,
                 semantic_gt_eval: Optional[Tensor] = None):
        semantic_logits = semantic_head(feats)
        if ignore_classes:
            masked = semantic_logits.detach().clone()
            masked[:, ignore_classes] = -float('inf')
            semantic_pred = F.channel_argmax(masked, torch.uint8)
        else:
            semantic_pred = F.channel_argmax(semantic_logits.detach(), torch.uint8)
        confusion_matrix = self._confusion(semantic_pred, semantic_gt_eval if semantic_gt_eval is not None else semantic_gt)
        semantic_loss = self.semantic_loss(semantic_logits, semantic_gt, semantic_weights)
        return semantic_loss, confusion_matrix, semantic_pred

    def inference(self, feats, semantic_head) -> Tuple[Tensor, Tensor]:
        semantic_logits = semantic_head(feats)
        return F.channel_argmax(semantic_logits, torch.uint8), semantic_logits

    def evaluation(self, feats, semantic_head, semantic_gt: Tensor):
        semantic_pred, semantic_logits = self.inference(feats, semantic_head)
        return self._confusion(semantic_pred, semantic_gt), semantic_pred, semantic_logits

    def adaptation_target(self, feats, semantic_head):
        """The target frame's prediction (algos/semantic_seg.py:208-213); None when this rank holds no target item."""
        if 'target' not in feats:       # (a rank of a batch-sharded step may hold no target item: nothing to predict there)
            return None
        with torch.no_grad():           # the target prediction only feeds argmax / panoptic fusion
            target_probs = semantic_head(feats['target'][0])
        return F.channel_argmax(target_probs)

    def adaptation_supervised(self, feats, semantic_head, semantic_gt, semantic_weights, batched=None):
        """The supervised losses of the source and the mix-up keys (algos/semantic_seg.py:215-234)."""
        losses = {'source': None}
        sup = [k for k in feats if k.endswith('mixup') or k == 'source']
        probs = {}
        for feats_b, slices in (batched or []):
            mine = sorted((slices[k], k) for k in sup if k in slices)
            if len(mine) < 2 or not hasattr(semantic_head, "forward_nhwc"):
                continue
            lo, hi = mine[0][0][0], mine[-1][0][1]
            if sum(b1 - b0 for (b0, b1), _ in mine) != hi - lo:
                continue
            base = semantic_head.forward_nhwc([f[lo:hi] for f in feats_b])
            parts = F.split_batch(base, [(b0 - lo, b1 - lo) for (b0, b1), _ in mine])
            for ((b0, b1), k), part in zip(mine, parts):
                probs[k] = F.to_nchw(part, semantic_head.num_classes)
        for key in sup:
            p = probs[key] if key in probs else semantic_head(feats[key][0])
            if key == 'source':
                losses['source'] = self.semantic_loss(p, semantic_gt['source'], semantic_weights['source'])
            else:
                losses[key] = self.semantic_loss(p, semantic_gt[key], None)          # weights = ones
        return losses

    def adaptation(self, feats: Dict[str, list], semantic_head, semantic_gt: Dict[str, Tensor],
                   semantic_weights: Dict[str, Tensor], batched=None, target_pred=None):
        """algos/semantic_seg.py:200-249.  `batched` (optional, from CodepsNet.adapt): the semantic head has no
        batch-dependent op, so all supervised keys of one resolution (source + *mixup) run through it in ONE call;
        the hard-pixel mining stays per key (separate top-k over each key's pixels, like the reference)."""
        # target_pred: the target prediction when the caller has already computed it (CodepsNet.adapt runs it on a side stream)
        semantic_pred = self.adaptation_target(feats, semantic_head) if target_pred is None else target_pred
        losses = self.adaptation_supervised(feats, semantic_head, semantic_gt, semantic_weights, batched)
        confusion_matrix = None if semantic_pred is None else self._confusion(semantic_pred, semantic_gt.get('target'))
        return losses, confusion_matrix, semantic_pred
