"""SemanticEvaluator: drop-in for eval/semantic.py:8-65; the confusion matrix is kernel K20 (codeps_confusion_matrix)."""
from typing import List, Optional

import numpy as np
import torch
from torch import Tensor, distributed

from .._lib import CodepsHipError, check, lib, ptr, require_cuda, stream_ptr


class SemanticEvaluator:
    def __init__(self, num_classes: int, ignore_classes: Optional[List[int]] = None, ignore_index: int = 255):
        self.num_classes = num_classes
        self.ignore_index = ignore_index
        self.ignore_classes = ignore_classes if ignore_classes is not None else []
        self._bad: List[Tensor] = []          # out-of-range flags of earlier calls (device), read by check()

    def compute_sem_miou(self, sem_conf_mat, sum_pixels: bool = False):
        """eval/semantic.py:21-43 (a num_classes^2 matrix: plain tensor arithmetic)."""
        keep = np.array([i for i in range(self.num_classes) if i not in self.ignore_classes])
        m = sem_conf_mat[keep, :][:, keep]
        inter = m.diag()
        union = m.sum(dim=1) + m.sum(dim=0) - m.diag() + 1e-8
        return inter.sum() / union.sum() if sum_pixels else inter / union

    def filter_sem_conf_mat(self, sem_conf_mat, device, debug):
        """eval/semantic.py:45-52."""
        sem_conf_mat = sem_conf_mat.to(device=device)
        if not debug:
            distributed.all_reduce(sem_conf_mat, distributed.ReduceOp.SUM)
        return sem_conf_mat.cpu()[:self.num_classes, :]

    def compute_confusion_matrix(self, semantic_pred: Tensor, semantic_gt: Tensor) -> Tensor:
        """eval/semantic.py:54-65: int32 [C,C], rows = target, columns = prediction, over all valid pixels of the batch."""
        require_cuda(semantic_pred, semantic_gt)
        pred = (semantic_pred if semantic_pred.dtype == torch.uint8 else semantic_pred.to(torch.uint8)).contiguous()
        gt = (semantic_gt if semantic_gt.dtype == torch.uint8 else semantic_gt.to(torch.uint8)).contiguous()
        assert pred.numel() == gt.numel(), "prediction and target differ in size"
        conf = torch.zeros(self.num_classes * self.num_classes, dtype=torch.int32, device=pred.device)
        bad = torch.zeros(1, dtype=torch.int32, device=pred.device)
        mask = 0
        for c in self.ignore_classes:
            mask |= 1 << int(c)
        check(lib().codeps_confusion_matrix(ptr(pred), ptr(gt), pred.numel(), self.num_classes, self.ignore_index, mask, ptr(conf), ptr(bad),
                                            stream_ptr()), "codeps_confusion_matrix")
        self._bad = self._bad[-15:] + [bad]
        return conf.view(self.num_classes, self.num_classes)

    def check(self):
        """Raise if a recent call met a valid pixel whose class is >= num_classes (the reference's index_add_ would have
        written outside the pixel's row).  Synchronises; call it where a sync is acceptable (logging, end of an epoch)."""
        flags, self._bad = self._bad, []
        if flags and int(torch.stack(flags).sum()) != 0:
            raise CodepsHipError("codeps_confusion_matrix: a valid pixel holds a class >= num_classes")
