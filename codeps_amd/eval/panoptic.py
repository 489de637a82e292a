"""PanopticEvaluator: drop-in for eval/panoptic.py:10-115 with the per-image matching on the device (kernel K21,
codeps_panoptic_stats).  `update` only enqueues work; the per-category (iou, tp, fp, fn) accumulators live on the device
and are read back when `pq_stats` / `evaluate()` are asked for."""
from collections import defaultdict
from typing import Dict, List

import torch
from torch import Tensor

from .._lib import CodepsHipError, check, lib, ptr, require_cuda, stream_ptr


class PQStatCat:
    def __init__(self):
        self.iou, self.tp, self.fp, self.fn = 0.0, 0, 0, 0

    def __iadd__(self, other):
        self.iou += other.iou
        self.tp += other.tp
        self.fp += other.fp
        self.fn += other.fn
        return self


class PQStat:
    def __init__(self):
        self.pq_per_cat = defaultdict(PQStatCat)

    def __getitem__(self, index: int):
        return self.pq_per_cat[index]

    def __iadd__(self, pq_stat):
        for label, cat in pq_stat.pq_per_cat.items():
            self.pq_per_cat[label] += cat
        return self

    def pq_average(self, categories: List[int]):
        """eval/panoptic.py:87-115."""
        pq, sq, rq, n = 0, 0, 0, 0
        per_class_results = {}
        for label in categories:
            c = self.pq_per_cat[label]
            iou, tp, fp, fn = c.iou, c.tp, c.fp, c.fn
            if tp + fp + fn == 0:
                per_class_results[label] = {"pq": 0.0, "sq": 0.0, "rq": 0.0}
                continue
            n += 1
            pq_class = iou / (tp + 0.5 * fp + 0.5 * fn)
            sq_class = iou / tp if tp != 0 else 0
            rq_class = tp / (tp + 0.5 * fp + 0.5 * fn)
            per_class_results[label] = {"pq": pq_class, "sq": sq_class, "rq": rq_class}
            pq += pq_class
            sq += sq_class
            rq += rq_class
        if n == 0:
            return {"pq": 0, "sq": 0, "rq": 0, "n": n}, per_class_results
        return {"pq": pq / n, "sq": sq / n, "rq": rq / n, "n": n}, per_class_results


class PanopticEvaluator:
    def __init__(self, stuff_list: List[int], thing_list: List[int], label_divisor: int = 1000, void_label: int = -1):
        self.stuff_list = stuff_list
        self.thing_list = thing_list
        self.label_divisor = label_divisor
        self.void_label = void_label
        self.num_categories = max(list(stuff_list) + list(thing_list) + [0]) + 1
        self._base = PQStat()                     # results merged in from elsewhere
        self._iou = self._counts = None
        self._status: List[Tensor] = []

    def update(self, pan_img_gt: Tensor, pan_img_pred: Tensor):
        """eval/panoptic.py:42-47.  pan_img_* [B,H,W] integer maps (category * label_divisor + instance, void_label)."""
        require_cuda(pan_img_gt, pan_img_pred)
        gt = (pan_img_gt if pan_img_gt.dtype == torch.int32 else pan_img_gt.to(torch.int32)).contiguous()
        pred = (pan_img_pred if pan_img_pred.dtype == torch.int32 else pan_img_pred.to(torch.int32)).contiguous()
        assert gt.shape == pred.shape and gt.dim() == 3
        dev = gt.device
        if self._iou is None:
            self._iou = torch.zeros(self.num_categories, dtype=torch.float64, device=dev)
            self._counts = torch.zeros(self.num_categories, 3, dtype=torch.int32, device=dev)
        b, h, w = gt.shape
        status = torch.empty(b, dtype=torch.int32, device=dev)
        nbytes = lib().codeps_panoptic_stats_workspace_bytes(b)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        check(lib().codeps_panoptic_stats(ptr(gt), ptr(pred), b, h, w, self.label_divisor, self.void_label, self.num_categories,
                                          ptr(self._iou), ptr(self._counts), ptr(status), ptr(ws), nbytes, stream_ptr()),
              "codeps_panoptic_stats")
        self._status.append(status)

    @property
    def pq_stats(self) -> PQStat:
        """Accumulated statistics as the reference's PQStat (reads the device accumulators back: one sync)."""
        out = PQStat()
        out += self._base
        if self._iou is not None:
            for status in self._status:
                s = status.cpu()
                if bool((s < 0).any()):
                    raise CodepsHipError(f"codeps_panoptic_stats: status {s.tolist()} (-1 table overflow, -2 negative id, "
                                         "-3 category out of range)")
            self._status = []
            iou, counts = self._iou.cpu().tolist(), self._counts.cpu().tolist()
            for cat, (v, (tp, fp, fn)) in enumerate(zip(iou, counts)):
                if tp or fp or fn:
                    c = out[cat]
                    c.iou += v
                    c.tp += tp
                    c.fp += fp
                    c.fn += fn
        return out

    @pq_stats.setter
    def pq_stats(self, value: PQStat):
        self._base = value
        self._iou = self._counts = None
        self._status = []

    def evaluate(self) -> Dict[str, dict]:
        stats = self.pq_stats
        results = {"Things": stats.pq_average(self.thing_list)[0], "Stuff": stats.pq_average(self.stuff_list)[0]}
        results["All"], results["per_class"] = stats.pq_average(self.stuff_list + self.thing_list)
        return results

    def reset(self):
        self.pq_stats = PQStat()
