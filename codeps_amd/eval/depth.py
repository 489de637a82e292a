"""DepthEvaluator: drop-in for eval/depth.py:7-131 on the device (kernel K19, codeps_depth_metrics)."""
from typing import Dict, Tuple

import torch
from torch import Tensor

from .._lib import check, lib, ptr, require_cuda, stream_ptr

_KEYS = ("d_a1", "d_a2", "d_a3", "d_rmse", "d_rmse_log", "d_abs_rel", "d_sq_rel")


def _f32(t):
    return (t if t.dtype == torch.float32 else t.float()).contiguous()


class DepthEvaluator:
    def __init__(self, use_gt_scale: bool, depth_ranges: Tuple[float, float], use_garg_crop: bool = False):
        self.use_gt_scale = use_gt_scale
        self.depth_ranges = depth_ranges
        self.use_garg_crop = use_garg_crop

    def _run(self, gt, pred, class_map, class_id, batch, h, w, garg):
        stats = torch.empty(batch, 8, dtype=torch.float32, device=gt.device)
        nbytes = lib().codeps_depth_metrics_workspace_bytes(batch)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=gt.device)
        check(lib().codeps_depth_metrics(ptr(gt), ptr(pred), ptr(class_map), int(class_id), batch, h, w, int(bool(self.use_gt_scale)),
                                         float(self.depth_ranges[0]), float(self.depth_ranges[1]), int(bool(garg)), ptr(stats), ptr(ws),
                                         nbytes, stream_ptr()), "codeps_depth_metrics")
        return stats

    def compute_depth_metrics(self, depth_gt: Tensor, depth_pred: Tensor) -> Dict[str, Tensor]:
        """eval/depth.py:21-68: per-sample median scaling, clamping and the seven statistics, averaged over the batch.
        depth_gt [B,H,W] or [B,1,H,W], depth_pred [B,1,H,W] -> dict of 0-dim device tensors.  No host sync."""
        require_cuda(depth_gt, depth_pred)
        b, h, w = depth_gt.shape[0], depth_gt.shape[-2], depth_gt.shape[-1]
        stats = self._run(_f32(depth_gt), _f32(depth_pred.detach()), None, 0, b, h, w, self.use_garg_crop)
        mean = stats[:, :7].sum(0) / b                       # "for value in depth_stats.values(): value /= batch_size"
        return {k: mean[i] for i, k in enumerate(_KEYS)}

    def compute_depth_metrics_per_class(self, depth_gt: Tensor, depth_pred: Tensor, semantic_gt: Tensor) -> Dict[str, Tensor]:
        """eval/depth.py:70-108: the statistics of each semantic class present in the batch (255 skipped), pixels of the
        whole batch pooled per class.  The class list is read back once (the reference's torch.unique does the same)."""
        require_cuda(depth_gt, depth_pred, semantic_gt)
        gt, pred = _f32(depth_gt), _f32(depth_pred.detach())
        sem = (semantic_gt if semantic_gt.dtype == torch.uint8 else semantic_gt.to(torch.uint8)).contiguous()
        n_rows, w = gt.numel() // gt.shape[-1], gt.shape[-1]
        out = {}
        present = torch.bincount(sem.reshape(-1).long(), minlength=256).cpu()
        for c in torch.nonzero(present).flatten().tolist():
            if c == 255:
                continue
            stats = self._run(gt, pred, sem, c, 1, n_rows, w, False)[0]
            if float(stats[7]) == 0:                            # "if not mask.any(): continue"
                continue
            for i, k in enumerate(_KEYS):
                out[f"{k}_c{c}"] = stats[i]
        return out
