"""CPU oracle: evaluation metrics (depth statistics, confusion matrix, panoptic quality).  TEST INFRA.

numpy restatement of eval/depth.py:21-131, eval/semantic.py:21-65 and eval/panoptic.py:26-115,134-209.
PINNED by tests/golden/metrics.npz (the reference's own evaluator classes run in the development container).
"""
from collections import defaultdict

import numpy as np

DEPTH_KEYS = ("d_a1", "d_a2", "d_a3", "d_rmse", "d_rmse_log", "d_abs_rel", "d_sq_rel")


def lower_median(x):
    """torch.median: the lower of the two middle elements for an even count."""
    return np.sort(x)[(x.size - 1) // 2]


def depth_stats(gt, pred):
    """eval/depth.py:111-131 on float32 vectors -> dict of float32."""
    f = np.float32
    thresh = np.maximum(gt / pred, pred / gt)
    d = gt - pred
    return {"d_a1": (thresh < f(1.25)).astype(f).mean(dtype=f), "d_a2": (thresh < f(1.25 ** 2)).astype(f).mean(dtype=f),
            "d_a3": (thresh < f(1.25 ** 3)).astype(f).mean(dtype=f), "d_rmse": np.sqrt((d ** 2).mean(dtype=f)),
            "d_rmse_log": np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean(dtype=f)), "d_abs_rel": (np.abs(d) / gt).mean(dtype=f),
            "d_sq_rel": (d ** 2 / gt).mean(dtype=f)}


def _scale_clamp(gt, pred, use_gt_scale, lo, hi):
    if use_gt_scale:
        pred = pred * (lower_median(gt) / lower_median(pred))
    return np.clip(gt, np.float32(lo), np.float32(hi)), np.clip(pred, np.float32(lo), np.float32(hi))


def depth_metrics(depth_gt, depth_pred, use_gt_scale, depth_ranges, use_garg_crop=False):
    """eval/depth.py:21-68: depth_gt [B,1,H,W] (or [B,H,W]), depth_pred [B,1,H,W] float32 -> dict (batch mean)."""
    if depth_gt.ndim == 3:
        depth_gt = depth_gt[:, None]
    mask = depth_gt > 0
    if use_garg_crop:
        h, w = depth_gt.shape[-2:]
        crop = np.zeros_like(mask)
        crop[:, :, int(0.4080 * h):int(0.9891 * h), int(0.0354 * w):int(0.9638 * w)] = True
        mask &= crop
    out = defaultdict(np.float32)
    for b in range(depth_gt.shape[0]):
        gt, pred = _scale_clamp(depth_gt[b][mask[b]], depth_pred[b][mask[b]], use_gt_scale, *depth_ranges)
        for k, v in depth_stats(gt, pred).items():
            out[k] += v
    return {k: v / np.float32(depth_gt.shape[0]) for k, v in out.items()}


def depth_metrics_per_class(depth_gt, depth_pred, semantic_gt, use_gt_scale, depth_ranges):
    """eval/depth.py:70-108: depth_gt [B,H,W], depth_pred [B,1,H,W], semantic_gt [B,H,W]."""
    out = {}
    for c in np.unique(semantic_gt):
        if c == 255:
            continue
        sel = semantic_gt == c
        gt, pred = depth_gt[sel], depth_pred[:, 0][sel]
        m = gt > 0
        if not m.any():
            continue
        gt, pred = _scale_clamp(gt[m], pred[m], use_gt_scale, *depth_ranges)
        for k, v in depth_stats(gt, pred).items():
            out[f"{k}_c{c}"] = v
    return out


def confusion_matrix(pred, target, num_classes, ignore_classes=(), ignore_index=255):
    """eval/semantic.py:54-65 -> int32 [C,C] (rows target, columns prediction)."""
    valid = target != ignore_index
    for c in ignore_classes:
        valid &= target != c
    idx = target[valid].astype(np.int64) * num_classes + pred[valid].astype(np.int64)
    return np.bincount(idx, minlength=num_classes * num_classes).astype(np.int32).reshape(num_classes, num_classes)


def sem_miou(conf, num_classes, ignore_classes=()):
    """eval/semantic.py:21-43 (per-class IoU, float32 like a torch int matrix divided by a float tensor)."""
    keep = np.array([i for i in range(num_classes) if i not in ignore_classes])
    m = conf[keep][:, keep]
    inter = np.diag(m)
    union = (m.sum(1) + m.sum(0) - np.diag(m)).astype(np.float32) + np.float32(1e-8)
    return (inter / union).astype(np.float32)


def panoptic_stats(pan_gt, pan_pred, label_divisor=1000, void_label=-1):
    """PanopticEvaluator.update for a batch (eval/panoptic.py:26-47,134-209) -> {category: [iou, tp, fp, fn]}."""
    stats = defaultdict(lambda: [0.0, 0, 0, 0])
    for g, p in zip(pan_gt, pan_pred):
        def meta(img):
            ids, area = np.unique(img, return_counts=True)
            return {int(i): (int(i) // label_divisor, int(a)) for i, a in zip(ids, area) if i != void_label}
        gt_segms, pred_segms = meta(g), meta(p)
        pairs, cnt = np.unique(np.stack([g.ravel(), p.ravel()], 1), axis=0, return_counts=True)
        inter = {(int(a), int(b)): int(c) for (a, b), c in zip(pairs, cnt)}
        gt_matched, pred_matched = set(), set()
        for (gl, pl), n in inter.items():
            if void_label in (gl, pl) or gt_segms[gl][0] != pred_segms[pl][0]:
                continue
            union = pred_segms[pl][1] + gt_segms[gl][1] - n - inter.get((void_label, pl), 0)
            iou = n / union
            if iou > 0.5:
                stats[gt_segms[gl][0]][1] += 1
                stats[gt_segms[gl][0]][0] += iou
                gt_matched.add(gl)
                pred_matched.add(pl)
        for gl, (cat, _) in gt_segms.items():
            if gl not in gt_matched:
                stats[cat][3] += 1
        for pl, (cat, area) in pred_segms.items():
            if pl in pred_matched or inter.get((void_label, pl), 0) / area > 0.5:
                continue
            stats[cat][2] += 1
    return stats


def pq_average(stats, categories):
    """PQStat.pq_average (eval/panoptic.py:87-115) -> (pq, sq, rq, n)."""
    pq = sq = rq = n = 0
    for c in categories:
        iou, tp, fp, fn = stats[c] if c in stats else (0.0, 0, 0, 0)
        if tp + fp + fn == 0:
            continue
        n += 1
        pq += iou / (tp + 0.5 * fp + 0.5 * fn)
        sq += iou / tp if tp else 0
        rq += tp / (tp + 0.5 * fp + 0.5 * fn)
    return (pq / n, sq / n, rq / n, n) if n else (0, 0, 0, 0)
