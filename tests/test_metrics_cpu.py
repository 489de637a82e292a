"""oracle/metrics.py against tests/golden/metrics.npz (the reference's DepthEvaluator / SemanticEvaluator /
PanopticEvaluator run in the development container) -- pins the oracle for SURVEY.md section 8f-4."""
import numpy as np

import inputs as gi
from helpers import load
from oracle import metrics as om


def depth_inputs():
    b, h, w = 2, 48, 96
    import torch
    gt = gi.depth_map(b, h, w, 81)
    gt = torch.where(gi.uniform((b, 1, h, w), 82) < 0.3, torch.zeros_like(gt), gt)
    pred = (gi.depth_map(b, h, w, 81) * (0.4 + 0.2 * gi.uniform((b, 1, h, w), 83)) + 0.05).contiguous()
    return gt.numpy(), pred.numpy(), gi.semantic_labels(b, h, w, 84).numpy()


def test_depth_metrics():
    z, meta = load("metrics")
    gt, pred, sem = depth_inputs()
    for name, c in meta["depth"].items():
        got = om.depth_metrics(gt, pred, c["use_gt_scale"], [0.1, 80], c["garg"])
        np.testing.assert_allclose([got[k] for k in om.DEPTH_KEYS], z[f"depth_{name}"], rtol=2e-6, err_msg=name)
    pc = om.depth_metrics_per_class(gt[:, 0], pred, sem, True, [0.1, 80])
    assert sorted(pc) == meta["per_class_keys"]
    np.testing.assert_allclose([pc[k] for k in sorted(pc)], z["depth_per_class"], rtol=2e-6)


def test_confusion_matrix_and_miou():
    z, _ = load("metrics")
    _, _, sem = depth_inputs()
    pred = gi.semantic_labels(2, 48, 96, 85, void_frac=0.0).numpy()
    conf = om.confusion_matrix(pred, sem, 14, [3])
    assert np.array_equal(conf, z["conf"]) and conf[3].sum() == 0
    assert np.array_equal(om.confusion_matrix(pred, sem, 14), z["conf_noignore"])
    np.testing.assert_allclose(om.sem_miou(conf, 14, [3]), z["miou"], rtol=1e-6)


def test_panoptic_quality():
    z, meta = load("metrics")
    gt, pred = (t.numpy() for t in gi.panoptic_maps(2, 64, 128, 86, shift=1))
    stats = om.panoptic_stats(gt, pred)
    for c, v in om.panoptic_stats(pred, gt).items():
        for i in range(4):
            stats[c][i] += v[i]
    cats = z["pq_cats"].tolist()
    assert sorted(stats) == cats
    assert np.array_equal(np.array([stats[c][1:] for c in cats]), z["pq_counts"])
    np.testing.assert_allclose([stats[c][0] for c in cats], z["pq_iou"], rtol=1e-12)
    np.testing.assert_allclose(om.pq_average(stats, meta["stuff_list"] + meta["thing_list"]), z["pq_all"], rtol=1e-12)
    np.testing.assert_allclose(om.pq_average(stats, meta["thing_list"]), z["pq_things"], rtol=1e-12)
