"""HIP evaluation-metric kernels (codeps_depth_metrics / codeps_confusion_matrix / codeps_panoptic_stats through the C ABI and
the codeps_amd.eval mirrors) against the oracle and the reference's golden vectors."""
import numpy as np
import pytest
import torch

import inputs as gi
from helpers import load
from oracle import metrics as om
from test_metrics_cpu import depth_inputs

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_depth_metrics_match_reference():
    from codeps_amd.eval import DepthEvaluator
    z, meta = load("metrics")
    gt, pred, sem = depth_inputs()
    for name, c in meta["depth"].items():
        ev = DepthEvaluator(c["use_gt_scale"], [0.1, 80], c["garg"])
        out = ev.compute_depth_metrics(dev(gt), dev(pred))
        assert list(out) == list(om.DEPTH_KEYS) and all(v.dim() == 0 and v.is_cuda for v in out.values())
        # tolerance: fp32 sums in a different order + device logf (the thresholds a1..a3 are counts: exact)
        np.testing.assert_allclose([float(out[k]) for k in om.DEPTH_KEYS], z[f"depth_{name}"], rtol=5e-6, err_msg=name)
        np.testing.assert_allclose([float(out[k]) for k in om.DEPTH_KEYS[:3]], z[f"depth_{name}"][:3], rtol=2e-7)
    ev = DepthEvaluator(True, [0.1, 80])
    pc = ev.compute_depth_metrics_per_class(dev(gt[:, 0]), dev(pred), dev(sem))
    assert sorted(pc) == meta["per_class_keys"]
    np.testing.assert_allclose([float(pc[k]) for k in sorted(pc)], z["depth_per_class"], rtol=5e-6)
    # [B,H,W] ground truth is accepted like the reference does (unsqueeze)
    out3 = DepthEvaluator(True, [0.1, 80]).compute_depth_metrics(dev(gt[:, 0]), dev(pred))
    np.testing.assert_allclose([float(out3[k]) for k in om.DEPTH_KEYS], z["depth_scaled"], rtol=5e-6)


def test_depth_median_is_the_lower_median_with_ties_and_negatives():
    """Radix select == torch.median on adversarial inputs: even counts, heavy ties, negative / tiny predictions."""
    from codeps_amd._lib import lib, ptr, stream_ptr, check
    g = np.random.default_rng(3)
    for n_valid in (1, 2, 7, 1000, 4096):
        h, w = 64, 80
        gt = np.zeros((1, h, w), np.float32)
        idx = g.permutation(h * w)[:n_valid]
        gt.reshape(-1)[idx] = g.choice(np.array([0.5, 1.0, 1.0, 2.0, 3e-3, 77.0], np.float32), n_valid)
        pred = g.choice(np.array([-2.0, -0.0, 0.0, 1e-30, 0.25, 0.25, 4.0, 1e20], np.float32), (1, h, w)).astype(np.float32)
        with np.errstate(all="ignore"):
            want = om.lower_median(gt[gt > 0]) / om.lower_median(pred[gt > 0])
        gd, pd = dev(gt), dev(pred)
        # recover the ratio from the kernel: with clamping disabled by a huge range, d_abs_rel of (gt, ratio * pred) is known
        stats = torch.empty(1, 8, device=DEV)
        nb = lib().codeps_depth_metrics_workspace_bytes(1)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        check(lib().codeps_depth_metrics(ptr(gd), ptr(pd), None, 0, 1, h, w, 1, -3e38, 3e38, 0, ptr(stats), ptr(ws), nb, stream_ptr()), "dm")
        assert int(stats[0, 7]) == n_valid
        if np.isfinite(want) and want != 0:
            m = gt > 0
            with np.errstate(all="ignore"):
                ref = np.mean(np.abs(gt[m] - pred[m] * np.float32(want)) / gt[m], dtype=np.float64)
            if np.isfinite(ref):
                np.testing.assert_allclose(float(stats[0, 5]), ref, rtol=1e-4)


def test_depth_metrics_fullsize_and_empty_sample():
    from codeps_amd.eval import DepthEvaluator
    b, h, w = 2, 512, 1024
    gt = gi.depth_map(b, h, w, 91).numpy()
    gt[gi.uniform((b, 1, h, w), 92).numpy() < 0.5] = 0
    pred = (gi.depth_map(b, h, w, 93) * 0.3 + 0.01).numpy()
    out = DepthEvaluator(True, [0.1, 80]).compute_depth_metrics(dev(gt), dev(pred))
    want = om.depth_metrics(gt, pred, True, [0.1, 80])
    np.testing.assert_allclose([float(out[k]) for k in om.DEPTH_KEYS], [want[k] for k in om.DEPTH_KEYS], rtol=2e-5)
    gt[1] = 0                                                            # a sample without ground truth: NaN like the reference
    out = DepthEvaluator(True, [0.1, 80]).compute_depth_metrics(dev(gt), dev(pred))
    assert all(torch.isnan(v) for v in out.values())


def test_confusion_matrix():
    from codeps_amd._lib import CodepsHipError
    from codeps_amd.eval import SemanticEvaluator
    z, _ = load("metrics")
    _, _, sem = depth_inputs()
    pred = gi.semantic_labels(2, 48, 96, 85, void_frac=0.0).numpy()
    ev = SemanticEvaluator(14, ignore_classes=[3])
    conf = ev.compute_confusion_matrix(dev(pred), dev(sem))
    assert conf.dtype == torch.int32 and np.array_equal(conf.cpu().numpy(), z["conf"])
    assert np.array_equal(SemanticEvaluator(14).compute_confusion_matrix(dev(pred).long(), dev(sem).long()).cpu().numpy(), z["conf_noignore"])
    np.testing.assert_allclose(ev.compute_sem_miou(conf.cpu()).numpy(), z["miou"], rtol=1e-6)
    ev.check()
    big_p, big_t = gi.semantic_labels(4, 512, 1024, 95, void_frac=0.0).numpy(), gi.semantic_labels(4, 512, 1024, 96).numpy()
    got = SemanticEvaluator(14).compute_confusion_matrix(dev(big_p), dev(big_t)).cpu().numpy()
    assert np.array_equal(got, om.confusion_matrix(big_p, big_t, 14)) and got.sum() == (big_t != 255).sum()
    small = SemanticEvaluator(5)                                         # classes 5..13 do not fit a 5 x 5 matrix
    small.compute_confusion_matrix(dev(pred), dev(sem))
    with pytest.raises(CodepsHipError):
        small.check()


def test_panoptic_quality_matches_reference():
    from codeps_amd.eval import PanopticEvaluator
    z, meta = load("metrics")
    gt, pred = gi.panoptic_maps(2, 64, 128, 86, shift=1)
    ev = PanopticEvaluator(meta["stuff_list"], meta["thing_list"], 1000, -1)
    ev.update(gt.to(DEV), pred.to(DEV))
    ev.update(pred.to(DEV).long(), gt.to(DEV).long())
    st = ev.pq_stats
    cats = z["pq_cats"].tolist()
    assert sorted(c for c in st.pq_per_cat) == cats
    assert np.array_equal(np.array([[st[c].tp, st[c].fp, st[c].fn] for c in cats]), z["pq_counts"])
    np.testing.assert_allclose([st[c].iou for c in cats], z["pq_iou"], rtol=1e-12)
    res = ev.evaluate()
    np.testing.assert_allclose([res["All"][k] for k in ("pq", "sq", "rq", "n")], z["pq_all"], rtol=1e-12)
    np.testing.assert_allclose([res["Things"][k] for k in ("pq", "sq", "rq", "n")], z["pq_things"], rtol=1e-12)
    ev.reset()
    assert ev.evaluate()["All"]["n"] == 0


def test_panoptic_quality_fullsize_and_void_rules():
    from codeps_amd._lib import CodepsHipError
    from codeps_amd.eval import PanopticEvaluator
    gt, pred = gi.panoptic_maps(2, 512, 1024, 97, block=32, shift=5)
    gt[:, :40] = -1                                                      # a void band: predicted segments mostly inside it are ignored
    ev = PanopticEvaluator(list(range(9)), [9, 10, 11, 12, 13], 1000, -1)
    ev.update(gt.to(DEV), pred.to(DEV))
    want = om.panoptic_stats(gt.numpy(), pred.numpy())
    st = ev.pq_stats
    assert sorted(st.pq_per_cat) == sorted(want)
    for c, (iou, tp, fp, fn) in want.items():
        assert (st[c].tp, st[c].fp, st[c].fn) == (tp, fp, fn), c
        assert abs(st[c].iou - iou) <= 1e-12 * max(1.0, iou)
    bad = PanopticEvaluator(list(range(9)), [9, 10], 1000, -1)           # categories 11..13 exceed the evaluator's list
    bad.update(gt.to(DEV), pred.to(DEV))
    with pytest.raises(CodepsHipError):
        bad.pq_stats


def test_eval_forward_with_device_evaluators():
    """CodepsNet "eval" mode with ground truth (codeps/online_adap.py:142-243) fills depth statistics and the confusion
    matrix through the device evaluators that gen_models(evaluators="device") builds like codeps/model_setup.py:82,110,124."""
    from codeps_amd import synthetic
    from codeps_amd.codeps import gen_models
    from codeps_amd.config import get_cfg_shipped
    cfg = get_cfg_shipped()
    cfg.model.po_depth_net.params.nof_layers = 18
    cfg.dataset.feed_img_size = [64, 128]
    cfg.eval.depth.depth_ranges = [0.1, 80]
    torch.manual_seed(3)
    net = gen_models(cfg, DEV, list(range(9)), [9, 10, 11, 12, 13], [], evaluators="device").eval()
    gen = torch.Generator(device=DEV).manual_seed(4)
    item = synthetic.triplet_item(2, 64, 128, gen, DEV, labels=True)
    item["depth"] = dev(np.where(gi.uniform((2, 64, 128), 5).numpy() < 0.4, 0, gi.depth_map(2, 64, 128, 6).numpy()[:, 0]).astype(np.float32))
    with torch.no_grad():
        _, res, stats = net(item, "eval", [0, -1, 1], True, True)
    want = om.depth_metrics(item["depth"].cpu().numpy(), res["depth"].cpu().numpy(), True, [0.1, 80])
    for k in om.DEPTH_KEYS:
        np.testing.assert_allclose(float(stats[k]), want[k], rtol=2e-5, err_msg=k)
    conf = om.confusion_matrix(res["semantic"].cpu().numpy(), item["semantic"].cpu().numpy(), 14)
    assert np.array_equal(stats["sem_conf"].cpu().numpy(), conf)
    pc = om.depth_metrics_per_class(item["depth"].cpu().numpy(), res["depth"].cpu().numpy(), item["semantic"].cpu().numpy(), True, [0.1, 80])
    assert {k for k in stats if k.startswith("d_") and "_c" in k} == set(pc)
    for k, v in pc.items():
        np.testing.assert_allclose(float(stats[k]), v, rtol=2e-5, err_msg=k)
    ev = net.instance_algo.evaluator
    ev.update(res["panoptic"], res["panoptic"])                       # a prediction scored against itself: every segment a TP
    st = ev.pq_stats
    assert sum(c.fp + c.fn for c in st.pq_per_cat.values()) == 0 and sum(c.tp for c in st.pq_per_cat.values()) > 0
    assert abs(ev.evaluate()["All"]["pq"] - 1.0) < 1e-12
