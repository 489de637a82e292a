"""oracle/mixup.py against tests/golden/mixup.npz (the REAL reference's Mixup.do_mixup / warp_c2c / get_off_cen run in
the development container) -- pins the oracle for SURVEY.md section 8f-2."""
import numpy as np
import pytest

import inputs as gi
from helpers import load
from oracle import mixup as om


def case_inputs(c):
    item = gi.mixup_item(c["b"], c["hs"], c["ws"], c["ht"], c["wt"], c["seed"])
    item = {k: (v[0].numpy() if isinstance(v, dict) else v.numpy()) for k, v in item.items()}
    np.random.seed(c["seed"])                                 # the reference's draw: one np.random.choice per batch item
    ids = [np.random.choice(8, 2, replace=False) for _ in range(c["b"])]
    return item, ids


@pytest.mark.parametrize("name", ["geom", "plain", "geom_same"])
def test_do_cut_mixup_matches_reference(name):
    z, meta = load("mixup")
    c = meta["cases"][name]
    item, ids = case_inputs(c)
    out = om.do_cut_mixup(item, meta["thing_list"], ids, c["geom"])
    for k in ("rgb", "semantic", "instance", "center", "offset"):
        assert out[k].dtype == z[f"{name}_{k}"].dtype and np.array_equal(out[k], z[f"{name}_{k}"]), f"{name}: {k} differs"


@pytest.mark.parametrize("name", ["geom", "geom_same"])
def test_warp_c2c_matches_reference(name):
    z, meta = load("mixup")
    c = meta["cases"][name]
    item, _ = case_inputs(c)
    cs, ct, hw = item["camera_model_src"], item["camera_model_tgt"], (c["hs"], c["ws"])
    for key, src, mode, pad in (("warp_rgb", item["rgb_tgt"], "bilinear", "zeros"), ("warp_rgb_border", item["rgb_tgt"], "bilinear", "border"),
                                ("warp_inst", z[f"{name}_inst_tgt"], "nearest", "zeros"),
                                ("warp_sem", item["semantic_pred_tgt_ema"], "nearest", "border")):
        got, ref = om.warp_c2c(ct, cs, src, hw, mode, pad), z[f"{name}_{key}"]
        assert got.dtype == np.float64 and got.shape == ref.shape
        if mode == "nearest":
            assert np.array_equal(got, ref), key
        else:
            # float64 sums: ATen's vectorised kernel contracts the four-term sum into FMAs -> <= 1 ulp(float64); the
            # reference's own cast to float32 (mixup.py:328) removes it
            assert np.abs(got - ref).max() <= 1e-15 * max(1.0, np.abs(ref).max()), key
            assert np.array_equal(got.astype(np.float32), ref.astype(np.float32)), key
        # the correctly rounded square root (what GPUs compute) moves a sampling coordinate by at most one float32 ulp of
        # the viewing ray: label maps keep their values, images move by ~1e-6
        ieee = om.warp_c2c(ct, cs, src, hw, mode, pad, ieee_sqrt=True)
        if mode == "nearest":
            assert (ieee != ref).mean() < 1e-3, key
        else:
            assert np.abs(ieee - ref).max() < 2e-5, key


def test_offset_center_matches_reference():
    z, _ = load("mixup")
    center, offset = om.get_off_cen(z["offcen_inst"])
    assert np.array_equal(center, z["offcen_center"]) and np.array_equal(offset, z["offcen_offset"])
    assert center.max() == 1.0 and (z["offcen_offset"][0, :, 0, 0] != 0).any()


def test_cut_rects_keep_last_row_and_column():
    r = om.cut_rects([0, 7], 48, 96, 2, 4)
    assert r.tolist() == [[0, 23, 0, 23], [24, 47, 72, 95]]


@pytest.mark.parametrize("name", ["cls_geom", "cls_plain"])
def test_do_class_mixup_matches_reference(name):
    z, meta = load("mixup")
    c = meta["cases"][name]
    item = gi.mixup_item(c["b"], c["hs"], c["ws"], c["ht"], c["wt"], c["seed"])
    item = {k: (v[0].numpy() if isinstance(v, dict) else v.numpy()) for k, v in item.items()}
    n = np.unique(item["semantic_src"]).size
    np.random.seed(c["seed"])                                 # the reference's single draw (mixup.py:72-75)
    pick = np.random.choice(n, int((n + n % 2) / 2), replace=False)
    out = om.do_class_mixup(item, meta["thing_list"], pick, c["geom"])
    for k in ("rgb", "semantic", "instance", "center", "offset"):
        assert out[k].dtype == z[f"{name}_{k}"].dtype and np.array_equal(out[k], z[f"{name}_{k}"]), f"{name}: {k} differs"


@pytest.mark.parametrize("name", ["geom", "plain", "big"])
def test_do_conf_instance_mixup_matches_reference(name):
    """Mixup.do_mixup("conf_instance_mixup") of the real reference (run with the oracle's restatement of kornia's distance
    transform in kornia's place, tests/golden/_ref_import.py) against the oracle's mask formulation of it."""
    z, meta = load("conf_mixup")
    c = meta["cases"][name]
    item = gi.mixup_item(1, c["hs"], c["ws"], c["ht"], c["wt"], c["seed"])
    item = {k: (v[0].numpy() if isinstance(v, dict) else v.numpy()) for k, v in item.items()}
    out = om.do_conf_instance_mixup(item, meta["thing_list"], c["geom"], c["conf_thresh"], c["min_inst_size"])
    pasted = out["instance"] != item["instance_src"][:, 0]
    assert pasted.sum() > 500, "the case must paste instances"
    for k in ("rgb", "semantic", "instance", "center", "offset"):
        assert out[k].dtype == z[f"{name}_{k}"].dtype and np.array_equal(out[k], z[f"{name}_{k}"]), f"{name}: {k} differs"


def test_kornia_distance_transform_restatement_properties():
    """No kornia here (parity unpinned): what the restatement must satisfy by construction -- zero on the objects, the
    Chebyshev ring index times 3 // 2 plus a term in (0, 1.42) elsewhere, all zeros without an object."""
    m = np.zeros((20, 31), np.float32)
    m[4:7, 5:9] = 1
    m[15, 28] = 1
    d = om.kornia_distance_transform(m)
    ys, xs = np.nonzero(m)
    yy, xx = np.mgrid[:20, :31]
    cheb = np.min(np.maximum(np.abs(yy[..., None] - ys), np.abs(xx[..., None] - xs)), -1)
    assert (d[m > 0] == 0).all()
    ring = (cheb - 1) * 3 // 2
    assert ((d - ring)[m == 0] > 0).all() and ((d - ring)[m == 0] < 1.42).all()
    assert (om.kornia_distance_transform(np.zeros((6, 9), np.float32)) == 0).all()


def test_restated_kornia_distance_transform_against_scipy():
    """oracle.mixup.kornia_distance_transform restates kornia.contrib.distance_transform (kornia is absent: parity unpinned).
    scipy IS installed and pins the structural claim the closed form of the device path relies on (csrc/mixup.hip:
    codeps_conf_place): a background pixel is reached in iteration D - 1, D its CHEBYSHEV distance to the nearest object pixel,
    and takes the value (D - 1) * 3 // 2 (kornia's `i * kernel_size // 2`: 0, 1, 3, 4, 6, ... -- not a Euclidean distance) plus a
    soft-min correction in (0, 1.5]."""
    import numpy as np
    from scipy import ndimage
    from oracle import mixup as M
    rng = np.random.default_rng(3)
    for case in range(4):
        img = np.zeros((48, 80), np.float32)
        for _ in range(3 + case):
            y, x = rng.integers(2, 44), rng.integers(2, 76)
            img[y:y + rng.integers(1, 6), x:x + rng.integers(1, 8)] = 1.0
        got = M.kornia_distance_transform(img)
        cheb = ndimage.distance_transform_cdt(img == 0, metric="chessboard")
        bg = img == 0
        assert np.all(got[~bg] == 0)
        base = (cheb[bg] - 1) * 3 // 2
        corr = got[bg] - base
        assert corr.min() > 0 and corr.max() <= 1.5 + 1e-5, (corr.min(), corr.max())
