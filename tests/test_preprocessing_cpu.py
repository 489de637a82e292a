"""oracle/preprocessing.py against Pillow ITSELF (installed here and on the GPU box): the Lanczos resize behind
`transforms.Resize(..., LANCZOS)` (datasets/dataset.py:66-67), the ImageEnhance blends and the RGB <-> HSV conversion behind
torchvision's colour-jitter functions (datasets/preprocessing.py:78-105) -- bit for bit; plus the committed fixtures
(tests/golden/preprocessing.npz, produced by tests/golden/make_golden.py with Pillow) and the host coefficient routine of the
product library."""
import ctypes

import numpy as np
import pytest

from helpers import load
from oracle import preprocessing as op

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance  # noqa: E402


def _img(h, w, seed):
    r = np.random.default_rng(seed)
    a = np.kron(r.random((h // 8 + 2, w // 8 + 2, 3)), np.ones((8, 8, 1)))[:h, :w]
    return (a * 255 + r.normal(0, 8, (h, w, 3))).clip(0, 255).astype(np.uint8)


@pytest.mark.parametrize("size", [(48, 80), (40, 72), (96, 100), (120, 200), (33, 57), (96, 160)])
def test_lanczos_resize_is_pillow_bit_for_bit(size):
    a = _img(96, 160, 1)
    ref = np.array(Image.fromarray(a).resize((size[1], size[0]), Image.LANCZOS))
    assert np.array_equal(op.lanczos_resize(a, *size), ref)


def test_product_coefficient_tables_match_the_oracle():
    from codeps_amd._lib import lib
    for n_in, n_out in ((160, 80), (96, 40), (100, 160), (1024, 512), (2048, 1024)):
        ks = lib().codeps_lanczos_coeffs(n_in, n_out, None, None)
        bounds, kk = np.zeros((n_out, 2), np.int32), np.zeros((n_out, ks), np.int32)
        assert lib().codeps_lanczos_coeffs(n_in, n_out, bounds.ctypes.data_as(ctypes.c_void_p), kk.ctypes.data_as(ctypes.c_void_p)) == ks
        ob, ok = op.resample_coeffs(n_in, n_out)
        assert ok.shape[1] == ks and np.array_equal(bounds, ob) and np.array_equal(kk, ok)


def test_colour_operations_are_pillow_bit_for_bit():
    a = _img(64, 96, 2)
    pa = Image.fromarray(a)
    assert np.array_equal(op.to_grey(a), np.array(pa.convert("L")))
    for f in (0.8, 1.2, 0.93, 1.07, 0.0, 1.0, 1.9):
        assert np.array_equal(op.adjust_brightness(a, f), np.array(ImageEnhance.Brightness(pa).enhance(f))), f
        assert np.array_equal(op.adjust_contrast(a, f), np.array(ImageEnhance.Contrast(pa).enhance(f))), f
        assert np.array_equal(op.adjust_saturation(a, f), np.array(ImageEnhance.Color(pa).enhance(f))), f
    g = np.stack(np.meshgrid(np.arange(0, 256, 5), np.arange(0, 256, 3), np.arange(0, 256, 7), indexing="ij"), -1).reshape(1, -1, 3).astype(np.uint8)
    assert np.array_equal(op.rgb_to_hsv(g), np.array(Image.fromarray(g, "RGB").convert("HSV")))
    assert np.array_equal(op.hsv_to_rgb(g), np.array(Image.fromarray(g, "HSV").convert("RGB")))
    # torchvision 0.13.1 F.adjust_hue on a PIL image, written out (functional_pil.py)
    for f in (-0.1, 0.07, 0.5, -0.5, 0.0):
        h, s, v = pa.convert("HSV").split()
        np_h = np.array(h, dtype=np.uint8)
        np_h = (np_h.astype(np.int32) + (int(f * 255) & 0xFF)).astype(np.uint8)
        ref = np.array(Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB"))
        assert np.array_equal(op.adjust_hue(a, f), ref), f


def test_committed_fixtures():
    z, meta = load("preprocessing")
    a = _img(meta["h"], meta["w"], meta["seed"])
    assert np.array_equal(op.lanczos_resize(a, *meta["size"]), z["resized"])
    ops = [tuple(o) for o in meta["ops"]]
    assert np.array_equal(op.color_jitter(z["resized"], ops), z["jittered"])
    assert np.array_equal(op.to_tensor_normalize(z["jittered"], meta["mean"], meta["std"]), z["tensor"])


def test_oracle_hsv_histogram_modes_against_pillow():
    """oracle.transfer_histogram_style("hsv" / "saturation") with Pillow doing the colour-space conversions of the expected image."""
    from PIL import Image
    a, r = _img(48, 64, 13), (_img(40, 56, 14) * 0.7 + 40).astype(np.uint8)
    a_hsv, r_hsv = np.array(Image.fromarray(a, "RGB").convert("HSV")), np.array(Image.fromarray(r, "RGB").convert("HSV"))
    assert np.array_equal(op.rgb_to_hsv(a), a_hsv)
    for mode in ("hsv", "saturation"):
        new = op.match_histograms_u8(a_hsv, r_hsv)
        if mode == "saturation":
            new[:, :, 0], new[:, :, 2] = a_hsv[:, :, 0], a_hsv[:, :, 2]
        want = np.array(Image.fromarray(new, "HSV").convert("RGB"))
        assert np.array_equal(op.transfer_histogram_style(a, r, mode), want), mode
    assert np.array_equal(op.transfer_histogram_style(a, r, "rgb"), op.match_histograms_u8(a, r)) and op.transfer_histogram_style(a, r, None) is a


def test_histogram_matching_and_label_path_properties():
    a, r = _img(48, 64, 3), (_img(40, 56, 4) * 0.6 + 60).astype(np.uint8)
    m = op.match_histograms_u8(a, r)
    assert m.dtype == np.uint8 and m.shape == a.shape
    for c in range(3):                                         # monotone per channel, values inside the reference's range
        lut = {}
        for s_, d_ in zip(a[..., c].ravel(), m[..., c].ravel()):
            assert lut.setdefault(s_, d_) == d_
        ks = sorted(lut)
        assert all(lut[x] <= lut[y] for x, y in zip(ks, ks[1:]))
        assert r[..., c].min() <= m[..., c].min() and m[..., c].max() <= r[..., c].max()
    assert np.array_equal(op.match_histograms_u8(a, a), a)
    sem = np.random.default_rng(5).choice([7, 8, 11, 24, 26, 33, 1], size=(32, 48)).astype(np.uint8)
    inst = np.where(np.isin(sem, [24, 26, 33]), sem.astype(np.int32) * 1000 + np.random.default_rng(6).integers(0, 4, sem.shape), sem).astype(np.uint16)
    lab = op.cityscapes_labels(sem, inst, 16, 24, 32, 48, small_instance_area_full_res=64)
    assert set(np.unique(lab["semantic"])) <= {0, 1, 2, 9, 11, 13, 255}
    assert ((lab["instance"] > 0) == (lab["offset_weights"][0] == 1)).all()
    assert (lab["semantic_weights"][lab["semantic"] == 255] == 0).all() and (lab["semantic_weights"] == 3).any()
