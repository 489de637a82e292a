"""SURVEY 8f-4, the image-file data path on the device (csrc/preprocess.hip, codeps_amd/datasets/preprocessing.py) -- bit for bit
against Pillow (the library behind `transforms.Resize(..., LANCZOS)` and behind torchvision's colour-jitter functions,
datasets/dataset.py:66-67, datasets/preprocessing.py:78-105), against the Pillow-made fixture, and against the oracle for the
pieces whose library is not installed (skimage histogram matching, cv2 nearest resize: parity unpinned there)."""
import random

import numpy as np
import pytest
import torch

from helpers import load
from oracle import mixup as omix
from oracle import preprocessing as op

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _img(h, w, seed):
    r = np.random.default_rng(seed)
    a = np.kron(r.random((h // 8 + 2, w // 8 + 2, 3)), np.ones((8, 8, 1)))[:h, :w]
    return (a * 255 + r.normal(0, 8, (h, w, 3))).clip(0, 255).astype(np.uint8)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("shape,size", [((96, 160), (48, 80)), ((96, 160), (40, 72)), ((96, 160), (120, 200)), ((97, 161), (33, 57)),
                                        ((1024, 2048), (512, 1024)), ((376, 1408), (384, 1408))])
def test_lanczos_resize_is_pillow_bit_for_bit(shape, size):
    from PIL import Image
    from codeps_amd.datasets import preprocessing as P
    a = _img(shape[0], shape[1], 1)
    ref = np.array(Image.fromarray(a).resize((size[1], size[0]), Image.LANCZOS))
    got = P.resize_lanczos(_dev(a), size).cpu().numpy()
    assert np.array_equal(got, ref)
    two = P.resize_lanczos(_dev(np.stack([a, a[::-1].copy()])), size).cpu().numpy()        # batched
    assert np.array_equal(two[0], ref)


def test_colour_jitter_every_order_is_pillow_bit_for_bit():
    from PIL import Image, ImageEnhance
    from codeps_amd.datasets import preprocessing as P
    import itertools
    a = _img(64, 96, 2)
    factors = {"brightness": 0.83, "contrast": 1.18, "saturation": 0.9, "hue": 0.07}

    def pil_chain(order):
        img = Image.fromarray(a)
        for name in order:
            f = factors[name]
            if name == "brightness":
                img = ImageEnhance.Brightness(img).enhance(f)
            elif name == "contrast":
                img = ImageEnhance.Contrast(img).enhance(f)
            elif name == "saturation":
                img = ImageEnhance.Color(img).enhance(f)
            else:
                h, s, v = img.convert("HSV").split()
                np_h = (np.array(h, dtype=np.uint8).astype(np.int32) + (int(f * 255) & 0xFF)).astype(np.uint8)
                img = Image.merge("HSV", (Image.fromarray(np_h, "L"), s, v)).convert("RGB")
        return np.array(img)

    orders = list(itertools.permutations(factors))                               # all 24 orders `random.shuffle` can produce
    for lo in range(0, len(orders), 12):
        chunk = orders[lo:lo + 12]
        got = P.color_jitter(_dev(np.stack([a] * len(chunk))), [[(n, factors[n]) for n in o] for o in chunk]).cpu().numpy()
        for o, g in zip(chunk, got):
            assert np.array_equal(g, pil_chain(o)), o
    # subsets, extrapolating factors, negative hue
    for ops in ([("contrast", 1.9)], [("hue", -0.31), ("contrast", 0.4)], [], [("brightness", 1.6), ("saturation", 0.0)]):
        factors.update(dict(ops))
        got = P.color_jitter(_dev(a), list(ops)).cpu().numpy()
        assert np.array_equal(got, pil_chain([n for n, _ in ops])), ops


def test_fixture_and_to_tensor():
    from codeps_amd.datasets import preprocessing as P
    z, meta = load("preprocessing")
    a = _img(meta["h"], meta["w"], meta["seed"])
    resized = P.resize_lanczos(_dev(a), meta["size"])
    assert np.array_equal(resized.cpu().numpy(), z["resized"])
    jit = P.color_jitter(resized, [tuple(o) for o in meta["ops"]])
    assert np.array_equal(jit.cpu().numpy(), z["jittered"])
    ten = P.to_tensor(jit, meta["mean"], meta["std"])
    assert np.array_equal(ten.cpu().numpy(), z["tensor"]), float(np.abs(ten.cpu().numpy() - z["tensor"]).max())
    flipped = P.to_tensor(torch.stack([jit, jit]), None, None, flip=[False, True]).cpu().numpy()
    assert np.array_equal(flipped[1], flipped[0][:, :, ::-1]) and np.array_equal(flipped[0], op.to_tensor_normalize(z["jittered"]))


def test_augment_data_draws_like_the_reference():
    """Same calls to Python's `random` in the same order as datasets/preprocessing.py:21-57 -> same factors, same op order, same
    flip decision; the data follows (images flipped and jittered, labels flipped, principal point mirrored, x offsets negated)."""
    from codeps_amd.config import CfgNode
    from codeps_amd.datasets import preprocessing as P
    cfg = CfgNode(active=True, horizontal_flipping=True, brightness_jitter=0.2, contrast_jitter=0.2, saturation_jitter=0.2, hue_jitter=0.1)
    for seed in range(6):
        a = _img(32, 48, 10 + seed)
        random.seed(seed)
        ops = []
        for name, rng in (("brightness", (0.8, 1.2)), ("contrast", (0.8, 1.2)), ("saturation", (0.8, 1.2)), ("hue", (-0.1, 0.1))):
            ops.append((name, random.uniform(*rng)))
        random.shuffle(ops)
        flip = random.random() > .5
        out = {"rgb": {0: _dev(a), -1: _dev(a[:, ::-1].copy())}, "camera_model": torch.tensor([100.0, 101.0, 20.0, 15.0]),
               "semantic": _dev(np.arange(32 * 48, dtype=np.int32).reshape(32, 48) % 14), "offset": torch.ones(2, 32, 48, device=DEV)}
        random.seed(seed)
        P.augment_data(out, cfg)
        src = a[:, ::-1] if flip else a
        assert np.array_equal(out["rgb"][0].cpu().numpy(), op.color_jitter(np.ascontiguousarray(src), ops)), seed
        assert float(out["camera_model"][2]) == (48 - 20.0 - 1 if flip else 20.0)
        assert float(out["offset"][1].mean()) == (-1.0 if flip else 1.0) and float(out["offset"][0].mean()) == 1.0
        assert int(out["semantic"][0, 0]) == (47 % 14 if flip else 0)


def test_histogram_matching_against_oracle():
    from codeps_amd.datasets import preprocessing as P
    a, r = _img(96, 128, 3), (_img(80, 112, 4) * 0.6 + 60).astype(np.uint8)
    got = P.transfer_histogram_style(_dev(a), _dev(r), "rgb").cpu().numpy()
    assert np.array_equal(got, op.match_histograms_u8(a, r))
    both = P.transfer_histogram_style(_dev(np.stack([a, a[::-1].copy()])), _dev(np.stack([r, r])), "rgb").cpu().numpy()
    assert np.array_equal(both[1], op.match_histograms_u8(a[::-1].copy(), r))
    assert P.transfer_histogram_style(_dev(a), _dev(r), None).data_ptr() is not None
    with pytest.raises(ValueError):
        P.transfer_histogram_style(_dev(a), _dev(r), "lab")


@pytest.mark.parametrize("mode", ["hsv", "saturation"])
def test_histogram_matching_in_hsv_space_against_pillow(mode):
    """datasets/preprocessing.py:64-71: Image.convert("HSV") -> match_histograms -> (saturation: keep H and V) -> convert("RGB"),
    with Pillow ITSELF doing both conversions of the expected image (the matching in between is the oracle's)."""
    from PIL import Image
    from codeps_amd.datasets import preprocessing as P
    a, r = _img(96, 128, 13), (_img(80, 112, 14) * 0.7 + 40).astype(np.uint8)
    a_hsv, r_hsv = np.array(Image.fromarray(a, "RGB").convert("HSV")), np.array(Image.fromarray(r, "RGB").convert("HSV"))
    new = op.match_histograms_u8(a_hsv, r_hsv)
    if mode == "saturation":
        new[:, :, 0], new[:, :, 2] = a_hsv[:, :, 0], a_hsv[:, :, 2]
    want = np.array(Image.fromarray(new, "HSV").convert("RGB"))
    got = P.transfer_histogram_style(_dev(a), _dev(r), mode).cpu().numpy()
    assert np.array_equal(got, want), f"{int((got != want).sum())} of {got.size} bytes differ from Pillow"
    assert np.array_equal(op.transfer_histogram_style(a, r, mode), want)          # and the oracle's restatement of the same path
    batch = P.transfer_histogram_style(_dev(np.stack([a, a])), _dev(np.stack([r, r])), mode).cpu().numpy()
    assert np.array_equal(batch[0], want) and np.array_equal(batch[1], want)


def test_cityscapes_label_path_against_oracle():
    from codeps_amd.datasets import preprocessing as P
    rng = np.random.default_rng(5)
    h, w, oh, ow = 128, 256, 64, 96
    sem = np.kron(rng.choice([7, 8, 11, 12, 21, 23, 24, 25, 26, 27, 32, 33, 1, 19], size=(h // 16, w // 16)), np.ones((16, 16))).astype(np.uint8)
    k = np.kron(rng.integers(0, 3, (h // 8, w // 8)), np.ones((8, 8))).astype(np.int32)
    inst = np.where(np.isin(sem, [24, 25, 26, 27, 32, 33]), sem.astype(np.int32) * 1000 + k, sem.astype(np.int32))
    inst[:16, :16] = sem[:16, :16]                                              # an "iscrowd" region: instance id < 1000
    want = op.cityscapes_labels(sem, inst.astype(np.uint16), oh, ow, h, w, small_instance_area_full_res=600)
    got = P.cityscapes_labels(_dev(sem), _dev(inst), (oh, ow), (h, w), small_instance_area_full_res=600)
    for key in ("semantic", "semantic_weights", "instance", "thing_mask", "center_weights", "offset_weights"):
        assert np.array_equal(got[key].cpu().numpy(), want[key]), key
    assert (want["semantic_weights"] == 3).any()
    off, ctr = omix.offset_center(want["instance"])
    assert np.array_equal(got["offset"].cpu().numpy(), off)
    assert np.abs(got["center"].cpu().numpy() - ctr).max() < 1e-6
    disp = rng.integers(0, 30000, (h, w)).astype(np.uint16)
    disp[rng.random((h, w)) < 0.2] = 0
    d = P.disparity_to_depth(_dev(disp.astype(np.int32)), 2262.52, 0.209313, (oh, ow)).cpu().numpy()
    assert np.allclose(d, op.disparity_to_depth(disp, 2262.52, 0.209313, oh, ow), rtol=1e-6, atol=0)


def test_cityscapes_dataset_item_from_png_files(tmp_path):
    """A miniature Cityscapes tree written as real PNG / JSON files (the layout of datasets/cityscapes.py:31-67) -> one
    `__getitem__` through host inflate + device kernels == the oracle's per-step restatement on the decoded arrays."""
    import json
    from PIL import Image
    from codeps_amd.config import CfgNode
    from codeps_amd.datasets import Cityscapes
    rng = np.random.default_rng(11)
    h, w, oh, ow = 128, 256, 64, 128
    city, stem = "aachen", "aachen_000000_000019"
    for d in ("gtFine/train", "leftImg8bit_sequence/train", "camera/train", "disparity_sequence/train"):
        (tmp_path / d / city).mkdir(parents=True)
    frames = {}
    for n in (18, 19, 20):
        frames[n] = _img(h, w, 30 + n)
        Image.fromarray(frames[n]).save(tmp_path / "leftImg8bit_sequence/train" / city / f"aachen_000000_0000{n}_leftImg8bit.png")
    sem = np.kron(rng.choice([7, 8, 11, 21, 23, 24, 26, 33, 1], size=(h // 16, w // 16)), np.ones((16, 16))).astype(np.uint8)
    k = np.kron(rng.integers(0, 3, (h // 8, w // 8)), np.ones((8, 8))).astype(np.int32)
    inst = np.where(np.isin(sem, [24, 26, 33]), sem.astype(np.int32) * 1000 + k, sem.astype(np.int32)).astype(np.uint16)
    disp = rng.integers(0, 30000, (h, w)).astype(np.uint16)
    Image.fromarray(sem).save(tmp_path / "gtFine/train" / city / f"{stem}_gtFine_labelIds.png")
    Image.fromarray(inst).save(tmp_path / "gtFine/train" / city / f"{stem}_gtFine_instanceIds.png")
    Image.fromarray(disp).save(tmp_path / "disparity_sequence/train" / city / f"{stem}_disparity.png")
    cam = {"intrinsic": {"fx": 2262.52, "fy": 2265.30, "u0": 1096.98, "v0": 513.137}, "extrinsic": {"baseline": 0.209313}}
    (tmp_path / "camera/train" / city / f"{stem}_camera.json").write_text(json.dumps(cam))
    cfg = CfgNode(name="cityscapes", path=str(tmp_path), feed_img_size=[oh, ow], offsets=[1], center_heatmap_sigma=8, small_instance_weight=3,
                  small_instance_area_full_res=600, remove_classes=[],
                  augmentation=CfgNode(active=True, horizontal_flipping=True, brightness_jitter=0.2, contrast_jitter=0.2,
                                       saturation_jitter=0.2, hue_jitter=0.1),
                  normalization=CfgNode(active=True, rgb_mean=(0.485, 0.456, 0.406), rgb_std=(0.229, 0.224, 0.225)))
    ds = Cityscapes("train", cfg, return_depth=True, device=DEV)
    assert len(ds) == 1
    plain = ds.__getitem__(0, do_network_preparation=False, do_augmentation=False)
    assert sorted(plain["rgb"]) == [-1, 0, 1]
    for off, n in ((-1, 18), (0, 19), (1, 20)):
        assert np.array_equal(plain["rgb"][off].cpu().numpy(), op.lanczos_resize(frames[n], oh, ow)), off
    want = op.cityscapes_labels(sem, inst, oh, ow, h, w, small_instance_area_full_res=600)
    for key, v in want.items():
        assert np.array_equal(plain[key].cpu().numpy(), v), key
    assert np.allclose(plain["depth"].cpu().numpy(), op.disparity_to_depth(disp, 2262.52, 0.209313, oh, ow), rtol=1e-6)
    assert np.allclose(plain["camera_model"].numpy(), [2262.52 * ow / w, 2265.30 * oh / h, 1096.98 * ow / w, 513.137 * oh / h])
    random.seed(4)
    item = ds[0]                                                     # augmentation + network preparation
    random.seed(4)
    ops = []
    for name, rg in (("brightness", (0.8, 1.2)), ("contrast", (0.8, 1.2)), ("saturation", (0.8, 1.2)), ("hue", (-0.1, 0.1))):
        ops.append((name, random.uniform(*rg)))
    random.shuffle(ops)
    flip = random.random() > .5
    src = op.lanczos_resize(frames[19], oh, ow)
    src = np.ascontiguousarray(src[:, ::-1]) if flip else src
    ref = op.to_tensor_normalize(op.color_jitter(src, ops), cfg.normalization.rgb_mean, cfg.normalization.rgb_std)
    assert item["rgb"][0].shape == (3, oh, ow) and np.array_equal(item["rgb"][0].cpu().numpy(), ref)
    sem_ref = want["semantic"][:, ::-1] if flip else want["semantic"]
    assert np.array_equal(item["semantic"].cpu().numpy(), sem_ref)


def test_kitti360_dataset_item_from_files(tmp_path):
    """A miniature KITTI-360 tree (frame list, raw images, semantic / instance PNGs, calibration, one Velodyne sweep) through
    `Kitti360.__getitem__` == the oracle's restatement of datasets/kitti_360.py:183-338 on the decoded arrays."""
    from PIL import Image
    from codeps_amd.config import CfgNode
    from codeps_amd.datasets import Kitti360
    rng = np.random.default_rng(21)
    h, w, oh, ow = 96, 256, 48, 128
    seq = "2013_05_28_drive_0000_sync"
    raw = tmp_path / "data_2d_raw" / seq / "image_00" / "data_rect"
    semd = tmp_path / "data_2d_semantics" / "train" / seq / "image_00" / "semantic"
    insd = tmp_path / "data_2d_semantics" / "train" / seq / "image_00" / "instance"
    velo = tmp_path / "data_3d_raw" / seq / "velodyne_points" / "data"
    for d in (raw, semd, insd, velo, tmp_path / "calibration"):
        d.mkdir(parents=True)
    frames = {}
    for n in (249, 250, 251):
        frames[n] = _img(h, w, 40 + n)
        Image.fromarray(frames[n]).save(raw / f"{n:010d}.png")
    sem = np.kron(rng.choice([7, 8, 11, 34, 35, 37, 21, 23, 24, 26, 33, 1], size=(h // 16, w // 16)), np.ones((16, 16))).astype(np.uint8)
    k = np.kron(rng.integers(0, 3, (h // 8, w // 8)), np.ones((8, 8))).astype(np.int32)
    inst = (sem.astype(np.int32) * 1000 + k).astype(np.uint16)
    inst[40:42, 100:103] = 26 * 1000 + 7
    sem[40:42, 100:103] = 26                                                      # a 6-pixel instance: dropped from semantic_eval
    Image.fromarray(sem).save(semd / "0000000250.png")
    Image.fromarray(inst).save(insd / "0000000250.png")
    (tmp_path / "data_2d_semantics" / "train" / "2013_05_28_drive_train_frames.txt").write_text(
        f"data_2d_raw/{seq}/image_00/data_rect/0000000250.png data_2d_semantics/train/{seq}/image_00/semantic/0000000250.png\n")
    fx, fy, cx, cy = 180.0, 182.0, 127.3, 47.6
    (tmp_path / "calibration" / "perspective.txt").write_text(
        f"S_rect_00: {w} {h}\nR_rect_00: 0.9999 0.01 -0.005 -0.01 0.9999 0.003 0.005 -0.003 0.9999\n"
        f"P_rect_00: {fx} 0.0 {cx} 0.0 0.0 {fy} {cy} 0.0 0.0 0.0 1.0 0.0\n")
    (tmp_path / "calibration" / "calib_cam_to_velo.txt").write_text("0.04 -0.99 0.02 0.26 -0.03 -0.02 -0.99 -0.11 0.99 0.04 -0.03 -0.83\n")
    pts = np.concatenate([rng.uniform(2, 60, (4000, 1)), rng.uniform(-20, 20, (4000, 1)), rng.uniform(-2, 1, (4000, 1)), rng.random((4000, 1))], 1)
    pts.astype(np.float32).tofile(velo / "0000000250.bin")
    cfg = CfgNode(name="kitti_360", path=str(tmp_path), feed_img_size=[oh, ow], offsets=[1], center_heatmap_sigma=8, small_instance_weight=3,
                  small_instance_area_full_res=800, remove_classes=[],
                  augmentation=CfgNode(active=False), normalization=CfgNode(active=True, rgb_mean=(0.485, 0.456, 0.406), rgb_std=(0.229, 0.224, 0.225)))
    ds = Kitti360("train", cfg, return_depth=True, device=DEV)
    assert len(ds) == 1
    # a sweep whose points land inside the image: camera-frame points of random pixels / depths taken back to the Velodyne frame
    n = 3000
    dpt = rng.uniform(1.0, 95.0, n)                                               # some beyond the 80 m cut
    uv = np.stack([rng.uniform(-10, w + 10, n), rng.uniform(-5, h + 5, n)], 1)     # some outside the image
    cam = np.stack([(uv[:, 0] - cx) / fx * dpt, (uv[:, 1] - cy) / fy * dpt, dpt, np.ones(n)], 1)
    velo_pts = (np.linalg.inv(ds.lidar_to_rect) @ cam.T).T
    velo_pts[:, 3] = rng.random(n)
    velo_pts.astype(np.float32).tofile(velo / "0000000250.bin")
    item = ds.__getitem__(0, do_network_preparation=False)
    for off, n in ((-1, 249), (0, 250), (1, 251)):
        assert np.array_equal(item["rgb"][off].cpu().numpy(), op.lanczos_resize(frames[n], oh, ow)), off
    want = op.kitti360_labels(sem, inst, oh, ow, h, w, small_instance_area_full_res=800)
    for key, v in want.items():
        assert np.array_equal(item[key].cpu().numpy(), v), key
    assert (want["semantic_eval"] != want["semantic"]).any() and (want["semantic_weights"] == 3).any()
    assert np.allclose(item["camera_model"].numpy(), [fx * ow / w, fy * oh / h, cx * ow / w, cy * oh / h])
    d_ref = op.lidar_depth(np.fromfile(velo / "0000000250.bin", dtype=np.float32), ds.lidar_to_rect, ds.camera_matrix, h, w, oh, ow)
    d = item["depth"].cpu().numpy()
    assert (d_ref > 0).sum() > 50 and np.array_equal(d > 0, d_ref > 0)
    assert np.allclose(d, d_ref.astype(np.float32), rtol=1e-6, atol=0)
    ten = ds[0]["rgb"][0]
    assert ten.shape == (3, oh, ow) and ten.dtype == torch.float32
