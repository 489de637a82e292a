"""Image-file data path on the device: drop-in for datasets/preprocessing.py:12-105 and the per-sample work of
datasets/dataset.py:66-67 / datasets/cityscapes.py:137-290, on DEVICE uint8 images instead of PIL images.

The PNG inflate itself stays on the host (`decode_png`: Pillow's libpng into pinned memory, one asynchronous H2D copy): a PNG is
one serial zlib stream.  Everything after it is byte / integer kernels of csrc/preprocess.hip that reproduce what the reference's
libraries compute -- Pillow's fixed-point Lanczos resize (`transforms.Resize(..., LANCZOS)`), Pillow's ImageEnhance blends and
HSV hue shift behind torchvision's ColorJitter functions, ToTensor + Normalize, skimage's histogram matching, cv2's nearest
resize on the label maps -- bit for bit where the library is available to check against (Pillow), by the published algorithm
otherwise (tests/test_preprocessing_{cpu,gpu}.py).

Image convention: uint8 tensors [H, W, 3] (or [B, H, W, 3]) on the GPU = `np.array(pil_image)`.
"""
import ctypes
import random
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib
from .._lib import check, lib, ptr, require_cuda, stream_ptr

OPS = {"brightness": 0, "contrast": 1, "saturation": 2, "hue": 3}
_coeff_cache: Dict[Tuple[int, int, int], Tuple[torch.Tensor, torch.Tensor, int]] = {}


def decode_png(path, device, mode: Optional[str] = "RGB") -> torch.Tensor:
    """PNG file -> device tensor (uint8 [H,W,3] for mode "RGB"; the file's own depth for mode None: 8-bit labelIds, 16-bit
    instanceIds / disparity as int32 -> uint16 range).  Host decode (Pillow), pinned staging, asynchronous copy."""
    from PIL import Image
    img = Image.open(path)
    if mode is not None:
        img = img.convert(mode)
    a = np.asarray(img)
    if a.dtype == np.int32 or a.dtype == np.uint16:
        a = a.astype(np.int32)
    t = torch.from_numpy(np.ascontiguousarray(a))
    if torch.cuda.is_available():
        t = t.pin_memory()
    return t.to(device, non_blocking=True)


def _as_u16(t: torch.Tensor) -> torch.Tensor:
    """16-bit unsigned values held in any integer dtype -> a contiguous tensor whose memory is their uint16 bit pattern."""
    if t.dtype in (torch.int16, getattr(torch, "uint16", torch.int16)):
        return t.contiguous()
    return (t.to(torch.int32) & 0xFFFF).to(torch.int16).contiguous()


def _batched(img: torch.Tensor) -> Tuple[torch.Tensor, bool]:
    require_cuda(img)
    if img.dtype != torch.uint8 or img.shape[-1] != 3 or img.dim() not in (3, 4):
        raise _lib.CodepsHipError(f"expected a uint8 image [H,W,3] or [B,H,W,3], got {tuple(img.shape)} {img.dtype}")
    return (img.unsqueeze(0), True) if img.dim() == 3 else (img, False)


def _coeffs(in_size: int, out_size: int, device) -> Tuple[torch.Tensor, torch.Tensor, int]:
    key = (in_size, out_size, device.index if device.index is not None else 0)
    hit = _coeff_cache.get(key)
    if hit is None:
        ksize = lib().codeps_lanczos_coeffs(in_size, out_size, None, None)
        if ksize <= 0:
            raise _lib.CodepsHipError("codeps_lanczos_coeffs: " + lib().codeps_last_error().decode())
        bounds = np.zeros((out_size, 2), np.int32)
        kk = np.zeros((out_size, ksize), np.int32)
        lib().codeps_lanczos_coeffs(in_size, out_size, bounds.ctypes.data_as(ctypes.c_void_p), kk.ctypes.data_as(ctypes.c_void_p))
        hit = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize)
        _coeff_cache[key] = hit
    return hit


def resize_lanczos(img: torch.Tensor, size: Sequence[int]) -> torch.Tensor:
    """`transforms.Resize(size=[H, W], interpolation=LANCZOS)` (datasets/dataset.py:66-67) = PIL `Image.resize((W, H), LANCZOS)`:
    horizontal pass, then vertical pass, 8-bit in between."""
    x, single = _batched(img.contiguous())
    b, h, w, _ = x.shape
    oh, ow = int(size[0]), int(size[1])
    if w != ow:
        bounds, kk, ks = _coeffs(w, ow, x.device)
        y = torch.empty(b, h, ow, 3, dtype=torch.uint8, device=x.device)
        check(lib().codeps_resample_u8(ptr(x), b, h, w, 1, ow, ptr(bounds), ptr(kk), ks, ptr(y), stream_ptr()), "codeps_resample_u8")
        x, w = y, ow
    if h != oh:
        bounds, kk, ks = _coeffs(h, oh, x.device)
        y = torch.empty(b, oh, w, 3, dtype=torch.uint8, device=x.device)
        check(lib().codeps_resample_u8(ptr(x), b, h, w, 0, oh, ptr(bounds), ptr(kk), ks, ptr(y), stream_ptr()), "codeps_resample_u8")
        x = y
    return x[0] if single else x


def color_jitter(img: torch.Tensor, ops: List[List[Tuple[str, float]]]) -> torch.Tensor:
    """Apply, per image of the batch, the colour operations [(name, factor), ...] in the given order (the order
    `random.shuffle` left them in, datasets/preprocessing.py:102-104); names: brightness, contrast, saturation, hue."""
    x, single = _batched(img.contiguous())
    b, h, w, _ = x.shape
    if not ops or (isinstance(ops[0], tuple) and len(ops[0]) == 2 and isinstance(ops[0][0], str)):
        ops = [list(ops)] * b                                     # one op list for the whole batch
    if len(ops) != b or b > 16:
        raise _lib.CodepsHipError("color_jitter: one op list per image, at most 16 images per call")
    n_ops = (ctypes.c_int * b)(*[len(o) for o in ops])
    codes = (ctypes.c_int * (4 * b))()
    factors = (ctypes.c_float * (4 * b))()
    for i, o in enumerate(ops):
        if len(o) > 4:
            raise _lib.CodepsHipError("color_jitter: at most four operations per image")
        for j, (name, f) in enumerate(o):
            if name == "hue" and not -0.5 <= f <= 0.5:
                raise ValueError(f"hue_factor ({f}) is not in [-0.5, 0.5].")
            codes[4 * i + j], factors[4 * i + j] = OPS[name], float(f)
    out, tmp = torch.empty_like(x), torch.empty_like(x)
    scratch = torch.empty(24, dtype=torch.int64, device=x.device)
    check(lib().codeps_color_jitter_u8(ptr(x), b, h, w, n_ops, codes, factors, ptr(out), ptr(tmp), ptr(scratch), stream_ptr()),
          "codeps_color_jitter_u8")
    return out[0] if single else out


def _get_random_color_jitter(brightness=None, contrast=None, saturation=None, hue=None) -> List[Tuple[str, float]]:
    """datasets/preprocessing.py:78-105: the same draws from Python's `random` in the same order (one uniform per active
    operation, then the shuffle) -> the op list `color_jitter` takes."""
    ops: List[Tuple[str, float]] = []
    if brightness is not None:
        ops.append(("brightness", random.uniform(brightness[0], brightness[1])))
    if contrast is not None:
        ops.append(("contrast", random.uniform(contrast[0], contrast[1])))
    if saturation is not None:
        ops.append(("saturation", random.uniform(saturation[0], saturation[1])))
    if hue is not None:
        ops.append(("hue", random.uniform(hue[0], hue[1])))
    random.shuffle(ops)
    return ops


def augment_data(output: Dict[str, Any], cfg) -> None:
    """datasets/preprocessing.py:21-57 on device data: `output["rgb"]` = {offset: uint8 [H,W,3]}, label maps as tensors
    ([H,W] or [C,H,W]), `camera_model` a 4-vector (fx, fy, cx, cy).  Same random draws as the reference."""
    if not cfg.active:
        return
    rng = lambda j: None if j is None else (1 - j, 1 + j)          # noqa: E731
    hue = None if cfg.hue_jitter is None else (-cfg.hue_jitter, cfg.hue_jitter)
    ops = _get_random_color_jitter(rng(cfg.brightness_jitter), rng(cfg.contrast_jitter), rng(cfg.saturation_jitter), hue)
    do_flip = cfg.horizontal_flipping and random.random() > .5
    for key, value in output.items():
        if key == "rgb":
            offsets = list(value)
            imgs = torch.stack([value[o] for o in offsets])
            if do_flip:
                imgs = torch.flip(imgs, dims=[2])                  # Image.FLIP_LEFT_RIGHT
            imgs = color_jitter(imgs, [ops] * len(offsets)) if ops else imgs
            for o, im in zip(offsets, imgs):
                output[key][o] = im
        elif key == "camera_model":
            if do_flip:
                img_width = next(iter(output["rgb"].values())).shape[1]
                value = value.clone()
                value[2] = img_width - value[2] - 1                # flip the principal point horizontally
                output[key] = value
        elif do_flip:
            flipped = torch.flip(value, dims=[-1]).contiguous() if torch.is_tensor(value) else np.flip(value, axis=-1).copy()
            if key == "offset":
                flipped[1] *= -1                                   # the x offsets change sign
            output[key] = flipped


def to_tensor(img: torch.Tensor, mean: Optional[Sequence[float]] = None, std: Optional[Sequence[float]] = None,
              flip: Optional[Sequence[bool]] = None) -> torch.Tensor:
    """ToTensor (+ Normalize) of uint8 [B,H,W,3] / [H,W,3] -> fp32 [B,3,H,W] / [3,H,W]; optional per-image horizontal flip."""
    x, single = _batched(img.contiguous())
    b, h, w, _ = x.shape
    out = torch.empty(b, 3, h, w, dtype=torch.float32, device=x.device)
    m = None if mean is None else (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = None if std is None else (ctypes.c_float * 3)(*[float(v) for v in std])
    mask = 0 if flip is None else sum(1 << i for i, f in enumerate(flip) if f)
    check(lib().codeps_u8_to_tensor(ptr(x), b, h, w, m, s, mask, ptr(out), stream_ptr()), "codeps_u8_to_tensor")
    return out[0] if single else out


def prepare_for_network(output: Dict[str, Any], cfg) -> None:
    """datasets/preprocessing.py:12-18."""
    offsets = list(output["rgb"])
    imgs = torch.stack([output["rgb"][o] for o in offsets])
    ten = to_tensor(imgs, cfg.rgb_mean if cfg.active else None, cfg.rgb_std if cfg.active else None)
    for o, t in zip(offsets, ten):
        output["rgb"][o] = t


def _convert_hsv(x: torch.Tensor, to_hsv: bool) -> torch.Tensor:
    out = torch.empty_like(x)
    check(lib().codeps_rgb_hsv_u8(ptr(x), x.numel() // 3, 1 if to_hsv else 0, ptr(out), stream_ptr()), "codeps_rgb_hsv_u8")
    return out


def _match(x: torch.Tensor, r: torch.Tensor) -> torch.Tensor:
    b, h, w, _ = x.shape
    out = torch.empty_like(x)
    nbytes = lib().codeps_match_histograms_workspace_bytes(b)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    check(lib().codeps_match_histograms_u8(ptr(x), ptr(r), b, h, w, r.shape[1], r.shape[2], ptr(out), ptr(ws), nbytes, stream_ptr()),
          "codeps_match_histograms_u8")
    return out


def transfer_histogram_style(img: torch.Tensor, reference_img: torch.Tensor, mode: Optional[str]) -> torch.Tensor:
    """datasets/preprocessing.py:60-75 on uint8 [B,H,W,3] / [H,W,3] device images.  "rgb": skimage match_histograms per channel;
    "hsv": the same in Pillow's HSV space (Image.convert("HSV") -> match -> convert("RGB")); "saturation": only the S channel is
    taken from the matched image; None: identity.  (The reference's callers pass "rgb": datasets/replay_buffer.py:159,290.)"""
    if mode is None:
        return img
    if mode not in ("rgb", "hsv", "saturation"):
        raise ValueError(f"Unknown histogram transfer mode: {mode}")
    x, single = _batched(img.contiguous())
    r, _ = _batched(reference_img.contiguous())
    if r.shape[0] != x.shape[0]:
        raise _lib.CodepsHipError("transfer_histogram_style: one reference image per image")
    if mode == "rgb":
        out = _match(x, r)
    else:
        x_hsv, r_hsv = _convert_hsv(x, True), _convert_hsv(r, True)
        new = _match(x_hsv, r_hsv)
        if mode == "saturation":
            new[..., 0] = x_hsv[..., 0]
            new[..., 2] = x_hsv[..., 2]
        out = _convert_hsv(new.contiguous(), False)
    return out[0] if single else out


CITYSCAPES_TO_CODEPS = [(7, 0), (8, 1), (11, 2), (12, 2), (13, 3), (17, 4), (20, 5), (21, 6), (22, 7), (23, 8), (24, 9), (25, 10),
                        (26, 11), (27, 12), (32, 13), (33, 13)]                                 # datasets/cityscapes.py:291-311
CITYSCAPES_TO_CITYSCAPES = [(7, 0), (8, 1), (11, 2), (12, 3), (13, 4), (17, 5), (19, 6), (20, 7), (21, 8), (22, 9), (23, 10), (24, 11),
                            (25, 12), (26, 13), (27, 14), (28, 15), (31, 16), (32, 17), (33, 18)]    # cityscapes.py:266-289


KITTI360_TO_CODEPS = [(7, 0), (8, 1), (11, 2), (34, 2), (12, 2), (13, 3), (35, 3), (17, 4), (37, 4), (20, 5), (21, 6), (22, 7), (23, 8), (24, 9),
                      (25, 10), (26, 11), (27, 12), (32, 13), (33, 13)]                          # datasets/kitti_360.py:370-390
KITTI360_TO_CITYSCAPES = CITYSCAPES_TO_CITYSCAPES + [(34, 2), (35, 4), (37, 5)]                 # kitti_360.py:342-367


def kitti360_labels(semantic_ids, instance_ids, out_size, full_size, mapping=None, thing_classes=(9, 10, 11, 12, 13),
                    small_instance_area_full_res: float = 4096, small_instance_weight: int = 3, sigma: float = 8.0):
    """The label half of Kitti360.__getitem__ (datasets/kitti_360.py:236-300), incl. `semantic_eval`."""
    return cityscapes_labels(semantic_ids, instance_ids, out_size, full_size, KITTI360_TO_CODEPS if mapping is None else mapping,
                             thing_classes, small_instance_area_full_res, small_instance_weight, sigma, _kitti=True)


def lidar_depth(points: torch.Tensor, lidar_to_rect, camera_matrix, full_size: Sequence[int], out_size: Sequence[int]) -> torch.Tensor:
    """datasets/kitti_360.py:303-330: Velodyne points (device fp32 [N,4]) -> sparse depth [out_h,out_w] (fp32; the reference's
    map is float64).  lidar_to_rect [>=3,4] / camera_matrix [3,>=3]: host arrays as `_read_lidar_to_rect` returns them."""
    require_cuda(points)
    pts = points.float().contiguous()
    m = np.ascontiguousarray(np.asarray(lidar_to_rect, np.float64)[:3, :4])
    k = np.ascontiguousarray(np.asarray(camera_matrix, np.float64)[:3, :3])
    h, w = int(full_size[0]), int(full_size[1])
    out = torch.empty(int(out_size[0]), int(out_size[1]), dtype=torch.float32, device=pts.device)
    nbytes = lib().codeps_lidar_depth_workspace_bytes(h, w)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
    check(lib().codeps_lidar_depth(ptr(pts), pts.shape[0], m.ctypes.data_as(ctypes.c_void_p), k.ctypes.data_as(ctypes.c_void_p), h, w,
                                   int(out_size[0]), int(out_size[1]), ptr(out), ptr(ws), nbytes, stream_ptr()), "codeps_lidar_depth")
    return out


def cityscapes_labels(semantic_ids: torch.Tensor, instance_ids: torch.Tensor, out_size: Sequence[int], full_size: Sequence[int],
                      mapping=None, thing_classes=(9, 10, 11, 12, 13), small_instance_area_full_res: float = 4096,
                      small_instance_weight: int = 3, sigma: float = 8.0, _kitti: bool = False) -> Dict[str, torch.Tensor]:
    """The label half of Cityscapes.__getitem__ (datasets/cityscapes.py:166-246): labelIds (uint8 [H,W]) and instanceIds
    (16-bit, any integer dtype [H,W]) at file resolution -> the network-resolution maps incl. centre heat map and offsets."""
    from .. import functional as F
    require_cuda(semantic_ids, instance_ids)
    h, w = semantic_ids.shape
    oh, ow = int(out_size[0]), int(out_size[1])
    dev = semantic_ids.device
    table = np.full(256, 255, np.uint8)
    for a, b_ in (CITYSCAPES_TO_CODEPS if mapping is None else mapping):
        table[a] = b_
    mask = 0
    for c in thing_classes:
        mask |= 1 << int(c)
    sem8 = semantic_ids.to(torch.uint8).contiguous()
    inst16 = _as_u16(instance_ids)
    out = {"semantic": torch.empty(oh, ow, dtype=torch.uint8, device=dev), "instance": torch.empty(oh, ow, dtype=torch.int32, device=dev),
           "thing_mask": torch.empty(oh, ow, dtype=torch.uint8, device=dev), "semantic_weights": torch.empty(oh, ow, dtype=torch.uint8, device=dev),
           "center_weights": torch.empty(1, oh, ow, dtype=torch.uint8, device=dev), "offset_weights": torch.empty(1, oh, ow, dtype=torch.uint8, device=dev)}
    nbytes = lib().codeps_cityscapes_labels_workspace_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    small = small_instance_area_full_res * (oh / full_size[0]) * (ow / full_size[1])
    if _kitti:
        out["semantic_eval"] = torch.empty(oh, ow, dtype=torch.uint8, device=dev)
        check(lib().codeps_kitti360_labels(ptr(sem8), ptr(inst16), h, w, oh, ow, table.ctypes.data_as(ctypes.c_void_p), mask, float(small),
                                           int(small_instance_weight), ptr(out["semantic"]), ptr(out["instance"]), ptr(out["thing_mask"]),
                                           ptr(out["semantic_weights"]), ptr(out["center_weights"]), ptr(out["offset_weights"]),
                                           ptr(out["semantic_eval"]), ptr(ws), nbytes, stream_ptr()), "codeps_kitti360_labels")
    else:
        check(lib().codeps_cityscapes_labels(ptr(sem8), ptr(inst16), h, w, oh, ow, table.ctypes.data_as(ctypes.c_void_p), mask, float(small),
                                             int(small_instance_weight), ptr(out["semantic"]), ptr(out["instance"]), ptr(out["thing_mask"]),
                                             ptr(out["semantic_weights"]), ptr(out["center_weights"]), ptr(out["offset_weights"]), ptr(ws), nbytes,
                                             stream_ptr()), "codeps_cityscapes_labels")
    center, offset, _status = F.offset_center(out["instance"].unsqueeze(0), sigma)          # Dataset.get_offset_center on the device
    out["center"], out["offset"] = center[0], offset[0]
    return out


def disparity_to_depth(disparity: torch.Tensor, fx: float, baseline: float, out_size: Sequence[int]) -> torch.Tensor:
    """datasets/cityscapes.py:249-262: 16-bit Cityscapes disparity [H,W] -> depth in metres at the network resolution."""
    require_cuda(disparity)
    h, w = disparity.shape
    d16 = _as_u16(disparity)
    out = torch.empty(int(out_size[0]), int(out_size[1]), dtype=torch.float32, device=disparity.device)
    check(lib().codeps_disparity_to_depth(ptr(d16), h, w, int(out_size[0]), int(out_size[1]), float(baseline * fx), ptr(out), stream_ptr()),
          "codeps_disparity_to_depth")
    return out
