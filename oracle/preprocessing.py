"""CPU oracle: the image-file data path.  TEST INFRASTRUCTURE -- imported by tests/ only.

Restates, in numpy, what the reference's data loader does to a decoded image before the network sees it
(datasets/dataset.py:66-67, datasets/cityscapes.py:137-290, datasets/preprocessing.py:12-105):

  * `transforms.Resize(size, LANCZOS)` = PIL `Image.resize(..., LANCZOS)`: two separable passes in 8.22 fixed point with an 8-bit
    intermediate (Pillow `src/libImaging/Resample.c`: precompute_coeffs, normalize_coeffs_8bpc, ImagingResample{Horizontal,
    Vertical}_8bpc).  PINNED: bit-exact against Pillow (installed here) in tests/test_preprocessing_cpu.py.
  * the colour jitter of `_get_random_color_jitter` (preprocessing.py:78-105): torchvision 0.13.1 `F.adjust_{brightness,
    contrast,saturation,hue}` on PIL images = Pillow `ImageEnhance.{Brightness,Contrast,Color}` (Image.blend with a black /
    mean-grey / greyscale image) and the uint8 hue shift in HSV space (Pillow `Convert.c` rgb2hsv / hsv2rgb).  torchvision is
    not installed: its four wrappers are restated from the published 0.13.1 source; the Pillow operations underneath are
    PINNED bit-exact against Pillow.
  * `ToTensor` + `Normalize` (preprocessing.py:12-18), horizontal flip incl. the principal point and the x offsets
    (preprocessing.py:41-57).
  * `transfer_histogram_style(mode="rgb")` = skimage.exposure.match_histograms per channel (preprocessing.py:60-63).
    scikit-image is not installed: restated from the published algorithm (bincount CDFs, np.interp, truncating cast back to
    uint8) -- PARITY UNPINNED for this function.
  * the label path of Cityscapes.__getitem__ (cityscapes.py:166-246): cv2.INTER_NEAREST resize (cv2 not installed: restated,
    PARITY UNPINNED: src = min(floor(dst * src_size / dst_size), src_size - 1)), label conversion table, instance ids of the
    thing classes, semantic weights by instance area, thing mask, centre / offset weights, disparity -> depth.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


# ----------------------------------------------------------------------------------------------- Lanczos resize
def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def resample_coeffs(in_size, out_size, support=3.0, filt=_lanczos):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc -> (bounds [out, 2] = (xmin, count), coeffs int32 [out, ksize])."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - sup + 0.5)
        xmin = max(xmin, 0)
        xmax = int(center + sup + 0.5)
        xmax = min(xmax, in_size) - xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img, out_size, axis):
    """One 8-bit pass along `axis` of a uint8 array [H, W, C]."""
    in_size = img.shape[axis]
    bounds, kk = resample_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)                 # [in, other, C]
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(cnt):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def lanczos_resize(img, out_h, out_w):
    """uint8 [H, W, C] -> uint8 [out_h, out_w, C]; horizontal pass first, then vertical (ImagingResample)."""
    if img.shape[1] != out_w:
        img = _resample_axis(img, out_w, 1)
    if img.shape[0] != out_h:
        img = _resample_axis(img, out_h, 0)
    return img


# ----------------------------------------------------------------------------------------------- colour operations
def to_grey(rgb):
    """Pillow rgb2l: (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16."""
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(degenerate, image, alpha):
    """Pillow ImagingBlend(im1 = degenerate, im2 = image, alpha) on uint8 arrays (Blend.c): interpolation truncates, extrapolation
    clips."""
    a = np.float32(alpha)
    d, i = degenerate.astype(np.int32), image.astype(np.int32)
    if alpha == 0.0:
        return degenerate.copy()
    if alpha == 1.0:
        return image.copy()
    tmp = (d.astype(np.float32) + a * (i - d).astype(np.float32)).astype(np.float32)
    if 0.0 <= alpha <= 1.0:
        return tmp.astype(np.int32).astype(np.uint8)                 # (UINT8) of a float in [0, 255]: truncation
    out = np.where(tmp <= 0.0, 0, np.where(tmp >= 255.0, 255, tmp.astype(np.int32)))
    return out.astype(np.uint8)


def adjust_brightness(rgb, factor):
    """torchvision F.adjust_brightness on a PIL image = ImageEnhance.Brightness(img).enhance(factor)."""
    return blend(np.zeros_like(rgb), rgb, factor)


def adjust_contrast(rgb, factor):
    """ImageEnhance.Contrast: degenerate = solid grey of int(mean(L) + 0.5)."""
    grey = to_grey(rgb)
    mean = int(grey.astype(np.float64).sum() / grey.size + 0.5)
    return blend(np.full_like(rgb, mean), rgb, factor)


def adjust_saturation(rgb, factor):
    """ImageEnhance.Color: degenerate = image.convert("L").convert("RGB")."""
    grey = to_grey(rgb)
    return blend(np.repeat(grey[..., None], 3, -1), rgb, factor)


def rgb_to_hsv(rgb):
    """Pillow Convert.c rgb2hsv_row (float arithmetic of colorsys, uint8 in / out)."""
    r, g, b = (rgb[..., i].astype(np.int32) for i in range(3))
    maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    ok = maxc != minc
    crs = np.where(ok, cr, np.float32(1))
    s = cr / np.where(maxc == 0, 1, maxc).astype(np.float32)
    rc, gc, bc = ((maxc - c).astype(np.float32) / crs for c in (r, g, b))
    # the C expression mixes float operands with double literals: evaluated in double
    h = np.where(r == maxc, bc.astype(np.float64) - gc, np.where(g == maxc, 2.0 + rc.astype(np.float64) - bc, 4.0 + gc.astype(np.float64) - rc))
    h = h.astype(np.float32)                                         # assigned to `float h`
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    out = np.stack([np.where(ok, uh, 0), np.where(ok, us, 0), maxc], -1)
    return out.astype(np.uint8)


def hsv_to_rgb(hsv):
    """Pillow Convert.c hsv2rgb."""
    h, s, v = (hsv[..., i].astype(np.int32) for i in range(3))
    hf = h.astype(np.float32).astype(np.float64) * 6.0 / 255.0
    i = np.floor(hf).astype(np.int32)
    f = (hf - i.astype(np.float32).astype(np.float64)).astype(np.float32)
    fs = (s.astype(np.float32).astype(np.float64) / 255.0).astype(np.float32)
    vf = v.astype(np.float32).astype(np.float64)
    rnd = lambda x: np.where(x >= 0, np.floor(x + 0.5), np.ceil(x - 0.5)).astype(np.int32)        # noqa: E731  C round()
    p = np.clip(rnd(vf * (1.0 - fs.astype(np.float64))), 0, 255)
    q = np.clip(rnd(vf * (1.0 - fs.astype(np.float64) * f.astype(np.float64))), 0, 255)
    t = np.clip(rnd(vf * (1.0 - fs.astype(np.float64) * (1.0 - f.astype(np.float64)))), 0, 255)
    sel = i % 6
    r = np.choose(sel, [v, q, p, p, t, v])
    g = np.choose(sel, [t, v, v, q, p, p])
    b = np.choose(sel, [p, p, t, v, v, q])
    grey = s == 0
    return np.stack([np.where(grey, v, r), np.where(grey, v, g), np.where(grey, v, b)], -1).astype(np.uint8)


def adjust_hue(rgb, hue_factor):
    """torchvision 0.13.1 F.adjust_hue (functional_pil.py): HSV, h += uint8(hue_factor * 255) with wrap-around, back to RGB."""
    if not -0.5 <= hue_factor <= 0.5:
        raise ValueError(f"hue_factor ({hue_factor}) is not in [-0.5, 0.5].")
    hsv = rgb_to_hsv(rgb)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + int(np.uint8(np.int64(hue_factor * 255) & 0xFF))).astype(np.uint8)
    return hsv_to_rgb(hsv)


JITTER = {"brightness": adjust_brightness, "contrast": adjust_contrast, "saturation": adjust_saturation, "hue": adjust_hue}


def color_jitter(rgb, ops):
    """ops = [(name, factor), ...] in the order `random.shuffle` left them (preprocessing.py:102-104)."""
    for name, factor in ops:
        rgb = JITTER[name](rgb, factor)
    return rgb


def to_tensor_normalize(rgb, mean=None, std=None):
    """ToTensor (uint8 HWC -> float32 CHW / 255) + Normalize (preprocessing.py:12-18)."""
    x = rgb.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    if mean is not None:
        m, s = np.asarray(mean, np.float32)[:, None, None], np.asarray(std, np.float32)[:, None, None]
        x = (x - m) / s
    return x


# ----------------------------------------------------------------------------------------------- histogram matching
def match_histograms_u8(image, reference):
    """skimage.exposure.match_histograms(image, reference, channel_axis=-1) for uint8 inputs (published algorithm, >= 0.19:
    exposure/histogram_matching.py): per channel, CDFs from bincounts (template values with a zero count omitted), np.interp of the
    source quantiles, result assigned into an array of the image's dtype (truncating cast)."""
    out = np.empty(image.shape, image.dtype)
    for c in range(image.shape[-1]):
        src, tmpl = image[..., c].reshape(-1), reference[..., c].reshape(-1)
        src_counts = np.bincount(src)
        tmpl_counts = np.bincount(tmpl)
        tmpl_values = np.nonzero(tmpl_counts)[0]
        tmpl_counts = tmpl_counts[tmpl_values]
        src_q = np.cumsum(src_counts) / src.size
        tmpl_q = np.cumsum(tmpl_counts) / tmpl.size
        lut = np.interp(src_q, tmpl_q, tmpl_values)
        out[..., c] = lut[src].reshape(image.shape[:-1])
    return out


def transfer_histogram_style(img, reference, mode):
    """datasets/preprocessing.py:60-75 on uint8 HWC arrays: "rgb" / "hsv" / "saturation" / None (Pillow's HSV space through
    rgb_to_hsv / hsv_to_rgb above, which are pinned against Pillow itself)."""
    if mode is None:
        return img
    if mode == "rgb":
        return match_histograms_u8(img, reference)
    if mode not in ("hsv", "saturation"):
        raise ValueError(f"Unknown histogram transfer mode: {mode}")
    img_hsv, ref_hsv = rgb_to_hsv(img), rgb_to_hsv(reference)
    new = match_histograms_u8(img_hsv, ref_hsv)
    if mode == "saturation":
        new[:, :, 0] = img_hsv[:, :, 0]
        new[:, :, 2] = img_hsv[:, :, 2]
    return hsv_to_rgb(new)


# ----------------------------------------------------------------------------------------------- label path
def nearest_resize(a, out_h, out_w):
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST) for a 2-D array: src = min(floor(dst * src / dst_size), src - 1)."""
    h, w = a.shape[:2]
    ys = np.minimum(np.floor(np.arange(out_h) * (h / out_h)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(out_w) * (w / out_w)).astype(np.int64), w - 1)
    return a[ys][:, xs]


CITYSCAPES_TO_CODEPS = [(7, 0), (8, 1), (11, 2), (12, 2), (13, 3), (17, 4), (20, 5), (21, 6), (22, 7), (23, 8), (24, 9), (25, 10),
                        (26, 11), (27, 12), (32, 13), (33, 13)]                              # cityscapes.py:291-311
CODEPS_THINGS = [9, 10, 11, 12, 13]


def cityscapes_labels(semantic_ids, instance_ids, out_h, out_w, full_h, full_w, mapping=CITYSCAPES_TO_CODEPS, things=CODEPS_THINGS,
                      small_instance_area_full_res=4096, small_instance_weight=3):
    """cityscapes.py:166-246 without the centre / offset maps (oracle.mixup.offset_center restates those): labelIds /
    instanceIds at file resolution -> dict of the network-resolution label maps."""
    sem = nearest_resize(semantic_ids, out_h, out_w)
    inst = nearest_resize(instance_ids, out_h, out_w)
    sem_city = np.full_like(sem, 255, dtype=np.uint8)
    for a, b in mapping:
        sem_city[sem == a] = b
    class_instance = inst.astype(np.int64) - sem.astype(np.int64) * 1000
    thing = np.isin(sem_city, things)
    msk = thing.copy()
    msk[inst < 1000] = False
    inst_city = np.zeros(inst.shape, np.uint16)
    inst_city[msk] = (sem_city[msk].astype(np.int64) * 1000 + class_instance[msk]).astype(np.uint16)
    weights = np.ones(inst.shape, np.uint8)
    weights[sem_city == 255] = 0
    small = small_instance_area_full_res * (out_h / full_h) * (out_w / full_w)
    ids, areas = np.unique(inst_city, return_counts=True)
    for i, a in zip(ids, areas):
        if i != 0 and a < small:
            weights[inst_city == i] = small_instance_weight
    center_w = np.ones((1,) + inst.shape, np.uint8)
    center_w[0][sem_city == 255] = 0
    return {"semantic": sem_city, "semantic_weights": weights, "instance": inst_city.astype(np.int32), "thing_mask": thing.astype(np.uint8),
            "center_weights": center_w, "offset_weights": msk.astype(np.uint8)[None]}


def disparity_to_depth(disparity_u16, fx, baseline, out_h, out_w):
    """cityscapes.py:249-262."""
    d = disparity_u16.astype(np.float32)
    pos = d > 0
    d[pos] = (d[pos] - 1) / 256
    depth = np.zeros_like(d)
    depth[pos] = (baseline * fx) / d[pos]
    return nearest_resize(depth, out_h, out_w)


KITTI360_TO_CODEPS = [(7, 0), (8, 1), (11, 2), (34, 2), (12, 2), (13, 3), (35, 3), (17, 4), (37, 4), (20, 5), (21, 6), (22, 7), (23, 8), (24, 9),
                      (25, 10), (26, 11), (27, 12), (32, 13), (33, 13)]                       # kitti_360.py:370-390


def kitti360_labels(semantic_ids, instance_ids, out_h, out_w, full_h, full_w, mapping=KITTI360_TO_CODEPS, things=CODEPS_THINGS,
                    small_instance_area_full_res=4096, small_instance_weight=3):
    """kitti_360.py:236-300 without the centre / offset maps: every thing pixel belongs to an instance, `semantic_eval` drops
    instances below a tenth of the small-instance area, centre and offset weights are the thing mask."""
    sem = nearest_resize(semantic_ids, out_h, out_w)
    inst = nearest_resize(instance_ids, out_h, out_w)
    sem_city = np.full_like(sem, 255, dtype=np.uint8)
    for a, b in mapping:
        sem_city[sem == a] = b
    class_instance = inst.astype(np.int64) - sem.astype(np.int64) * 1000
    thing = np.isin(sem_city, things)
    inst_city = np.zeros(inst.shape, np.uint16)
    inst_city[thing] = (sem_city[thing].astype(np.int64) * 1000 + class_instance[thing]).astype(np.uint16)
    weights = np.ones(inst.shape, np.uint8)
    weights[sem_city == 255] = 0
    sem_eval = sem_city.copy()
    small = small_instance_area_full_res * (out_h / full_h) * (out_w / full_w)
    ids, areas = np.unique(inst_city, return_counts=True)
    for i, a in zip(ids, areas):
        if i == 0:
            continue
        if a < small:
            weights[inst_city == i] = small_instance_weight
        if a < small * .1:
            sem_eval[inst_city == i] = 255
    tm = thing.astype(np.uint8)
    return {"semantic": sem_city, "semantic_eval": sem_eval, "semantic_weights": weights, "instance": inst_city.astype(np.int32),
            "thing_mask": tm, "center_weights": tm[None], "offset_weights": tm[None]}


def lidar_depth(points, lidar_to_rect, camera_matrix, full_h, full_w, out_h, out_w):
    """kitti_360.py:303-330: Velodyne points [N,4] float32 -> sparse depth map at the network resolution."""
    pcl = points.astype(np.float32).reshape(-1, 4).copy()
    pcl[:, 3] = 1
    cam = (lidar_to_rect @ pcl.T).T[:, :3].T[None]
    proj = camera_matrix[:3, :3].reshape(1, 3, 3) @ cam
    depth = proj[:, 2, :]
    depth[depth == 0] = -1e-6
    u = np.round(proj[:, 0, :] / np.abs(depth)).astype(int)
    v = np.round(proj[:, 1, :] / np.abs(depth)).astype(int)
    dm = np.zeros((full_h, full_w))
    mask = (u >= 0) & (u < full_w) & (v >= 0) & (v < full_h) & (depth > 0) & (depth < 80)
    dm[v[mask], u[mask]] = depth[mask]
    return nearest_resize(dm, out_h, out_w)
