"""CPU oracle: pseudo-label mix-up (camera-to-camera warp, cut mix, centre / offset regeneration).  TEST INFRA.

numpy restatement of datasets/mixup.py:28-66,89-146,211-229,243-257,314-344,406-489 (the `cut_mixup` strategy both
adaptation YAMLs select, cfg/adapt_cityscapes_kitti_360.yaml:31-43), misc/image_warper.py:54-87,
misc/camera_model.py:43-70 and datasets/dataset.py:126-177.  Float64 arithmetic in the reference's operation order;
integer outputs bit-exact.  PINNED by tests/golden/mixup.npz (real reference run, tests/golden/make_golden.py).
"""
import numpy as np

from . import panoptic


def _sqrt32(x, ieee):
    """float32 square root.  ieee=False: torch's CPU kernel, which is what the reference computes in the development
    container and what the golden vectors hold -- its vectorised sqrtf is NOT correctly rounded (0.9 % of inputs are
    off by one ulp).  ieee=True: the correctly rounded root every GPU (the reference's CUDA sqrtf as well as HIP's)
    returns; the HIP kernels are compared bit for bit against this variant."""
    if ieee:
        return np.sqrt(x)
    import torch
    return torch.sqrt(torch.from_numpy(np.ascontiguousarray(x))).numpy()


def c2c_grid(intr_in, intr_out, hs, ws, ho, wo, depth_val=1.0, ieee_sqrt=False):
    """Unnormalised sampling coordinates (ix, iy) [ho, wo] float64 in an input image of size (hs, ws) for one sample:
    viewing rays of the OUTPUT camera in float32 (camera_model.py:52-70 on the float32 pixel grid of
    image_warper.py:62-66), the point cloud in float64 (image_warper.py:83-85; depth = ones * depth_val,
    mixup.py:223), projection with the INPUT camera (camera_model.py:43-50; z clamped at 1e-5, mixup.py:49),
    normalisation to [-1, 1] (mixup.py:61-62) and ATen's align_corners=True un-normalisation."""
    fx_o, fy_o, cx_o, cy_o = (np.float32(v) for v in intr_out)
    fx_i, fy_i, cx_i, cy_i = (np.float64(np.float32(v)) for v in intr_in)
    u = np.broadcast_to(np.arange(wo, dtype=np.float32)[None, :], (ho, wo))
    v = np.broadcast_to(np.arange(ho, dtype=np.float32)[:, None], (ho, wo))
    rx = (u - cx_o) / fx_o
    ry = (v - cy_o) / fy_o
    norm = _sqrt32(rx * rx + ry * ry + np.float32(1.0), ieee_sqrt)
    rx, ry, rz = rx / norm, ry / norm, np.float32(1.0) / norm
    assert rx.dtype == np.float32 and rz.dtype == np.float32
    s = np.float64(depth_val) / np.abs(rz.astype(np.float64))
    x3, y3, z3 = s * rx.astype(np.float64), s * ry.astype(np.float64), s * rz.astype(np.float64)
    z3 = np.maximum(z3, 1e-5)
    u2 = (x3 / z3) * fx_i + cx_i
    v2 = (y3 / z3) * fy_i + cy_i
    un = (u2 / (ws - 1) - 0.5) * 2
    vn = (v2 / (hs - 1) - 0.5) * 2
    return ((un + 1) / 2) * (ws - 1), ((vn + 1) / 2) * (hs - 1)


def grid_sample(planes, ix, iy, mode, padding):
    """planes [C,hs,ws] -> float64 [C,ho,wo]: F.grid_sample(planes.double(), ., mode, padding, align_corners=True)
    (mixup.py:226-227) from unnormalised coordinates."""
    c, hs, ws = planes.shape
    src = planes.astype(np.float64)
    if padding == "border":
        ix, iy = np.clip(ix, 0, ws - 1), np.clip(iy, 0, hs - 1)

    def gather(yy, xx):
        ok = (xx >= 0) & (xx <= ws - 1) & (yy >= 0) & (yy <= hs - 1)
        yi, xi = np.where(ok, yy, 0).astype(np.int64), np.where(ok, xx, 0).astype(np.int64)
        return np.where(ok[None], src[:, yi, xi], 0.0)

    if mode == "nearest":
        return gather(np.rint(iy), np.rint(ix))
    xw, yn = np.floor(ix), np.floor(iy)
    w = ix - xw
    e = 1 - w
    n = iy - yn
    s = 1 - n
    return gather(yn, xw) * (s * e) + gather(yn, xw + 1) * (s * w) + gather(yn + 1, xw) * (n * e) + gather(yn + 1, xw + 1) * (n * w)


def warp_c2c(intr_in, intr_out, in_maps, out_hw, mode="bilinear", padding="border", depth_val=1.0, ieee_sqrt=False):
    """Mixup.warp_c2c(cam_model_src=intr_in, cam_model_tgt=intr_out, in_src=in_maps, in_tgt=<anything of out_hw>):
    in_maps [B,C,hs,ws] (or [B,hs,ws]) -> float64 [B,C,ho,wo]."""
    if in_maps.ndim == 3:
        in_maps = in_maps[:, None]
    b, c, hs, ws = in_maps.shape
    out = np.empty((b, c) + tuple(out_hw), np.float64)
    for i in range(b):
        ix, iy = c2c_grid(intr_in[i], intr_out[i], hs, ws, out_hw[0], out_hw[1], depth_val, ieee_sqrt)
        out[i] = grid_sample(in_maps[i], ix, iy, mode, padding)
    return out


def cut_rects(split_ids, h, w, nof_vert_split, nof_hor_split):
    """(y0, y1, x0, x1) of each chosen cell; end-exclusive, INCLUDING the reference's '- 1' on the slice ends
    (mixup.py:136-144): the last row / column of every cell stays source."""
    ch, cw = h // nof_vert_split, w // nof_hor_split
    out = []
    for sid in split_ids:
        mx, my = int(sid) % nof_hor_split, int(sid) // nof_hor_split
        out.append((my * ch, (my + 1) * ch - 1, mx * cw, (mx + 1) * cw - 1))
    return np.asarray(out, np.int32).reshape(-1, 4)


def offset_center(inst, sigma=8):
    """Dataset.get_offset_center (dataset.py:126-177) for one [H,W] id map -> (offset [2,H,W], center [1,H,W]) fp32."""
    size = 6 * sigma + 3
    x = np.arange(0, size, 1, float)
    y = x[:, None]
    x0 = y0 = 3 * sigma + 1
    gaussian = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))
    h, w = inst.shape
    center = np.zeros((1, h, w), np.float32)
    offset = np.zeros((2, h, w), np.float32)
    yc, xc = np.mgrid[0:h, 0:w].astype(np.float32)
    for iid in np.unique(inst):
        if iid == 0:
            continue
        ys, xs = np.where(inst == iid)
        cy, cx = np.mean(ys), np.mean(xs)                       # float64
        yy, xx = int(np.round(cy)), int(np.round(cx))
        if not 0 <= xx < w or not 0 <= yy < h:
            continue
        ul = int(np.round(xx - 3 * sigma - 1)), int(np.round(yy - 3 * sigma - 1))
        br = int(np.round(xx + 3 * sigma + 2)), int(np.round(yy + 3 * sigma + 2))
        c, d = max(0, -ul[0]), min(br[0], w) - ul[0]
        a, b = max(0, -ul[1]), min(br[1], h) - ul[1]
        cc, dd = max(0, ul[0]), min(br[0], w)
        aa, bb = max(0, ul[1]), min(br[1], h)
        center[0, aa:bb, cc:dd] = np.maximum(center[0, aa:bb, cc:dd], gaussian[a:b, c:d])
        offset[0, ys, xs] = cy - yc[ys, xs].astype(np.float64)   # float64 difference, rounded once on assignment
        offset[1, ys, xs] = cx - xc[ys, xs].astype(np.float64)
    return offset, center


def get_off_cen(inst_maps, sigma=8):
    """Mixup.get_off_cen (mixup.py:243-257): [B,(1,)H,W] -> (center [B,1,H,W], offset [B,2,H,W])."""
    maps = inst_maps.reshape((inst_maps.shape[0],) + inst_maps.shape[-2:])
    pairs = [offset_center(m, sigma) for m in maps]
    return np.stack([p[1] for p in pairs]), np.stack([p[0] for p in pairs])


def do_cut_mixup(item, thing_list, split_ids, geom_augment=True, nof_vert_split=2, nof_hor_split=4, ieee_sqrt=False):
    """Mixup.do_mixup("cut_mixup", ...) (mixup.py:406-489) on numpy inputs; split_ids [B][nof_segments] are the cells
    np.random.choice drew (mixup.py:108).  -> dict(rgb, semantic, instance, center, offset)."""
    img_src, img_tgt = item["rgb_src"], item["rgb_tgt"]
    sem_src, inst_src = item["semantic_src"], item["instance_src"]              # [B,1,H,W]
    sem_tgt = item["semantic_pred_tgt_ema"]
    b, _, h, w = img_src.shape
    inst_tgt = panoptic.panoptic_fusion(sem_tgt, item["center_pred_tgt_ema"], item["offset_pred_tgt_ema"], thing_list)[1]
    if geom_augment:
        cs, ct = item["camera_model_src"], item["camera_model_tgt"]
        img_tgt = warp_c2c(ct, cs, img_tgt, (h, w), "bilinear", "zeros", ieee_sqrt=ieee_sqrt)
        inst_tgt = warp_c2c(ct, cs, inst_tgt, (h, w), "nearest", "zeros", ieee_sqrt=ieee_sqrt)[:, 0].astype(inst_tgt.dtype)
        sem_tgt = warp_c2c(ct, cs, sem_tgt, (h, w), "nearest", "border", ieee_sqrt=ieee_sqrt)[:, 0].astype(sem_tgt.dtype)
    assert img_tgt.shape == img_src.shape, "mixup.py:91,150: differently sized images need geom_augment"
    rgb = img_src.astype(np.float32).copy()
    sem, inst = sem_src[:, 0].copy(), inst_src[:, 0].copy()
    for i in range(b):
        for y0, y1, x0, x1 in cut_rects(split_ids[i], h, w, nof_vert_split, nof_hor_split):
            rgb[i, :, y0:y1, x0:x1] = img_tgt[i, :, y0:y1, x0:x1].astype(np.float32)
            sem[i, y0:y1, x0:x1] = sem_tgt[i, y0:y1, x0:x1]
            inst[i, y0:y1, x0:x1] = inst_tgt[i, y0:y1, x0:x1].astype(np.int32)
    center, offset = get_off_cen(inst)
    return {"rgb": rgb, "semantic": sem, "instance": inst, "center": center, "offset": offset}


def do_class_mixup(item, thing_list, pick, geom_augment=True, ieee_sqrt=False):
    """Mixup.do_mixup("class_mixup", ...) (mixup.py:261-311,406-449) on numpy inputs; `pick` = the indices np.random.choice drew
    from the sorted unique source labels (mixup.py:69-75).  -> dict(rgb, semantic, instance, center, offset)."""
    import torch
    img_src, img_tgt = item["rgb_src"], item["rgb_tgt"]
    sem_src, inst_src = item["semantic_src"], item["instance_src"]              # [B,1,H,W]
    sem_tgt = item["semantic_pred_tgt_ema"]
    b, _, h, w = img_src.shape
    inst_tgt = panoptic.panoptic_fusion(sem_tgt, item["center_pred_tgt_ema"], item["offset_pred_tgt_ema"], thing_list)[1]
    if geom_augment:
        cs, ct = item["camera_model_src"], item["camera_model_tgt"]
        img_tgt = warp_c2c(ct, cs, img_tgt, (h, w), "bilinear", "zeros", ieee_sqrt=ieee_sqrt).astype(np.float32)
        inst_tgt = warp_c2c(ct, cs, inst_tgt, (h, w), "nearest", "zeros", ieee_sqrt=ieee_sqrt)[:, 0].astype(inst_tgt.dtype)
        sem_tgt = warp_c2c(ct, cs, sem_tgt, (h, w), "nearest", "border", ieee_sqrt=ieee_sqrt)[:, 0].astype(sem_tgt.dtype)
    else:                                                                        # mixup.py:270-278: plain resizes
        f = torch.nn.functional.interpolate
        nthreads = torch.get_num_threads()
        torch.set_num_threads(1)             # the CPU kernel's rounding depends on how the work is split (goldens: one thread)
        try:
            img_tgt = f(torch.from_numpy(img_tgt), (h, w), mode="bilinear", align_corners=False).numpy()
        finally:
            torch.set_num_threads(nthreads)
        sem_tgt = f(torch.from_numpy(sem_tgt)[:, None], (h, w), mode="nearest")[:, 0].numpy()
        inst_tgt = f(torch.from_numpy(inst_tgt)[:, None].float(), (h, w), mode="nearest")[:, 0].numpy().astype(np.int32)
    classes = np.unique(sem_src)[np.asarray(pick)]
    msk = np.isin(sem_src, classes)                                              # [B,1,H,W]
    rgb = np.where(msk, img_tgt.astype(np.float32), img_src.astype(np.float32))
    sem = np.where(msk[:, 0], sem_tgt, sem_src[:, 0])
    inst = np.where(msk[:, 0], inst_tgt.astype(np.int32), inst_src[:, 0])
    center, offset = get_off_cen(inst)
    return {"rgb": rgb, "semantic": sem, "instance": inst, "center": center, "offset": offset}


# ------------------------------------------------------------------------------------------ conf_instance_mixup
def kornia_distance_transform(image, kernel_size=3, h=0.35):
    """kornia.contrib.distance_transform (kornia is listed UNPINNED in the reference's requirements.txt:16 and is not in this
    image: **parity unpinned** for this one function -- restated from kornia's published algorithm, Pham et al., "DTLayer",
    2021, as kornia 0.6 implements it): cascaded 3x3 convolutions of the growing boundary with exp(-|d| / h), replicate
    border; a pixel reached in iteration i gets  i * kernel_size // 2 - h log(sum)  once, float32 throughout.
    image [H, W] float32 (non-zero = object) -> [H, W] float32.  The nine products are summed in row-major kernel order;
    that order is this restatement's definition (a library convolution leaves it open) and the HIP kernel follows it."""
    img = np.ascontiguousarray(image, np.float32)
    hh, ww = img.shape
    n_iters = int(np.ceil(max(hh, ww) / (kernel_size // 2)))
    r = kernel_size // 2
    ax = np.arange(kernel_size, dtype=np.float32) - np.float32(r)
    kern = np.exp(np.hypot(ax[:, None], ax[None, :]).astype(np.float32) / np.float32(-h)).astype(np.float32)
    out = np.zeros_like(img)
    boundary = img.copy()
    for i in range(n_iters):
        pad = np.pad(boundary, r, mode="edge")
        s = np.zeros_like(img)
        for ky in range(kernel_size):
            for kx in range(kernel_size):
                s = (s + kern[ky, kx] * pad[ky:ky + hh, kx:kx + ww]).astype(np.float32)
        with np.errstate(divide="ignore"):
            cdt = (np.float32(-h) * np.log(s)).astype(np.float32)
        cdt[np.isposinf(cdt)] = 0.0
        mask = cdt > 0
        if not mask.any():
            break
        offset = i * kernel_size // 2
        out = np.where(mask, (out + (np.float32(offset) + cdt)).astype(np.float32), out)
        boundary = np.where(mask, np.float32(1.0), boundary)
    return out


def conf_mask(inst_tgt, thing_src, min_inst_size):
    """Mixup.get_conf_mask (mixup.py:542-652) for ONE batch item, on masks instead of index lists.  inst_tgt [H,W]: instance ids
    of the teacher's panoptic fusion (0 = no thing); thing_src [H,W] bool: thing pixels of the source labels.
    Instances are placed in ascending id order, each at the column of its centre row that is farthest (kornia distance
    transform) from every thing already in the image, shifted back inside the image; where a new instance overlaps one pasted
    earlier, the NEW one is cropped.  -> (dst [H,W] int64: id of the instance pasted on a source pixel, 0 = none,
    shift {id: dx}: the pasted pixel (y, x) shows target pixel (y, x - dx))."""
    h, w = inst_tgt.shape
    things = thing_src.astype(bool).copy()
    dst = np.zeros((h, w), np.int64)
    shift = {}
    for cid in np.unique(inst_tgt):
        cid = int(cid)
        t = inst_tgt == cid
        if cid == 0 or t.sum() < min_inst_size:
            continue
        ys, xs = np.nonzero(t)
        # round half to even of the float32 mean (torch.round of .float().mean(0), mixup.py:580-581)
        cy, cx = (int(np.rint(np.float32(v.astype(np.float32).mean(dtype=np.float32)))) for v in (ys, xs))
        dist_row = kornia_distance_transform(things.astype(np.float32))[cy]
        dx = int(np.argmax(dist_row)) - cx
        lo, hi = int(xs.min()) + dx, int(xs.max()) + dx
        if lo < 0:
            dx -= lo
        if hi > w - 1:
            dx -= hi - (w - 1)
        s = np.zeros_like(t)
        s[ys, xs + dx] = True
        shift[cid] = dx
        dst[s & (dst == 0)] = cid                       # overlap with an earlier instance: the earlier one keeps the pixel
        things |= s
    return dst, shift


def do_conf_instance_mixup(item, thing_list, geom_augment, conf_thresh, min_inst_size, ieee_sqrt=False):
    """Mixup.do_mixup("conf_instance_mixup", ...) (mixup.py:347-404, 406-466, 542-652) on numpy inputs, batch size 1 (the
    reference reads batch item 0's instance map for every item's centre, mixup.py:581).
    -> dict(rgb, semantic, instance, center, offset)."""
    img_src, img_tgt = item["rgb_src"], item["rgb_tgt"]
    sem_src, inst_src = item["semantic_src"], item["instance_src"]              # [B,1,H,W]
    sem_tgt = item["semantic_pred_tgt_ema"]
    b, _, h, w = img_src.shape
    assert b == 1, "mixup.py:581 indexes batch item 0"
    inst_tgt = panoptic.panoptic_fusion(sem_tgt, item["center_pred_tgt_ema"], item["offset_pred_tgt_ema"], thing_list)[1]
    if geom_augment:
        cs, ct = item["camera_model_src"], item["camera_model_tgt"]
        img_tgt = warp_c2c(ct, cs, img_tgt, (h, w), "bilinear", "zeros", ieee_sqrt=ieee_sqrt).astype(np.float32)
        inst_tgt = warp_c2c(ct, cs, inst_tgt, (h, w), "nearest", "zeros", ieee_sqrt=ieee_sqrt)[:, 0].astype(inst_tgt.dtype)
        sem_tgt = warp_c2c(ct, cs, sem_tgt, (h, w), "nearest", "border", ieee_sqrt=ieee_sqrt)[:, 0].astype(sem_tgt.dtype)
    assert img_tgt.shape == img_src.shape, "differently sized images need geom_augment"
    center_t, offset_t = get_off_cen(inst_tgt)                                   # mixup.py:444: labels of the (warped) prediction
    inst_conf = panoptic.panoptic_fusion(sem_tgt, center_t, offset_t, thing_list, conf_thresh)[1]       # mixup.py:546-550
    rgb = img_src.astype(np.float32).copy()
    sem, inst = sem_src[:, 0].copy(), inst_src[:, 0].copy()
    for i in range(b):
        dst, shift = conf_mask(inst_conf[i].astype(np.int64), inst_src[i, 0] != 0, min_inst_size)
        for cid, dx in shift.items():
            ys, xs = np.nonzero(dst == cid)
            rgb[i][:, ys, xs] = img_tgt[i][:, ys, xs - dx].astype(np.float32)
            sem[i][ys, xs] = sem_tgt[i][ys, xs - dx]
            inst[i][ys, xs] = inst_conf[i][ys, xs - dx].astype(np.int32)
    center, offset = get_off_cen(inst)
    return {"rgb": rgb, "semantic": sem, "instance": inst, "center": center, "offset": offset}
