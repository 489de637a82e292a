"""CPU oracle: camera geometry, image warping and every per-pixel loss on the hot path.  TEST INFRA.

Vectorised over the batch (intrinsics are a [B,4] fp32 tensor (fx,fy,cx,cy)) instead of the
reference's per-sample Python loops; arithmetic order per element follows the reference so results
agree to fp32 rounding.  Pinned by tests/golden/{warp,ssim,recon,smooth,semloss,instloss}_*.npz.
"""
import torch
import torch.nn.functional as F

SSIM_C1, SSIM_C2 = 0.01 ** 2, 0.03 ** 2       # algos/depth.py:125-126


# ----------------------------------------------------------------------------- geometry
def scale_intrinsics(intr, full_w, full_h, w, h):
    """misc/camera_model.py:37-43: (fx,cx) * w/W, (fy,cy) * h/H.  intr: [B,4] = fx,fy,cx,cy."""
    su, sv = w / full_w, h / full_h
    return intr * torch.tensor([su, sv, su, sv], dtype=intr.dtype)


def backproject(intr, depth):
    """depth [B,1,H,W] -> points [B,3,H,W] (misc/image_warper.py:56-89 + camera_model.py:54-71):
    ray = ((u-cx)/fx, (v-cy)/fy, 1)/|.| ;  P = depth/|ray_z| * ray."""
    b, _, h, w = depth.shape
    u = torch.arange(w, dtype=depth.dtype).view(1, 1, w).expand(b, h, w)
    v = torch.arange(h, dtype=depth.dtype).view(1, h, 1).expand(b, h, w)
    fx, fy, cx, cy = (intr[:, i].view(b, 1, 1) for i in range(4))
    rx = (u - cx) / fx
    ry = (v - cy) / fy
    norm = torch.sqrt(rx ** 2 + ry ** 2 + 1.0)
    rx, ry, rz = rx / norm, ry / norm, 1.0 / norm
    d = depth[:, 0]
    return torch.stack([d / rz.abs() * rx, d / rz.abs() * ry, d / rz.abs() * rz], 1)


def warp_grid(intr, depth, T):
    """-> normalised sampling grid [B,H,W,2] (misc/image_warper.py:100-144,20-50): back-project,
    T @ [P;1], divide by w, clamp z >= 1e-5, project, normalise with (W-1),(H-1)."""
    b, _, h, w = depth.shape
    p = backproject(intr, depth).view(b, 3, -1)
    ph = torch.cat([p, torch.ones(b, 1, h * w, dtype=p.dtype)], 1)
    q = torch.bmm(T, ph)
    q = q[:, :3] / q[:, 3:4]
    x, y, z = q[:, 0], q[:, 1], q[:, 2].clamp(min=1e-5)
    fx, fy, cx, cy = (intr[:, i].view(b, 1) for i in range(4))
    u2 = (x / z) * fx + cx
    v2 = (y / z) * fy + cy
    un = (u2 / (w - 1) - 0.5) * 2
    vn = (v2 / (h - 1) - 0.5) * 2
    return torch.stack([un, vn], 2).view(b, h, w, 2)


def warp_image(intr, src, depth, T, mode="bilinear"):
    """misc/image_warper.py:153-184: grid_sample(border, align_corners=True)."""
    return F.grid_sample(src, warp_grid(intr, depth, T), mode=mode, padding_mode="border",
                         align_corners=True)


# ----------------------------------------------------------------------------- photometric
def ssim_loss(x, y):
    """[B,C,H,W] x2 -> clamp((1-SSIM)/2,0,1) per channel (algos/depth.py:110-155): reflect pad 1,
    3x3 mean (stride 1), C1=1e-4, C2=9e-4."""
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    y = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x, mu_y = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
    sig_x = F.avg_pool2d(x ** 2, 3, 1) - mu_x ** 2
    sig_y = F.avg_pool2d(y ** 2, 3, 1) - mu_y ** 2
    sig_xy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + SSIM_C1) * (2 * sig_xy + SSIM_C2)
    d = (mu_x ** 2 + mu_y ** 2 + SSIM_C1) * (sig_x + sig_y + SSIM_C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def photometric(pred, target, alpha=0.85):
    """-> [B,1,H,W] (algos/depth.py:221-237)."""
    l1 = (pred - target).abs().mean(1, True)
    return alpha * ssim_loss(pred, target).mean(1, True) + (1 - alpha) * l1


def _down(x, h, w):
    return F.interpolate(x, (h, w), mode="bilinear", align_corners=False)   # algos/depth.py:158-173


def reconstruction_loss(intr, images, depth, poses, num_scales=5, noise=None, alpha=0.85,
                        return_maps=False):
    """Multi-scale photometric loss with auto-masking (algos/depth.py:239-326).
    images = (I_t, I_t-1, I_t+1) each [B,3,H,W]; depth [B,1,H,W]; poses = (T_t->t-1, T_t->t+1) [B,4,4];
    intr [B,4] at full resolution.  ``noise``: list of [B,2,Hs,Ws] tensors (already scaled by 1e-5)
    or None -> ``torch.randn(shape) * 1e-5`` drawn per scale in scale order like the reference."""
    b, _, full_h, full_w = images[0].shape
    loss = torch.zeros((), dtype=depth.dtype)
    maps = []
    for s in range(num_scales):
        w, h = full_w // 2 ** s, full_h // 2 ** s
        k = scale_intrinsics(intr, full_w, full_h, w, h)
        tgt, dep = _down(images[0], h, w), _down(depth, h, w)
        srcs = [_down(f, h, w) for f in images[1:]]
        recon = [photometric(warp_image(k, src, dep, poses[i]), tgt, alpha)
                 for i, src in enumerate(srcs)]
        ident = torch.cat([photometric(src, tgt, alpha) for src in srcs], 1)
        n = torch.randn(ident.shape) * .00001 if noise is None else noise[s]
        comb = torch.cat(recon + [ident + n], 1)
        per_px = comb.min(1)[0]
        maps.append(per_px)
        loss = loss + per_px.mean() / 2 ** s
    loss = loss / num_scales
    return (loss, maps) if return_maps else loss


def smoothness_loss(img, disp):
    """Edge-aware smoothness (algos/depth.py:58-107)."""
    d = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
    gdx = (d[:, :, :, :-1] - d[:, :, :, 1:]).abs()
    gdy = (d[:, :, :-1, :] - d[:, :, 1:, :]).abs()
    gix = (img[:, :, :, :-1] - img[:, :, :, 1:]).abs().mean(1, True)
    giy = (img[:, :, :-1, :] - img[:, :, 1:, :]).abs().mean(1, True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


# ----------------------------------------------------------------------------- semantic / instance
def semantic_loss(probs, target, pixel_weights, class_weights=None, top_k=0.2, ignore_index=255, ignore_labels=None):
    """Hard-pixel-mined CE applied to *softmax probabilities* (algos/semantic_seg.py:45-67):
    CE(=log_softmax again + weighted NLL, ignore 255, reduction none) * pixel_weights; flatten the
    whole batch; top-k (k=int(top_k*numel)); mean.  ignore_labels (semantic_seg.py:50-55, with the weight filter of
    :34-40): targets with those labels become ignore_index IN PLACE, their channels are sliced out of the probabilities
    (the remaining targets keep indexing the sliced tensor with their original values) and class_weights (one per original
    class) is filtered the same way."""
    if ignore_labels:
        for label in ignore_labels:
            target[target == label] = ignore_index
        keep = [c for c in range(probs.shape[1]) if c not in ignore_labels]
        probs = probs[:, keep]
        if class_weights is not None:
            class_weights = [w_ for c, w_ in enumerate(class_weights) if c not in ignore_labels]
    w = None if class_weights is None else torch.as_tensor(class_weights, dtype=probs.dtype)
    per_px = F.cross_entropy(probs, target.long(), w, ignore_index=ignore_index, reduction="none")
    per_px = per_px * pixel_weights
    if top_k < 1.0:
        flat = per_px.contiguous().view(-1)
        flat, _ = torch.topk(flat, int(top_k * flat.numel()))
        return flat.mean()
    return per_px.mean()


def center_loss(pred, target):
    """algos/instance_seg.py:12-25 (pixel weights are ignored by the reference)."""
    return ((pred - target) ** 2).mean()


def offset_loss(pred, target):
    """algos/instance_seg.py:28-41."""
    return (pred - target).abs().mean()


def binary_mask_loss(probs, target, ignore_index=255):
    """algos/instance_seg.py:44-51: CE on softmaxed thing probabilities, plain mean over all pixels."""
    return F.cross_entropy(probs, target.long(), ignore_index=ignore_index, reduction="none").mean()
