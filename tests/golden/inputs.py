"""Deterministic synthetic inputs shared by tests/golden/make_golden.py (which feeds them to the
real reference) and by the parity tests (which feed them to the oracle and to the HIP path).

Pure numpy (PCG64, platform independent) -> torch; nothing here touches the reference.  Fixtures
therefore only need to store seeds, shapes and *expected outputs*, not multi-megabyte weights.
"""
import math
from collections import OrderedDict

import numpy as np
import torch


def _rng(seed):
    return np.random.default_rng(seed)


def normal(shape, seed, scale=1.0, shift=0.0):
    return torch.from_numpy((_rng(seed).standard_normal(shape) * scale + shift).astype(np.float32))


def uniform(shape, seed, lo=0.0, hi=1.0):
    return torch.from_numpy(_rng(seed).uniform(lo, hi, shape).astype(np.float32))


def fill_state_dict(shapes, seed):
    """shapes: OrderedDict name -> tuple.  Conv/linear weights ~ N(0, 2/fan_in), biases small, BN
    affine / running stats non-trivial so that BN folding bugs show."""
    out = OrderedDict()
    for i, (name, shape) in enumerate(shapes.items()):
        shape = tuple(shape)
        s = seed * 1000 + i
        if name.endswith("num_batches_tracked"):
            out[name] = torch.tensor(0, dtype=torch.long)
        elif name.endswith("running_var"):
            out[name] = uniform(shape, s, 0.5, 1.5)
        elif name.endswith("running_mean"):
            out[name] = normal(shape, s, 0.1)
        elif len(shape) == 1 and name.endswith("weight"):          # BN gamma
            out[name] = uniform(shape, s, 0.5, 1.5)
        elif len(shape) == 1:                                      # bias / BN beta
            out[name] = normal(shape, s, 0.05)
        else:
            fan_in = int(np.prod(shape[1:]))
            out[name] = normal(shape, s, math.sqrt(2.0 / fan_in))
    return out


def shapes_of(state_dict):
    return OrderedDict((k, tuple(v.shape)) for k, v in state_dict.items())


def feature_pyramid(num_ch_enc, batch, img_h, img_w, seed):
    """Encoder-like features (post-ReLU => non-negative) at strides 2..32."""
    feats = []
    for i, c in enumerate(num_ch_enc):
        h, w = img_h // 2 ** (i + 1), img_w // 2 ** (i + 1)
        feats.append(normal((batch, int(c), h, w), seed * 100 + i).clamp_(min=0))
    return feats


def smooth_image(batch, h, w, seed, channels=3):
    """Low-pass noise in [0,1], then ImageNet-style normalisation (SURVEY.md §8d synthetic rgb)."""
    g = _rng(seed)
    coarse = g.uniform(0, 1, (batch, channels, max(h // 8, 2), max(w // 8, 2))).astype(np.float32)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)            # the interpolation kernel's rounding depends on how the work is split across threads
    try:
        img = torch.nn.functional.interpolate(torch.from_numpy(coarse), (h, w), mode="bilinear",
                                              align_corners=False)
    finally:
        torch.set_num_threads(nthreads)
    img = img + torch.from_numpy(g.normal(0, 0.02, (batch, channels, h, w)).astype(np.float32))
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)[:, :channels]
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)[:, :channels]
    return ((img.clamp(0, 1) - mean) / std).contiguous()


def triplet(batch, h, w, seed):
    """(I_t, I_t-1, I_t+1): neighbours are the centre frame shifted by -/+2 px + noise."""
    f0 = smooth_image(batch, h, w, seed)
    g = _rng(seed + 7)
    fm = torch.roll(f0, -2, 3) + torch.from_numpy(g.normal(0, 0.01, f0.shape).astype(np.float32))
    fp = torch.roll(f0, 2, 3) + torch.from_numpy(g.normal(0, 0.01, f0.shape).astype(np.float32))
    return f0, fm.contiguous(), fp.contiguous()


def intrinsics(batch, h, w, seed=0):
    """Cityscapes-like (fx,fy,cx,cy) scaled to (h,w), jittered per sample."""
    base = np.array([2262.52 * w / 2048, 2265.30 * h / 1024, 1096.98 * w / 2048, 513.137 * h / 1024])
    jit = 1.0 + 0.05 * _rng(seed + 11).standard_normal((batch, 4))
    return torch.from_numpy((base[None] * jit).astype(np.float32))


def depth_map(batch, h, w, seed):
    """Positive depth in ~[0.5, 80] with smooth structure."""
    g = _rng(seed + 3)
    coarse = g.uniform(0, 1, (batch, 1, max(h // 16, 2), max(w // 16, 2))).astype(np.float32)
    d = torch.nn.functional.interpolate(torch.from_numpy(coarse), (h, w), mode="bilinear",
                                        align_corners=False)
    return (0.5 + 80.0 * d ** 2).contiguous()


def pose(batch, seed, rot=0.02, trans=0.3):
    """Small rigid transforms [B,4,4] from axis-angle / translation."""
    g = _rng(seed + 5)
    aa = (g.standard_normal((batch, 3)) * rot).astype(np.float64)
    t = (g.standard_normal((batch, 3)) * trans).astype(np.float64)
    out = np.tile(np.eye(4), (batch, 1, 1))
    for b in range(batch):
        th = np.linalg.norm(aa[b])
        k = aa[b] / (th + 1e-12)
        kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        out[b, :3, :3] = np.eye(3) + math.sin(th) * kx + (1 - math.cos(th)) * kx @ kx
        out[b, :3, 3] = t[b]
    return torch.from_numpy(out.astype(np.float32))


def semantic_labels(batch, h, w, seed, num_classes=14, block=8, void_frac=0.05):
    g = _rng(seed + 13)
    bh, bw = -(-h // block), -(-w // block)
    lab = g.integers(0, num_classes, (batch, bh, bw))
    lab = np.repeat(np.repeat(lab, block, 1), block, 2)[:, :h, :w]
    void = g.uniform(0, 1, (batch, bh, bw)) < void_frac
    void = np.repeat(np.repeat(void, block, 1), block, 2)[:, :h, :w]
    lab = np.where(void, 255, lab)
    return torch.from_numpy(lab.astype(np.uint8))


def pixel_weights(batch, h, w, seed):
    g = _rng(seed + 17)
    return torch.from_numpy(g.choice(np.array([1, 3], np.uint8), (batch, h, w)))


def center_heatmap(batch, h, w, seed, n_max=20, sigma=8.0):
    g = _rng(seed + 19)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.zeros((batch, 1, h, w), np.float32)
    for b in range(batch):
        for _ in range(int(g.integers(1, n_max + 1))):
            cy, cx = g.uniform(0, h), g.uniform(0, w)
            out[b, 0] = np.maximum(out[b, 0], np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sigma ** 2)))
    return torch.from_numpy(out)


def offsets(batch, h, w, seed, mag=64.0):
    return uniform((batch, 2, h, w), seed + 23, -mag, mag)


def instance_labels(semantic, seed, things=(9, 10, 11, 12, 13), block=8, per_class=5):
    """Instance-id map [B,1,H,W] int32 for a blocky semantic map [B,H,W]: thing blocks get class*1000 + k (several
    disjoint blocks may share an id, like occluded objects), stuff / void pixels 0."""
    g = _rng(seed + 29)
    sem = semantic.numpy().astype(np.int64)
    b, h, w = sem.shape
    k = g.integers(1, per_class + 1, (b, -(-h // block), -(-w // block)))
    k = np.repeat(np.repeat(k, block, 1), block, 2)[:, :h, :w]
    inst = np.where(np.isin(sem, things), sem * 1000 + k, 0)
    return torch.from_numpy(inst.astype(np.int32)).unsqueeze(1)


def mixup_item(batch, hs, ws, ht, wt, seed, num_classes=14):
    """The `mixup_data` dict Mixup.do_mixup receives (datasets/replay_buffer.py:281-305 + the teacher's predictions,
    scripts/adapt_codeps.py:173-199): source image / labels at (hs, ws), target image / predictions at (ht, wt)."""
    sem_src = semantic_labels(batch, hs, ws, seed, num_classes)
    return {
        "rgb_src": {0: smooth_image(batch, hs, ws, seed + 1)},
        "rgb_tgt": {0: smooth_image(batch, ht, wt, seed + 2)},
        "semantic_src": sem_src.unsqueeze(1).contiguous(),
        "instance_src": instance_labels(sem_src, seed + 3),
        "semantic_pred_tgt_ema": semantic_labels(batch, ht, wt, seed + 4, num_classes, void_frac=0.0),   # uint8 like SemanticSegAlgo.inference
        "center_pred_tgt_ema": center_heatmap(batch, ht, wt, seed + 5, 10, 3.0),
        "offset_pred_tgt_ema": offsets(batch, ht, wt, seed + 6, 16.0),
        "camera_model_src": intrinsics(batch, hs, ws, seed + 7),
        "camera_model_tgt": intrinsics(batch, ht, wt, seed + 8) * torch.tensor([0.8, 0.8, 1.0, 1.0]),
        "tgt_is_replay": torch.tensor([i % 2 for i in range(batch)]),
    }


def panoptic_maps(batch, h, w, seed, num_classes=14, things=(9, 10, 11, 12, 13), block=8, shift=3):
    """(ground truth, prediction) panoptic id maps [B,H,W] int32 (class * 1000 + instance, -1 void): the prediction is the
    ground truth of a shifted / partly relabelled scene, so that matched, missed and spurious segments all occur."""
    sem = semantic_labels(batch, h, w, seed, num_classes, block).numpy().astype(np.int64)
    inst = instance_labels(torch.from_numpy(sem.astype(np.uint8)), seed + 1, things, block)[:, 0].numpy().astype(np.int64)
    gt = np.where(sem == 255, -1, np.where(inst > 0, inst, sem * 1000))
    g = _rng(seed + 31)
    pred = np.roll(gt, (shift, -shift), (1, 2))
    flip = np.repeat(np.repeat(g.uniform(0, 1, (batch, -(-h // block), -(-w // block))) < 0.15, block, 1), block, 2)[:, :h, :w]
    pred = np.where(flip & (pred >= 0), ((pred // 1000 + 1) % num_classes) * 1000 + np.where(pred % 1000 > 0, 7, 0), pred)
    pred = np.where((pred >= 0) & (pred // 1000 < things[0]), (pred // 1000) * 1000, pred)   # stuff classes carry no instance id
    return torch.from_numpy(gt.astype(np.int32)), torch.from_numpy(pred.astype(np.int32))


def replay_features(n, c, h, w, seed, clusters=5):
    """n post-ReLU-like encoder maps [1, c, h, w] around a few cluster centres: some frames are near-duplicates of an earlier one
    (the replay buffer rejects them: cosine similarity above the threshold), some are new (stored; evictions once it is full)."""
    g = _rng(seed)
    centres = np.abs(g.standard_normal((clusters, c, 1, 1))).astype(np.float32)
    out = []
    for _ in range(n):
        k = int(g.integers(clusters))
        spread = float(g.choice([0.05, 0.5]))
        f = centres[k] * (1 + spread * g.standard_normal((1, c, 1, 1))) + 0.3 * g.standard_normal((1, c, h, w))
        out.append(torch.from_numpy(np.maximum(f, 0).astype(np.float32)))
    return out
