"""CPU oracle: Panoptic-DeepLab style fusion (centre NMS, pixel grouping, majority vote).  TEST INFRA.

numpy restatement of misc/post_processing_panoptic.py:31-300 as driven by
algos/instance_seg.py:207-258 (label_divisor 1000, stuff_area 0, void 255, threshold 0.1,
nms_kernel 7, top_k 200).  Integer outputs -> bit-exact; pinned by tests/golden/panoptic_*.npz.
"""
import numpy as np

LABEL_DIVISOR, VOID_LABEL, THRESHOLD, NMS_KERNEL, TOP_K = 1000, 255, 0.1, 7, 200


def find_centers(ctr, threshold=THRESHOLD, nms_kernel=NMS_KERNEL, top_k=TOP_K):
    """ctr [H,W] fp32 -> [K,2] int64 (y,x), row-major order (post_processing_panoptic.py:31-70).
    value <= threshold -> -1; keep local maxima of the k x k window (-inf padding); if there are
    >= top_k survivors keep those STRICTLY greater than the top_k-th largest value of the map."""
    h, w = ctr.shape
    hm = np.where(ctr > np.float32(threshold), ctr, np.float32(-1)).astype(np.float32)
    p = (nms_kernel - 1) // 2
    padded = np.full((h + 2 * p, w + 2 * p), -np.inf, np.float32)
    padded[p:p + h, p:p + w] = hm
    pooled = np.full((h, w), -np.inf, np.float32)
    for dy in range(nms_kernel):
        for dx in range(nms_kernel):
            np.maximum(pooled, padded[dy:dy + h, dx:dx + w], out=pooled)
    hm = np.where(hm != pooled, np.float32(-1), hm)
    pts = np.argwhere(hm > 0)
    if top_k is None or pts.shape[0] < top_k:
        return pts.astype(np.int64)
    kth = np.partition(hm.ravel(), hm.size - top_k)[hm.size - top_k]
    return np.argwhere(hm > kth).astype(np.int64)


def group_pixels(centers, offsets):
    """centers [K,2] (y,x); offsets [2,H,W] (dy,dx) -> 1-based nearest-centre id [H,W] int64
    (post_processing_panoptic.py:73-109): argmin_k sqrt((y+dy-cy)^2 + (x+dx-cx)^2), first index on
    ties, all in fp32."""
    _, h, w = offsets.shape
    yy = (np.arange(h, dtype=np.float32)[:, None] + offsets[0]).astype(np.float32)
    xx = (np.arange(w, dtype=np.float32)[None, :] + offsets[1]).astype(np.float32)
    best = np.full((h, w), np.inf, np.float32)
    idx = np.zeros((h, w), np.int64)
    for k, (cy, cx) in enumerate(centers):
        dy = (np.float32(cy) - yy).astype(np.float32)
        dx = (np.float32(cx) - xx).astype(np.float32)
        d = np.sqrt((dy * dy + dx * dx).astype(np.float32)).astype(np.float32)
        upd = d < best
        best[upd] = d[upd]
        idx[upd] = k
    return idx + 1


def instance_segmentation(sem, ctr, offsets, thing_list, threshold=THRESHOLD):
    """-> (ins_seg [H,W] int64, centers) (post_processing_panoptic.py:112-152)."""
    thing = np.isin(sem, np.asarray(thing_list))
    centers = find_centers(ctr, threshold)
    if centers.shape[0] == 0:
        return np.zeros(sem.shape, np.int64), centers
    return thing.astype(np.int64) * group_pixels(centers, offsets), centers


def merge(sem, ins, thing_list):
    """-> panoptic int16 [H,W] (post_processing_panoptic.py:155-220): per instance (ascending id)
    majority class (smallest label on ties, = torch.mode) inside the thing region, id = class*1000 +
    per-class running counter; stuff classes -> class*1000; everything else -1."""
    pan = np.full(sem.shape, -1, np.int16)
    sem_thing = np.isin(sem, np.asarray(thing_list))
    counters = {}
    for iid in np.unique(ins):
        if iid == 0:
            continue
        mask = (ins == iid) & sem_thing
        if not mask.any():
            continue
        vals, cnt = np.unique(sem[mask], return_counts=True)
        cls = int(vals[np.argmax(cnt)])               # first max = smallest label on ties
        new_id = counters.get(cls, 1)
        counters[cls] = new_id + 1
        pan[mask] = np.int16(cls * LABEL_DIVISOR + new_id)
    for cls in np.unique(sem):
        if int(cls) in thing_list or int(cls) == VOID_LABEL:
            continue
        pan[(sem == cls) & ~(ins > 0)] = np.int16(int(cls) * LABEL_DIVISOR)
    return pan


def panoptic_fusion(semantic, center, offset, thing_list, threshold=THRESHOLD):
    """semantic [B,H,W] (uint8/int64), center [B,1,H,W], offset [B,2,H,W] (numpy) ->
    (panoptic int16 [B,H,W], instance [B,H,W] of semantic's dtype, [centers per image])
    (algos/instance_seg.py:207-258)."""
    b = semantic.shape[0]
    pan = np.empty(semantic.shape, np.int16)
    inst = np.empty(semantic.shape, semantic.dtype)
    pts = []
    for i in range(b):
        ins, c = instance_segmentation(semantic[i], center[i, 0], offset[i], thing_list, threshold)
        pan[i] = merge(semantic[i], ins, thing_list)
        inst[i] = ins.astype(semantic.dtype)
        pts.append(c)
    return pan, inst, pts
