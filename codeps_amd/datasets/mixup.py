"""Mixup: device-side mirror of datasets/mixup.py for the strategies the adaptation YAMLs select.

`Mixup.do_mixup(strategy, mixup_data, instance_algo, cfg_mixup)` has the reference's signature, input dict and output dict
(datasets/mixup.py:406-489) and is what CodepsNet.adapt calls for every `*mixup` key (codeps/online_adap.py:331-335;
install it with `net.mixup_fn = Mixup.do_mixup`).  Differences in HOW, not WHAT:
  * the three float64 `grid_sample` warps and the three boolean-mask scatters of `cut_mixup` run as ONE kernel that only
    warps pixels inside the cut rectangles (codeps_cut_mixup);
  * the centre / offset ground truth is regenerated on the device (codeps_offset_center) instead of the reference's
    per-image D2H copy -> numpy loop -> H2D copy (mixup.py:243-257);
  * nothing synchronises the host: the cell choice is the reference's own `np.random.choice` call sequence (host RNG,
    mixup.py:108), everything else is enqueued on the current stream.
`class_mixup` (mixup.py:261-311, not selected by either adaptation YAML) is offered through the same kernels: the float64
warps as three codeps_warp_c2c launches, the class mask and the masked copies as tensor ops, the centre / offset ground
truth on the device; its `torch.unique` over the source labels synchronises like the reference's.
`conf_instance_mixup` (mixup.py:347-404, 542-652; in the DEFAULT strategy list, default_config_adapt.py:138, overridden by both
YAMLs) places the teacher's instances with kornia's distance transform: the reference runs the transform's cascade of
max(H, W) convolutions per instance and reads one row of it; here that row comes from the cascade's closed form
(codeps_conf_place), the per-instance bookkeeping from one statistics table per image (one host read per image instead of
several per instance) and mask kernels (codeps_conf_apply / codeps_conf_paste).  kornia is not vendored by the reference
(requirements.txt:16, unpinned) and not installed here: oracle/mixup.py restates its published algorithm (parity unpinned
for that one function), everything built around it is pinned by tests/golden/conf_mixup.npz -- the reference's own
conf_instance_mixup run with that restatement in kornia's place.  Like the reference (mixup.py:581 reads batch item 0's
instance map for every item's centre) it is meaningful at batch size 1.
"""
from typing import Any, Dict, List, Optional

import numpy as np
import torch
from torch import Tensor

from .. import functional as F
from .._lib import CodepsHipError
from ..misc.camera_model import CameraModel


def _intr(models_or_tensor) -> Tensor:
    if isinstance(models_or_tensor, Tensor):
        return models_or_tensor
    return CameraModel.batch_tensor(models_or_tensor)


class Mixup:
    _pending_status: List[Any] = []             # (pinned host copy of an offset_center status word, event), checked lazily

    def __init__(self):
        pass

    # ------------------------------------------------------------------ building blocks (reference names)
    @staticmethod
    def warp_c2c(cam_model_src, cam_model_tgt, in_src: Tensor, in_tgt: Tensor, depth_val: Optional[float] = 1,
                 interp_mode="bilinear", padding_mode="border") -> Tensor:
        """mixup.py:211-229: `in_src` (seen by cam_model_src) resampled into cam_model_tgt's view, the size of `in_tgt`.
        Camera models: lists of CameraModel like the reference, or a [B,4] intrinsics tensor.  Returns in_src's dtype
        (the reference returns the float64 intermediate and casts afterwards; the rounding is the same single one)."""
        dev = in_src.device
        return F.warp_c2c(in_src, _intr(cam_model_src).to(dev), _intr(cam_model_tgt).to(dev), in_tgt.shape[-2:], interp_mode,
                          padding_mode, depth_val)

    @staticmethod
    def get_off_cen(lbl_inst_mixup: Tensor):
        """mixup.py:243-257 -> (center [B,1,H,W], offset [B,2,H,W]) fp32, sigma 8 (dataset.py:126)."""
        center, offset, status = F.offset_center(lbl_inst_mixup, 8.0)
        host = torch.empty(status.shape, dtype=status.dtype, pin_memory=True)
        host.copy_(status, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        Mixup._pending_status.append((host, done))
        return center, offset

    @staticmethod
    def check():
        """Raise if an earlier get_off_cen met more distinct instance ids than the device table holds.  Called at the start
        of every do_mixup: only status words whose copy to pinned host memory has already completed are looked at (at most
        a handful stay queued), so the host never waits for the stream here."""
        keep = []
        for host, done in Mixup._pending_status:
            if not done.query():
                if len(Mixup._pending_status) > 16:
                    done.synchronize()
                else:
                    keep.append((host, done))
                    continue
            if bool((host < 0).any()):
                Mixup._pending_status = []
                raise CodepsHipError("codeps_offset_center: more than 4096 distinct instance ids in one image")
        Mixup._pending_status = keep

    @staticmethod
    def _cut_cells(batch: int, nof_vert_split: int, nof_hor_split: int, height: int, width: int, nof_segments: int) -> np.ndarray:
        """The reference's cell draw (mixup.py:96-146; the rare-class branch is dead code there: `or True`, :106) ->
        rectangles [B,nof_segments,4] = (y0, y1, x0, x1), end-exclusive with the reference's '-1' (mixup.py:138-144)."""
        crop_h, crop_w = height // nof_vert_split, width // nof_hor_split
        rects = np.empty((batch, nof_segments, 4), np.int32)
        for b in range(batch):
            split_ids = np.random.choice(nof_vert_split * nof_hor_split, nof_segments, replace=False)
            for s, split_id in enumerate(split_ids):
                mul_x, mul_y = int(split_id) % nof_hor_split, int(split_id) // nof_hor_split
                rects[b, s] = (mul_y * crop_h, (mul_y + 1) * crop_h - 1, mul_x * crop_w, (mul_x + 1) * crop_w - 1)
        return rects

    @staticmethod
    def _get_cls_msk(lbl_sem: Tensor) -> Tensor:
        """mixup.py:69-87: half of the classes present in the (whole batch of) source labels, drawn with the reference's
        np.random.choice call -> [B,H,W] count mask (> 0 where the pixel's class was drawn)."""
        all_classes = torch.unique(lbl_sem)                                     # host sync, as in the reference
        num_classes = all_classes.shape[0]
        pick = np.random.choice(num_classes, int((num_classes + num_classes % 2) / 2), replace=False)
        classes = all_classes[torch.as_tensor(pick, device=all_classes.device).long()]
        return torch.eq(lbl_sem, classes.view(1, -1, 1, 1)).sum(1)

    @staticmethod
    def class_mixup(img_src, img_tgt, lbl_instance_src, lbl_sem_src, semantic_pred_tgt_ema, instance_pred_tgt_ema, geom_augment,
                    conf_thresh, get_off_cen=True):
        """mixup.py:261-311.  img_* [B,3,H,W]; lbl_*_src [B,1,H,W]; *_pred_tgt_ema [B,H,W] (already warped when geom_augment)."""
        lbl_instance_tgt, lbl_sem_tgt = instance_pred_tgt_ema, semantic_pred_tgt_ema
        if not geom_augment:
            size = img_src.shape[-2:]
            img_tgt = torch.nn.functional.interpolate(img_tgt, size, mode="bilinear", align_corners=False)
            lbl_sem_tgt = torch.nn.functional.interpolate(lbl_sem_tgt.unsqueeze(1), size, mode="nearest").squeeze(1)
            lbl_instance_tgt = torch.nn.functional.interpolate(lbl_instance_tgt.unsqueeze(1).float(), size,
                                                               mode="nearest").type(torch.int32).squeeze(1)
        cls_msk = (Mixup._get_cls_msk(lbl_sem_src) > 0).unsqueeze(1)            # [B,1,H,W]
        rgb_mixup = torch.where(cls_msk, img_tgt.float(), img_src.float())
        lbl_sem_mixup = torch.where(cls_msk, lbl_sem_tgt.unsqueeze(1), lbl_sem_src).squeeze(1)
        lbl_inst_mixup = torch.where(cls_msk, lbl_instance_tgt.unsqueeze(1).type(torch.int32), lbl_instance_src).squeeze(1)
        if get_off_cen:
            lbl_center_mixup, lbl_offset_mixup = Mixup.get_off_cen(lbl_inst_mixup)
            return rgb_mixup, lbl_sem_mixup, lbl_inst_mixup, lbl_center_mixup, lbl_offset_mixup
        return rgb_mixup, lbl_sem_mixup, lbl_inst_mixup

    @staticmethod
    def _do_class_mixup(mixup_data: Dict[str, Any], instance_algo, cfg_mixup) -> Dict[str, Any]:
        img_src, img_tgt = mixup_data["rgb_src"][0], mixup_data["rgb_tgt"][0]
        semantic_src, instance_src = mixup_data["semantic_src"], mixup_data["instance_src"]
        semantic_pred = mixup_data["semantic_pred_tgt_ema"].detach()
        _, instance_pred = instance_algo.panoptic_fusion(semantic_pred, mixup_data["center_pred_tgt_ema"].detach(),
                                                         mixup_data["offset_pred_tgt_ema"].detach())
        geom = bool(cfg_mixup.general.geom_augment)
        if geom:                                                                 # mixup.py:419-441
            cs, ct = mixup_data["camera_model_src"], mixup_data["camera_model_tgt"]
            img_tgt = Mixup.warp_c2c(ct, cs, img_tgt, img_src, interp_mode="bilinear", padding_mode="zeros")
            instance_pred = Mixup.warp_c2c(ct, cs, instance_pred, instance_src, interp_mode="nearest", padding_mode="zeros").squeeze(1)
            semantic_pred = Mixup.warp_c2c(ct, cs, semantic_pred, semantic_src, interp_mode="nearest", padding_mode="border").squeeze(1)
        rgb, sem, inst, center, offset = Mixup.class_mixup(img_src, img_tgt, instance_src, semantic_src, semantic_pred, instance_pred,
                                                           geom, cfg_mixup.class_mix.conf_thresh)
        return {"rgb": {0: rgb}, "camera_model": mixup_data["camera_model_src"], "semantic": sem, "center": center, "offset": offset,
                "instance": inst}

    # ------------------------------------------------------------------ conf_instance_mixup
    @staticmethod
    def get_conf_mask(semantic_pred_ema, center_ema_plabel_det, offset_ema_plabel_det, instance_algo, src_shape, threshold: float,
                      min_inst_size: int, thing_msk_src):
        """mixup.py:542-652, on masks instead of index lists: the teacher's instances (panoptic fusion at `threshold`) are placed
        one after the other, in ascending id order, at the column of their centre row that kornia's distance transform
        (closed form, `F.conf_place`) ranks farthest from every thing already in the image; a later instance is cropped where it
        overlaps an earlier one.  -> (dst uint8 [B,H,W]: instance held by a source pixel, 0 = none; shift int32 [B,256]: a held
        pixel (y, x) shows target pixel (y, x - shift[id]); instance_plabel_ema [B,H,W]).  One host read per image (the
        per-instance table) -- the reference synchronises several times per instance."""
        _, inst = instance_algo.panoptic_fusion(semantic_pred_ema, center_ema_plabel_det, offset_ema_plabel_det,
                                                threshold_center=threshold)
        b, h, w = inst.shape
        if tuple(src_shape[-2:]) != (h, w):
            raise AssertionError("Not implemented")                                    # the reference's index arithmetic needs equal sizes
        dev = inst.device
        inst32 = inst.to(torch.int32).contiguous()
        dst = torch.zeros(b, h, w, dtype=torch.uint8, device=dev)
        shift = torch.zeros(b, 256, dtype=torch.int32, device=dev)
        things = (thing_msk_src.reshape(b, h, w) != 0).to(torch.uint8).contiguous()
        tables = [F.conf_instance_stats(inst32[i]) for i in range(b)]
        host = torch.stack(tables).cpu()                                               # the one synchronisation
        for i in range(b):
            for cid in range(1, 256):
                n = int(host[i, cid, 0])
                if n == 0 or n < min_inst_size:
                    continue
                n0 = int(host[0, cid, 0])                                              # mixup.py:581 reads batch item 0's map
                if n0 == 0:
                    raise ValueError(f"conf_instance_mixup: instance {cid} of batch item {i} does not exist in item 0, whose map the "
                                     "reference reads for every item's centre (datasets/mixup.py:581): use batch size 1")
                # torch.round (half to even) of the float32 mean of the pixel coordinates
                cy, cx = (int(np.rint(np.float32(int(host[0, cid, k]) / n0))) for k in (1, 2))
                argmax, _ = F.conf_place(things[i], cy)
                F.conf_apply(inst32[i], cid, tables[i], cx, argmax, shift[i], dst[i], things[i])
        return dst, shift, inst

    @staticmethod
    def conf_instance_mixup(img_src, img_tgt, lbl_instance_src, lbl_sem_src, semantic_pred_tgt_ema, center_pred_tgt_ema,
                            offset_pred_tgt_ema, instance_algo, threshold: float, min_inst_size: int, get_off_cen: bool = True):
        """mixup.py:347-404.  img_* [B,3,H,W]; lbl_*_src [B,1,H,W]; *_pred_tgt_ema: the teacher's (warped) predictions."""
        dst, shift, inst_conf = Mixup.get_conf_mask(semantic_pred_tgt_ema, center_pred_tgt_ema, offset_pred_tgt_ema, instance_algo,
                                                    img_src.shape, threshold, min_inst_size, lbl_instance_src != 0)
        b, _, h, w = img_src.shape
        i32 = lambda t: t.reshape(b, h, w).to(torch.int32).contiguous()                 # noqa: E731
        sem_src, sem_tgt, inst_src, inst_tgt = i32(lbl_sem_src), i32(semantic_pred_tgt_ema), i32(lbl_instance_src), i32(inst_conf)
        img_src, img_tgt = img_src.float().contiguous(), img_tgt.float().contiguous()
        outs = [F.conf_paste(img_src[i], img_tgt[i], sem_src[i], sem_tgt[i], inst_src[i], inst_tgt[i], dst[i], shift[i]) for i in range(b)]
        rgb_mixup = torch.stack([o[0] for o in outs])
        lbl_sem_mixup = torch.stack([o[1] for o in outs]).to(lbl_sem_src.dtype)
        lbl_inst_mixup = torch.stack([o[2] for o in outs]).unsqueeze(1).to(lbl_instance_src.dtype)
        if get_off_cen:
            lbl_center_mixup, lbl_offset_mixup = Mixup.get_off_cen(lbl_inst_mixup)
            return rgb_mixup, lbl_sem_mixup, lbl_inst_mixup, lbl_center_mixup, lbl_offset_mixup
        return rgb_mixup, lbl_sem_mixup, lbl_inst_mixup

    @staticmethod
    def _do_conf_instance_mixup(mixup_data: Dict[str, Any], instance_algo, cfg_mixup) -> Dict[str, Any]:
        img_src, img_tgt = mixup_data["rgb_src"][0], mixup_data["rgb_tgt"][0]
        semantic_src, instance_src = mixup_data["semantic_src"], mixup_data["instance_src"]
        semantic_pred = mixup_data["semantic_pred_tgt_ema"].detach()
        _, instance_pred = instance_algo.panoptic_fusion(semantic_pred, mixup_data["center_pred_tgt_ema"].detach(),
                                                         mixup_data["offset_pred_tgt_ema"].detach())
        if bool(cfg_mixup.general.geom_augment):                                        # mixup.py:419-441
            cs, ct = mixup_data["camera_model_src"], mixup_data["camera_model_tgt"]
            img_tgt = Mixup.warp_c2c(ct, cs, img_tgt, img_src, interp_mode="bilinear", padding_mode="zeros")
            instance_pred = Mixup.warp_c2c(ct, cs, instance_pred, instance_src, interp_mode="nearest", padding_mode="zeros").squeeze(1)
            semantic_pred = Mixup.warp_c2c(ct, cs, semantic_pred, semantic_src, interp_mode="nearest", padding_mode="border").squeeze(1)
        center_pred, offset_pred = Mixup.get_off_cen(instance_pred)                     # mixup.py:444: labels of the (warped) prediction
        cm = cfg_mixup.conf_instance_mix
        rgb, sem, inst, center, offset = Mixup.conf_instance_mixup(img_src, img_tgt, instance_src, semantic_src, semantic_pred, center_pred,
                                                                   offset_pred, instance_algo, cm.conf_thresh, cm.min_inst_size)
        return {"rgb": {0: rgb}, "camera_model": mixup_data["camera_model_src"], "semantic": sem, "center": center, "offset": offset,
                "instance": inst.squeeze(1)}

    # ------------------------------------------------------------------ entry point
    @staticmethod
    def do_mixup(mixup_strategy: str, mixup_data: Dict[str, Any], instance_algo, cfg_mixup) -> Dict[str, Any]:
        Mixup.check()
        if mixup_strategy == "class_mixup":
            return Mixup._do_class_mixup(mixup_data, instance_algo, cfg_mixup)
        if mixup_strategy == "conf_instance_mixup":
            return Mixup._do_conf_instance_mixup(mixup_data, instance_algo, cfg_mixup)
        if mixup_strategy != "cut_mixup":
            raise NotImplementedError("The requested mixup strategy is not implemented yet")      # mixup.py:477
        img_src = mixup_data["rgb_src"][0]
        img_tgt = mixup_data["rgb_tgt"][0]
        semantic_src, instance_src = mixup_data["semantic_src"], mixup_data["instance_src"]          # [B,1,H,W]
        semantic_pred = mixup_data["semantic_pred_tgt_ema"].detach()
        center_pred = mixup_data["center_pred_tgt_ema"].detach()
        offset_pred = mixup_data["offset_pred_tgt_ema"].detach()
        _, instance_pred = instance_algo.panoptic_fusion(semantic_pred, center_pred, offset_pred)
        b, _, h, w = img_src.shape
        geom = bool(cfg_mixup.general.geom_augment)
        if not geom and img_tgt.shape != img_src.shape:
            raise AssertionError("Not implemented")                                      # mixup.py:150
        cm = cfg_mixup.cut_mix
        rects = torch.from_numpy(Mixup._cut_cells(b, cm.nof_vert_splits, cm.nof_hor_splits, h, w, cm.nof_segments))
        dev = img_src.device
        rects = rects.to(dev, non_blocking=True)
        intr_src = mixup_data["camera_model_src"].to(dev) if geom else None
        intr_tgt = mixup_data["camera_model_tgt"].to(dev) if geom else None
        rgb, sem, inst = F.cut_mixup(img_src, img_tgt, semantic_src.reshape(b, h, w), semantic_pred, instance_src.reshape(b, h, w),
                                     instance_pred, rects, intr_src, intr_tgt)
        center, offset = Mixup.get_off_cen(inst)
        return {
            "rgb": {0: rgb},
            "camera_model": mixup_data["camera_model_src"],
            "semantic": sem.to(semantic_src.dtype),
            "center": center,
            "offset": offset,
            "instance": inst.to(instance_src.dtype),
        }
