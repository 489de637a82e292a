"""CodepsNet: drop-in for codeps/online_adap.py:20-455 (modes train / eval / infer / adapt, EMA teacher,
state-dict export).  Differences, all result-neutral for losses / gradients:
  * the shared encoder runs on the centre frame only -- features of frames t-1 / t+1 are never consumed
    when there is no flow head (algos/depth.py:389-393).  `faithful_bn_updates=True` restores the
    reference's two extra forwards for bit-faithful BN running statistics in train mode;
  * camera intrinsics stay on the device (no per-sample D2H sync, online_adap.py:95-100,304-313);
  * '*mixup' items holding the reference's raw `mixup_data` (rgb_src / rgb_tgt / ..., datasets/replay_buffer.py:281-305)
    go through the device-side Mixup.do_mixup (codeps_amd/datasets/mixup.py) exactly where the
    reference calls it (online_adap.py:331-335); items that are already mixed ("rgb" present) pass through, which is
    what the graph-captured benchmark step uses.  `mixup_fn` overrides the function.
"""
import copy
from collections import OrderedDict
from typing import Any, Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from ..datasets.mixup import Mixup
from ..misc.camera_model import CameraModel


class _StreamRef:
    """Holder of a HIP stream that deep-copies to an empty holder (CodepsNet.ema_model deep-copies the network; a stream is a
    process resource, not model state)."""

    def __init__(self, stream):
        self.stream = stream

    def __deepcopy__(self, memo):
        return _StreamRef(None)


class CodepsNet(nn.Module):

    def __init__(self, cfg_mixup, backbone_po_depth, backbone_pose_sflow, depth_head, pose_head, flow_head, semantic_head,
                 instance_head, depth_algo, semantic_algo, instance_algo):
        super().__init__()
        if depth_algo is not None:
            assert depth_head is not None and pose_head is not None
        if semantic_algo is not None:
            assert semantic_head is not None
        if instance_algo is not None:
            assert instance_head is not None
        assert flow_head is None, "flow head is out of scope (make_sflow is False in every shipped config)"
        self.cfg_mixup = cfg_mixup
        self.backbone_po_depth = backbone_po_depth
        self.backbone_pose_sflow = backbone_pose_sflow
        self.depth_head, self.pose_head, self.flow_head = depth_head, pose_head, flow_head
        self.semantic_head, self.instance_head = semantic_head, instance_head
        self.depth_algo, self.semantic_algo, self.instance_algo = depth_algo, semantic_algo, instance_algo
        self.faithful_bn_updates = False
        self.mixup_fn = None
        # optional side stream for the pose networks: they depend on the images only, so their forward (and, through autograd's
        # stream bookkeeping, their backward) can overlap the shared encoder and the heads (StepRunner sets it; None = serial)
        self._pose_stream_ref = _StreamRef(None)
        self._branch_stream_ref = _StreamRef(None)
        self.check_car_moving = "device"       # the rule of algos/depth.py:537-558 on the stream (True: host form, False: off)

    @property
    def pose_stream(self):
        return self._pose_stream_ref.stream

    @pose_stream.setter
    def pose_stream(self, stream):
        self._pose_stream_ref = _StreamRef(stream)

    @property
    def branch_stream(self):
        return self._branch_stream_ref.stream

    @branch_stream.setter
    def branch_stream(self, stream):
        self._branch_stream_ref = _StreamRef(stream)

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _camera_models(rgb0: Tensor, intr: Tensor) -> List[CameraModel]:
        h, w = rgb0.shape[2], rgb0.shape[3]
        return [CameraModel.from_tensor(w, h, intr[i]) for i in range(rgb0.shape[0])]

    def _encode_window(self, rgb: Dict[int, Tensor], offsets) -> list:
        feats = [self.backbone_po_depth(rgb[offsets[0]])]
        if self.faithful_bn_updates and self.backbone_po_depth.training:
            with torch.no_grad():
                for o in offsets[1:]:
                    self.backbone_po_depth(rgb[o])
        return feats + [None] * (len(offsets) - 1)

    # ------------------------------------------------------------------ forward
    def forward(self, in_data: Dict[str, Any], mode: str = "infer", depth_offsets: Optional[List[int]] = None,
                do_panoptic_fusion: bool = False, do_class_wise_depth_stats: bool = False,
                sem_ignore_classes: Optional[List[int]] = None, feats: Optional[List[Tensor]] = None,
                heads: Optional[Tuple[str, ...]] = None):
        """feats ("eval" / "infer" only, not in the reference): the shared encoder's features of the centre frame computed
        by the caller, so that a teacher and a student with the same frozen encoder run it once (codeps.frame.pre_step).
        heads ("eval" / "infer" only, not in the reference): subset of ("depth", "semantic", "instance") to evaluate -- the
        per-frame passes over the mix-up items consume one head's predictions each; None = all, as the reference does."""
        assert mode in ["train", "eval", "infer", "adapt"], f"Unsupported mode: {mode}"
        if mode == "adapt":
            return self.adapt(in_data, depth_offsets, do_panoptic_fusion)
        if mode == "train":
            assert depth_offsets is not None, "'depth_offsets' must be set in 'train' mode."
            assert depth_offsets[0] == 0 and depth_offsets[1] < 0 and depth_offsets[1] == -depth_offsets[2]
        if do_panoptic_fusion:
            assert self.semantic_algo is not None and self.instance_algo is not None

        L = OrderedDict((k, None) for k in ("depth_recon", "depth_smth", "flow_smth", "flow_sparsity", "semantic", "center",
                                            "offset", "thing_mask"))
        R = OrderedDict((k, None) for k in ("depth", "object_motion_map", "semantic", "semantic_soft", "center", "offset",
                                            "panoptic", "instance", "thing_mask"))
        stats = OrderedDict(sem_conf=None)
        depth_stats = None

        if mode == "train":
            rgb = in_data["rgb"]
            images_window = [rgb[o] for o in depth_offsets]
            if self.depth_algo is not None:
                pre_pose, side = None, self.pose_stream
                if side is not None:                    # the pose networks need the images only: side stream, next to the encoder
                    cur = torch.cuda.current_stream()
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        pre_pose = self.depth_algo._pose_pair(images_window, self.backbone_pose_sflow, self.pose_head)
                    for m in pre_pose:
                        m.record_stream(cur)
                feats = self._encode_window(rgb, depth_offsets)
                cams = self._camera_models(rgb[0], in_data["camera_model"])
                if side is not None:
                    torch.cuda.current_stream().wait_stream(side)
                L["depth_recon"], L["depth_smth"], _, _, R["depth"], _, _, depth_stats = self.depth_algo.training(
                    images_window, feats, in_data["camera_model"] if isinstance(in_data["camera_model"], Tensor) else cams,
                    self.depth_head, self.backbone_pose_sflow, self.pose_head, None, in_data.get("depth"), pre_pose=pre_pose)
            else:
                feats = [self.backbone_po_depth(rgb[0])]
            if self.semantic_algo is not None:
                L["semantic"], stats["sem_conf"], R["semantic"] = self.semantic_algo.training(
                    feats[0], self.semantic_head, in_data["semantic"], in_data["semantic_weights"], sem_ignore_classes,
                    in_data.get("semantic_eval"))
            if self.instance_algo is not None:
                L["center"], L["offset"], R["center"], R["offset"], L["thing_mask"], R["thing_mask"] = \
                    self.instance_algo.training(feats[0], center=in_data["center"], offset=in_data["offset"],
                                                center_weights=in_data.get("center_weights"),
                                                offset_weights=in_data.get("offset_weights"),
                                                thing_mask=in_data.get("thing_mask"), instance_head=self.instance_head)
        else:       # eval / infer: predictions (and statistics) only
            key_rgb = "rgb" if "rgb" in in_data else "rgb_tgt"
            feats0 = feats if feats is not None else self.backbone_po_depth(in_data[key_rgb][0])
            want = (lambda h: True) if heads is None else (lambda h: h in heads)
            if self.depth_algo is not None and want("depth"):
                depth_gt = in_data.get("depth") if mode == "eval" else None
                if depth_gt is None:
                    R["depth"] = self.depth_algo.inference(feats0, self.depth_head)
                else:
                    depth_stats, R["depth"] = self.depth_algo.evaluation(feats0, self.depth_head, depth_gt)
            if self.semantic_algo is not None and want("semantic"):
                semantic_gt = in_data.get("semantic") if mode == "eval" else None
                if semantic_gt is None:
                    R["semantic"], R["semantic_soft"] = self.semantic_algo.inference(feats0, self.semantic_head)
                else:
                    stats["sem_conf"], R["semantic"], R["semantic_soft"] = self.semantic_algo.evaluation(
                        feats0, self.semantic_head, in_data.get("semantic_eval", semantic_gt))
            if self.instance_algo is not None and want("instance"):
                R["center"], R["offset"], R["thing_mask"] = self.instance_algo.inference(feats0, self.instance_head)
                if mode == "infer" and not self.instance_head.use_thing_mask:
                    R["thing_mask"] = None

        if do_panoptic_fusion:
            R["panoptic"], R["instance"] = self.instance_algo.panoptic_fusion(R["semantic"], R["center"], R["offset"])
        depth_stats_classes = None
        if do_class_wise_depth_stats and R["depth"] is not None and "semantic" in in_data:
            depth_stats_classes = self.depth_algo.evaluator.compute_depth_metrics_per_class(in_data["depth"], R["depth"],
                                                                                            in_data["semantic"])
        if depth_stats is not None:                 # key order of the reference's stats dict (online_adap.py:262-268)
            stats.update(depth_stats)
        if depth_stats_classes is not None:
            stats.update(depth_stats_classes)
        return L, R, stats

    # ------------------------------------------------------------------ EMA
    @classmethod
    def ema_model(cls, model):
        ema_model = copy.deepcopy(model)
        for param in ema_model.parameters():
            param.detach_()
        return ema_model

    def update_weights(self, student_model: nn.Module, modules: List[str], alpha_teacher: float, iteration: Optional[int] = None):
        """codeps/online_adap.py:279-289 (per-parameter reference form).  The fused flat-buffer form used by
        the step harness is codeps_amd.functional.adam_ema_step."""
        if iteration is not None:
            alpha_teacher = min(1 - 1 / (iteration + 1), alpha_teacher)
        with torch.no_grad():
            for module in modules:
                for (name, p_ema), p in zip(self.named_parameters(), student_model.parameters()):
                    if name.startswith(module):
                        p_ema.mul_(alpha_teacher).add_(p.detach(), alpha=1 - alpha_teacher)

    # ------------------------------------------------------------------ adapt
    def adapt(self, in_data: Dict[str, Dict[str, Any]], depth_offsets: Optional[List[int]] = None, do_panoptic_fusion: bool = False):
        assert depth_offsets[0] == 0 and depth_offsets[1] < 0 and depth_offsets[1] == -depth_offsets[2]
        if do_panoptic_fusion:
            assert self.semantic_algo is not None and self.instance_algo is not None
        items = {k: v for k, v in in_data.items() if isinstance(v, dict)}
        camera_models = {k: v["camera_model"] for k, v in items.items() if not k.endswith("mixup")}
        images_window = {k: [v["rgb"][o] for o in depth_offsets] for k, v in items.items() if not k.endswith("mixup")}
        for key in list(items):
            if key.endswith("mixup") and (self.mixup_fn is not None or "rgb_src" in items[key]):
                in_data[key] = items[key] = (self.mixup_fn or Mixup.do_mixup)(key, items[key], self.instance_algo, self.cfg_mixup)
        pre_pose, side = None, None
        if self.pose_stream is not None and self.depth_algo is not None and not self.backbone_po_depth.training:
            side, cur = self.pose_stream, torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                pre_pose = self.depth_algo.poses_batched(images_window, [k for k in ("source", "target", "target_replay") if k in images_window],
                                                         self.backbone_pose_sflow, self.pose_head)
            for mats in pre_pose.values():
                for m in mats:
                    m.record_stream(cur)
        po_depth_feats, batched = {}, []
        if not self.backbone_po_depth.training and hasattr(self.backbone_po_depth, "forward_batch"):
            # frozen encoder (eval BatchNorm) is batch-independent: ONE forward per resolution over the centre frames of
            # every key (order: target, target_replay, source, mixups -> depth keys and supervised-semantic keys are
            # contiguous batch ranges), instead of one small-batch forward per key
            order = [k for k in ("target", "target_replay", "source") if k in items] + [k for k in items if k.endswith("mixup")]
            groups = OrderedDict()
            for k in order:
                groups.setdefault(tuple(items[k]["rgb"][0].shape[2:]), []).append(k)
            for _res, ks in groups.items():
                feats_b = self.backbone_po_depth.forward_batch([items[k]["rgb"][0] for k in ks])
                slices, off = {}, 0
                for k in ks:
                    n = items[k]["rgb"][0].shape[0]
                    slices[k] = (off, off + n)
                    po_depth_feats[k] = [[f[off:off + n] for f in feats_b]] + ([None, None] if not k.endswith("mixup") else [])
                    off += n
                batched.append((feats_b, slices))
        else:
            for k, v in items.items():
                po_depth_feats[k] = [self.backbone_po_depth(v["rgb"][0])] if k.endswith("mixup") else \
                    self._encode_window(v["rgb"], depth_offsets)

        # second optional side stream: the (frozen, small-grid) instance head next to the semantic branch; joined before the
        # panoptic fusion.  (The whole semantic + instance branch on a stream of its own next to the depth branch measured
        # SLOWER than serial: three concurrent GEMM streams evict each other's operands from L2.)
        import contextlib
        branch = self.branch_stream if self.depth_algo is not None else None
        if branch is not None:
            branch.wait_stream(torch.cuda.current_stream())
        panoptic_pred, instance_pred, target_pred = None, None, None
        with (torch.cuda.stream(branch) if branch is not None else contextlib.nullcontext()):
            # everything of the (B = 1) target frame that feeds no loss: instance head, semantic prediction, panoptic fusion
            center_losses, offset_losses, center_pred, offset_pred, thing_mask_pred = self._instance_branch(items, po_depth_feats)
            if self.semantic_algo is not None:
                target_pred = self.semantic_algo.adaptation_target(po_depth_feats, self.semantic_head)
            if do_panoptic_fusion and target_pred is not None:
                panoptic_pred, instance_pred = self.instance_algo.panoptic_fusion(target_pred, center_pred, offset_pred)
        semantic_losses, confusion_matrix, semantic_pred = self._semantic_branch(items, po_depth_feats, batched, target_pred)
        if branch is not None:
            torch.cuda.current_stream().wait_stream(branch)

        depth_losses, depth_pred = {}, None
        if self.depth_algo is not None:
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
            depth_losses, _, depth_pred, _, _, _ = self.depth_algo.adaptation(
                images_window, po_depth_feats, camera_models, self.depth_head, self.backbone_pose_sflow, self.pose_head, None,
                check_car_moving=self.check_car_moving, batched=batched, pre_pose=pre_pose)
        return self._adapt_outputs(in_data, po_depth_feats, depth_losses, depth_pred, semantic_losses, center_losses, offset_losses,
                                   semantic_pred, center_pred, offset_pred, panoptic_pred, instance_pred, thing_mask_pred, confusion_matrix)

    def _instance_branch(self, items, po_depth_feats):
        center_losses, offset_losses, center_pred, offset_pred, thing_mask_pred = {}, {}, None, None, None
        if self.instance_algo is not None:
            get = lambda name: {k: v.get(name) for k, v in items.items()}      # noqa: E731
            out = self.instance_algo.adaptation(po_depth_feats, get("center"), get("offset"), get("center_weights"),
                                                get("offset_weights"), get("thing_mask"), self.instance_head,
                                                items.get("target", {}).get("center_ema"), items.get("target", {}).get("offset_ema"))
            center_losses, offset_losses, center_pred, offset_pred = out[:4]
            if self.instance_head.use_thing_mask:
                thing_mask_pred = out[5]
        return center_losses, offset_losses, center_pred, offset_pred, thing_mask_pred

    def _semantic_branch(self, items, po_depth_feats, batched, target_pred):
        semantic_losses, confusion_matrix, semantic_pred = {"source": None}, None, None
        if self.semantic_algo is not None:
            semantic_losses, confusion_matrix, semantic_pred = self.semantic_algo.adaptation(
                po_depth_feats, self.semantic_head, {k: v.get("semantic") for k, v in items.items()},
                {k: v.get("semantic_weights") for k, v in items.items()}, batched=batched, target_pred=target_pred)
        return semantic_losses, confusion_matrix, semantic_pred

    @staticmethod
    def _adapt_outputs(in_data, po_depth_feats, depth_losses, depth_pred, semantic_losses, center_losses, offset_losses, semantic_pred,
                       center_pred, offset_pred, panoptic_pred, instance_pred, thing_mask_pred, confusion_matrix):
        losses = OrderedDict()
        losses["depth_recon"] = depth_losses.get("recon")
        losses["depth_smth"] = depth_losses.get("smth")
        losses["flow_smth"] = None
        losses["flow_sparsity"] = None
        losses["semantic_source"] = semantic_losses["source"]
        losses["semantic_cut_mixup"] = semantic_losses.get("cut_mixup")
        losses["center_source"] = center_losses.get("source")
        losses["offset_source"] = offset_losses.get("source")
        result = OrderedDict(depth=depth_pred, object_motion_map=None, semantic=semantic_pred, center=center_pred,
                             offset=offset_pred, panoptic=panoptic_pred, instance=instance_pred, thing_mask=thing_mask_pred)
        # (batch-sharded step: a rank without a target item has no frame to offer to the replay buffer)
        result["image_features"] = po_depth_feats["target"][0][-1].detach() if "target" in po_depth_feats else None
        stats = OrderedDict(sem_conf=confusion_matrix)
        return losses, result, stats, in_data

    def get_state_dict(self) -> Dict[str, Any]:
        def sd(m):
            return None if m is None else m.state_dict()
        return {"backbone_po_depth": sd(self.backbone_po_depth), "backbone_pose_sflow": sd(self.backbone_pose_sflow),
                "depth_head": sd(self.depth_head), "pose_head": sd(self.pose_head), "flow_head": sd(self.flow_head),
                "semantic_head": sd(self.semantic_head), "instance_head": sd(self.instance_head)}
