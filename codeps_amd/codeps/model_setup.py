"""gen_models: drop-in for codeps/model_setup.py:30-147 (same cfg keys, same construction order)."""
from typing import List, Optional

from ..algos import (BinaryMaskLoss, CenterLoss, DepthAlgo, EdgeAwareSmoothnessLoss, InstanceSegAlgo, OffsetLoss,
                     ReconstructionLoss, SemanticLoss, SemanticSegAlgo, SSIMLoss)
from ..models import DepthHead, InstanceHead, PoseHead, ResnetEncoder, SemanticHead
from .online_adap import CodepsNet


def gen_models(cfg, device, stuff_classes: List[int], thing_classes: List[int], ignore_classes: List[int],
               label_mode: Optional[str] = None, adaptation_mode: bool = False, evaluators=None) -> CodepsNet:
    """`evaluators` = optional dict(depth=, semantic=, panoptic=) of metric objects, or "device" for the device-side mirrors of
    eval/* built from cfg exactly like codeps/model_setup.py:82,110,124-126; without them the algos skip
    metric bookkeeping."""
    if evaluators == "device":
        from ..eval import DepthEvaluator, PanopticEvaluator, SemanticEvaluator
        evaluators = {"depth": DepthEvaluator(cfg.eval.depth.use_gt_scale, cfg.eval.depth.depth_ranges or [0.1, 80]),
                      "semantic": SemanticEvaluator(num_classes=len(stuff_classes) + len(thing_classes), ignore_classes=ignore_classes),
                      "panoptic": PanopticEvaluator(stuff_list=stuff_classes, thing_list=thing_classes, label_divisor=1000, void_label=-1)}
    evaluators = evaluators or {}
    ref_h = {"target": cfg.dataset.feed_img_size[0], "source": None}
    ref_w = {"target": cfg.dataset.feed_img_size[1], "source": None}
    if adaptation_mode:
        ref_h["source"], ref_w["source"] = cfg.adapt.source_dataset.feed_img_size

    backbone_po_depth = ResnetEncoder(cfg.model.po_depth_net.params.nof_layers,
                                      cfg.model.po_depth_net.params.weights_init == "pretrained")
    if cfg.model.make_depth:
        assert not cfg.model.make_sflow, "scene-flow head is out of scope"
        backbone_pose_sflow = ResnetEncoder(cfg.model.pose_sflow_net.params.nof_layers,
                                            cfg.model.pose_sflow_net.params.weights_init == "pretrained",
                                            num_input_images=2, num_channels_input=3)
        depth_head = DepthHead(backbone_po_depth.num_ch_enc, use_skips=True)
        pose_head = PoseHead(backbone_pose_sflow.num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)
        ssim_loss = SSIMLoss()
        rec_target = ReconstructionLoss(ref_w["target"], ref_h["target"], ssim_loss, cfg.depth.num_recon_scales, device)
        rec_source = None
        if adaptation_mode:
            rec_source = ReconstructionLoss(ref_w["source"], ref_h["source"], ssim_loss, cfg.depth.num_recon_scales, device)
        depth_algo = DepthAlgo(rec_target, EdgeAwareSmoothnessLoss(), evaluators.get("depth"), None, None, rec_source, label_mode)
    else:
        backbone_pose_sflow = depth_head = pose_head = depth_algo = None

    if cfg.model.make_semantic:
        num_classes = len(stuff_classes) + len(thing_classes)
        semantic_head = SemanticHead(backbone_po_depth.num_ch_enc, num_classes, use_skips=True,
                                     use_guda_fusion=cfg.model.semantic_head.use_guda_fusion)
        class_weights = [wt for idx, wt in enumerate(cfg.semantics.class_weights) if idx not in cfg.dataset.remove_classes]
        sem_loss = SemanticLoss(device=device, class_weights=class_weights, top_k_percent_pixels=cfg.semantics.top_k,
                                ignore_labels=ignore_classes)
        sem_algo = SemanticSegAlgo(sem_loss, evaluators.get("semantic"), None)
    else:
        semantic_head = sem_algo = None

    if cfg.model.make_instance:
        instance_head = InstanceHead(backbone_po_depth.num_ch_enc, use_thing_mask=cfg.model.instance_head.use_thing_mask)
        instance_algo = InstanceSegAlgo(CenterLoss(), OffsetLoss(), evaluators.get("panoptic"), BinaryMaskLoss(),
                                        thing_list=list(thing_classes))
    else:
        instance_head = instance_algo = None

    net = CodepsNet(cfg_mixup=cfg.adapt.mixup, backbone_po_depth=backbone_po_depth, backbone_pose_sflow=backbone_pose_sflow,
                    depth_head=depth_head, pose_head=pose_head, flow_head=None, semantic_head=semantic_head,
                    instance_head=instance_head, depth_algo=depth_algo, semantic_algo=sem_algo, instance_algo=instance_algo)
    return net.to(device) if device is not None else net
