"""gen_models: drop-in for codeps/model_setup.py:30-147 (same signature, same cfg keys, same module / algo objects).

Built in three independent parts -- geometry (pose encoder, depth / pose heads, photometric + smoothness losses), semantics,
instances -- around the shared encoder; a part that cfg.model switches off contributes None, as in the reference."""
from typing import Any, Dict, List, Optional, Tuple

from .. import algos, models
from .online_adap import CodepsNet


def _device_evaluators(cfg, stuff: List[int], things: List[int], ignore: List[int]) -> Dict[str, Any]:
    """The reference's three metric objects (model_setup.py:82,110,124-126) as their device-side mirrors."""
    from ..eval import DepthEvaluator, PanopticEvaluator, SemanticEvaluator
    return {"depth": DepthEvaluator(cfg.eval.depth.use_gt_scale, cfg.eval.depth.depth_ranges or [0.1, 80]),
            "semantic": SemanticEvaluator(num_classes=len(stuff) + len(things), ignore_classes=ignore),
            "panoptic": PanopticEvaluator(stuff_list=stuff, thing_list=things, label_divisor=1000, void_label=-1)}


def _geometry_part(cfg, device, enc_channels, adaptation: bool, label_mode, evaluator) -> Tuple[Any, Any, Any, Any]:
    """(pose encoder, depth head, pose head, DepthAlgo) -- model_setup.py:47-89 without the scene-flow branch."""
    if not cfg.model.make_depth:
        return None, None, None, None
    assert not cfg.model.make_sflow, "scene-flow head is out of scope"
    pose_cfg = cfg.model.pose_sflow_net.params
    pose_encoder = models.ResnetEncoder(pose_cfg.nof_layers, pose_cfg.weights_init == "pretrained", num_input_images=2,
                                        num_channels_input=3)
    pose_encoder.only_last_feature = True          # PoseHead reads features[-1] only (models/pose_head.py:30)
    ssim = algos.SSIMLoss()
    sizes = {"target": cfg.dataset.feed_img_size, "source": cfg.adapt.source_dataset.feed_img_size if adaptation else None}
    recon = {k: (algos.ReconstructionLoss(hw[1], hw[0], ssim, cfg.depth.num_recon_scales, device) if hw is not None else None)
             for k, hw in sizes.items()}
    algo = algos.DepthAlgo(recon["target"], algos.EdgeAwareSmoothnessLoss(), evaluator, None, None, recon["source"], label_mode)
    depth_head = models.DepthHead(enc_channels, use_skips=True)         # (construction order = the reference's: same random initialisation)
    pose_head = models.PoseHead(pose_encoder.num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)
    pose_head.owns_input = True                    # ... and is that map's only reader (algos/depth.py:408-420)
    return pose_encoder, depth_head, pose_head, algo


def _semantic_part(cfg, device, enc_channels, n_classes: int, ignore: List[int], evaluator) -> Tuple[Any, Any]:
    """(semantic head, SemanticSegAlgo) -- model_setup.py:94-113."""
    if not cfg.model.make_semantic:
        return None, None
    removed = set(cfg.dataset.remove_classes)
    weights = [w for c, w in enumerate(cfg.semantics.class_weights) if c not in removed]
    loss = algos.SemanticLoss(device=device, class_weights=weights, top_k_percent_pixels=cfg.semantics.top_k, ignore_labels=ignore)
    head = models.SemanticHead(enc_channels, n_classes, use_skips=True, use_guda_fusion=cfg.model.semantic_head.use_guda_fusion)
    return head, algos.SemanticSegAlgo(loss, evaluator, None)


def _instance_part(cfg, enc_channels, things: List[int], evaluator) -> Tuple[Any, Any]:
    """(instance head, InstanceSegAlgo) -- model_setup.py:118-130."""
    if not cfg.model.make_instance:
        return None, None
    head = models.InstanceHead(enc_channels, use_thing_mask=cfg.model.instance_head.use_thing_mask)
    return head, algos.InstanceSegAlgo(algos.CenterLoss(), algos.OffsetLoss(), evaluator, algos.BinaryMaskLoss(), thing_list=list(things))


def gen_models(cfg, device, stuff_classes: List[int], thing_classes: List[int], ignore_classes: List[int],
               label_mode: Optional[str] = None, adaptation_mode: bool = False, evaluators=None) -> CodepsNet:
    """`evaluators`: dict(depth=, semantic=, panoptic=) of metric objects, "device" for the device-side mirrors of eval/*
    (SURVEY.md §8f-4), or None: the algos then skip metric bookkeeping."""
    if evaluators == "device":
        evaluators = _device_evaluators(cfg, stuff_classes, thing_classes, ignore_classes)
    ev = evaluators or {}
    enc_cfg = cfg.model.po_depth_net.params
    encoder = models.ResnetEncoder(enc_cfg.nof_layers, enc_cfg.weights_init == "pretrained")
    channels = encoder.num_ch_enc
    pose_encoder, depth_head, pose_head, depth_algo = _geometry_part(cfg, device, channels, adaptation_mode, label_mode, ev.get("depth"))
    semantic_head, semantic_algo = _semantic_part(cfg, device, channels, len(stuff_classes) + len(thing_classes), ignore_classes,
                                                  ev.get("semantic"))
    instance_head, instance_algo = _instance_part(cfg, channels, thing_classes, ev.get("panoptic"))
    net = CodepsNet(cfg_mixup=cfg.adapt.mixup, backbone_po_depth=encoder, backbone_pose_sflow=pose_encoder, depth_head=depth_head,
                    pose_head=pose_head, flow_head=None, semantic_head=semantic_head, instance_head=instance_head,
                    depth_algo=depth_algo, semantic_algo=semantic_algo, instance_algo=instance_algo)
    return net if device is None else net.to(device)
