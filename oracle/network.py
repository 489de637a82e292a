"""CPU oracle: network-level composition (CodepsNet "train" forward and the adaptation step).
TEST INFRASTRUCTURE.  Follows codeps/online_adap.py:76-139 ("train"), :291-436 ("adapt") with the
dead work removed (encoder features of frames t-1/t+1 are never consumed when
flow_head is None, algos/depth.py:389-393).  Pinned by tests/golden/network_{train,adapt}.npz.

``sds`` = dict of per-module state dicts keyed like CodepsNet.get_state_dict
(codeps/online_adap.py:438-455); encoder dicts carry torchvision keys without the ``encoder.`` prefix.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import heads, losses, panoptic, resnet

THINGS = [9, 10, 11, 12, 13]                   # datasets/dataset.py:212-229 (label mode "codeps")


def _strip(sd):
    return {k[len("encoder."):] if k.startswith("encoder.") else k: v for k, v in sd.items()}


def encode(sd, x, num_layers, training=False):
    return resnet.features(_strip(sd), x, num_layers, training)


def depth_forward(sds, cfg, rgb, training_pose=False):
    """algos/depth.py:374-423 (_forward) without flow head.  rgb = {0,-1,1: [B,3,H,W]}."""
    feats = encode(sds["backbone_po_depth"], rgb[0], cfg["nof_layers"], cfg.get("train_encoder", False))
    depth, disp = heads.depth_head(sds["depth_head"], feats)
    pf = encode(sds["backbone_pose_sflow"], torch.cat([rgb[-1], rgb[0]], 1), cfg.get("pose_layers", 18),
                training_pose)
    t0 = heads.pose_head(sds["pose_head"], pf, invert_pose=True)
    pf = encode(sds["backbone_pose_sflow"], torch.cat([rgb[0], rgb[1]], 1), cfg.get("pose_layers", 18),
                training_pose)
    t1 = heads.pose_head(sds["pose_head"], pf)
    return feats, depth, disp, (t0, t1)


def train_forward(sds, cfg, sample, noise=None, do_panoptic=True):
    """-> (losses, result) like CodepsNet.forward(mode="train") (codeps/online_adap.py:76-139,216-268)."""
    rgb = sample["rgb"]
    feats, depth, disp, poses = depth_forward(sds, cfg, rgb)
    out, res = OrderedDict(), OrderedDict()
    out["depth_recon"] = losses.reconstruction_loss(sample["camera_model"], (rgb[0], rgb[-1], rgb[1]),
                                                    depth, poses, cfg.get("num_scales", 5), noise)
    out["depth_smth"] = losses.smoothness_loss(rgb[0], disp)
    if "semantic_head" not in sds:                 # depth-only network (BASELINE configs[0]: make_semantic = make_instance = False)
        res.update(depth=depth)
        return out, res
    probs = heads.semantic_head(sds["semantic_head"], feats, cfg.get("use_guda_fusion", True))
    out["semantic"] = losses.semantic_loss(probs, sample["semantic"], sample["semantic_weights"],
                                           cfg.get("class_weights"), cfg.get("top_k", 0.2))
    sem_pred = probs.argmax(1).to(torch.uint8)
    center, offset, thing = heads.instance_head(sds["instance_head"], feats, cfg.get("use_thing_mask", False),
                                                cfg.get("train_instance_bn", False))
    out["center"] = losses.center_loss(center, sample["center"])
    out["offset"] = losses.offset_loss(offset, sample["offset"])
    if thing is not None:
        out["thing_mask"] = losses.binary_mask_loss(thing, sample["thing_mask"].squeeze(1))
        res["thing_mask"] = thing.argmax(1).to(torch.uint8)
    res.update(depth=depth, semantic=sem_pred, center=center, offset=offset)
    if do_panoptic:
        pan, inst, _ = panoptic.panoptic_fusion(sem_pred.numpy(), center.detach().numpy(),
                                                offset.detach().numpy(), THINGS)
        res["panoptic"], res["instance"] = torch.from_numpy(pan), torch.from_numpy(inst)
    return out, res


def adapt_forward(sds, cfg, data, noise=None, mixup_cells=None):
    """Adaptation-step losses (codeps/online_adap.py:291-436 with algos/depth.py:507-577,
    algos/semantic_seg.py:200-249, algos/instance_seg.py:141-206).  data: {"target","target_replay","source": sample dicts,
    "<x>mixup": pre-mixed sample dicts, or RAW mixup items (keys rgb_src / rgb_tgt / ..., online_adap.py:331-335) that are
    mixed here by oracle.mixup.do_cut_mixup with the cells `mixup_cells[key]` np.random.choice drew}.  Depth losses are the
    image-count weighted mean over source/target/target_replay; the rolling "car not moving" rule (depth.py:537-558) is host
    state and not part of this function.  ``noise``: {key: [per-scale tensors]} or None."""
    from . import mixup as omix
    out, res = OrderedDict(), OrderedDict()
    data = dict(data)
    for key, item in list(data.items()):
        if key.endswith("mixup") and "rgb_src" in item:
            raw = {k: (v[0].numpy() if isinstance(v, dict) else v.numpy()) for k, v in item.items()}
            m = omix.do_cut_mixup(raw, THINGS, mixup_cells[key], cfg.get("geom_augment", True), cfg.get("nof_vert_splits", 2),
                                  cfg.get("nof_hor_splits", 4))
            data[key] = {"rgb": {0: torch.from_numpy(m["rgb"])}, "semantic": torch.from_numpy(m["semantic"])}
            res["mix_" + key] = data[key]
    recon, smth, n = {}, {}, {}
    feats = {}
    for key in ("source", "target", "target_replay"):
        if key not in data:
            continue
        rgb = data[key]["rgb"]
        feats[key], depth, disp, poses = depth_forward(sds, cfg, rgb, cfg.get("train_pose_bn", False))
        recon[key] = losses.reconstruction_loss(data[key]["camera_model"], (rgb[0], rgb[-1], rgb[1]), depth,
                                                poses, cfg.get("num_scales", 5),
                                                None if noise is None else noise[key])
        smth[key] = losses.smoothness_loss(rgb[0], disp)
        n[key] = rgb[0].shape[0]
        if key == "target":
            res["depth"], res["T"] = depth, poses
    tot = sum(n.values())
    out["depth_recon"] = sum(recon[k] * n[k] for k in recon) / tot
    out["depth_smth"] = sum(smth[k] * n[k] for k in smth) / tot
    sem = lambda f: heads.semantic_head(sds["semantic_head"], f, cfg.get("use_guda_fusion", True))  # noqa: E731
    res["semantic"] = sem(feats["target"]).argmax(1)
    for key, item in data.items():
        if key.endswith("mixup"):
            f = encode(sds["backbone_po_depth"], item["rgb"][0], cfg["nof_layers"])
            out["semantic_" + key] = losses.semantic_loss(sem(f), item["semantic"],
                                                          torch.ones_like(item["semantic"]),
                                                          cfg.get("class_weights"), cfg.get("top_k", 0.2))
    if "source" in data:
        out["semantic_source"] = losses.semantic_loss(sem(feats["source"]), data["source"]["semantic"],
                                                      data["source"]["semantic_weights"],
                                                      cfg.get("class_weights"), cfg.get("top_k", 0.2))
    if "instance_head" in sds and cfg.get("instance", True):
        inst = lambda f: heads.instance_head(sds["instance_head"], f, cfg.get("use_thing_mask", False))[:2]  # noqa: E731
        res["center"], res["offset"] = inst(feats["target"])               # instance_seg.py:155-157 (target inference)
        if "source" in data and "center" in data["source"]:                 # instance_seg.py:176-183
            c_src, o_src = inst(feats["source"])
            out["center_source"] = losses.center_loss(c_src, data["source"]["center"])
            out["offset_source"] = losses.offset_loss(o_src, data["source"]["offset"])
        pan, ins, _ = panoptic.panoptic_fusion(res["semantic"].to(torch.uint8).numpy(), res["center"].detach().numpy(),
                                               res["offset"].detach().numpy(), THINGS)
        res["panoptic"], res["instance"] = torch.from_numpy(pan), torch.from_numpy(ins)
    res["image_features"] = feats["target"][-1].detach()
    return out, res


def eval_forward(sds, cfg, sample, mode="eval", do_panoptic=True, class_wise_depth_stats=False):
    """CodepsNet.forward in modes "eval" / "infer" (codeps/online_adap.py:140-268): predictions of the centre frame and, in
    "eval" with ground truth present, the depth statistics (eval/depth.py) and the confusion matrix (eval/semantic.py).
    -> (result, stats)."""
    from . import metrics
    rgb = sample["rgb"] if "rgb" in sample else sample["rgb_tgt"]              # online_adap.py:143 (mix-up items)
    feats = encode(sds["backbone_po_depth"], rgb[0], cfg["nof_layers"])
    res, stats = OrderedDict(), OrderedDict(sem_conf=None)
    res["depth"], _ = heads.depth_head(sds["depth_head"], feats)
    probs = heads.semantic_head(sds["semantic_head"], feats, cfg.get("use_guda_fusion", True))
    res["semantic"], res["semantic_soft"] = probs.argmax(1).to(torch.uint8), probs
    res["center"], res["offset"], _ = heads.instance_head(sds["instance_head"], feats, cfg.get("use_thing_mask", False))
    if mode == "eval":
        if sample.get("depth") is not None:
            stats.update(metrics.depth_metrics(sample["depth"].numpy(), res["depth"].numpy(), cfg.get("use_gt_scale", True), cfg["depth_ranges"]))
        if sample.get("semantic") is not None:
            stats["sem_conf"] = metrics.confusion_matrix(res["semantic"].numpy(), sample.get("semantic_eval", sample["semantic"]).numpy(),
                                                         probs.shape[1], cfg.get("ignore_classes", ()))
    if do_panoptic:
        pan, ins, _ = panoptic.panoptic_fusion(res["semantic"].numpy(), res["center"].numpy(), res["offset"].numpy(), THINGS)
        res["panoptic"], res["instance"] = torch.from_numpy(pan), torch.from_numpy(ins)
    if class_wise_depth_stats and "semantic" in sample and sample.get("depth") is not None:
        stats.update(metrics.depth_metrics_per_class(sample["depth"].numpy(), res["depth"].numpy(), sample["semantic"].numpy(),
                                                     cfg.get("use_gt_scale", True), cfg["depth_ranges"]))
    return res, stats


def adam_step(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults used by misc/train_utils.py:194-201 (no weight decay, no amsgrad)."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    p.addcdiv_(m, (v.sqrt() / np.sqrt(bc2)).add_(eps), value=-lr / bc1)


def ema_update(p_ema, p, alpha):
    """codeps/online_adap.py:279-289."""
    p_ema.copy_(alpha * p_ema + (1 - alpha) * p)
