"""Tiny attribute-dict config honouring the reference's yacs keys (cfg/default_config.py,
cfg/default_config_adapt.py) for the hot path.  yacs itself is not installed here; only the keys that
gen_models / the step harness read are given defaults.  YAML overrides load with `merge_from_file`."""
import copy

import yaml


class CfgNode(dict):
    def __getattr__(self, k):
        if k in self:
            return self[k]
        raise AttributeError(k)

    __setattr__ = dict.__setitem__

    def clone(self):
        return copy.deepcopy(self)

    def merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].merge(v)
            else:
                self[k] = _wrap(v)
        return self

    def merge_from_file(self, path):
        with open(path) as f:
            return self.merge(yaml.safe_load(f) or {})


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, CfgNode):
        return CfgNode({k: _wrap(x) for k, x in v.items()})
    return v


def get_cfg_defaults(adapt=False):
    """The reference's DEFAULTS for the keys the hot path reads: cfg/default_config.py:6-165 (adapt False) or
    cfg/default_config_adapt.py:6-199 (adapt True) -- value for value (tests/test_boundary_cpu.py compares them with the reference
    files when those are present), with ONE documented exception: `weights_init` is "random" where the reference says
    "pretrained", because that default makes torchvision download ImageNet weights and this build has no network
    (ResnetEncoder(pretrained=True) raises; load a checkpoint instead).  The shipped YAMLs override these defaults:
    `get_cfg_shipped` / `merge_from_file`."""
    model = {"make_depth": True, "make_sflow": False, "make_semantic": True, "make_instance": True,
             "po_depth_net": {"params": {"nof_layers": 101, "weights_init": "random"}},
             "pose_sflow_net": {"input": "pairs", "params": {"nof_layers": 18, "weights_init": "random"}},
             "semantic_head": {"use_guda_fusion": True}, "instance_head": {"use_thing_mask": False}}
    if not adapt:
        return _wrap({
            "model": model,
            "train": {"nof_epochs": 20, "nof_workers_per_gpu": 1, "batch_size_per_gpu": 1,
                      "optimizer": {"type": "Adam", "learning_rate": 1e-4}},
            "val": {"batch_size_per_gpu": 1, "nof_workers_per_gpu": 1},
            "dataset": {"name": "", "path": "", "feed_img_size": [], "offsets": [1], "center_heatmap_sigma": 8, "return_only_rgb": False,
                        "small_instance_area_full_res": 4096, "small_instance_weight": 3, "remove_classes": [], "label_mode": "codeps"},
            "eval": {"depth": {"use_gt_scale": True, "depth_ranges": [0.1, 80]}, "semantic": {"ignore_classes": []}},
            "losses": {"weights": {"depth_recon": 1.0, "depth_smth": 0.0001, "flow_smth": 1.0, "flow_sparsity": 1.0, "semantic": 1.0,
                                   "center": 1.0, "offset": 1.0, "thing_mask": 1.0}},
            "semantics": {"class_weights": [1.0] * 19, "top_k": 0.2},
            "depth": {"num_recon_scales": 5},
            # not a key of default_config.py: CodepsNet takes its mix-up settings from cfg.adapt.mixup in adaptation mode only;
            # gen_models reads the sub-tree unconditionally, so the training config carries the adaptation defaults of it
            "adapt": {"mixup": _mixup_defaults(), "source_dataset": {"feed_img_size": []}},
        })
    return _wrap({
        "adapt": {"mode": "online",
                  "model": {"backbone_po_depth": True, "backbone_pose_sflow": True, "depth": True, "pose": True, "sflow": True,
                            "semantic": True, "instance": True},
                  "ema": {"alpha": 0.999, "depth": False, "semantic": False, "instance": False},
                  "source_dataset": {"name": "", "path": "", "feed_img_size": [], "offsets": [1]},
                  "replay_sampler": {"nof_source_samples": 1, "nof_target_samples": 0, "seed": 42},
                  "replay_buffer": {"source_size": None, "target_size": None, "maximize_diversity": False, "similarity_threshold": 0.95,
                                    "seed": 42},
                  "target_dataset_adapt_ratio": 0.7,
                  "mixup": _mixup_defaults()},
        "train": {"nof_adaptation_steps": 1, "nof_workers_per_gpu": 10, "optimizer": {"type": "Adam", "learning_rate": 1e-4}},
        "val": {"batch_size_per_gpu": 2, "nof_workers_per_gpu": 10},
        "model": model,
        "dataset": {"name": "", "path": "", "sequences": [], "feed_img_size": [], "offsets": [1], "batch_size_per_gpu": 2,
                    "center_heatmap_sigma": 8, "return_only_rgb": False, "small_instance_area_full_res": 4096,
                    "small_instance_weight": 3, "remove_classes": [], "label_mode": "codeps"},
        "eval": {"depth": {"use_gt_scale": True, "depth_ranges": [0.1, 80]}, "semantic": {"ignore_classes": []}},
        "losses": {"weights": {"depth_recon": 10.0, "depth_smth": 0.001, "flow_smth": 10.0, "flow_sparsity": 10.0,
                               "semantic_source": 1.0, "semantic_cut_mixup": 1.0, "center_source": 20.0, "offset_source": 0.1}},
        "depth": {"num_recon_scales": 5},
        "semantics": {"class_weights": [1.0] * 14, "top_k": 0.2},
    })


def _mixup_defaults():
    """cfg/default_config_adapt.py:131-147."""
    return {"general": {"active": True, "mixup_strategies": ["cut_mixup", "conf_instance_mixup"], "nof_samples": 2, "geom_augment": True},
            "cut_mix": {"nof_hor_splits": 4, "nof_vert_splits": 4, "nof_segments": 1},
            "conf_instance_mix": {"conf_thresh": 0.0, "min_inst_size": 0}, "class_mix": {"conf_thresh": 0.0}}


# What the reference's shipped YAML files set on top of the defaults, for the keys above (dataset paths, logging and
# visualisation keys are not hot-path keys).  tests/test_boundary_cpu.py checks, where the reference tree is present, that
# merging the YAML FILE into the defaults gives exactly defaults + these.
SHIPPED = {
    # cfg/train_cityscapes.yaml
    "train_cityscapes": {
        "model": {"make_depth": True, "make_sflow": False, "make_semantic": True, "make_instance": True,
                  "instance_head": {"use_thing_mask": True}},
        "train": {"nof_epochs": 600, "batch_size_per_gpu": 6, "nof_workers_per_gpu": 10, "optimizer": {"learning_rate": 1e-4}},
        "val": {"batch_size_per_gpu": 2, "nof_workers_per_gpu": 10},
        "dataset": {"name": "cityscapes", "feed_img_size": [512, 1024]},
        "semantics": {"class_weights": [1.0] * 14, "top_k": 0.2},
        "depth": {"num_recon_scales": 5},
        "losses": {"weights": {"depth_recon": 10.0, "depth_smth": 0.001, "flow_smth": 10.0, "flow_sparsity": 10.0, "semantic": 1.0,
                               "center": 20.0, "offset": 0.1, "thing_mask": 0.5}},
    },
    # cfg/adapt_cityscapes_kitti_360.yaml
    "adapt_cityscapes_kitti_360": {
        "adapt": {"mode": "online",
                  "model": {"backbone_po_depth": False, "backbone_pose_sflow": True, "depth": True, "pose": True, "sflow": False,
                            "semantic": True, "instance": False},
                  "ema": {"alpha": 0.99, "semantic": True, "instance": False},
                  "source_dataset": {"name": "cityscapes", "feed_img_size": [512, 1024]},
                  "target_dataset_adapt_ratio": 0.7,
                  "replay_sampler": {"seed": 42, "nof_source_samples": 2, "nof_target_samples": 2},
                  "replay_buffer": {"seed": 20, "source_size": 300, "target_size": 300, "maximize_diversity": True,
                                    "similarity_threshold": 0.95},
                  "mixup": {"general": {"active": True, "mixup_strategies": ["cut_mixup", "cut_mixup"], "nof_samples": 2,
                                        "geom_augment": True},
                            "conf_instance_mix": {"conf_thresh": 0.98, "min_inst_size": 2000},
                            "cut_mix": {"nof_hor_splits": 4, "nof_vert_splits": 2, "nof_segments": 2}}},
        "dataset": {"name": "kitti_360", "feed_img_size": [384, 1408], "sequences": ["03"]},
        "train": {"nof_adaptation_steps": 1, "nof_workers_per_gpu": 1},
    },
}


def get_cfg_shipped(adapt=False):
    """Defaults + the shipped YAML of the mode: cfg/adapt_cityscapes_kitti_360.yaml (adapt; the configuration the headline
    metric is quoted on) or cfg/train_cityscapes.yaml -- what `gen_config(args, adapt)` (io_utils/io_utils.py:220-237) returns
    for those files, restricted to the hot-path keys."""
    return get_cfg_defaults(adapt).merge(copy.deepcopy(SHIPPED["adapt_cityscapes_kitti_360" if adapt else "train_cityscapes"]))
