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
    """Hot-path subset of cfg/default_config.py:48-165 (+ default_config_adapt.py:31-37,169-192)."""
    cfg = _wrap({
        "model": {"make_depth": True, "make_sflow": False, "make_semantic": True, "make_instance": True,
                  "po_depth_net": {"params": {"nof_layers": 101, "weights_init": "random"}},
                  "pose_sflow_net": {"params": {"nof_layers": 18, "weights_init": "random"}},
                  "semantic_head": {"use_guda_fusion": True}, "instance_head": {"use_thing_mask": False}},
        "dataset": {"feed_img_size": [512, 1024], "remove_classes": []},
        "depth": {"num_recon_scales": 5},
        "semantics": {"class_weights": [1.0] * 14, "top_k": 0.2},
        "eval": {"depth": {"use_gt_scale": True, "depth_ranges": None}, "semantic": {"ignore_classes": []}},
        "train": {"optimizer": {"type": "Adam", "learning_rate": 1e-4}},
        "losses": {"weights": {"depth_recon": 10.0, "depth_smth": 0.001, "semantic": 1.0, "center": 20.0, "offset": 0.1,
                               "thing_mask": 0.5}},
        "adapt": {"mixup": {"general": {"active": True, "mixup_strategies": ["cut_mixup", "conf_instance_mixup"], "nof_samples": 2,
                                        "geom_augment": True},                            # default_config_adapt.py:131-147
                            "cut_mix": {"nof_hor_splits": 4, "nof_vert_splits": 4, "nof_segments": 1},
                            "conf_instance_mix": {"conf_thresh": 0.0, "min_inst_size": 0}, "class_mix": {"conf_thresh": 0.0}},
                  "source_dataset": {"feed_img_size": [512, 1024]},
                  "model": {"backbone_po_depth": False, "backbone_pose_sflow": True, "depth": True, "pose": True, "sflow": False,
                            "semantic": True, "instance": False},
                  "ema": {"alpha": 0.99, "semantic": True, "instance": False}},
    })
    if adapt:
        cfg.losses.weights = _wrap({"depth_recon": 10.0, "depth_smth": 0.001, "semantic_source": 1.0, "semantic_cut_mixup": 1.0,
                                    "center_source": 20.0, "offset_source": 0.1})
    return cfg
