"""Checkpoint and config plumbing at the drop-in boundary, in the reference's formats (io_utils/io_utils.py):

    checkpoint = {"config": <YAML string>, "state_dict": {<module name>: state_dict or None, ...}, "training_meta": {...}}

`CodepsNet.get_state_dict()` (codeps/online_adap.py:438-455) produces the per-module dictionary; loading is per module,
non-strict, with the reference's extra rule that a key present on both sides must have the same shape.  Logging, run
directories and visualisation stay out of scope."""
from typing import Any, Dict, List

import torch
import yaml

from ..config import CfgNode, get_cfg_defaults


def make_modules_list(config) -> List[str]:
    """io_utils/io_utils.py:18-33: the modules a configuration instantiates (order is irrelevant: the reference returns list(set))."""
    m = config.model
    wanted = (("backbone_po_depth", m.make_depth or m.make_semantic or m.make_instance),
              ("backbone_pose_sflow", m.make_depth), ("depth_head", m.make_depth), ("pose_head", m.make_depth),
              ("flow_head", m.make_depth and m.make_sflow), ("semantic_head", m.make_semantic), ("instance_head", m.make_instance))
    return [name for name, on in wanted if on]


def _plain(node):
    return {k: _plain(v) for k, v in node.items()} if isinstance(node, dict) else (list(node) if isinstance(node, tuple) else node)


def save_checkpoint(file: str, config, epoch: int, global_step: int, model_state_dict: Dict[str, Any], **kwargs):
    """io_utils/io_utils.py:76-88 (`config.dump()` is yacs' YAML string of the node)."""
    state_dict = dict(kwargs)
    state_dict.update(model_state_dict)                 # some modules might be None
    torch.save({"config": yaml.safe_dump(_plain(config)), "state_dict": state_dict,
                "training_meta": {"epoch": epoch, "global_step": global_step}}, file)


def _load_pretraining_dict(model, state_dict):
    """io_utils/io_utils.py:91-115: `load_state_dict(..., strict=False)` that refuses a shape mismatch on a shared key."""
    for k, v in model.state_dict().items():
        if k in state_dict and v.shape != state_dict[k].shape:
            raise AssertionError(f"The shape of the layer does not match: {k} - {tuple(state_dict[k].shape)} vs {tuple(v.shape)}")
    return model.load_state_dict(state_dict, False)


def _module_state(ckpt, name):
    """state_dict of one module inside a loaded checkpoint; a missing or None entry is the reference's KeyError."""
    sd = ckpt["state_dict"].get(name)
    if sd is None:
        raise KeyError(f"The given checkpoint does not contain a state_dict for module {name}")
    return sd


def resume_from_checkpoint(model, checkpoint: str, modules: List[str]):
    """io_utils/io_utils.py:36-49: every listed module must be in the file; modules the model does not have are skipped."""
    ckpt = torch.load(checkpoint, map_location="cpu")
    for name in modules:
        sd, target = _module_state(ckpt, name), getattr(model, name)
        if target is not None:
            _load_pretraining_dict(target, sd)
    return ckpt


def pretrained_from_checkpoints(model, checkpoints: List[str], modules: List[str], rank: int = 0):
    """io_utils/io_utils.py:52-73: "module:path" entries load one module each from another checkpoint; a bare path is a
    whole-body state dict."""
    for entry in checkpoints:
        name, _, path = entry.rpartition(":")
        ckpt = torch.load(path, map_location="cpu")
        if not name:
            _load_pretraining_dict(model.body, ckpt)
        elif name in modules:
            _load_pretraining_dict(getattr(model, name), _module_state(ckpt, name))
        else:
            raise ValueError(f"Unrecognized network module {name}")


def gen_config(filename_config=None, adapt=False) -> CfgNode:
    """io_utils/io_utils.py:220-237 for the hot-path keys: defaults, the YAML merged over them, and -- in adaptation mode --
    the source dataset inheriting the target dataset's label settings."""
    cfg = get_cfg_defaults(adapt)
    if filename_config is not None:
        cfg.merge_from_file(filename_config)
    if adapt:
        for k in ("center_heatmap_sigma", "small_instance_area_full_res", "small_instance_weight", "remove_classes"):
            cfg.adapt.source_dataset[k] = cfg.dataset[k]
    return cfg
