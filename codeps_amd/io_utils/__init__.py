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
    modules = []
    if config.model.make_depth:
        modules += ["backbone_po_depth", "backbone_pose_sflow", "depth_head", "pose_head"]
        if config.model.make_sflow:
            modules.append("flow_head")
    if config.model.make_semantic:
        modules += ["backbone_po_depth", "semantic_head"]
    if config.model.make_instance:
        modules += ["backbone_po_depth", "instance_head"]
    return list(dict.fromkeys(modules))


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


def resume_from_checkpoint(model, checkpoint: str, modules: List[str]):
    """io_utils/io_utils.py:36-49."""
    checkpoint = torch.load(checkpoint, map_location="cpu")
    state_dict = checkpoint["state_dict"]
    for module in modules:
        if module in state_dict and state_dict[module] is not None:
            model_module = getattr(model, module)
            if model_module is not None:
                _load_pretraining_dict(model_module, state_dict[module])
        else:
            raise KeyError(f"The given checkpoint does not contain a state_dict for module {module}")
    return checkpoint


def pretrained_from_checkpoints(model, checkpoints: List[str], modules: List[str], rank: int = 0):
    """io_utils/io_utils.py:52-73: "module:path" entries load one module each from another checkpoint."""
    for entry in checkpoints:
        module, path = entry.split(":") if ":" in entry else (None, entry)
        checkpoint = torch.load(path, map_location="cpu")
        if module is None:
            _load_pretraining_dict(getattr(model, "body"), checkpoint)
            continue
        if module not in modules:
            raise ValueError(f"Unrecognized network module {module}")
        state_dict = checkpoint["state_dict"]
        if module not in state_dict or state_dict[module] is None:
            raise KeyError(f"The given checkpoint does not contain a state_dict for module {module}")
        _load_pretraining_dict(getattr(model, module), state_dict[module])


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
