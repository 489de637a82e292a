"""Drop-in boundary, host side: configuration defaults / shipped YAMLs and the checkpoint format.

The comparisons against the reference's own files (cfg/default_config*.py, cfg/*.yaml) run where /root/reference exists (the
development container) and are skipped on the GPU box; the checkpoint round trip and the self-consistency of the shipped
overrides run everywhere.  Reference: io_utils/io_utils.py:18-115 (checkpoints), :220-237 (gen_config)."""
import copy
import os
import sys
import types

import pytest
import torch
import yaml

from codeps_amd import config as C
from codeps_amd import io_utils

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "cfg")), reason="reference tree not present (GPU box)")


class _CN(dict):
    """What the reference's config files need from yacs.CfgNode to be READ (yacs is not installed): attribute access."""

    def __getattr__(self, k):
        if k in self:
            return self[k]
        raise AttributeError(k)

    __setattr__ = dict.__setitem__

    def clone(self):
        return copy.deepcopy(self)


def _reference_defaults(adapt):
    path = os.path.join(REF, "cfg", "default_config_adapt.py" if adapt else "default_config.py")
    y, yc = types.ModuleType("yacs"), types.ModuleType("yacs.config")
    yc.CfgNode = _CN
    y.config = yc
    saved = {k: sys.modules.get(k) for k in ("yacs", "yacs.config")}
    sys.modules.update({"yacs": y, "yacs.config": yc})
    try:
        ns = {"__name__": "ref_cfg"}
        with open(path) as f:
            exec(compile(f.read(), path, "exec"), ns)                 # noqa: S102  (reads values from a config file)
        return ns["get_cfg_defaults"]()
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _norm(v):
    return [_norm(x) for x in v] if isinstance(v, (list, tuple)) else v


def _leaves(node, prefix=()):
    for k, v in node.items():
        if isinstance(v, dict):
            yield from _leaves(v, prefix + (k,))
        else:
            yield prefix + (k,), _norm(v)


def _get(node, path):
    for k in path:
        node = node[k]
    return node


# the ONE documented deviation of the defaults (codeps_amd/config.py::get_cfg_defaults) and the sub-tree the training config
# borrows from the adaptation defaults
DEVIATIONS = {("model", "po_depth_net", "params", "weights_init"), ("model", "pose_sflow_net", "params", "weights_init")}


@needs_ref
@pytest.mark.parametrize("adapt", [False, True])
def test_defaults_equal_the_reference_defaults(adapt):
    ours, ref = C.get_cfg_defaults(adapt), _reference_defaults(adapt)
    checked = 0
    for path, val in _leaves(ours):
        if path in DEVIATIONS:
            assert val == "random" and _get(ref, path) == "pretrained"
            continue
        if not adapt and path[0] == "adapt":
            # cfg.adapt.mixup in the training config: borrowed from the adaptation defaults (gen_models reads it unconditionally)
            if path[1] == "mixup":
                assert val == _norm(_get(_reference_defaults(True), path)), path
            continue
        try:
            want = _norm(_get(ref, path))
        except KeyError:
            raise AssertionError(f"{'.'.join(path)} is not a key of the reference defaults") from None
        assert val == want, f"{'.'.join(path)}: {val!r} here, {want!r} in the reference"
        checked += 1
    assert checked >= 35


@needs_ref
@pytest.mark.parametrize("adapt,fname,name", [(True, "adapt_cityscapes_kitti_360.yaml", "adapt_cityscapes_kitti_360"),
                                              (False, "train_cityscapes.yaml", "train_cityscapes")])
def test_shipped_yaml_merge_equals_the_shipped_dict(adapt, fname, name):
    """merge_from_file(<reference YAML>) over the defaults == get_cfg_shipped (what bench.build consumes), key by key; and the
    same merge over the REFERENCE's defaults agrees on every key this tree carries."""
    path = os.path.join(REF, "cfg", fname)
    merged = C.get_cfg_defaults(adapt).merge_from_file(path)
    shipped = C.get_cfg_shipped(adapt)
    ours = dict(_leaves(shipped))
    got = dict(_leaves(merged))
    for p, v in ours.items():
        if p[-1] == "path":                                 # where the data sets live on the authors' machine: not shipped
            continue
        assert got[p] == v, f"{'.'.join(p)}: YAML gives {got[p]!r}, shipped dict {v!r}"
    # keys the YAML adds that the tree does not carry are paths / logging / visualisation keys only
    extra = sorted(".".join(p) for p in got if p not in ours)
    allowed = ("dataset.path", "adapt.source_dataset.path", "visualization.", "logging.", "train.scheduler.")
    assert all(e.startswith(allowed) for e in extra), extra
    # the SHIPPED dict holds no key the YAML does not set
    with open(path) as f:
        raw = dict(_leaves(yaml.safe_load(f)))
    for p, v in _leaves(C.SHIPPED[name]):
        assert p in raw and _norm(raw[p]) == v, f"SHIPPED[{name}] {'.'.join(p)} = {v!r} is not what {fname} says"
    ref = C._wrap(_plain(_reference_defaults(adapt))).merge_from_file(path)
    for p, v in ours.items():
        if p in DEVIATIONS or (not adapt and p[0] == "adapt") or p[-1] == "path":
            continue
        assert _norm(_get(ref, p)) == v, f"{'.'.join(p)}: reference defaults + YAML give {_get(ref, p)!r}, here {v!r}"


def _plain(node):
    return {k: _plain(v) for k, v in node.items()} if isinstance(node, dict) else node


def test_shipped_adapt_config_is_the_headline_configuration():
    cfg = C.get_cfg_shipped(True)
    assert cfg.adapt.ema.alpha == 0.99 and cfg.adapt.ema.semantic is True and cfg.adapt.model.backbone_po_depth is False
    assert cfg.adapt.mixup.general.mixup_strategies == ["cut_mixup", "cut_mixup"]
    assert (cfg.adapt.mixup.cut_mix.nof_hor_splits, cfg.adapt.mixup.cut_mix.nof_vert_splits, cfg.adapt.mixup.cut_mix.nof_segments) == (4, 2, 2)
    assert cfg.adapt.replay_buffer.target_size == 300 and cfg.adapt.replay_buffer.maximize_diversity is True
    d = C.get_cfg_defaults(True)
    assert d.adapt.ema.alpha == 0.999 and d.adapt.ema.semantic is False and d.adapt.model.backbone_po_depth is True
    t = C.get_cfg_shipped(False)
    assert t.model.instance_head.use_thing_mask is True and t.losses.weights.center == 20.0 and len(t.semantics.class_weights) == 14
    assert len(C.get_cfg_defaults(False).semantics.class_weights) == 19


def _small_net(seed):
    from codeps_amd.codeps import gen_models
    cfg = C.get_cfg_shipped(True)
    cfg.model.po_depth_net.params.nof_layers = 50
    cfg.dataset.feed_img_size = [64, 128]
    cfg.adapt.source_dataset.feed_img_size = [64, 128]
    torch.manual_seed(seed)
    return gen_models(cfg, None, list(range(9)), [9, 10, 11, 12, 13], [], adaptation_mode=True), cfg


def test_checkpoint_round_trip_in_the_reference_format(tmp_path):
    """{"config", "state_dict": {<module>: sd}, "training_meta"} -> torch.save -> per-module, shape-checked, non-strict load
    (io_utils/io_utils.py:36-49,76-115), incl. the torchvision `encoder.fc.*` keys a reference checkpoint carries."""
    net, cfg = _small_net(1)
    sd = net.get_state_dict()
    assert set(sd) == {"backbone_po_depth", "backbone_pose_sflow", "depth_head", "pose_head", "flow_head", "semantic_head", "instance_head"}
    assert sd["flow_head"] is None
    assert "encoder.fc.weight" in sd["backbone_po_depth"] and "encoder.fc.bias" in sd["backbone_pose_sflow"]
    f = str(tmp_path / "ckpt.pth")
    io_utils.save_checkpoint(f, cfg, epoch=3, global_step=1234, model_state_dict=sd)
    raw = torch.load(f, map_location="cpu")
    assert set(raw) == {"config", "state_dict", "training_meta"} and raw["training_meta"] == {"epoch": 3, "global_step": 1234}
    assert yaml.safe_load(raw["config"])["adapt"]["ema"]["alpha"] == 0.99

    other, _ = _small_net(2)
    modules = io_utils.make_modules_list(cfg)
    assert sorted(modules) == ["backbone_po_depth", "backbone_pose_sflow", "depth_head", "instance_head", "pose_head", "semantic_head"]
    before = other.depth_head.state_dict()
    assert any(not torch.equal(v, sd["depth_head"][k]) for k, v in before.items() if v.is_floating_point())
    ck = io_utils.resume_from_checkpoint(other, f, modules)
    assert ck["training_meta"]["global_step"] == 1234
    for m in modules:
        for k, v in getattr(other, m).state_dict().items():
            assert torch.equal(v, sd[m][k]), f"{m}.{k} differs after the round trip"

    # non-strict: unknown keys are ignored, missing keys keep their values; a shape mismatch on a shared key is refused
    third, _ = _small_net(3)
    part = {k: v for k, v in sd["pose_head"].items() if not k.startswith("net.0.")}
    part["not.a.key"] = torch.zeros(3)
    keep = {k: v.clone() for k, v in third.pose_head.state_dict().items() if k.startswith("net.0.")}
    assert keep, "the pose head's first layer is the squeeze conv net.0 (models/pose_head.py:36-44)"
    io_utils._load_pretraining_dict(third.pose_head, part)
    for k, v in third.pose_head.state_dict().items():
        assert torch.equal(v, keep[k] if k in keep else sd["pose_head"][k])
    bad = dict(sd["depth_head"])
    k0 = next(k for k, v in bad.items() if v.dim() == 4)
    bad[k0] = bad[k0][:, :-1]
    with pytest.raises(AssertionError, match="shape of the layer does not match"):
        io_utils._load_pretraining_dict(third.depth_head, bad)
    # a checkpoint without a configured module raises like the reference
    sd2 = dict(sd)
    sd2["semantic_head"] = None
    f2 = str(tmp_path / "ckpt2.pth")
    io_utils.save_checkpoint(f2, cfg, 0, 0, sd2)
    with pytest.raises(KeyError, match="semantic_head"):
        io_utils.resume_from_checkpoint(third, f2, modules)
    # "module:path" form (io_utils.py:52-73)
    fourth, _ = _small_net(4)
    io_utils.pretrained_from_checkpoints(fourth, ["depth_head:" + f], modules)
    for k, v in fourth.depth_head.state_dict().items():
        assert torch.equal(v, sd["depth_head"][k])


def test_gen_config_mirrors_the_reference_merge_rules(tmp_path):
    y = tmp_path / "x.yaml"
    y.write_text("dataset:\n  remove_classes: [3]\n  feed_img_size: [64, 128]\nadapt:\n  ema:\n    alpha: 0.9\n")
    cfg = io_utils.gen_config(str(y), adapt=True)
    assert cfg.adapt.ema.alpha == 0.9 and cfg.adapt.ema.semantic is False            # only what the file sets moves
    assert cfg.adapt.source_dataset.remove_classes == [3] and cfg.adapt.source_dataset.center_heatmap_sigma == 8
