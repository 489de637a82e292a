"""Per-frame evaluation passes that precede every adaptation step (reference:
scripts/adapt_codeps.py:164-199): the EMA teacher is evaluated on the incoming target frame, it produces the pseudo labels
of the mix-up items, and -- when the instance head is not EMA-filtered -- the student completes them with its centre /
offset predictions.  The teacher is a deep copy of the student whose encoder stays frozen, so both hold the same encoder
weights: the encoder then runs ONCE per distinct image and its features feed both networks (`CodepsNet.forward(feats=)`),
with results identical to the reference's separate calls.
"""
import weakref
from typing import Any, Dict, List, Optional, Tuple

import torch


_same_cache: Dict[Tuple[int, int], Tuple[Any, Any, Tuple[int, ...], bool]] = {}


def _same_frozen_encoder(a, b) -> bool:
    """True when both networks' shared encoders are in eval mode, frozen and hold equal tensors.  The value comparison runs
    once per (pair, tensor versions): frozen tensors are never written, so later frames pay a few attribute reads."""
    ea, eb = a.backbone_po_depth, b.backbone_po_depth
    if ea is eb:
        return not ea.training
    if ea.training or eb.training or any(p.requires_grad for p in ea.parameters()):
        return False
    sa, sb = ea.state_dict(), eb.state_dict()
    if sa.keys() != sb.keys():
        return False
    sig = tuple(t._version for t in sa.values()) + tuple(t._version for t in sb.values()) + tuple(t.data_ptr() for t in sb.values())
    hit = _same_cache.get((id(ea), id(eb)))
    if hit is not None and hit[0]() is ea and hit[1]() is eb and hit[2] == sig:
        return hit[3]
    same = all(torch.equal(sa[k], sb[k]) for k in sa)
    _same_cache[(id(ea), id(eb))] = (weakref.ref(ea), weakref.ref(eb), sig, same)
    return same


def pre_step(model, ema_model, ema_modules: List[str], sample: Dict[str, Dict[str, Any]], rgb_frame_offsets: List[int],
             do_panoptic_fusion: bool, do_class_wise_depth_stats: bool = False,
             share_encoder: bool = True, batch_encoder: bool = True, needed_heads_only: bool = True) -> Tuple[Dict[str, Any], Dict[str, Any]]:
    """-> (results_eval, stats_eval) of the target frame; fills `semantic_pred_tgt_ema`, `center_pred_tgt_ema`,
    `offset_pred_tgt_ema` of every `*mixup` item in place, exactly as scripts/adapt_codeps.py:164-199 does.
    `model` is left in whatever train / eval state it is in (the reference does not switch the student here).

    Three things the reference's call sequence leaves on the table (each switchable, results unchanged):
      share_encoder      teacher and student hold the same frozen encoder -> it runs once per mix-up batch, not twice;
      batch_encoder      ... and the target frame joins that pass when it has the mix-up images' size (one B = 1 + n launch
                         sequence instead of a B = 1 and a B = n one; the convolution kernels then tile / split K for the
                         larger batch, so values agree to fp32 rounding rather than bit for bit);
      needed_heads_only  the passes over the mix-up items evaluate the heads whose predictions are kept (teacher: semantic,
                         plus instance when it is EMA-filtered; student: instance), not all three."""
    mixup_keys = [k for k in sample if k.endswith("mixup")]
    shared: Dict[str, Optional[List[torch.Tensor]]] = {k: None for k in mixup_keys}
    teacher_heads = student_heads = None
    if needed_heads_only:
        teacher_heads = tuple(h for h, on in (("semantic", "semantic_head" in ema_modules or "instance_head" in ema_modules),
                                              ("instance", "instance_head" in ema_modules)) if on)
        student_heads = ("instance",)
    with torch.no_grad():
        if ema_model is not None:
            ema_model.eval()
            student_needs = "instance_head" not in ema_modules
            same_enc = share_encoder and bool(mixup_keys) and _same_frozen_encoder(model, ema_model)
            share = same_enc and student_needs
            tgt = sample["target"]
            tgt_rgb = tgt["rgb" if "rgb" in tgt else "rgb_tgt"][0]
            mix_rgb = {k: sample[k]["rgb" if "rgb" in sample[k] else "rgb_tgt"][0] for k in mixup_keys}
            tgt_feats = None
            joined = [k for k in mixup_keys if mix_rgb[k].shape[1:] == tgt_rgb.shape[1:]] if (batch_encoder and share_encoder) else []
            if joined:                                      # one encoder pass over the target frame and the mix-up images
                cat = torch.cat([tgt_rgb] + [mix_rgb[k] for k in joined], 0)
                feats_all = ema_model.backbone_po_depth(cat)
                n0 = tgt_rgb.shape[0]
                tgt_feats = [f[:n0] for f in feats_all]
                for k in joined:
                    n1 = n0 + mix_rgb[k].shape[0]
                    shared[k] = [f[n0:n1] for f in feats_all]
                    n0 = n1
            results_eval, stats_eval = ema_model(tgt, "eval", rgb_frame_offsets, do_panoptic_fusion,
                                                 do_class_wise_depth_stats, feats=tgt_feats)[1:]
            for key in mixup_keys:
                item = sample[key]
                feats = shared[key]
                if feats is None and share:                 # one encoder pass per mix-up batch for teacher AND student
                    feats = shared[key] = ema_model.backbone_po_depth(mix_rgb[key])
                if teacher_heads is None or teacher_heads:
                    res = ema_model(item, "eval", rgb_frame_offsets, False, False, feats=feats, heads=teacher_heads)[1]
                    if "semantic_head" in ema_modules or "instance_head" in ema_modules:
                        item["semantic_pred_tgt_ema"] = res["semantic"].detach().clone()
                    if "instance_head" in ema_modules:
                        item["center_pred_tgt_ema"] = res["center"].detach().clone()
                        item["offset_pred_tgt_ema"] = res["offset"].detach().clone()
                if not (same_enc and student_needs):
                    shared[key] = None                      # the student's encoder differs: it runs its own
        else:
            model.eval()
            results_eval, stats_eval = model(sample["target"], "eval", rgb_frame_offsets, do_panoptic_fusion,
                                             do_class_wise_depth_stats)[1:]
    # complete the predictions for the mix-up strategy (the reference runs this part with gradients enabled; the results
    # are detached, so it is evaluated without them here)
    if "instance_head" not in ema_modules:
        with torch.no_grad():
            for key in mixup_keys:
                item = sample[key]
                res = model(item, "eval", rgb_frame_offsets, False, False, feats=shared[key], heads=student_heads)[1]
                item["center_pred_tgt_ema"] = res["center"].detach().clone()
                item["offset_pred_tgt_ema"] = res["offset"].detach().clone()
    return results_eval, stats_eval


class OnlineAdapter:
    """The per-frame loop of scripts/adapt_codeps.py:146-310 around one model: teacher / student evaluation passes
    (`pre_step`), `nof_adaptation_steps` adaptation steps (StepRunner: forward, backward, one gradient all-reduce, fused
    Adam + EMA), and the frame's offer to the replay buffer.

    The reference keeps the teacher as a deep copy and, after every step, rewrites it parameter by parameter: an EMA for the
    modules in cfg.adapt.ema, a plain copy (alpha = 0) of the student for every other module (adapt_codeps.py:293-299,
    online_adap.py:279-289).  Here the teacher's parameters are VIEWS: EMA'd modules look into the EMA buffer that the fused
    optimizer kernel maintains, the others share the student's storage -- the teacher is current after every step without a
    single copy kernel.  Frozen modules are equal by construction."""

    def __init__(self, model, cfg, world_size: int = 1, group=None, replay=None, rgb_frame_offsets=(0, -1, 1),
                 do_panoptic_fusion: bool = True, nof_adaptation_steps: int = 1):
        from ..step import StepRunner, ema_module_lists
        from .online_adap import CodepsNet
        self.model, self.cfg, self.replay = model, cfg, replay
        self.offsets, self.do_panoptic_fusion, self.steps = list(rgb_frame_offsets), do_panoptic_fusion, int(nof_adaptation_steps)
        self.runner = StepRunner(model, cfg, "adapt", world_size, group)
        self.ema_modules, _ = ema_module_lists(cfg, self.runner.frozen)
        self.ema_model = None
        if self.ema_modules:                                             # adapt_codeps.py:146-150
            self.ema_model = CodepsNet.ema_model(model)
            self._bind_teacher()

    def _bind_teacher(self):
        teacher_of = {id(p): t for (_, t), (_, p) in zip(self.ema_model.named_parameters(), self.model.named_parameters())}
        for g in self.runner.groups:
            off = 0
            for p in g.params:
                k = p.numel()
                t = teacher_of[id(p)]
                src = g.ema if g.ema is not None else g.flat
                t.data = src[off:off + k].view_as(p)
                t._codeps_volatile = True            # rewritten by the optimizer kernel: weight packs must not outlive a step
                off += k

    def process(self, sample: Dict[str, Dict[str, Any]], index: Optional[int] = None, do_class_wise_depth_stats: bool = False):
        """One incoming frame -> (results_eval, stats_eval, losses, result) exactly in the script's order: evaluate, adapt,
        store.  `sample` is the data loader's dict (target / target_replay / source / *mixup raw items)."""
        results_eval, stats_eval = pre_step(self.model, self.ema_model, self.ema_modules, sample, self.offsets,
                                            self.do_panoptic_fusion, do_class_wise_depth_stats)
        losses = result = None
        for i in range(self.steps):          # the EMA filter runs once per frame, after the last optimizer step (adapt_codeps.py:293-299)
            _, losses, result = self.runner.step(sample, apply_ema=(i == self.steps - 1))
        if self.replay is not None and index is not None:                # adapt_codeps.py:307
            self.replay.add(int(index), result["image_features"][:1])
        return results_eval, stats_eval, losses, result
