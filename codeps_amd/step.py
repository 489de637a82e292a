"""The timed unit: one continual-adaptation step (reference region scripts/adapt_codeps.py:260-299) or
one pre-training step (scripts/train_codeps.py:97-108).

    zero flat grads -> CodepsNet forward (adapt / train) -> sum_k w_k loss_k -> backward
    -> ONE all-reduce of the flat gradient buffer (world > 1) -> fused Adam (+ EMA teacher) on flat buffers

All trainable parameters live in ONE flat fp32 buffer (views handed back to the nn.Parameters), likewise
their gradients, Adam moments and the EMA copy; conv weight / bias gradients are accumulated straight into
the flat buffer by the wgrad / bias kernels (codeps_amd.functional.register_grad_sink).
"""
from collections import OrderedDict

import torch

from . import functional as F
from .distributed import allreduce_flat_gradients

ADAPT_MODULES = OrderedDict([("backbone_po_depth", "backbone_po_depth"), ("backbone_pose_sflow", "backbone_pose_sflow"),
                             ("depth", "depth_head"), ("pose", "pose_head"), ("semantic", "semantic_head"),
                             ("instance", "instance_head")])


def freeze_for_adaptation(net, cfg):
    """misc/train_utils.py:204-226 + scripts/adapt_codeps.py:261-265: modules with cfg.adapt.model.<x> False get
    requires_grad False and stay in eval mode (BN uses running statistics)."""
    frozen = []
    for key, attr in ADAPT_MODULES.items():
        mod = getattr(net, attr, None)
        if mod is None:
            continue
        train = bool(cfg.adapt.model.get(key, False))
        for p in mod.parameters():
            p.requires_grad_(train)
        mod.train(train)
        if not train:
            frozen.append(attr)
    return frozen


class FlatState:
    """Flat parameter / gradient / Adam / EMA buffers over a list of parameters."""

    def __init__(self, params, with_ema):
        self.params = params
        n = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in params:
            k = p.numel()
            self.flat[off:off + k] = p.data.reshape(-1)
            p.data = self.flat[off:off + k].view_as(p)
            g = self.grad[off:off + k].view_as(p)
            p.grad = g
            if p.dim() == 4 and p.shape[1] != 1 or (p.dim() == 1 and (getattr(p, "_codeps_conv_bias", False) or
                                                                     getattr(p, "_codeps_bn_affine", False))):
                F.register_grad_sink(p, g)
            off += k
        self.ema = self.flat.clone() if with_ema else None
        self.numel = n


def _mark_conv_biases(net):
    """Tag the 1-D parameters whose backward kernels can accumulate straight into the flat gradient buffer."""
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d) and m.bias is not None:
            m.bias._codeps_conv_bias = True
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.affine:
            m.weight._codeps_bn_affine = True
            m.bias._codeps_bn_affine = True


class StepRunner:
    """mode "adapt": CodepsNet.adapt with the adapt loss weights; mode "train": CodepsNet.forward("train")."""

    def __init__(self, net, cfg, mode="adapt", world_size=1, group=None, lr=None):
        self.net, self.cfg, self.mode, self.world, self.group = net, cfg, mode, world_size, group
        self.weights = dict(cfg.losses.weights)
        self.lr = float(cfg.train.optimizer.learning_rate if lr is None else lr)
        _mark_conv_biases(net)
        ema_mods = []
        if mode == "adapt":
            self.frozen = freeze_for_adaptation(net, cfg)
            ema_mods = [ADAPT_MODULES[k] for k in ("semantic", "instance", "depth", "pose") if cfg.adapt.ema.get(k, False)]
            self.ema_alpha = float(cfg.adapt.ema.alpha)
        else:
            net.train()
            self.frozen, self.ema_alpha = [], 0.0
        # parameters that actually receive gradients (dead dispconvs.1-3 / fc are excluded instead of relying on
        # find_unused_parameters)
        def live(name):
            return not (".fc." in name or name.startswith("fc.") or any(f"dispconvs.{i}." in name for i in (1, 2, 3)))
        ema_params, other = [], []
        for attr in ADAPT_MODULES.values():
            mod = getattr(net, attr, None)
            if mod is None:
                continue
            for name, p in mod.named_parameters():
                if p.requires_grad and live(name):
                    (ema_params if attr in ema_mods else other).append(p)
        self.groups = [FlatState(ps, with_ema) for ps, with_ema in ((ema_params, True), (other, False)) if ps]
        self.step_count = 0
        self.net.check_car_moving = False          # host-side rolling rule needs a D2H sync; evaluated by the caller

    @property
    def grad_bytes(self):
        return sum(g.numel * 4 for g in self.groups)

    def forward_backward(self, data):
        for g in self.groups:
            g.grad.zero_()
        if self.mode == "adapt":
            losses, result, _, _ = self.net(data, "adapt", [0, -1, 1], do_panoptic_fusion=True)
        else:
            losses, result, _ = self.net(data, "train", [0, -1, 1])
        total = None
        for k, v in losses.items():
            w = self.weights.get(k)
            if v is None or w is None or not v.requires_grad:        # scripts/adapt_codeps.py:274-288
                continue
            total = v * w if total is None else total + v * w
        total.backward()
        return total.detach(), losses, result

    def optimizer_step(self):
        self.step_count += 1
        for g in self.groups:
            allreduce_flat_gradients(g.grad, self.group)
            F.adam_ema_step(g.flat, g.grad, g.m, g.v, self.step_count, lr=self.lr, grad_scale=1.0 / self.world,
                            ema_params=g.ema, ema_alpha=self.ema_alpha)

    def step(self, data):
        out = self.forward_backward(data)
        self.optimizer_step()
        return out
