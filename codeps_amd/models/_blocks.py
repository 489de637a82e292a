"""Shared helpers: conv + BatchNorm (+ReLU, +residual) on NHWC through the C ABI."""
import os
import weakref

import torch
from torch import nn

from .. import functional as F

_fold_cache = {}

# num_batches_tracked increments of the train-mode BatchNorm layers of the module forward in flight: applied by ONE
# multi-tensor kernel in flush_bn_counters() (called at the end of every public forward that contains BatchNorm layers)
# instead of one tiny kernel per layer
_nbt_pending = []


_stats_epoch = [0]
_FORK = os.environ.get("CODEPS_FORK", "1") != "0"      # A/B switch: 0 = autograd sums the gradients of a tensor with two consumers


def _count_batch(bn, n):
    bn._codeps_stats_live = True          # its running statistics are written by kernels from now on (folded_bn)
    _stats_epoch[0] += 1
    if bn.num_batches_tracked is not None:
        _nbt_pending.append((bn.num_batches_tracked, n))


def flush_bn_counters():
    if _nbt_pending:
        F.counters_add([t for t, _ in _nbt_pending], [n for _, n in _nbt_pending])
        _nbt_pending.clear()


def folded_bn(bn: nn.BatchNorm2d):
    """(scale, shift) of an eval-mode BatchNorm, cached while its tensors are unchanged."""
    key = id(bn)
    # values that change behind torch's version counters: affine parameters the fused optimizer writes through raw pointers
    # (trainable, or teacher views), running statistics a train-mode forward of this library has written (_count_batch marks the
    # module).  Such a fold lives for one weights epoch; a frozen BatchNorm's fold lives as long as its tensors are unchanged.
    volatile = (F._is_volatile(bn.weight) or F._is_volatile(bn.bias) or getattr(bn, "_codeps_stats_live", False))
    ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), (F._weights_epoch[0], _stats_epoch[0]) if volatile else 0)
    hit = _fold_cache.get(key)
    if hit is not None and hit[0] == ver and hit[3]() is bn:
        return hit[1], hit[2]
    with torch.no_grad():
        scale, shift = F.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    if volatile:
        scale._codeps_volatile = shift._codeps_volatile = True
    _fold_cache[key] = (ver, scale, shift, weakref.ref(bn, lambda _r, k=key: _fold_cache.pop(k, None)))
    return scale, shift


def conv_bn(x, conv: nn.Conv2d, bn: nn.BatchNorm2d, act=None, residual=None, sync=None, groups=None, stem=False, fork=False,
            exclusive=False, defer=False):
    """act(BN(conv(x)) + residual).  Eval-mode BN is folded into the conv epilogue (one kernel);
    train-mode BN uses batch statistics (`sync` = cross-rank sum reducer for SyncBatchNorm).
    stem=True: x is in the double-pixel stem layout and conv is ResNet's conv1 (F.stem_conv2d).
    fork=True -> (out, x'): x' is x for the block's other consumer of the tensor (identity / shortcut); when x needs a gradient the
    convolution's data-gradient kernel then adds the gradient that comes back through x' in its epilogue (F.conv2d(fork=True)).
    exclusive=True: this call (with the taker of its fork alias) is x's only consumer, so the backward of the activation that produced
    x may run in this convolution's data-gradient epilogue (F.conv2d(exclusive=True)).  Train-mode BatchNorm takes its batch statistics
    from the convolution's epilogue (F.conv2d(bn_stats=True)) where the launch plan allows.
    defer=True: the result's ONLY reader is the next conv_bn / F.conv2d call -- a train-mode BatchNorm + ReLU then leaves its apply pass to
    that convolution's operand load (F.batch_norm_train(defer=True)); the tensor returned holds BatchNorm's input and must go nowhere else."""
    if fork:
        if not (_FORK and torch.is_grad_enabled() and x.requires_grad) or stem:
            # (without the fork the alias is x itself: a second consumer autograd sums separately -> not exclusive)
            return conv_bn(x, conv, bn, act, residual, sync, groups, stem, defer=defer), x
        return _conv_bn_fork(x, conv, bn, act, residual, sync, groups, exclusive, defer)
    stride, pad = conv.stride[0], conv.padding[0]
    if stem == "s3":                   # frozen 3-channel stem (eval BN folded, no gradients): F.stem3_conv2d, K = 224
        scale, shift = folded_bn(bn)
        return F.stem3_conv2d(x, conv.weight, scale=scale, shift=shift, act=act)
    if stem:
        if bn.training or (torch.is_grad_enabled() and (bn.weight.requires_grad or bn.bias.requires_grad)):
            y = F.stem_conv2d(x, conv.weight, bn_stats=bn.training)
            if bn.training:
                out = F.batch_norm_train(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                         residual=residual, act=act, reduce_fn=sync, groups=groups)
                _count_batch(bn, len(groups) if groups else 1)
                return out
            return F.batch_norm_eval(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, residual=residual, act=act)
        scale, shift = folded_bn(bn)
        return F.stem_conv2d(x, conv.weight, scale=scale, shift=shift, act=act)
    if bn.training:
        y = F.conv2d(x, conv.weight, stride=stride, pad=pad, exclusive=exclusive, bn_stats=True)
        out = F.batch_norm_train(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                 residual=residual, act=act, reduce_fn=sync, groups=groups, defer=defer)
        _count_batch(bn, len(groups) if groups else 1)
        return out
    if torch.is_grad_enabled() and (bn.weight.requires_grad or bn.bias.requires_grad):
        # eval statistics but trainable affine parameters: keep BN as its own op so gamma/beta get gradients
        y = F.conv2d(x, conv.weight, stride=stride, pad=pad, exclusive=exclusive)
        return F.batch_norm_eval(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, residual=residual, act=act)
    scale, shift = folded_bn(bn)
    return F.conv2d(x, conv.weight, scale=scale, shift=shift, residual=residual, stride=stride, pad=pad, act=act, exclusive=exclusive)


def _conv_bn_fork(x, conv, bn, act, residual, sync, groups, exclusive=False, defer=False):
    """conv_bn for an input that needs a gradient and has a second consumer (BN kept as its own op: x needs a gradient, so either BN is
    in train mode or, with eval statistics, its output still has to carry the graph)."""
    stride, pad = conv.stride[0], conv.padding[0]
    if bn.training:
        y, xa = F.conv2d(x, conv.weight, stride=stride, pad=pad, fork=True, exclusive=exclusive, bn_stats=True)
        out = F.batch_norm_train(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                 residual=residual, act=act, reduce_fn=sync, groups=groups, defer=defer)
        _count_batch(bn, len(groups) if groups else 1)
        return out, xa
    if bn.weight.requires_grad or bn.bias.requires_grad:
        y, xa = F.conv2d(x, conv.weight, stride=stride, pad=pad, fork=True, exclusive=exclusive)
        return F.batch_norm_eval(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, residual=residual, act=act), xa
    scale, shift = folded_bn(bn)
    return F.conv2d(x, conv.weight, scale=scale, shift=shift, residual=residual, stride=stride, pad=pad, act=act, fork=True,
                    exclusive=exclusive)


def dwconv_bn_relu(x, conv: nn.Conv2d, bn: nn.BatchNorm2d, sync=None, defer=False):
    """5x5 depthwise + BN + ReLU (models/instance_head.py:38-46).  defer: see conv_bn."""
    if bn.training:
        y = F.dwconv5x5(x, conv.weight)
        out = F.batch_norm_train(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, act="relu",
                                 reduce_fn=sync, defer=defer)
        _count_batch(bn, 1)
        return out
    if torch.is_grad_enabled() and (bn.weight.requires_grad or bn.bias.requires_grad):
        y = F.dwconv5x5(x, conv.weight)
        return F.batch_norm_eval(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, act="relu")
    scale, shift = folded_bn(bn)
    return F.dwconv5x5(x, conv.weight, scale, shift, "relu")
