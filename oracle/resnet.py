"""CPU oracle: ResNet trunk as used by the reference's ``ResnetEncoder``.  TEST INFRASTRUCTURE.

Reference call sites: models/resnet_encoder.py:19-43 (ResNetMultiImageInput: conv1 takes
num_input_images*num_channels_input channels, kaiming-normal fan_out init, BN weight 1 / bias 0),
models/resnet_encoder.py:82-126 (ResnetEncoder: picks torchvision resnet{18,34,50,101,152}; forward
returns [relu(bn1(conv1 x)), layer1(maxpool(.)), layer2, layer3, layer4]; avgpool/fc never called).

The block arithmetic itself lives in torchvision==0.13.1 (reference requirements.txt:2), which is not
vendored and not installed here, so this file restates the *published* torchvision v0.13 ResNet:
  - stem conv 7x7/2 pad 3 no bias, BN(eps 1e-5, momentum 0.1), ReLU, maxpool 3x3/2 pad 1
  - BasicBlock (expansion 1):  3x3(stride s) BN ReLU 3x3 BN (+identity) ReLU
  - Bottleneck v1.5 (expansion 4): 1x1 BN ReLU 3x3(stride s) BN ReLU 1x1 BN (+identity) ReLU
  - downsample = conv1x1(stride s, no bias)+BN iff s != 1 or inplanes != planes*expansion
  - zero_init_residual=False
PARITY UNPINNED against a reference run (no torchvision here); pinned by self-consistency tests and, as a second opinion,
against transformers.ResNetModel (an independent implementation of the same architecture) loaded with these weights:
tests/test_trunk_crosscheck_cpu.py.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

LAYERS = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]),
          101: ("bottleneck", [3, 4, 23, 3]), 152: ("bottleneck", [3, 8, 36, 3])}
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def block_plan(num_layers):
    """Yield (prefix, kind, inplanes, planes, stride, has_downsample) for every residual block."""
    kind, counts = LAYERS[num_layers]
    exp = 1 if kind == "basic" else 4
    inplanes = 64
    plan = []
    for li, (planes, n) in enumerate(zip([64, 128, 256, 512], counts), start=1):
        for bi in range(n):
            stride = 2 if (bi == 0 and li > 1) else 1
            ds = bi == 0 and (stride != 1 or inplanes != planes * exp)
            plan.append((f"layer{li}.{bi}", kind, inplanes, planes, stride, ds))
            inplanes = planes * exp
    return plan


def num_ch_enc(num_layers):
    c = [64, 64, 128, 256, 512]
    if num_layers > 34:
        c = [c[0]] + [4 * v for v in c[1:]]
    return c


def _bn_state(sd, prefix, c):
    sd[prefix + ".weight"] = torch.ones(c)
    sd[prefix + ".bias"] = torch.zeros(c)
    sd[prefix + ".running_mean"] = torch.zeros(c)
    sd[prefix + ".running_var"] = torch.ones(c)
    sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def _kaiming_fan_out(shape, gen):
    fan_out = shape[0] * shape[2] * shape[3]
    std = (2.0 / fan_out) ** 0.5
    return torch.randn(shape, generator=gen) * std


def make_state_dict(num_layers, in_channels=3, seed=0, with_fc=True):
    """Random-init state dict with torchvision's key names (prefix-free, i.e. what sits under
    ``encoder.`` in the reference's ResnetEncoder).  Init = kaiming-normal(fan_out) convs, BN 1/0
    (models/resnet_encoder.py:38-43; torchvision uses the same for the single-image variants)."""
    gen = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    sd["conv1.weight"] = _kaiming_fan_out((64, in_channels, 7, 7), gen)
    _bn_state(sd, "bn1", 64)
    for prefix, kind, inpl, planes, stride, ds in block_plan(num_layers):
        if kind == "basic":
            sd[prefix + ".conv1.weight"] = _kaiming_fan_out((planes, inpl, 3, 3), gen)
            _bn_state(sd, prefix + ".bn1", planes)
            sd[prefix + ".conv2.weight"] = _kaiming_fan_out((planes, planes, 3, 3), gen)
            _bn_state(sd, prefix + ".bn2", planes)
            out = planes
        else:
            sd[prefix + ".conv1.weight"] = _kaiming_fan_out((planes, inpl, 1, 1), gen)
            _bn_state(sd, prefix + ".bn1", planes)
            sd[prefix + ".conv2.weight"] = _kaiming_fan_out((planes, planes, 3, 3), gen)
            _bn_state(sd, prefix + ".bn2", planes)
            sd[prefix + ".conv3.weight"] = _kaiming_fan_out((planes * 4, planes, 1, 1), gen)
            _bn_state(sd, prefix + ".bn3", planes * 4)
            out = planes * 4
        if ds:
            sd[prefix + ".downsample.0.weight"] = _kaiming_fan_out((out, inpl, 1, 1), gen)
            _bn_state(sd, prefix + ".downsample.1", out)
    if with_fc:
        feat = 512 * (1 if LAYERS[num_layers][0] == "basic" else 4)
        bound = 1.0 / feat ** 0.5
        sd["fc.weight"] = (torch.rand((1000, feat), generator=gen) * 2 - 1) * bound
        sd["fc.bias"] = (torch.rand((1000,), generator=gen) * 2 - 1) * bound
    return sd


def _bn(sd, prefix, x, training):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], training, BN_MOMENTUM, BN_EPS)


def features(sd, x, num_layers, training=False):
    """[feat0..feat4] as models/resnet_encoder.py:115-126.  ``training`` selects batch statistics
    (and updates running stats in ``sd`` in place, like nn.BatchNorm2d.train())."""
    x = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], None, 2, 3), training))
    feats = [x]
    x = F.max_pool2d(x, 3, 2, 1)
    cur_layer = "layer1"
    for prefix, kind, _inpl, _planes, stride, ds in block_plan(num_layers):
        layer = prefix.split(".")[0]
        if layer != cur_layer:
            feats.append(x)
            cur_layer = layer
        identity = x
        if kind == "basic":
            out = F.relu(_bn(sd, prefix + ".bn1",
                             F.conv2d(x, sd[prefix + ".conv1.weight"], None, stride, 1), training))
            out = _bn(sd, prefix + ".bn2",
                      F.conv2d(out, sd[prefix + ".conv2.weight"], None, 1, 1), training)
        else:
            out = F.relu(_bn(sd, prefix + ".bn1",
                             F.conv2d(x, sd[prefix + ".conv1.weight"]), training))
            out = F.relu(_bn(sd, prefix + ".bn2",
                             F.conv2d(out, sd[prefix + ".conv2.weight"], None, stride, 1), training))
            out = _bn(sd, prefix + ".bn3", F.conv2d(out, sd[prefix + ".conv3.weight"]), training)
        if ds:
            identity = _bn(sd, prefix + ".downsample.1",
                           F.conv2d(x, sd[prefix + ".downsample.0.weight"], None, stride), training)
        x = F.relu(out + identity)
    feats.append(x)
    return feats


def conv_macs(num_layers, h, w, in_channels=3):
    """Analytic conv multiply-accumulates per image (self-consistency pin vs. published numbers)."""
    ho, wo = (h + 1) // 2, (w + 1) // 2
    macs = ho * wo * 64 * in_channels * 49
    ho, wo = (ho + 1) // 2, (wo + 1) // 2
    for _prefix, kind, inpl, planes, stride, ds in block_plan(num_layers):
        hi, wi = ho, wo
        if stride == 2:
            ho, wo = (ho + 1) // 2, (wo + 1) // 2
        if kind == "basic":
            macs += ho * wo * planes * inpl * 9 + ho * wo * planes * planes * 9
            out = planes
        else:
            macs += hi * wi * planes * inpl + ho * wo * planes * planes * 9 + ho * wo * planes * 4 * planes
            out = planes * 4
        if ds:
            macs += ho * wo * out * inpl
    return macs
