"""InstanceHead: drop-in for models/instance_head.py:10-161 (Panoptic-DeepLab decoder + centre / offset /
optional thing-mask heads)."""
from typing import Optional, Tuple

import numpy as np
from numpy.typing import ArrayLike
from torch import Tensor, nn

from .. import functional as F
from ._blocks import conv_bn, dwconv_bn_relu, flush_bn_counters


def _dwsep(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cin, kernel_size=5, stride=1, padding=2, groups=cin, bias=False),
                         nn.BatchNorm2d(cin), nn.ReLU(inplace=True),
                         nn.Conv2d(cin, cout, kernel_size=1, stride=1, padding=0, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


def _run_dwsep(x, seq, sync):
    x = dwconv_bn_relu(x, seq[0], seq[1], sync, defer=True)          # (... and its apply pass in that convolution's operand load)
    # (the 1 x 1 convolution is the only consumer of the depthwise stage's output: BatchNorm's ReLU mask and backward sums of that stage
    # ride in its data-gradient epilogue)
    return conv_bn(x, seq[3], seq[4], "relu", sync=sync, exclusive=True)


class _InstanceDecoder(nn.Module):

    def __init__(self, num_ch_enc: ArrayLike):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec_project = np.array([64, 32, 16])
        self.num_ch_dec = np.array([256, 128, 128])
        self.net = nn.ModuleDict()
        for i, _ in enumerate(self.num_ch_dec):
            cin, cout = int(self.num_ch_enc[-(i + 2)]), int(self.num_ch_dec_project[i])
            self.net[f"project_{i}"] = nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, bias=False),
                                                     nn.BatchNorm2d(cout), nn.ReLU(inplace=True))
            cin = (int(self.num_ch_dec[0]) if i == 0 else int(self.num_ch_dec[i - 1])) + int(self.num_ch_dec_project[i])
            self.net[f"fuse_{i}"] = _dwsep(cin, int(self.num_ch_dec[i]))
        self.net["conv"] = nn.Sequential(nn.Conv2d(int(self.num_ch_enc[-1]), int(self.num_ch_dec[0]), kernel_size=1,
                                                   stride=1, padding=0, bias=False))

    def forward_nhwc(self, in_feats, sync=None):
        feat = F.conv2d(F.from_nchw(in_feats[-1]), self.net["conv"][0].weight)
        for i, _ in enumerate(self.num_ch_dec):
            proj = self.net[f"project_{i}"]
            skip = conv_bn(F.from_nchw(in_feats[-(i + 2)]), proj[0], proj[1], "relu", sync=sync)
            _, h, w, _ = skip.shape
            feat = F.bilinear_concat([feat, skip], h, w, True)          # resize (align_corners=True) + concat
            feat = _run_dwsep(feat, self.net[f"fuse_{i}"], sync)
        return feat

    def forward(self, in_feats) -> Tensor:
        out = F.to_nchw(self.forward_nhwc(in_feats), int(self.num_ch_dec[-1]))
        flush_bn_counters()
        return out

    def __repr__(self):
        return self.net.__repr__()


class _InstanceHead(nn.Module):

    def __init__(self, num_ch_dec: ArrayLike, num_classes: int):
        super().__init__()
        self.num_ch_dec = num_ch_dec
        self.num_ch_head = np.array([32, num_classes])
        self.net = nn.ModuleDict()
        self.net["depth_conv"] = _dwsep(int(self.num_ch_dec[-1]), int(self.num_ch_head[0]))
        self.net["conv"] = nn.Conv2d(int(self.num_ch_head[0]), int(self.num_ch_head[1]), kernel_size=1)

    def forward_nhwc(self, feat, sync=None):
        _, h, w, _ = feat.shape
        feat = F.bilinear_resize(feat, 2 * h, 2 * w, True)
        feat = _run_dwsep(feat, self.net["depth_conv"], sync)
        feat = F.bilinear_resize(feat, 4 * h, 4 * w, True)
        return F.conv2d(feat, self.net["conv"].weight, self.net["conv"].bias)

    def forward(self, in_feats: Tensor) -> Tensor:
        out = F.to_nchw(self.forward_nhwc(F.from_nchw(in_feats)), int(self.num_ch_head[1]))
        flush_bn_counters()
        return out

    def __repr__(self):
        return self.net.__repr__()


class InstanceHead(nn.Module):

    def __init__(self, num_ch_enc: ArrayLike, use_thing_mask: bool = False):
        super().__init__()
        self.use_thing_mask = use_thing_mask
        self.decoder = _InstanceDecoder(num_ch_enc)
        self.center_head = _InstanceHead(self.decoder.num_ch_dec, num_classes=1)
        self.offset_head = _InstanceHead(self.decoder.num_ch_dec, num_classes=2)
        if use_thing_mask:
            self.thing_mask_head = nn.Sequential(_InstanceHead(self.decoder.num_ch_dec, num_classes=2), nn.Softmax(dim=1))
        else:
            self.thing_mask_head = None
        self.sync_bn = None

    def forward_nhwc(self, in_feats):
        dec = self.decoder.forward_nhwc(in_feats, self.sync_bn)
        center = self.center_head.forward_nhwc(dec, self.sync_bn)
        offset = self.offset_head.forward_nhwc(dec, self.sync_bn)
        thing = None
        if self.use_thing_mask:
            thing = F.channel_softmax(self.thing_mask_head[0].forward_nhwc(dec, self.sync_bn), 2)
        return center, offset, thing

    def forward(self, in_feats) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
        center, offset, thing = self.forward_nhwc(in_feats)
        flush_bn_counters()
        return F.to_nchw(center, 1), F.to_nchw(offset, 2), (None if thing is None else F.to_nchw(thing, 2))
