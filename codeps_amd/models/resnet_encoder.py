"""ResnetEncoder: drop-in for the reference's models/resnet_encoder.py:82-126.

The reference wraps torchvision's ResNet (requirements.txt:2, not vendored); here the trunk is defined
directly with torchvision's module / parameter names (conv1, bn1, layerL.B.conv{1,2,3}, bn{1,2,3},
downsample.{0,1}, fc) so that reference checkpoints load unchanged, and its forward runs the NHWC
f32-MFMA conv engine: every conv+BN(+ReLU)(+residual) is ONE kernel in eval mode.
"""
import numpy as np
import torch
from torch import nn

from .. import functional as F
from ._blocks import _count_batch, conv_bn, flush_bn_counters, folded_bn

_LAYERS = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]),
           101: ("bottleneck", [3, 4, 23, 3]), 152: ("bottleneck", [3, 8, 36, 3])}


class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward_nhwc(self, x, sync, groups=None, own_input=False):
        # x has two consumers (conv1 and the identity / shortcut): the second one takes conv1's alias of it, so that the gradients
        # meet inside conv1's data-gradient kernel.  own_input: this block is x's only reader (true between the blocks of a stage), so
        # the ReLU / BatchNorm backward of the block that made x runs in conv1's data-gradient epilogue; `out` is conv2's alone.
        out, xa = conv_bn(x, self.conv1, self.bn1, "relu", sync=sync, groups=groups, fork=True, exclusive=own_input)
        identity = xa if self.downsample is None else conv_bn(xa, self.downsample[0], self.downsample[1], sync=sync, groups=groups)
        return conv_bn(out, self.conv2, self.bn2, "relu", residual=identity, sync=sync, groups=groups, exclusive=True)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)          # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward_nhwc(self, x, sync, groups=None, own_input=False):
        out, xa = conv_bn(x, self.conv1, self.bn1, "relu", sync=sync, groups=groups, fork=True, exclusive=own_input)
        identity = xa if self.downsample is None else conv_bn(xa, self.downsample[0], self.downsample[1], sync=sync, groups=groups)
        # (bn2's normalise + ReLU happens on conv3's operand load when bn2 is in train mode: conv3, a 1x1, is `out`'s only reader)
        out = conv_bn(out, self.conv2, self.bn2, "relu", sync=sync, groups=groups, exclusive=True, defer=True)
        return conv_bn(out, self.conv3, self.bn3, "relu", residual=identity, sync=sync, groups=groups, exclusive=True)


def _frozen_eval(*mods):
    """Eval-mode BatchNorm (foldable) and nothing that needs a gradient."""
    if any(m.training for m in mods if isinstance(m, nn.modules.batchnorm._BatchNorm)):
        return False
    return not torch.is_grad_enabled() or not any(p.requires_grad for m in mods for p in m.parameters())


def _run_layer(layer, x, sync, groups, own_input=False):
    """One ResNet stage.  Frozen bottlenecks in eval mode whose mid width is 256 (ResNet-50 / 101 layer3: 22 of the 101's 33
    bottlenecks; 128 and 64 -- layer2, layer1 -- as well on the fp16-pair arithmetic) hand their last layer to the NEXT bottleneck's first one inside one launch (F.chain_expand_reduce: conv3 + bn3
    + residual + ReLU, then conv1' + bn1' + ReLU, the 1024-channel tensor passing from one product to the other in
    registers); every other block runs layer by layer.
    own_input: the stage is the only reader of x (the caller keeps no other use of the previous stage's map); between the blocks of a
    stage that is always so -- see _BasicBlock.forward_nhwc."""
    blocks = list(layer)
    y0 = None                                                  # relu(bn1(conv1(x))) of the current block, when the previous launch made it
    for i, blk in enumerate(blocks):
        own = own_input if i == 0 else True
        if not isinstance(blk, _Bottleneck):
            x = blk.forward_nhwc(x, sync, groups, own)
            continue
        nxt = blocks[i + 1] if i + 1 < len(blocks) else None
        xa = x
        if y0 is None:
            y0, xa = conv_bn(x, blk.conv1, blk.bn1, "relu", sync=sync, groups=groups, fork=True, exclusive=own)
        identity = xa if blk.downsample is None else conv_bn(xa, blk.downsample[0], blk.downsample[1], sync=sync, groups=groups)
        y1 = conv_bn(y0, blk.conv2, blk.bn2, "relu", sync=sync, groups=groups, exclusive=True, defer=True)
        y0 = None
        if (nxt is not None and nxt.downsample is None and nxt.conv1.out_channels == blk.conv3.in_channels
                and F.chain_ok(y1, blk.conv3.out_channels) and _frozen_eval(blk.bn3, blk.conv3, nxt.bn1, nxt.conv1)
                and getattr(y1, "_codeps_defer", None) is None          # (a train-mode bn2 left its apply pass to F.conv2d)
                and not (torch.is_grad_enabled() and (y1.requires_grad or identity.requires_grad))):
            s2, t2 = folded_bn(blk.bn3)
            s3, t3 = folded_bn(nxt.bn1)
            x, y0 = F.chain_expand_reduce(y1, identity, blk.conv3.weight, s2, t2, nxt.conv1.weight, s3, t3)
        else:
            x = conv_bn(y1, blk.conv3, blk.bn3, "relu", residual=identity, sync=sync, groups=groups, exclusive=True)
    return x


class _ResNet(nn.Module):
    """torchvision-named ResNet trunk (+ unused fc kept for state_dict compatibility)."""

    def __init__(self, num_layers, in_channels=3):
        super().__init__()
        kind, counts = _LAYERS[num_layers]
        block = _BasicBlock if kind == "basic" else _Bottleneck
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, counts[0], 1)
        self.layer2 = self._make_layer(block, 128, counts[1], 2)
        self.layer3 = self._make_layer(block, 256, counts[2], 2)
        self.layer4 = self._make_layer(block, 512, counts[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, 1000)
        for m in self.modules():                       # models/resnet_encoder.py:38-43 (= torchvision's init)
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes, 1, None) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


class ResnetEncoder(nn.Module):
    """Signature and outputs of the reference class (models/resnet_encoder.py:82-126):
    forward(input_image [B, num_input_images*num_channels_input, H, W]) -> 5 NCHW feature maps at
    strides 2..32 (zero-copy channels-last views of the internal NHWC tensors)."""

    def __init__(self, num_layers, pretrained, num_input_images=1, num_channels_input=3):
        super().__init__()
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        if num_layers not in _LAYERS:
            raise ValueError(f"{num_layers} is not a valid number of resnet layers")
        if num_input_images > 1:
            assert num_channels_input >= 3, "Require at least RGB image (3 channels)"
            assert num_layers in [18, 50], "Can only run with 18 or 50 layer resnet"
        if pretrained:
            raise RuntimeError("pretrained=True needs the ImageNet download the reference does through torchvision; "
                               "this build has no network: construct with pretrained=False and load a checkpoint "
                               "(state_dict keys are the reference's).")
        self.encoder = _ResNet(num_layers, num_input_images * num_channels_input)
        self.num_layers = num_layers
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4
        self.sync_bn = None            # set by codeps_amd.distributed for SyncBatchNorm semantics
        # True = the caller reads the LAST feature map only (the pose net: models/pose_head.py:30): forward() then returns None in
        # place of the stride-2 map and, in train mode, runs bn1 -> relu -> maxpool as one pass each way (F.bn_relu_maxpool_train)
        self.only_last_feature = False

    accepts_frame_tuple = True

    def forward(self, input_image):
        """input_image: NCHW tensor, or a tuple of NCHW frames to be stacked along channels (the pose net's
        frame pair) -- the tuple form skips the reference's torch.cat (algos/depth.py:408-413)."""
        enc = self.encoder
        groups = None
        first = input_image[0][0] if isinstance(input_image, _PairGroups) else (
            input_image[0] if isinstance(input_image, (tuple, list)) else input_image)
        # conv1 on <= 8 input channels and an even width runs in the double-pixel stem layout (F.stem_conv2d: K = 448
        # instead of 784 for the 16-channel padded form); anything else takes the generic 16-channel path
        stem = enc.conv1.in_channels <= 8 and first.shape[-1] % 2 == 0
        cp, xpad = (8, F.STEM_XPAD) if stem else (16, (0, 0))
        # frozen 3-channel conv1 (the shared encoder of the adaptation step, inference passes): rows of 4-channel pixels and a
        # 7 x 1 convolution over overlapping 8-pixel windows, K = 224 instead of the double-pixel form's 448 (F.stem3_conv2d)
        if stem and F._STEM3 and enc.conv1.in_channels == 3 and first.shape[-2] % 2 == 0 and _frozen_eval(enc.conv1, enc.bn1):
            stem, cp, xpad = "s3", 4, F.STEM3_XPAD
        if isinstance(input_image, _PairGroups):
            x = F.frame_pairs_to_nhwc(list(input_image), cp, xpad)
            groups, off = _Groups(), 0
            for fa, _fb in input_image:
                groups.append((off, off + fa.shape[0]))
                off += fa.shape[0]
            groups.plan = getattr(input_image, "plan", None)
        elif isinstance(input_image, _BatchOf):
            x = F.image_batches_to_nhwc(list(input_image), cp, xpad)
        elif isinstance(input_image, (tuple, list)):
            x = F.images_to_nhwc(list(input_image), cp, xpad)
        elif stem:
            x = F.images_to_nhwc([input_image], cp, xpad)
        else:
            x = F.from_nchw(input_image, 16)
        if self.only_last_feature and F._BNPOOL and enc.bn1.training and stem != "s3":
            # train-mode stem whose ReLU output nobody reads: statistics, then relu(bn(z)) max-pooled straight from z (and the backward
            # from the pooled gradient) -- 2.5 GB of the stem's 4.4 GB of BatchNorm / pooling traffic per step are not moved
            z = F.stem_conv2d(x, enc.conv1.weight, bn_stats=True) if stem else \
                F.conv2d(x, enc.conv1.weight, stride=enc.conv1.stride[0], pad=enc.conv1.padding[0], bn_stats=True)
            x = F.bn_relu_maxpool_train(z, enc.bn1.weight, enc.bn1.bias, enc.bn1.running_mean, enc.bn1.running_var, enc.bn1.momentum,
                                        enc.bn1.eps, groups, reduce_fn=self.sync_bn)        # (SyncBatchNorm too, since round 5)
            _count_batch(enc.bn1, len(groups) if groups else 1)
            feats = [None]
        else:
            x = conv_bn(x, enc.conv1, enc.bn1, "relu", sync=self.sync_bn, groups=groups, stem=stem)
            feats = [x]
            x = F.max_pool3x3s2(x)
        for layer in (enc.layer1, enc.layer2, enc.layer3, enc.layer4):
            # only_last_feature: nobody but the next stage reads a stage's map
            x = _run_layer(layer, x, self.sync_bn, groups, own_input=self.only_last_feature)
            feats.append(x)
        flush_bn_counters()
        return [None if f is None else F.to_nchw(f, int(c)) for f, c in zip(feats, self.num_ch_enc)]

    def forward_pair_groups(self, pairs, plan=None):
        """pairs = [(frame_a, frame_b), ...]: every pair is what the reference feeds to ONE call of this encoder
        (`torch.cat([a, b], 1)`, algos/depth.py:408-413).  All pairs run through one set of batched convolutions while
        every BatchNorm keeps per-pair batch statistics and applies the running-stat momentum updates pair by pair in
        list order -- bit-for-bit the semantics of len(pairs) separate calls, at len(pairs) x larger GEMMs."""
        pg = _PairGroups(pairs)
        pg.plan = plan          # SyncBatchNorm over unequal shards: slots of the global statistic groups (algos.depth._group_plan)
        return self.forward(pg)

    def forward_batch(self, images):
        """Several NCHW image batches of one resolution -> ONE forward over their concatenation along the batch
        axis (built directly in the stem's NHWC input buffer, no torch.cat).  Result-neutral only when BatchNorm
        uses running statistics (eval mode): the caller checks `not self.training`."""
        return self.forward(_BatchOf(images))


class _BatchOf(list):
    """Marker: list of NCHW tensors to be concatenated along the batch axis."""


class _Groups(list):
    """Batch ranges of the BatchNorm statistic groups; `.plan` = their slots in the global group list when ranks hold unequal shards."""
    plan = None


class _PairGroups(list):
    """Marker: list of (frame_a, frame_b) pairs = independent BatchNorm groups of one batched forward."""
