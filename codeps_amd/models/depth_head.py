"""DepthHead: drop-in for models/depth_head.py:10-78 (Monodepth2-style decoder -> sigmoid disparity ->
depth).  Each decoder stage is two fused kernels: conv3x3+bias+ELU, and
[nearest x2 upsample + concat skip + conv3x3 + bias + ELU] with the upsample / concat done on load."""
from typing import Tuple, Union

import os

import numpy as np
import torch
from numpy.typing import ArrayLike
from torch import Tensor, nn

from .. import functional as F


def _make_decoder(module, num_ch_enc, num_ch_dec, use_skips):
    module.upconvs_0, module.upconvs_1 = nn.ModuleDict(), nn.ModuleDict()
    for i in range(4, -1, -1):
        cin = num_ch_enc[-1] if i == 4 else num_ch_dec[i + 1]
        module.upconvs_0[str(i)] = nn.Sequential(nn.Conv2d(int(cin), int(num_ch_dec[i]), 3, stride=1, padding=1),
                                                 nn.ELU(inplace=True))
        cin = num_ch_dec[i]
        if use_skips and i > 0:
            cin = cin + num_ch_enc[i - 1]
        module.upconvs_1[str(i)] = nn.Sequential(nn.Conv2d(int(cin), int(num_ch_dec[i]), 3, stride=1, padding=1),
                                                 nn.ELU(inplace=True))


_UPCONV16 = os.environ.get("CODEPS_UPCONV16", "1") != "0"      # A/B switch: 0 = the generic engine's fused-gather form
_FORK = os.environ.get("CODEPS_FORK", "1") != "0"              # A/B switch: 0 = autograd sums the gradients of a tensor with two consumers


def _run_decoder(module, in_feats, keep=()):
    """-> NHWC decoder output at scale 0 (and the scales listed in `keep`)."""
    x = F.from_nchw(in_feats[-1])
    kept = {}
    for i in range(4, -1, -1):
        c0, c1 = module.upconvs_0[str(i)][0], module.upconvs_1[str(i)][0]
        # exclusive: the previous stage's output (i < 4; the encoder map of stage 4 has other readers) is read by this convolution alone,
        # or by it and the taker of its fork alias -- its ELU backward and bias gradient then come out of this convolution's
        # data-gradient epilogue (F.conv2d(exclusive=True)); likewise for the stage's own two convolutions
        if (i + 1) in kept and _FORK and torch.is_grad_enabled() and x.requires_grad:
            # the kept stage output has a second consumer (the GUDA fusion): it takes this convolution's alias of the tensor, so
            # that the two gradients meet in the data-gradient kernel's epilogue instead of a separate full-size sum
            x, kept[i + 1] = F.conv2d(x, c0.weight, c0.bias, pad=1, act="elu", fork=True, exclusive=True)
        else:
            x = F.conv2d(x, c0.weight, c0.bias, pad=1, act="elu", exclusive=i < 4 and (i + 1) not in kept)
        skip = F.from_nchw(in_feats[i - 1]) if (module.use_skips and i > 0) else None
        if skip is None and c1.in_channels == 16 and c1.out_channels == 16 and x.shape[-1] == 16 and _UPCONV16:
            x = F.upconv16(x, c1.weight, c1.bias, act="elu", exclusive=True)      # scale 0: four 2x2 convolutions of the half-resolution tensor
        else:
            x = F.conv2d(x, c1.weight, c1.bias, x1=skip, pad=1, act="elu", up0=True, exclusive=True)
        if i in keep:
            kept[i] = x
    return x, kept


class DepthHead(nn.Module):

    def __init__(self, num_ch_enc: ArrayLike, use_skips: bool):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.use_skips = use_skips
        self.scales = range(4)
        self.outputs = {}
        _make_decoder(self, self.num_ch_enc, self.num_ch_dec, use_skips)
        self.dispconvs = nn.ModuleDict()
        for s in self.scales:       # scales 1..3 exist for state_dict compatibility; nothing reads them
            self.dispconvs[str(s)] = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(int(self.num_ch_dec[s]), 1, 3))
        self.sigmoid = nn.Sigmoid()
        # parity instrumentation (off by default, not part of the reference's surface): a list here makes every forward append its
        # pre-sigmoid disparity logits [B,1,H,W] (models/depth_head.py:73), written by the SAME kernel launch that produces depth
        self.logit_log = None

    @staticmethod
    def disp_to_depth(disp: Tensor, min_depth: float = 0.1, max_depth: float = 100) -> Tensor:
        min_disp = 1 / max_depth
        max_disp = 1 / min_depth
        return 1 / (min_disp + (max_disp - min_disp) * disp)

    def forward(self, in_feats, return_disparity: bool = False) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        x, _ = _run_decoder(self, in_feats)
        conv = self.dispconvs["0"][1]
        if conv.in_channels == 16 and x.shape[-1] == 16:
            tap = None if self.logit_log is None else x.new_empty(x.shape[0], 1, x.shape[1], x.shape[2])
            # (the decoder's output is read by this stage alone: its ELU backward rides in the stage's data-gradient kernel)
            depth_map, disp_map = F.disp_head(x, conv.weight, conv.bias, logit_out=tap, exclusive=True)    # padded conv + sigmoid + disp_to_depth, one pass each way
            if tap is not None:
                self.logit_log.append(tap)
        else:
            logits = F.conv2d(x, conv.weight, conv.bias, pad=1, pad_mode="reflect")
            depth_map, disp_map = F.disp_depth(logits)      # sigmoid + disp_to_depth, one kernel
            if self.logit_log is not None:
                self.logit_log.append(F.to_nchw(logits.detach(), 1).clone())
        # detached: a module attribute that holds the autograd graph of the last call keeps its AccumulateGrad nodes (and
        # their streams) alive across iterations, which breaks HIP-graph capture of the next step
        self.outputs = {0: disp_map.detach()}
        if return_disparity:
            return depth_map, disp_map
        return depth_map
