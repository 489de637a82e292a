"""SemanticHead: drop-in for models/semantic_head.py:8-76.  Returns softmax PROBABILITIES like the
reference (its `sem_logits` name notwithstanding)."""
import numpy as np
from numpy.typing import ArrayLike
from torch import Tensor, nn

from .. import functional as F
from .depth_head import _make_decoder, _run_decoder


class SemanticHead(nn.Module):

    def __init__(self, num_ch_enc: ArrayLike, num_classes: int, use_skips: bool, use_guda_fusion: bool = False):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.use_skips = use_skips
        self.use_guda_fusion = use_guda_fusion
        self.num_classes = num_classes
        _make_decoder(self, self.num_ch_enc, self.num_ch_dec, use_skips)
        if self.use_guda_fusion:
            cat_ch = int(self.num_ch_dec.sum() - self.num_ch_dec[-1])
            self.semconv_guda = nn.Conv2d(cat_ch, num_classes, 3, padding=1)
        else:
            self.semconv = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(int(self.num_ch_dec[0]), num_classes, 3))
        self.softmax = nn.Softmax(dim=1)
        # parity instrumentation (off by default): a list here makes every forward append its pre-softmax logits [B,C,H,W]
        # (models/semantic_head.py:72-74, `sem_feat`)
        self.logit_log = None

    def forward_nhwc(self, in_feats):
        """-> NHWC probabilities [B,H,W,pad16(num_classes)] (padding channels are 0)."""
        x, kept = _run_decoder(self, in_feats, keep=(3, 2, 1, 0) if self.use_guda_fusion else ())
        if self.use_guda_fusion:
            _, h, w, _ = x.shape
            if kept[0].shape[1:3] == (h, w) and self.num_classes <= 16:
                logits = F.guda_fusion([kept[3], kept[2], kept[1]], kept[0], self.semconv_guda.weight, self.semconv_guda.bias)
            else:
                cat = F.bilinear_concat([kept[3], kept[2], kept[1], kept[0]], h, w, False)   # 240 channels, one buffer
                logits = F.conv2d(cat, self.semconv_guda.weight, self.semconv_guda.bias, pad=1)
        else:
            conv = self.semconv[1]
            logits = F.conv2d(x, conv.weight, conv.bias, pad=1, pad_mode="reflect")
        if self.logit_log is not None:
            self.logit_log.append(F.to_nchw(logits.detach(), self.num_classes).clone())
        return F.channel_softmax(logits, self.num_classes)

    def forward(self, in_feats) -> Tensor:
        return F.to_nchw(self.forward_nhwc(in_feats), self.num_classes)
