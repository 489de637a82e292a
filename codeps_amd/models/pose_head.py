"""PoseHead: drop-in for models/pose_head.py:14-137."""
from collections import OrderedDict
from typing import List

import torch
from torch import Tensor, nn

from .. import functional as F


class PoseHead(nn.Module):

    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_input_features = num_input_features
        if num_frames_to_predict_for is None:
            num_frames_to_predict_for = num_input_features - 1
        self.num_frames_to_predict_for = num_frames_to_predict_for
        self.stride = stride
        self.convs = OrderedDict()
        self.convs[("squeeze")] = nn.Conv2d(int(self.num_ch_enc[-1]), 256, 1)
        self.convs[("pose", 0)] = nn.Conv2d(num_input_features * 256, 256, 3, stride, 1)
        self.convs[("pose", 1)] = nn.Conv2d(256, 256, 3, stride, 1)
        self.convs[("pose", 2)] = nn.Conv2d(256, 6 * num_frames_to_predict_for, 1)
        self.relu = nn.ReLU()
        self.net = nn.ModuleList(list(self.convs.values()))
        # True = this head is the ONLY reader of input_features[-1] (set with ResnetEncoder.only_last_feature by codeps.model_setup):
        # the backward of the encoder's last BatchNorm + ReLU then runs in the squeeze convolution's data-gradient epilogue
        self.owns_input = False

    def _features(self, input_features):
        x = F.from_nchw(input_features[-1])
        sq, p0, p1, p2 = self.net
        x = F.conv2d(x, sq.weight, sq.bias, act="relu", exclusive=self.owns_input)
        x = F.conv2d(x, p0.weight, p0.bias, stride=self.stride, pad=1, act="relu", exclusive=True)
        x = F.conv2d(x, p1.weight, p1.bias, stride=self.stride, pad=1, act="relu", exclusive=True)
        return F.conv2d(x, p2.weight, p2.bias, exclusive=True)

    def forward(self, input_features: List[Tensor], invert_pose: bool = False) -> Tensor:
        # spatial mean, *0.01, frame 0 of num_frames_to_predict_for, Rodrigues, (inverse) composition: one kernel
        return F.pose_matrix(self._features(input_features), invert_pose)

    def forward_spans(self, input_features: List[Tensor], spans) -> List[Tensor]:
        """One batched pass of the (batch-independent) convs, then one pose-matrix kernel per (b0, b1, invert) span:
        equivalent to calling forward() on every slice."""
        x = self._features(input_features)
        b = x.shape[0]
        flags, pos = [False] * b, 0
        for b0, b1, inv in spans:
            if b0 != pos:
                return [F.pose_matrix(x[s0:s1], i) for s0, s1, i in spans]      # (spans that do not tile the batch: one launch each)
            flags[b0:b1] = [bool(inv)] * (b1 - b0)
            pos = b1
        if pos != b or b > 64:
            return [F.pose_matrix(x[s0:s1], i) for s0, s1, i in spans]
        T = F.pose_matrix(x, flags)                  # all calls in ONE launch (per-item invert flags)
        return list(F.split_batch(T, [(b0, b1) for b0, b1, _ in spans]))

    @staticmethod
    def transformation_from_parameters(axisangle: Tensor, translation: Tensor, invert: bool = False) -> Tensor:
        """[B,1,3] axis-angle + [B,1,3] translation -> [B,4,4] (models/pose_head.py:56-78).  Utility kept
        for callers of the reference API; runs the same HIP kernel on a 1x1 'feature map'."""
        vec = torch.cat([axisangle.reshape(-1, 3), translation.reshape(-1, 3)], 1) * 100.0     # kernel applies 0.01
        feat = torch.zeros(vec.shape[0], 1, 1, 16, dtype=torch.float32, device=vec.device)
        feat[:, 0, 0, :6] = vec
        return F.pose_matrix(feat, invert)
