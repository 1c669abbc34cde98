"""Pose decoder (reference: networks/pose_decoder.py:12-52)."""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import ops


class PoseDecoder(nn.Module):
    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        self.num_ch_enc, self.num_input_features = num_ch_enc, num_input_features
        if num_frames_to_predict_for is None:
            num_frames_to_predict_for = num_input_features - 1
        self.num_frames_to_predict_for = num_frames_to_predict_for
        self.convs = OrderedDict()
        self.convs["squeeze"] = ops.Conv2d(int(num_ch_enc[-1]), 256, 1)
        self.convs[("pose", 0)] = ops.Conv2d(num_input_features * 256, 256, 3, stride, 1)
        self.convs[("pose", 1)] = ops.Conv2d(256, 256, 3, stride, 1)
        self.convs[("pose", 2)] = ops.Conv2d(256, 6 * num_frames_to_predict_for, 1)
        self.relu = nn.ReLU()
        self.net = nn.ModuleList(list(self.convs.values()))

    def forward(self, input_features):
        from .. import ops
        last = [f[-1] for f in input_features]

        def conv(c, x, relu):                      # bias (+ ReLU) in the conv kernel's epilogue on the bf16 step
            y = ops.conv_module(c, x, "relu" if relu else "none") if x.is_cuda else None
            if y is None:
                y = c(x)
                y = self.relu(y) if relu else y
            return y
        out = torch.cat([conv(self.convs["squeeze"], f, True) for f in last], 1)
        for i in range(3):
            out = conv(self.convs[("pose", i)], out, i != 2)
        out = out.mean(3).mean(2)
        out = 0.01 * out.view(-1, self.num_frames_to_predict_for, 1, 6)
        return out[..., :3], out[..., 3:]
