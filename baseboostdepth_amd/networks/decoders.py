"""Decoder heads with the reference's interfaces and checkpoint layouts.

* `DepthDecoder` (reference networks/depth_decoder.py:11-59): U-Net disparity decoder; state dict
  `decoder.{0..13}.conv.conv.{weight,bias}` = ten ConvBlocks (upconv i,0 / i,1 for i = 4..0)
  followed by the four dispconvs.
* `PoseDecoder` (reference networks/pose_decoder.py:9-48): `net.{0..3}.{weight,bias}` = squeeze
  1x1, two 3x3, final 1x1; output scaled by 0.01 and split into axis-angle / translation.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..layers import ConvBlock, Conv3x3

_DECODER_WIDTHS = (16, 32, 64, 128, 256)


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels, self.use_skips, self.scales = num_output_channels, use_skips, scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array(_DECODER_WIDTHS)
        stages, self._slot = [], {}

        def add(key, module):
            self._slot[key] = len(stages)
            stages.append(module)

        for level in reversed(range(5)):
            width = int(self.num_ch_dec[level])
            fan_in = int(self.num_ch_enc[-1]) if level == 4 else int(self.num_ch_dec[level + 1])
            add(("upconv", level, 0), ConvBlock(fan_in, width))
            skip = int(self.num_ch_enc[level - 1]) if (use_skips and level > 0) else 0
            add(("upconv", level, 1), ConvBlock(width + skip, width))
        for s in self.scales:
            add(("dispconv", s), Conv3x3(int(self.num_ch_dec[s]), num_output_channels))
        self.decoder = nn.ModuleList(stages)

    def _stage(self, *key):
        return self.decoder[self._slot[key]]

    def forward(self, input_features):
        self.outputs = {}
        x = input_features[-1]
        for level in reversed(range(5)):
            x = self._stage("upconv", level, 0)(x)
            skip = input_features[level - 1] if (self.use_skips and level > 0) else None
            if ops.upcat_pad_supported(x, skip):
                # nearest x2 + skip concatenation + the next block's reflection border in ONE HIP pass
                x = self._stage("upconv", level, 1).forward_padded(ops.upcat_pad(x, skip))
            else:
                x = F.interpolate(x, scale_factor=2, mode="nearest")
                if skip is not None:
                    x = torch.cat([x, skip], 1)
                x = self._stage("upconv", level, 1)(x)
            if level in self.scales:
                self.outputs[("disp", level)] = torch.sigmoid(self._stage("dispconv", level)(x))
        return self.outputs


class PoseDecoder(nn.Module):
    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        frames = num_input_features - 1 if num_frames_to_predict_for is None else num_frames_to_predict_for
        self.num_ch_enc, self.num_input_features, self.num_frames_to_predict_for = num_ch_enc, num_input_features, frames
        self.net = nn.ModuleList([
            nn.Conv2d(int(num_ch_enc[-1]), 256, 1),                      # squeeze
            nn.Conv2d(num_input_features * 256, 256, 3, stride, 1),
            nn.Conv2d(256, 256, 3, stride, 1),
            nn.Conv2d(256, 6 * frames, 1)])

    def forward(self, input_features):
        x = torch.cat([F.relu(self.net[0](feats[-1])) for feats in input_features], 1)
        x = F.relu(self.net[1](x))
        x = F.relu(self.net[2](x))
        pose = 0.01 * self.net[3](x).mean(3).mean(2).view(-1, self.num_frames_to_predict_for, 1, 6)
        return pose[..., :3], pose[..., 3:]
