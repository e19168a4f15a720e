"""U-Net disparity decoder with the reference's interface and state-dict layout
(networks/depth_decoder.py:11-59): `decoder.{0..13}.conv.conv.{weight,bias}` - ten ConvBlocks
(upconv i,0 / i,1 for i = 4..0) followed by the four dispconvs."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..layers import ConvBlock, Conv3x3


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.scales = scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        blocks, self._index = [], {}
        for i in range(4, -1, -1):
            cin = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self._index[("upconv", i, 0)] = len(blocks)
            blocks.append(ConvBlock(cin, self.num_ch_dec[i]))
            cin = self.num_ch_dec[i] + (self.num_ch_enc[i - 1] if (use_skips and i > 0) else 0)
            self._index[("upconv", i, 1)] = len(blocks)
            blocks.append(ConvBlock(cin, self.num_ch_dec[i]))
        for s in self.scales:
            self._index[("dispconv", s)] = len(blocks)
            blocks.append(Conv3x3(self.num_ch_dec[s], num_output_channels))
        self.decoder = nn.ModuleList(blocks)

    def _conv(self, *key):
        return self.decoder[self._index[key]]

    def forward(self, input_features):
        self.outputs = {}
        x = input_features[-1]
        for i in range(4, -1, -1):
            x = self._conv("upconv", i, 0)(x)
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            if self.use_skips and i > 0:
                x = torch.cat([x, input_features[i - 1]], 1)
            x = self._conv("upconv", i, 1)(x)
            if i in self.scales:
                self.outputs[("disp", i)] = torch.sigmoid(self._conv("dispconv", i)(x))
        return self.outputs
