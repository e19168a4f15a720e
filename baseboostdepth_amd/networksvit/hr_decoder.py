"""MonoViT's depth decoder (HR-Depth style nested skip decoder with squeeze-excite fusion), with the
reference's interface and checkpoint layout (`networksvit/hr_decoder.py:10-126`, the modules it takes
from `networksvit/hr_layers.py:137-176, 361-382, 452-509`).

`DepthDecoder(ch_enc, scales, num_ch_enc, num_output_channels).forward(features) -> {("disp", s)}`; the
state dict has every module twice, as `convs.<name>.*` and as `decoder.<index>.*` (the reference registers
its ModuleDict's values again as a ModuleList, hr_decoder.py:70), in the reference's insertion order.

The body is table-driven: the decoder is a triangular grid of nodes X_rc (row r = resolution level,
column c = refinement step).  Node X_rc fuses the up-sampled node X_(r+1)(c-1) with all earlier nodes of
its own row; the nodes on the grid's diagonal use squeeze-excite fusion (`fSEModule`), the others a plain
`ConvBlock` (after a 1x1 bottleneck when more than one earlier node is concatenated).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..layers import ConvBlock, Conv3x3, upsample

_GRID = ["01", "11", "21", "31", "02", "12", "22", "03", "13", "04"]     # evaluation order
_DIAGONAL = ["31", "22", "13", "04"]                                       # squeeze-excite fusion nodes


class Conv1x1(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 1, stride=1, bias=False)

    def forward(self, x):
        return self.conv(x)


def _excite(fc, x):
    """Channel gates sigmoid(fc(mean_HW x)) applied to x (squeeze-and-excitation)."""
    b, c = x.shape[:2]
    return x * torch.sigmoid(fc(x.mean(dim=(2, 3)))).view(b, c, 1, 1)


def _gate_mlp(channels, reduction=16):
    return nn.Sequential(nn.Linear(channels, channels // reduction, bias=False), nn.ReLU(inplace=True),
                         nn.Linear(channels // reduction, channels, bias=False))


class ChannelAttention(nn.Module):
    def __init__(self, in_planes, ratio=16):
        super().__init__()
        self.fc = _gate_mlp(in_planes, ratio)

    def forward(self, x):
        return _excite(self.fc, x)


class Attention_Module(nn.Module):
    """Encoder-feature adapter: channel attention, then a zero-padded 3x3 conv + ReLU to the decoder width."""

    def __init__(self, high_feature_channel, output_channel=None):
        super().__init__()
        self.ca = ChannelAttention(high_feature_channel)
        self.conv_se = nn.Conv2d(high_feature_channel, output_channel or high_feature_channel, 3, 1, 1)

    def forward(self, high_features):
        return F.relu(self.conv_se(self.ca(high_features)))


class fSEModule(nn.Module):
    """Fusion node: up-sampled coarse features ++ same-level features -> channel gates -> 1x1 conv + ReLU."""

    def __init__(self, high_feature_channel, low_feature_channels, output_channel=None):
        super().__init__()
        channels = high_feature_channel + low_feature_channels
        self.fc = _gate_mlp(channels, 16)
        self.conv_se = nn.Conv2d(channels, output_channel or high_feature_channel, 1, 1)

    def forward(self, high_features, low_features):
        x = torch.cat([upsample(high_features)] + list(low_features), 1)
        return F.relu(self.conv_se(_excite(self.fc, x)))


class DepthDecoder(nn.Module):
    def __init__(self, ch_enc=[64, 128, 216, 288, 288], scales=range(4), num_ch_enc=[64, 64, 128, 256, 512],
                 num_output_channels=1):
        super().__init__()
        self.num_output_channels, self.num_ch_enc, self.ch_enc, self.scales = num_output_channels, num_ch_enc, ch_enc, scales
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        enc, dec = list(num_ch_enc), [int(c) for c in self.num_ch_dec]
        self.all_position, self.attention_position = list(_GRID), list(_DIAGONAL)
        self.non_attention_position = [n for n in _GRID if n not in _DIAGONAL]
        convs = nn.ModuleDict()
        for level in (4, 3, 2, 1):                                   # adapters from the ViT widths
            convs["f%d" % level] = Attention_Module(ch_enc[level], enc[level])
        for col in range(5):                                         # the halving conv in front of every up-sampling
            for row in range(5 - col):
                cin = enc[row] // 2 if (row == 0 and col != 0) else enc[row]
                convs["X_%d%d_Conv_0" % (row, col)] = ConvBlock(cin, cin // 2)
                if (row, col) == (0, 4):
                    convs["X_04_Conv_1"] = ConvBlock(cin // 2, dec[0])
        for node in self.attention_position:
            row, col = int(node[0]), int(node[1])
            convs["X_%s_attention" % node] = fSEModule(enc[row + 1] // 2, enc[row] + dec[row + 1] * (col - 1))
        for node in self.non_attention_position:
            row, col = int(node[0]), int(node[1])
            fused = enc[row + 1] // 2 + enc[row] + dec[row + 1] * (col - 1)
            if col == 1:
                convs["X_%d%d_Conv_1" % (row + 1, col - 1)] = ConvBlock(fused, dec[row + 1])
            else:
                convs["X_%s_downsample" % node] = Conv1x1(fused, dec[row + 1] * 2)
                convs["X_%d%d_Conv_1" % (row + 1, col - 1)] = ConvBlock(dec[row + 1] * 2, dec[row + 1])
        for i in range(4):
            convs["dispconv%d" % i] = Conv3x3(dec[i], num_output_channels)
        self.convs = convs
        self.decoder = nn.ModuleList(list(convs.values()))           # the reference's second registration

    def forward(self, input_features):
        c = self.convs
        X = {"00": input_features[0]}
        for level in (4, 3, 2, 1):
            X["%d0" % level] = c["f%d" % level](input_features[level])
        for node in _GRID:
            row, col = int(node[0]), int(node[1])
            below = "%d%d" % (row + 1, col - 1)
            same_row = [X["%d%d" % (row, k)] for k in range(col)]
            halved = c["X_%s_Conv_0" % below](X[below])
            if node in _DIAGONAL:
                X[node] = c["X_%s_attention" % node](halved, same_row)
            elif col == 1 and ops.upcat_pad_supported(halved, same_row[0]):
                # up-sampling + concatenation + the block's reflection border in one HIP pass (csrc/bbd_nn.hip)
                X[node] = c["X_%s_Conv_1" % below].forward_padded(ops.upcat_pad(halved, same_row[0]))
            else:
                x = torch.cat([upsample(halved)] + same_row, 1)
                if col != 1:
                    x = c["X_%s_downsample" % node](x)
                X[node] = c["X_%s_Conv_1" % below](x)
        last = c["X_04_Conv_0"](X["04"])
        if ops.upcat_pad_supported(last, None):
            top = c["X_04_Conv_1"].forward_padded(ops.upcat_pad(last, None))
        else:
            top = c["X_04_Conv_1"](upsample(last))
        heads = {0: top, 1: X["04"], 2: X["13"], 3: X["22"]}
        return {("disp", s): torch.sigmoid(c["dispconv%d" % s](heads[s])) for s in range(4)}
