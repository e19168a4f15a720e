"""Pose head with the reference's interface and state-dict layout
(networks/pose_decoder.py:9-48): `net.{0..3}.{weight,bias}` = squeeze 1x1, two 3x3, final 1x1."""
import torch
import torch.nn as nn


class PoseDecoder(nn.Module):
    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_input_features = num_input_features
        if num_frames_to_predict_for is None:
            num_frames_to_predict_for = num_input_features - 1
        self.num_frames_to_predict_for = num_frames_to_predict_for
        self.net = nn.ModuleList([
            nn.Conv2d(int(num_ch_enc[-1]), 256, 1),
            nn.Conv2d(num_input_features * 256, 256, 3, stride, 1),
            nn.Conv2d(256, 256, 3, stride, 1),
            nn.Conv2d(256, 6 * num_frames_to_predict_for, 1)])
        self.relu = nn.ReLU()

    def forward(self, input_features):
        squeezed = [self.relu(self.net[0](f[-1])) for f in input_features]
        out = torch.cat(squeezed, 1)
        out = self.relu(self.net[1](out))
        out = self.relu(self.net[2](out))
        out = self.net[3](out)
        out = out.mean(3).mean(2)
        out = 0.01 * out.view(-1, self.num_frames_to_predict_for, 1, 6)
        return out[..., :3], out[..., 3:]
