"""Tiny deterministic stand-in for the pose *encoder* used by golden-vector cases.

The reference's `networks.ResnetEncoder` cannot be built in the fixture container
(torchvision is absent), so pose-path fixtures are generated with this small conv
stack in its place, feeding the reference's real `networks.PoseDecoder`.  The same
class (with weights loaded from the fixture) is used by the tests on our side, so
`predict_poses` is compared like for like.
"""
import numpy as np
import torch
import torch.nn as nn


class FakePoseEncoder(nn.Module):
    """[n, 6, H, W] -> [feat] with feat [n, 16, H/4, W/4]; `gain` makes poses non-trivial."""

    def __init__(self, gain=40.0):
        super().__init__()
        self.num_ch_enc = np.array([16])
        self.conv = nn.Conv2d(6, 16, 3, stride=4, padding=1)
        self.gain = gain

    def forward(self, x):
        return [self.gain * torch.tanh(self.conv(x - 0.5))]


def fill_deterministic(module, phase=0.0):
    """Closed-form, seed-free weights so fixtures need not store multi-MB state dicts.

    Every parameter/buffer i-th element becomes a/sqrt(fan_in) * sin(0.7*i + k + phase)
    (biases: 0.05*sin).  Computed in float64 with numpy, then cast to float32, so the
    generator (build container) and the tests (GPU box, same image) agree bit for bit.
    """
    with torch.no_grad():
        for k, (name, t) in enumerate(sorted(module.state_dict().items())):
            if not t.is_floating_point():
                continue
            n = t.numel()
            idx = np.arange(n, dtype=np.float64)
            if t.dim() >= 2:
                fan_in = int(np.prod(t.shape[1:]))
                vals = (1.7 / np.sqrt(fan_in)) * np.sin(0.7 * idx + k + phase)
            elif name.endswith("running_var") or name.endswith("weight"):
                vals = 1.0 + 0.1 * np.sin(0.7 * idx + k + phase)
            else:
                vals = 0.05 * np.sin(0.7 * idx + k + phase)
            t.copy_(torch.from_numpy(vals.astype(np.float32)).view_as(t))
    return module
