"""ResNet encoder with the reference's interface (networks/resnet_encoder.py:56-91).

torchvision is not available on the target image, so the ResNet-18/34/50 trunk is defined
here with torchvision-identical module names - `conv1, bn1, layer{1..4}.{i}.conv{j}/bn{j},
layer{k}.0.downsample.{0,1}, fc` - so `encoder.pth` checkpoints written by the reference
(trainer.py:783-805) load with `load_state_dict` unchanged, including the unused `fc`.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class FusedBatchNorm2d(nn.BatchNorm2d):
    """`nn.BatchNorm2d` (same parameters, buffers and state-dict keys) whose training-mode forward on
    the GPU also absorbs what follows it in a ResNet block: `relu(bn(x) + residual)` runs as two HIP
    launches each way (csrc/bbd_nn.hip) instead of MIOpen batch-norm + add + clamp kernels.  Host
    tensors, eval mode and exotic configurations take the stock PyTorch ops."""

    fused = os.environ.get("BBD_FUSED_BN", "1") != "0"

    def forward(self, x, residual=None, relu=False):
        if (self.fused and x.is_cuda and self.training and self.affine and self.track_running_stats
                and self.momentum is not None and x.dtype == torch.float32 and x.dim() == 4):
            from .. import ops
            return ops.batch_norm_act(x, self.weight, self.bias, residual, self.running_mean, self.running_var,
                                      self.momentum, self.eps, relu, num_batches_tracked=self.num_batches_tracked)
        if self.training and x.is_cuda:
            from .. import ops
            assert ops._bn_groups is None and ops._bn_device is None, \
                "a batched pass with call groups needs the fused BatchNorm path"
        y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y


class MaxPool3s2(nn.MaxPool2d):
    """The stem's `nn.MaxPool2d(3, 2, 1)`; fp32 GPU tensors take the HIP kernels (atomic-free backward)."""

    def __init__(self):
        super().__init__(kernel_size=3, stride=2, padding=1)

    def forward(self, x):
        from .. import ops
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and ops.FUSED_NN
                and x.shape[0] * x.shape[1] <= 65535):
            return ops.maxpool3s2(x)
        return super().forward(x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = FusedBatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = FusedBatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.bn1(self.conv1(x), relu=True)
        return self.bn2(self.conv2(out), residual=identity, relu=True)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = FusedBatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = FusedBatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = FusedBatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.bn1(self.conv1(x), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        return self.bn3(self.conv3(out), residual=identity, relu=True)


_CONFIGS = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
            101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


class ResNetTrunk(nn.Module):
    """ImageNet-style ResNet; `num_input_images` widens conv1 like ResNetMultiImageInput
    (networks/resnet_encoder.py:12-33)."""

    def __init__(self, num_layers, num_input_images=1, num_classes=1000):
        super().__init__()
        block, layers = _CONFIGS[num_layers]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(num_input_images * 3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = FusedBatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = MaxPool3s2()
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)   # never used; kept for checkpoint parity
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                FusedBatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


class ResnetEncoder(nn.Module):
    """`ResnetEncoder(num_layers, pretrained, num_input_images=1)`; forward -> 5 feature maps."""

    def __init__(self, num_layers, pretrained, num_input_images=1):
        super().__init__()
        if num_layers not in _CONFIGS:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        if pretrained:
            raise RuntimeError("ImageNet weights cannot be downloaded here (no network); load a checkpoint "
                               "with load_state_dict instead (--weights_init scratch)")
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNetTrunk(num_layers, num_input_images)
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4

    def forward(self, input_image, normalized=False):
        """`normalized`: the caller hands in `(image - 0.45) / 0.225` already (the pooled step's pair gather does it in the
        same pass, `ops.gather_pairs`)."""
        e = self.encoder
        x = input_image if normalized else (input_image - 0.45) / 0.225
        f0 = e.bn1(e.conv1(x), relu=True)
        f1 = e.layer1(e.maxpool(f0))
        f2 = e.layer2(f1)
        f3 = e.layer3(f2)
        f4 = e.layer4(f3)
        self.features = [f0, f1, f2, f3, f4]
        return self.features
