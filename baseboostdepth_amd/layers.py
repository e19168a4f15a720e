"""`layers` API of the reference (layers.py), MI355X-native.

Same names, constructor arguments and return shapes as the reference so that `trainer.py`- and
`evaluate_depth.py`-style callers are drop-in.  The geometry / photometric classes run as HIP
kernels through the C ABI (include/bbd_hip.h), including the pose-matrix composition
(`transformation_from_parameters`, one launch each way on GPU tensors).
"""
import torch
import torch.nn as nn

from . import ops


def disp_to_depth(disp, min_depth, max_depth):
    """Sigmoid output -> (scaled disparity, depth)  (layers.py:13-22)."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    scaled_disp = min_disp + (max_disp - min_disp) * disp
    return scaled_disp, 1 / scaled_disp


def rot_from_axisangle(vec):
    """[n,1,3] axis-angle -> [n,4,4] rotation, Rodrigues entries of layers.py:61-100."""
    angle = torch.norm(vec, 2, 2, True)
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1 - ca
    x, y, z = axis[..., 0:1], axis[..., 1:2], axis[..., 2:3]
    xs, ys, zs = x * sa, y * sa, z * sa
    xC, yC, zC = x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    zero, one = torch.zeros_like(ca), torch.ones_like(ca)
    rows = [x * xC + ca, xyC - zs, zxC + ys, zero,
            xyC + zs, y * yC + ca, yzC - xs, zero,
            zxC - ys, yzC + xs, z * zC + ca, zero,
            zero, zero, zero, one]
    return torch.cat(rows, dim=2).view(-1, 4, 4)


def get_translation_matrix(translation_vector):
    """[n,1,3] -> [n,4,4] homogeneous translation (layers.py:45-58)."""
    t = translation_vector.contiguous().view(-1, 3, 1)
    n = t.shape[0]
    eye = torch.eye(4, device=t.device, dtype=t.dtype).expand(n, 4, 4)
    top = torch.cat([eye[:, :3, :3], t], dim=2)
    return torch.cat([top, eye[:, 3:, :]], dim=1)


def transformation_from_parameters(axisangle, translation, invert=False):
    """(axisangle, translation) [n,1,3] -> [n,4,4]; inverted form is R^T @ T(-t)  (layers.py:25-42).

    GPU tensors go through the fused pose-matrix kernel (one launch forward, one backward, instead
    of ~35 element-wise launches each way); host tensors - fixtures, synthetic-pose helpers - use the
    op-for-op torch form below, which is what the golden vectors pin."""
    if axisangle.is_cuda:
        return ops.pose_matrix(axisangle, translation, invert)
    return _transformation_from_parameters_torch(axisangle, translation, invert)


def _transformation_from_parameters_torch(axisangle, translation, invert=False):
    R = rot_from_axisangle(axisangle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    T = get_translation_matrix(t)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


class ReflectionPad1(nn.ReflectionPad2d):
    """`nn.ReflectionPad2d(1)`; fp32 GPU tensors take the HIP kernels (gather-form backward)."""

    def __init__(self):
        super().__init__(1)

    def forward(self, x):
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and min(x.shape[2:]) >= 2 and ops.FUSED_NN
                and x.shape[0] * x.shape[1] <= 65535):
            return ops.reflect_pad1(x)
        return super().forward(x)


class Conv3x3(nn.Module):
    """Reflection-padded 3x3 convolution (layers.py:118-133)."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.pad = ReflectionPad1() if use_refl else nn.ZeroPad2d(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)

    def forward(self, x):
        if (self.conv.out_channels == 1 and x.is_cuda and x.dtype == torch.float32 and ops.FUSED_NN
                and isinstance(self.pad, ReflectionPad1) and self.conv.in_channels <= 256 and min(x.shape[2:]) >= 2):
            return ops.dispconv(x, self.conv.weight, self.conv.bias)    # disparity head: one streaming kernel
        return self.conv(self.pad(x))


class ConvBlock(nn.Module):
    """Conv3x3 + ELU (layers.py:103-115).  fp32 GPU tensors: bias-free MIOpen convolution, then bias + ELU as
    one in-place HIP pass (and ELU' + bias gradient as one pass backward) instead of conv / add / ELU."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=True)

    def _fused(self, x):
        return (x.is_cuda and x.dtype == torch.float32 and ops.FUSED_NN and self.conv.conv.out_channels > 1
                and self.conv.conv.bias is not None and (x.shape[2] * x.shape[3]) % 4 == 0)

    def forward(self, x):
        if self._fused(x):
            return self.forward_padded(self.conv.pad(x))
        return self.nonlin(self.conv(x))

    def forward_padded(self, xp):
        """`xp` already carries the 1-pixel reflection border (e.g. from `ops.upcat_pad`)."""
        conv = self.conv.conv
        out_hw = (xp.shape[2] - 2) * (xp.shape[3] - 2)
        if xp.is_cuda and xp.dtype == torch.float32 and ops.FUSED_NN and conv.bias is not None and out_hw % 4 == 0:
            y = torch.nn.functional.conv2d(xp, conv.weight, None)
            if y.is_contiguous():        # channels-last inputs / NHWC policies can hand back a strided result
                return ops.bias_elu_(y, conv.bias)
            return self.nonlin(y + conv.bias.view(1, -1, 1, 1))
        return self.nonlin(conv(xp))


def upsample(x):
    return torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")


class BackprojectDepth(nn.Module):
    """depth [n,1,H,W], inv_K [n,4,4] -> camera points [n,4,H*W]  (layers.py:136-167); differentiable
    w.r.t. depth like the reference's module.

    The reference keeps [batch,3,H*W] pixel-grid and ones buffers; the kernel derives pixel
    coordinates from the thread index, so this module holds no tensors."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width

    def forward(self, depth, inv_K, backend=None):
        return ops.backproject(depth, inv_K, self.height, self.width, backend)


class Project3D(nn.Module):
    """points [n,4,H*W], K, T [n,4,4] -> sampling grid [n,H,W,2] in [-1,1]  (layers.py:170-195);
    differentiable w.r.t. points, T and K."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T, backend=None):
        return ops.project3d(points, K, T, self.height, self.width, self.eps, backend)


class SSIM(nn.Module):
    """(1 - SSIM)/2 map between image pairs, [n,3,H,W] -> [n,3,H,W]  (layers.py:219-249); differentiable
    w.r.t. both images."""

    def forward(self, x, y, backend=None):
        return ops.ssim_map(x, y, backend)


def get_smooth_loss(disp, img):
    """Edge-aware first-order smoothness (layers.py:203-216)."""
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


def compute_depth_errors(gt, pred):
    """KITTI depth metrics abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 (layers.py:271-286)."""
    thresh = torch.max(gt / pred, pred / gt)
    a1 = (thresh < 1.25).float().mean()
    a2 = (thresh < 1.25 ** 2).float().mean()
    a3 = (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean((gt - pred) ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3
