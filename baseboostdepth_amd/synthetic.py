"""Synthetic KITTI-shaped batches in the reference's post-collate layout.

There is no dataset on the benchmark machines: images are seeded random textures quantised to
k/255 (what `ToTensor` yields), intrinsics are the reference loader's fixed KITTI matrix
(datasets/kitti_dataset.py:14-23), `stereo_T[0,3] = 0.1`, and `color_aug == color`.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .plan import STEREO
from .layers import transformation_from_parameters


def kitti_intrinsics(H, W):
    K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    K[0, :] *= W
    K[1, :] *= H
    return K, np.linalg.pinv(K)


def _texture(gen, n, H, W, device):
    low = torch.rand(n, 3, H // 8, W // 8, generator=gen, device=device)
    img = 0.7 * F.interpolate(low, size=(H, W), mode="bilinear", align_corners=False)
    img = img + 0.3 * torch.rand(n, 3, H, W, generator=gen, device=device)
    return torch.round(img.clamp(0, 1) * 255) / 255


def synthetic_batch(ms, H=192, W=640, scales=(0, 1, 2, 3), device="cpu", seed=42):
    """ms: per-sample largest usable frame offset (0 = stereo pair only)."""
    gen = torch.Generator(device=device).manual_seed(seed)
    B = len(ms)
    M = max(ms)
    frames = list(range(-M, M + 1))
    if M == 0:
        frames = [0]
    if any(m < 3 for m in ms):
        frames.append(STEREO)
    inputs = {}
    base = _texture(gen, B, H, W + 32, device)
    for f in frames:
        own = [b for b, m in enumerate(ms) if (m < 3 if f == STEREO else m >= abs(f))]
        shift = 0 if f == 0 else (4 if f == STEREO else int(round(1.3 * f)))
        img = base[own][:, :, :, 16 + shift:16 + shift + W]
        if f != 0:
            img = torch.round((img + 0.02 * torch.randn(img.shape, generator=gen, device=device)).clamp(0, 1) * 255) / 255
        inputs[("color", f, 0)] = img.contiguous()
        if f != STEREO:
            inputs[("color_aug", f, 0)] = inputs[("color", f, 0)]
    for s in scales:
        if s:
            small = F.interpolate(inputs[("color", 0, 0)], size=(H >> s, W >> s), mode="area")
            inputs[("color", 0, s)] = torch.round(small * 255) / 255
    K, iK = kitti_intrinsics(H, W)
    inputs[("K", 0)] = torch.from_numpy(K)[None].repeat(B, 1, 1).to(device)
    inputs[("inv_K", 0)] = torch.from_numpy(iK)[None].repeat(B, 1, 1).to(device)
    sT = torch.eye(4)[None].repeat(B, 1, 1)
    sT[:, 0, 3] = 0.1
    inputs["stereo_T"] = sT.to(device)
    inputs["frames"] = frames
    inputs["ordering"] = [[0, STEREO] if m == 0 else [0, m, -m] for m in ms]
    inputs["cutt"] = torch.tensor(0.3)
    inputs["to_use"] = torch.tensor(max(M, 1))
    inputs["noise"] = torch.randn(B, H, W, generator=gen, device=device) * 0.00001
    return inputs


def synthetic_disp(B, H, W, scales, device="cpu", seed=1):
    gen = torch.Generator(device=device).manual_seed(seed)
    return {s: torch.rand(B, 1, H >> s, W >> s, generator=gen, device=device) for s in scales}


def synthetic_poses(plan, device="cpu", seed=2, pose_error=5.5):
    """{("cam_T_cam",0,f): [n_job,4,4]} (+ error poses when the plan has error-induced warps)."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    out = {}
    for f in plan.frames:
        if f == STEREO:
            continue
        n = len(plan.jobs[f])
        aa = 0.01 * torch.randn(n, 1, 3, generator=gen)
        tt = 0.05 * torch.randn(n, 1, 3, generator=gen)
        T = transformation_from_parameters(aa, tt, invert=(f < 0)).to(device)
        out[("cam_T_cam", 0, f)] = T
        if plan.decomp:
            Te = T.clone()
            Te[:, :3, 3:] /= pose_error
            out[("cam_T_cam_error", 0, f)] = Te
    return out


def draw_offsets(rnd, batch_size, epoch, trimin):
    """Per-sample largest usable frame offsets of one batch, drawn like the reference loader does for `epoch`
    (mono_dataset.py:87-109 over the KITTI baselines): before epoch 10 m = 1 (with tri-minimisation
    m in {0,1,2} with the epoch-5 probabilities), from epoch 10 the epoch-15 distribution over 1..7."""
    if epoch < 10:
        return [rnd.choices(range(0, 3), [.062, .573, .366])[0] if trimin else 1 for _ in range(batch_size)]
    return [rnd.choices(range(1, 8), [.050, .050, .077, .094, .139, .142, .448])[0] for _ in range(batch_size)]


def synthetic_loader(batch_size, steps, H=192, W=640, scales=(0, 1, 2, 3), device="cpu", seed=42, trimin=False,
                     epoch=0, canonical=True):
    """Generator of `steps` synthetic batches shaped like the reference loader's output for `epoch`:
    before epoch 10 the largest offset is 1 (2 with tri-minimisation, stereo for small baselines), from
    epoch 10 it follows the epoch-15 offset distribution of SURVEY.md 8d (m in 1..7)."""
    import random
    rnd = random.Random(seed + 1000 * epoch)
    for it in range(steps):
        ms = draw_offsets(rnd, batch_size, epoch, trimin)
        if canonical:           # stacked like the device loader's collate: largest offset first (plan.canonical_permutation)
            ms = sorted(ms, reverse=True)
        batch = synthetic_batch(ms, H, W, scales, device=device, seed=seed + it)
        batch["cutt"] = torch.tensor(0.1 + 0.04 * epoch if epoch < 10 else 0.15 * epoch - 0.9)
        batch.pop("noise")        # let the trainer draw its own identity noise, like the reference
        yield batch


def synthetic_kitti_tree(root, drives=(("2011_09_26/2011_09_26_drive_0001_sync", 375, 1242),
                                       ("2011_09_30/2011_09_30_drive_0020_sync", 370, 1226)), frames=24, seed=0):
    """A small KITTI-raw-shaped directory of synthetic JPEGs (both cameras, KITTI's image sizes) and the split lines
    with baselines that go with it (`splits/eigen_zhou/train_files_baselines.txt` rows: folder, frame, side, 'kt',
    baseline) - what `datasets.KITTIRAWDataset` reads.  There is no dataset on the benchmark machines: the loader-fed
    bench line and the loader tests decode these."""
    import os
    from PIL import Image
    rng = np.random.default_rng(seed)
    lines = []
    for folder, h, w in drives:
        base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 40, 3), dtype=np.uint8)
        big = np.array(Image.fromarray(base).resize((w + 300, h), Image.BICUBIC))
        for cam, dx in (("image_02", 0), ("image_03", 9)):
            d = os.path.join(root, folder, cam, "data")
            os.makedirs(d, exist_ok=True)
            for t in range(frames):
                img = big[:, 5 * t + dx: 5 * t + dx + w]
                Image.fromarray(img).save(os.path.join(d, "%010d.jpg" % t), quality=92)
        for t in range(8, frames - 8):
            for side in "lr":
                lines.append("%s %d %s kt %.6f" % (folder, t, side, float(rng.uniform(0.02, 0.6))))
    return lines
