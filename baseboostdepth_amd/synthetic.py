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


# ------------------------------------------------------------------------------------------ structured (coherent) batches
STRUCT_DISP = 0.05            # constant disparity of the planar scene: depth = 1 / (0.01 + 9.99 * 0.05) = 1.963 m
STRUCT_SHIFT = 2.0            # image shift per unit of pose translation multiplier, pixels


def _structured_depth(min_depth=0.1, max_depth=100.0, disp=STRUCT_DISP):
    lo, hi = 1.0 / max_depth, 1.0 / min_depth
    return 1.0 / (lo + (hi - lo) * disp)


def structured_translation(H=192, W=640, shift_px=STRUCT_SHIFT):
    """Camera translation along x that moves the planar scene by `shift_px` pixels (KAT K3: fx * tx / depth)."""
    K, _ = kitti_intrinsics(H, W)
    return float(shift_px * _structured_depth() / K[0, 0])


def pose_multipliers(ms, frame_ids, trimin=True, decomp=True, incremental=True, partial=True, cutt=1.35):
    """k[(b, f)]: by how many unit translations the pose the TRAINER will use for sample b's warp of frame f moves the
    camera, when every pose-network call returns the same unit translation along x and no rotation.  Found by running
    `Trainer.predict_poses` itself (CPU, constant stand-in networks), so every mode quirk of trainer.py:348-418 is in it:
    chained steps add up (k = f), negative offsets beyond -1 keep the identity pose in incremental mode (k = 0), the
    partial swap replaces the translation by the direct call's (k = +-1) except where |f| = m - 2."""
    import types
    import torch.nn as nn
    from .plan import get_plan
    from .trainer import Trainer

    class _Enc(nn.Module):
        def forward(self, x):
            return [x.new_zeros(x.shape[0], 1, 1, 1)]

    class _Dec(nn.Module):
        def forward(self, feats):
            n = feats[0][0].shape[0]
            t = torch.zeros(n, 2, 1, 3)
            t[:, 0, 0, 0] = 1.0
            return torch.zeros(n, 2, 1, 3), t

    tr = Trainer.__new__(Trainer)
    tr.opt = types.SimpleNamespace(height=32, width=32, scales=[0], trimin=trimin, decomp=decomp, pose_error=5.5,
                                   incremental_skip=incremental, partial_skip=partial, batched_pose=False,
                                   frame_ids=sorted(frame_ids, key=lambda f: float("inf") if f == STEREO else abs(f)))
    tr.device = torch.device("cpu")
    tr.models = {"pose_encoder": _Enc().eval(), "pose": _Dec().eval()}
    ordering = [[0, STEREO] if m == 0 else [0, m, -m] for m in ms]
    plan = get_plan(ordering, trimin, decomp)
    inputs = {"ordering": ordering, "cutt": torch.tensor(cutt)}
    for f in frame_ids:
        if f != STEREO:
            inputs[("color_aug", f, 0)] = torch.zeros(len(plan.owners(f)), 3, 2, 2)
    tr.valid_frames_trimin(inputs)
    outputs = tr.predict_poses(inputs)
    per_source_rows = bool(incremental and cutt > 0.5)
    k = {}
    for f in plan.frames:
        if f == STEREO:
            continue
        T = outputs[("cam_T_cam", 0, f)]
        rows = plan.owners(f) if per_source_rows else plan.jobs[f]
        for r, b in enumerate(rows):
            k[(b, f)] = float(T[r, 0, 3])
    return k


def structured_batch(ms, H=192, W=640, scales=(0,), device="cpu", seed=42, trimin=True, decomp=True, incremental=True,
                     partial=True, cutt=1.35, occluders=4, gain=0.08):
    """A batch whose arg-min maps are COHERENT, like a trained network's (random-initialised networks give salt-and-pepper
    maps with every candidate alive in every tile): a fronto-parallel textured plane at constant depth, every source
    frame a true shift of the target by exactly what the pose used for it predicts (`pose_multipliers` x STRUCT_SHIFT
    px; the stereo frame by fx * 0.1 / depth), so all true-pose warps register; per (sample, frame) a smooth brightness
    field (+-`gain`) decides which registered candidate is closest in a region, and `occluders` rectangles of foreign
    texture knock single frames out locally.  Use with `constant_heads(trainer)`: the depth network then predicts
    STRUCT_DISP everywhere and every pose-network call the unit translation."""
    gen = torch.Generator(device=device).manual_seed(seed)
    B, M = len(ms), max(ms)
    frames = list(range(-M, M + 1)) if M > 0 else [0]
    if any(m < 3 for m in ms):
        frames.append(STEREO)
    frame_ids = sorted(frames, key=lambda f: float("inf") if f == STEREO else abs(f))
    k = pose_multipliers(ms, frame_ids, trimin, decomp, incremental, partial, cutt)
    K, iK = kitti_intrinsics(H, W)
    stereo_px = int(round(float(K[0, 0]) * 0.1 / _structured_depth()))
    margin = int(max(stereo_px, STRUCT_SHIFT * (M + 1))) + 8
    base = _texture(gen, B, H, W + 2 * margin, device)
    foreign = _texture(gen, B, H, W + 2 * margin, device)
    inputs = {}
    for f in frames:
        own = [b for b, m in enumerate(ms) if (m < 3 if f == STEREO else m >= abs(f))]
        rows = []
        for b in own:
            shift = 0 if f == 0 else (stereo_px if f == STEREO else int(round(STRUCT_SHIFT * k.get((b, f), 0.0))))
            # the warp samples the source at x + shift: the source must hold base(x) there
            img = base[b, :, :, margin - shift:margin - shift + W]
            if f != 0:
                low = torch.rand(1, 1, 4, 10, generator=gen, device=device) * 2 - 1
                field = 1 + gain * F.interpolate(low, size=(H, W), mode="bicubic", align_corners=False)[0]
                img = img * field
                for _ in range(occluders):
                    h = int(torch.randint(max(2, H // 12), max(3, H // 3), (1,), generator=gen, device=device))
                    w = int(torch.randint(max(2, W // 20), max(3, W // 5), (1,), generator=gen, device=device))
                    y0, x0 = int(torch.randint(0, H - h, (1,), generator=gen, device=device)), int(torch.randint(0, W - w, (1,), generator=gen, device=device))
                    img = img.clone()
                    img[:, y0:y0 + h, x0:x0 + w] = foreign[b, :, y0:y0 + h, margin + x0:margin + x0 + w]
            rows.append(torch.round(img.clamp(0, 1) * 255) / 255)
        inputs[("color", f, 0)] = torch.stack(rows).contiguous()
        if f != STEREO:
            inputs[("color_aug", f, 0)] = inputs[("color", f, 0)]
    for s in scales:
        if s:
            small = F.interpolate(inputs[("color", 0, 0)], size=(H >> s, W >> s), mode="area")
            inputs[("color", 0, s)] = torch.round(small * 255) / 255
    inputs[("K", 0)] = torch.from_numpy(K)[None].repeat(B, 1, 1).to(device)
    inputs[("inv_K", 0)] = torch.from_numpy(iK)[None].repeat(B, 1, 1).to(device)
    sT = torch.eye(4)[None].repeat(B, 1, 1)
    sT[:, 0, 3] = 0.1
    inputs["stereo_T"] = sT.to(device)
    inputs["frames"] = frames
    inputs["ordering"] = [[0, STEREO] if m == 0 else [0, m, -m] for m in ms]
    inputs["cutt"] = torch.tensor(cutt)
    inputs["to_use"] = torch.tensor(max(M, 1))
    inputs["noise"] = torch.randn(B, H, W, generator=gen, device=device) * 0.00001
    return inputs


def constant_heads(trainer, disp=STRUCT_DISP, tx=None):
    """Makes the trainer's networks predict the planar scene of `structured_batch`: the last layer of every disparity
    head and of the pose decoder gets zero weights and the bias that yields `disp` / a translation of `tx` along x
    (the pose head's output is 0.01 * mean(conv), pose_decoder.py:42-44).  Everything in front of them runs (and trains:
    the weights move away from zero at the optimizer's pace) - the step costs what it costs."""
    import math
    opt = trainer.opt
    tx = structured_translation(opt.height, opt.width) if tx is None else tx
    with torch.no_grad():
        depth = trainer.models["depth"]
        for s in depth.scales:
            conv = depth._stage("dispconv", s).conv
            conv.weight.zero_()
            conv.bias.fill_(math.log(disp / (1 - disp)))
        last = trainer.models["pose"].net[3]
        last.weight.zero_()
        last.bias.zero_()
        last.bias[3] = tx / 0.01
    return tx


def live_candidates_per_tile(argmin, tile_h=16, tile_w=32, ncand=20):
    """Histogram {live candidates: tiles} of an arg-min map [B,H,W] over the fused backward's 32x16 tiles."""
    B, H, W = argmin.shape
    a = argmin[:, :H - H % tile_h, :W - W % tile_w].long()
    a = a.reshape(B, H // tile_h, tile_h, W // tile_w, tile_w).permute(0, 1, 3, 2, 4).reshape(-1, tile_h * tile_w)
    present = torch.zeros(a.shape[0], ncand, dtype=torch.bool, device=a.device)
    present.scatter_(1, a, True)
    counts = present.sum(1)
    return {int(c): int((counts == c).sum()) for c in torch.unique(counts).tolist()}
