#!/usr/bin/env python3
"""The hot path alone (generate_images_pred + compute_losses + backward w.r.t. disparities and poses, MD2 B=12, 4 scales) a few
times - run under `rocprofv3 --kernel-trace` and feed the CSV to tools/step_sequence.py to see its launch sequence."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseboostdepth_amd.synthetic import synthetic_batch, synthetic_disp, synthetic_poses  # noqa: E402
from baseboostdepth_amd.trainer import Trainer  # noqa: E402

dev, H, W, B, scales = "cuda:0", 192, 640, 12, [0, 1, 2, 3]
inputs = synthetic_batch([1] * B, H, W, scales, device=dev, seed=42)
opt = types.SimpleNamespace(height=H, width=W, batch_size=B, scales=scales, frame_ids=[0], min_depth=0.1, max_depth=100.0,
                            disparity_smoothness=1e-3, no_ssim=False, trimin=False, decomp=False, pose_error=5.5,
                            incremental_skip=False, partial_skip=False, materialize_warps=False)
tr = Trainer.__new__(Trainer)
tr.opt, tr.device, tr.num_scales, tr.backend, tr.maxing_valid_frames = opt, torch.device(dev), 4, None, False
tr._backend()
plan = tr.valid_frames_trimin(inputs)
disp = {s: d.requires_grad_(True) for s, d in synthetic_disp(B, H, W, scales, device=dev, seed=1).items()}
poses = {k: v.clone().requires_grad_(True) for k, v in synthetic_poses(plan, device=dev, seed=2).items()}
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    out = {("disp", s): disp[s] for s in scales}
    out.update(poses)
    out.update(tr.generate_images_pred(inputs, out))
    tr.compute_losses(inputs, out)["loss"].backward()
torch.cuda.synchronize()
