#!/usr/bin/env python3
"""Golden vectors for the validation metrics, from the live reference.

Runs ONLY in the build container: imports the reference's `Trainer.compute_depth_losses`
(trainer.py:572-617) unmodified through tools/refshim.py and executes its KITTI branch on seeded
synthetic predictions / sparse ground truth.  Output: tests/golden/eval_cases.npz (data only).

    python tools/make_golden_eval.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "eval_cases.npz")
torch.set_num_threads(1)

# (name, pred h, w, gt GH, GW, valid fraction)
CASES = [("kitti_375", 96, 320, 375, 1242, 0.05), ("kitti_370", 96, 320, 370, 1226, 0.04),
         ("full_res", 192, 640, 375, 1242, 0.05), ("small_out", 24, 40, 48, 76, 0.5),
         ("even_count", 32, 64, 50, 120, 0.3), ("downsample", 192, 640, 120, 400, 0.2)]


def synth(gen, h, w, gh, gw, frac):
    low = torch.rand(1, 1, 6, 12, generator=gen)
    pred = torch.nn.functional.interpolate(low, size=(h, w), mode="bicubic", align_corners=True).clamp(0.01, 1)
    pred = (1.0 / (0.01 + pred * 0.3) + 0.05 * torch.rand(1, 1, h, w, generator=gen)).float()   # depth ~3..100
    gt_dense = torch.nn.functional.interpolate(pred, size=(gh, gw), mode="bilinear", align_corners=True)[0, 0]
    gt = gt_dense * 1.7 * (1 + 0.25 * torch.randn(gh, gw, generator=gen))       # other scale + noise
    keep = torch.rand(gh, gw, generator=gen) < frac
    gt = torch.where(keep, gt, torch.zeros(()))
    gt[0:3, :] = 90.0                                                          # > max_depth rows
    return pred.contiguous(), gt.numpy().astype(np.float32)


def main():
    ref_trainer, _, _ = refshim.import_reference()
    tr = ref_trainer.Trainer.__new__(ref_trainer.Trainer)
    tr.device = torch.device("cpu")
    tr.depth_metric_names = ["de/abs_rel", "de/sq_rel", "de/rms", "de/log_rms", "da/a1", "da/a2", "da/a3"]
    out = {}
    gen = torch.Generator().manual_seed(77)
    for name, h, w, gh, gw, frac in CASES:
        pred, gt = synth(gen, h, w, gh, gw, frac)
        tr.gt_depths = [gt]
        losses = {}
        tr.compute_depth_losses({("depth", 0, 0): pred.clone()}, losses, 0)
        out[name + "/pred"] = pred.numpy()
        out[name + "/gt"] = gt
        out[name + "/metrics"] = np.array([float(losses[k]) for k in tr.depth_metric_names], dtype=np.float64)
        print(name, out[name + "/metrics"])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KB")


if __name__ == "__main__":
    main()
