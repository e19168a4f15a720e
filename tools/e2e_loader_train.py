#!/usr/bin/env python3
"""End-to-end training throughput WITH the loader in the loop (synthetic KITTI-shaped JPEG tree, MD2 frame
set, 640x192, batch 12): decode in worker processes -> shared ring -> device collate -> Trainer.train_step.
Compare with bench.py's number on pre-resident batches (the PCIe / loader-inclusive rate is never `value`)."""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--workers", type=int, default=16)
    a = ap.parse_args()
    import bench
    import image_checks
    from baseboostdepth_amd import Trainer, datasets
    tmp = tempfile.mkdtemp(prefix="bbd_kitti_")
    lines = image_checks.make_kitti_tree(tmp, frames=40) * 40
    opt = bench.make_options(12, 0, "md2")
    tr = Trainer(opt)
    tr.set_train()
    ds = datasets.KITTIRAWDataset(lines, 0, bench.H, bench.W, kt_path=tmp, rand=False, is_train=True, scales=opt.scales,
                                  kt=True, naive_mix=True, trimin=False, seed=1)
    loader = datasets.DeviceLoader(ds, 12, datasets.DeviceCollate(bench.H, bench.W, opt.scales, "cuda:0"),
                                   num_workers=a.workers, prefetch=3, seed=0, workers="process")
    n, t0 = 0, None
    for i, batch in enumerate(loader):
        if i == 20:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        _, losses = tr.train_step(batch)
        last = losses["loss"]
        if i >= 20:
            n += 12
        if i == 20 + a.steps - 1:
            break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(last.detach()).item(), "loss diverged"
    print(json.dumps({"final_loss": round(float(last.detach()), 5), "images_per_s_with_loader": round(n / dt, 1), "ms_per_step": round(dt / (n / 12) * 1e3, 2),
                      "workers": a.workers, "steps": n // 12}))


if __name__ == "__main__":
    main()
