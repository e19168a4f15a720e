#!/usr/bin/env python3
"""Throughput of the index-table loader on a synthetic KITTI tree: host decode rate (frames/s over the
thread pool) and the device part of collation (ms per batch, HIP events), for the MD2 frame set
(4 frames per sample) at 640x192, batch 12.

    python tools/loader_bench.py [--batches 20] [--workers 16]
"""
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
    ap.add_argument("--batches", type=int, default=100)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--mode", default="process", choices=["process", "thread"])
    a = ap.parse_args()
    import image_checks
    from baseboostdepth_amd import datasets
    tmp = tempfile.mkdtemp(prefix="bbd_kitti_")
    lines = image_checks.make_kitti_tree(tmp, frames=40) * 40      # 3 840 split lines over the same files
    H, W = 192, 640
    ds = datasets.KITTIRAWDataset(lines, 0, H, W, kt_path=tmp, rand=False, is_train=True, scales=[0, 1, 2, 3], kt=True,
                                  naive_mix=True, trimin=False, seed=1)
    collate = datasets.DeviceCollate(H, W, [0, 1, 2, 3], "cuda:0")
    # device part alone: recipes decoded once, collated repeatedly
    recipes = [ds[i] for i in range(a.batch)]
    for _ in range(3):
        collate(recipes)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(10):
        collate(recipes)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 10
    dev_ms = e0.elapsed_time(e1) / 10
    frames = sum(len(r["images"]) for r in recipes)
    # end to end: decode on the pool + collate
    loader = datasets.DeviceLoader(ds, a.batch, collate, num_workers=a.workers, prefetch=3, seed=0, workers=a.mode)
    n, t0 = 0, None
    for i, batch in enumerate(loader):
        if i == 5:                       # past worker start-up and the prefetch backlog
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if i >= 5:
            n += batch[("color", 0, 0)].shape[0]
        if i == a.batches + 4:
            break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"collate_device_ms_per_batch": round(dev_ms, 3), "collate_wall_ms_per_batch": round(wall * 1e3, 3),
                      "frames_per_batch": frames, "loader_samples_per_s": round((n - a.batch) / dt, 1),
                      "loader_frames_per_s": round((n - a.batch) * frames / a.batch / dt, 1), "workers": a.workers, "mode": a.mode,
                      "cpu_count": os.cpu_count(), "batch": a.batch, "size": "1242x375 -> 640x192, 4 scales for frame 0"}))


if __name__ == "__main__":
    main()
