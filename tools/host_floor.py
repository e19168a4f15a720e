#!/usr/bin/env python3
"""Host-side floor of the training step: the same MD2 step on a tiny problem (64x128, batch 1), where
every kernel takes microseconds, so the wall time per step is the Python / launch overhead alone."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from baseboostdepth_amd import Trainer  # noqa: E402
from baseboostdepth_amd.synthetic import synthetic_batch  # noqa: E402

H, W, B = 64, 128, 1
opt = bench.make_options(B, 0, "md2")
opt.height, opt.width = H, W
tr = Trainer(opt)
tr.set_train()
inputs = synthetic_batch([1] * B, H, W, opt.scales, device="cuda:0", seed=42)
for _ in range(20):
    tr.train_step(dict(inputs))
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    tr.train_step(dict(inputs))
torch.cuda.synchronize()
print("tiny problem: %.2f ms/step (host floor)" % ((time.perf_counter() - t0) / n * 1e3))
