#!/usr/bin/env python3
"""Whole-step hipGraph experiment: same training run eager and graph-replayed (BBD_STEP_GRAPH=1)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from baseboostdepth_amd import Trainer  # noqa: E402
from baseboostdepth_amd.synthetic import synthetic_batch  # noqa: E402

opt = bench.make_options(12, 0, "md2")
torch.manual_seed(1)
tr = Trainer(opt)
tr.set_train()
batches = [synthetic_batch([1] * 12, bench.H, bench.W, opt.scales, device="cuda:0", seed=s) for s in range(4)]
losses = []
for i in range(12):
    _, l = tr.train_step(dict(batches[i % 4]))
    losses.append(l["loss"].detach().clone())
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for i in range(n):
    tr.train_step(dict(batches[i % 4]))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("graph=%s: %.2f ms/step, %.1f images/s; losses %s" % (tr.use_graph, dt * 1e3, 12 / dt,
                                                           [round(float(x), 5) for x in losses]))
