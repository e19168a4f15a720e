#!/usr/bin/env python3
"""Is the training step host-bound?  Enqueue time of the Python side (no synchronisation inside the loop)
against the wall time including the final device synchronisation, MD2 config of bench.py."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from baseboostdepth_amd import Trainer  # noqa: E402
from baseboostdepth_amd.synthetic import synthetic_batch  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 12
opt = bench.make_options(batch, 0, "md2")
tr = Trainer(opt)
tr.set_train()
inputs = synthetic_batch([1] * batch, bench.H, bench.W, opt.scales, device="cuda:0", seed=42)
for _ in range(10):
    tr.train_step(dict(inputs))
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    tr.train_step(dict(inputs))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("batch %d: host enqueue %.2f ms/step, wall %.2f ms/step" % (batch, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
