#!/usr/bin/env python3
"""Per-sample cost of one pose-network pass (ResNet-18 pose encoder + PoseDecoder, forward + backward,
fp32, 640x192) as a function of the batch size: is a boosted step (26 passes of <= 12 samples) limited by
small-batch inefficiency or by the convolution FLOPs themselves?"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseboostdepth_amd import networks  # noqa: E402

dev = "cuda:0"
enc = networks.ResnetEncoder(18, False, num_input_images=2).to(dev).train()
dec = networks.PoseDecoder(enc.num_ch_enc, 1, 2).to(dev).train()
for n in (4, 12, 24, 48, 96, 192):
    x = torch.rand(n, 6, 192, 640, device=dev)

    def step():
        a, t = dec([enc(x)])
        (a.sum() + t.sum()).backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    reps = max(3, 96 // n)
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("batch %3d: %7.2f ms per pass, %.3f ms per sample" % (n, ms, ms / n))
