#!/usr/bin/env python3
"""Time every distinct convolution of the MD2 networks (forward + backward, fp32, batch 12 at 640x192) in
isolation on MIOpen, with the bytes and FLOPs each moves: which layers are far from their roofline?"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseboostdepth_amd import networks  # noqa: E402

dev = "cuda:0"
B, H, W = 12, 192, 640
enc = networks.ResnetEncoder(18, False).to(dev)
dec = networks.DepthDecoder(enc.num_ch_enc, [0, 1, 2, 3]).to(dev)
shapes = {}


def hook(name):
    def f(m, inp, out):
        x = inp[0]
        key = (tuple(x.shape), tuple(m.weight.shape), m.stride, m.padding)
        shapes.setdefault(key, []).append(name)
    return f


for n, m in list(enc.named_modules()) + list(dec.named_modules()):
    if isinstance(m, torch.nn.Conv2d):
        m.register_forward_hook(hook(n))
with torch.no_grad():
    dec(enc(torch.rand(B, 3, H, W, device=dev)))
rows = []
for (xs, ws, st, pd), names in shapes.items():
    x = torch.randn(xs, device=dev, requires_grad=True)
    w = torch.randn(ws, device=dev, requires_grad=True)

    def step():
        y = F.conv2d(x, w, None, st, pd)
        y.backward(torch.ones_like(y))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    y = F.conv2d(x, w, None, st, pd)
    flops = 3 * 2 * y.numel() * ws[1] * ws[2] * ws[3]
    bytes_ = 4 * (3 * x.numel() + 3 * y.numel())
    rows.append((ms * len(names), ms, len(names), xs, ws, st, flops / ms / 1e9, bytes_ / ms / 1e6, names[0]))
rows.sort(reverse=True)
print("total ms (x uses) | ms | uses | input | weight | stride | TFLOP/s | GB/s(min traffic) | first use")
tot = 0
for r in rows:
    tot += r[0]
    print("%6.2f | %5.2f | %d | %s | %s | %s | %6.1f | %6.0f | %s" % r)
print("sum over depth encoder+decoder convs (one pass each): %.2f ms" % tot)
