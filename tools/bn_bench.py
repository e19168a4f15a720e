#!/usr/bin/env python3
"""HBM throughput of the fused BatchNorm(+residual)+ReLU launches at the ResNet-18 encoder's shapes (MD2 step: the
depth encoder at batch 12, the pose encoder's batched pass at batch 24 in two call groups).  Each C-ABI call is a
pair of launches; bytes = the activation passes the pair makes (forward: statistics read + apply read/write
[+ residual read]; backward: reduce reads x, dy [, y] + apply reads the same and writes dx [, d residual]).
usage: python tools/bn_bench.py [> profiles/rNN/bn_throughput.txt]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from baseboostdepth_amd import ops
from baseboostdepth_amd.networks.encoder import FusedBatchNorm2d

dev = "cuda:0"
be = ops.default_backend()
shapes = [("stem bn1", 64, 96, 320, False), ("layer1 bn1", 64, 48, 160, False), ("layer1 bn2 + identity", 64, 48, 160, True),
          ("layer2 bn2 + identity", 128, 24, 80, True), ("layer3 bn2 + identity", 256, 12, 40, True),
          ("layer4 bn2 + identity", 512, 6, 20, True)]
print("%-26s %-10s %9s %9s %9s %9s" % ("layer", "batch", "fwd us", "fwd TB/s", "bwd us", "bwd TB/s"))
for name, C, H, W, res in shapes:
    for rows in ([12], [12, 12]):
        N = sum(rows)
        bn = FusedBatchNorm2d(C).to(dev).train()
        x = torch.randn(N, C, H, W, device=dev, requires_grad=True)
        r = torch.randn(N, C, H, W, device=dev, requires_grad=True) if res else None
        g = torch.randn(N, C, H, W, device=dev)
        timer = ops.KernelTimer()
        for it in range(25):
            if it == 5:
                be.timer = timer
            with ops.bn_call_groups(rows):
                y = bn(x, residual=r, relu=True)
            y.backward(g)
            x.grad = None
            if res:
                r.grad = None
        torch.cuda.synchronize()
        be.timer = None
        s = timer.summary()
        fwd = s["bbd_bn_act_grouped_fwd"][1] * 1e3
        bwd = s["bbd_bn_act_grouped_bwd"][1] * 1e3
        t = N * C * H * W * 4
        fb = t * (3 + (1 if res else 0))
        bb = t * ((2 + (1 if res else 0)) * 2 + 1 + (1 if res else 0))
        print("%-26s %-10s %9.1f %9.2f %9.1f %9.2f" % (name, "+".join(map(str, rows)), fwd, fb / fwd / 1e6, bwd, bb / bwd / 1e6))
