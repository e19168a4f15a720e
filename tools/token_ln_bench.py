#!/usr/bin/env python3
"""Residual + DropPath + LayerNorm on token activations: the fused passes (csrc/bbd_tokens.hip) against the eager ATen
sequence, at MPViT-small's four stage shapes (batch 12, 192x640).  HIP events, mean of 50."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from baseboostdepth_amd import ops

dev = "cuda:0"


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for N, C in ((7680, 64), (1920, 128), (480, 216), (120, 288)):
    B = 12
    x = torch.randn(B, N, C, device=dev, requires_grad=True)
    br = torch.randn(B, N, C, device=dev, requires_grad=True)
    mask = (torch.rand(B, device=dev) < 0.9).float() / 0.9
    norm = torch.nn.LayerNorm(C, eps=1e-6).to(dev)
    gy, gz = torch.randn(B, N, C, device=dev), torch.randn(B, N, C, device=dev)

    def eager_f():
        y = x + br * mask.view(B, 1, 1)
        return y, norm(y)

    def fused_f():
        return ops.residual_layernorm(x, br, mask, norm)

    res = {}
    for name, f in (("eager", eager_f), ("fused", fused_f)):
        with torch.no_grad():
            res[name + " fwd"] = timeit(f)
        y, z = f()

        def bwd():
            torch.autograd.grad([y, z], [x, br, norm.weight, norm.bias], [gy, gz], retain_graph=True)
        res[name + " bwd"] = timeit(bwd)
    mb = B * N * C * 4 / 1e6
    print("tokens %6d x C %3d (%.1f MB per tensor): forward eager %6.1f us fused %6.1f us | backward eager %6.1f us fused %6.1f us"
          % (B * N, C, mb, res["eager fwd"], res["fused fwd"], res["eager bwd"], res["fused bwd"]))
