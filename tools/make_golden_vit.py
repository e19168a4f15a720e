#!/usr/bin/env python3
"""Golden vectors for BASELINE configs[4] (MonoViT): run the upstream `networksvit` package on CPU.

Runs ONLY in the build container (needs /root/reference).  `networksvit.mpvit_small()` and
`networksvit.DepthDecoder()` are imported unmodified under the stubs of tools/refshim.py (timm / mmcv /
mmseg are absent; the ImageNet checkpoint the reference loads unconditionally does not exist and is
answered with an empty state dict), filled with closed-form weights (`fake_nets.fill_deterministic`, so no
multi-MB state dict is stored) and executed.  Written to tests/golden/vit_small.npz - data only:

    keys/enc, keys/dec        the reference's state-dict key lists, in order
    shape/enc, shape/dec      flattened shapes of those entries (rank-padded to 4)
    small/x                   uint8 [2,3,64,128] input * 255
    small/eval/feat/<i>       encoder features of sample 0, eval mode (feature 0: every 4th channel)
    small/eval/disp/<s>       decoder outputs, eval mode, both samples
    small/train/disp/<s>      train mode (BatchNorm batch statistics, DropPath drawn after manual_seed(0))
    small/train/featsum/<i>   (sum, abs-sum) of every train-mode encoder feature
    grad/enc, grad/dec        per state-dict-unique parameter (sum, abs-sum) of d loss / d parameter for
                              loss = sum_s mean(disp_s * w_s) in train mode; NaN rows = no gradient reached it
    small/train0/..., grad0/  the same train-mode pass with stochastic depth off (drop_prob = 0): reproducible on the GPU
    full/x, full/disp/<s>     one 192x640 sample, train mode, stochastic depth off (disp 0/1: every 4th row)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402
from fake_nets import fill_deterministic  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "vit_small.npz")
ENC_PHASE, DEC_PHASE = 0.1, 0.2


def build_reference():
    nv = refshim.import_reference_vit()
    real = torch.load
    torch.load = nv._bbd_fake_load
    try:
        enc = nv.mpvit_small()
    finally:
        torch.load = real
    dec = nv.DepthDecoder()
    return fill_deterministic(enc, ENC_PHASE), fill_deterministic(dec, DEC_PHASE)


def images(seed, n, H, W):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(n, 3, H // 8, W // 8, generator=g)
    img = 0.7 * torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=False)
    img = img + 0.3 * torch.rand(n, 3, H, W, generator=g)
    return torch.round(img.clamp(0, 1) * 255).to(torch.uint8)


def loss_weights(n, H, W):
    g = torch.Generator().manual_seed(77)
    return {s: torch.rand(n, 1, H >> s, W >> s, generator=g) for s in range(4)}


def unique_named_parameters(module):
    return list(module.named_parameters())          # de-duplicated, first registration name


def main():
    torch.set_num_threads(8)
    enc, dec = build_reference()
    out = {}
    for tag, m in (("enc", enc), ("dec", dec)):
        sd = m.state_dict()
        out["keys/" + tag] = np.array(list(sd.keys()))
        out["shape/" + tag] = np.array([list(v.shape) + [0] * (4 - v.dim()) for v in sd.values()], dtype=np.int64)
    x8 = images(3, 2, 64, 128)
    x = x8.float() / 255
    out["small/x"] = x8.numpy()
    enc.eval(); dec.eval()
    with torch.no_grad():
        feats = enc(x)
        disp = dec(feats)
    for i, f in enumerate(feats):
        out["small/eval/feat/%d" % i] = (f[0, ::4] if i == 0 else f[0]).numpy()
    for s in range(4):
        out["small/eval/disp/%d" % s] = disp[("disp", s)].numpy()
    enc.train(); dec.train()
    torch.manual_seed(0)
    feats = enc(x)
    disp = dec(feats)
    w = loss_weights(2, 64, 128)
    sum((disp[("disp", s)] * w[s]).mean() for s in range(4)).backward()
    for i, f in enumerate(feats):
        out["small/train/featsum/%d" % i] = np.array([f.double().sum().item(), f.double().abs().sum().item()])
    for s in range(4):
        out["small/train/disp/%d" % s] = disp[("disp", s)].detach().numpy()
    for tag, m in (("enc", enc), ("dec", dec)):
        rows, names = [], []
        for name, p in unique_named_parameters(m):
            names.append(name)
            rows.append([float("nan")] * 2 if p.grad is None else
                        [p.grad.double().sum().item(), p.grad.double().abs().sum().item()])
        out["grad/" + tag] = np.array(rows)
        out["gradnames/" + tag] = np.array(names)
    # the same train-mode pass with stochastic depth OFF: what the GPU tier can reproduce (DropPath draws come from the
    # device generator there) - pins the ASSEMBLED network's gradients, every parameter, on the GPU
    for m in enc.modules():
        if type(m).__name__ == "DropPath":
            m.drop_prob = 0.0
    for m in (enc, dec):
        for q in m.parameters():
            q.grad = None
    feats = enc(x)
    disp = dec(feats)
    sum((disp[("disp", s)] * w[s]).mean() for s in range(4)).backward()
    for i, f in enumerate(feats):
        out["small/train0/featsum/%d" % i] = np.array([f.double().sum().item(), f.double().abs().sum().item()])
    for s in range(4):
        out["small/train0/disp/%d" % s] = disp[("disp", s)].detach().numpy()
    for tag, m in (("enc", enc), ("dec", dec)):
        rows = []
        for name, q in unique_named_parameters(m):
            rows.append([float("nan")] * 2 if q.grad is None else
                        [q.grad.double().sum().item(), q.grad.double().abs().sum().item()])
        out["grad0/" + tag] = np.array(rows)
    # full size in TRAIN mode: with closed-form weights the eval-mode network (running statistics ~N(0,1),
    # i.e. no normalisation) saturates its sigmoids at 192x640 and the comparison would be ill-conditioned
    # (stochastic depth off for this one: its draws come from the device generator and the GPU tier
    # cannot reproduce the CPU's; the draw order itself is pinned by small/train above)
    for m in enc.modules():
        if type(m).__name__ == "DropPath":
            m.drop_prob = 0.0
    xf8 = images(4, 1, 192, 640)
    out["full/x"] = xf8.numpy()
    with torch.no_grad():
        disp = dec(enc(xf8.float() / 255))
    for s in range(4):
        d = disp[("disp", s)].numpy()
        out["full/disp/%d" % s] = d[:, :, ::4] if s < 2 else d
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB;",
          "params enc %d dec %d" % (sum(p.numel() for p in enc.parameters()), sum(p.numel() for p in dec.parameters())))
    print("gradient-free:", [n for n, r in zip(out["gradnames/enc"], out["grad/enc"]) if np.isnan(r[0])],
          [n for n, r in zip(out["gradnames/dec"], out["grad/dec"]) if np.isnan(r[0])])


if __name__ == "__main__":
    main()
