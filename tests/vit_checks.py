"""Shared by the CPU and GPU tiers: this build's `networksvit` (MPViT-small encoder + HR decoder) against
tests/golden/vit_small.npz, captured from the live reference package by tools/make_golden_vit.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
from fake_nets import fill_deterministic  # noqa: E402

GOLDEN = os.path.join(HERE, "golden", "vit_small.npz")
ENC_PHASE, DEC_PHASE = 0.1, 0.2


def build(device="cpu"):
    from baseboostdepth_amd import networksvit
    enc = fill_deterministic(networksvit.mpvit_small(checkpoint=None), ENC_PHASE).to(device)
    dec = fill_deterministic(networksvit.DepthDecoder(), DEC_PHASE).to(device)
    return enc, dec


def loss_weights(n, H, W):
    g = torch.Generator().manual_seed(77)
    return {s: torch.rand(n, 1, H >> s, W >> s, generator=g) for s in range(4)}


def _close(got, want, rel, what):
    want = torch.from_numpy(np.asarray(want))
    err = float((got.detach().cpu() - want).abs().max())
    scale = float(want.abs().max()) + 1e-12
    assert err <= rel * scale, "%s: max err %.3e vs scale %.3e (rel %.2e > %.0e)" % (what, err, scale, err / scale, rel)


def check_state_dict_layout():
    z = np.load(GOLDEN)
    enc, dec = build()
    for tag, m in (("enc", enc), ("dec", dec)):
        sd = m.state_dict()
        assert list(sd.keys()) == list(z["keys/" + tag]), "%s state-dict keys differ from the reference's" % tag
        shapes = [list(v.shape) + [0] * (4 - v.dim()) for v in sd.values()]
        assert shapes == z["shape/" + tag].tolist()
    assert sum(p.numel() for p in enc.parameters()) == 22603400
    assert sum(p.numel() for p in dec.parameters()) == 5266708
    assert list(enc.num_ch_enc) == [64, 128, 216, 288, 288]      # what trainer.py:54 assigns by hand


def check_forward_and_gradients(device, rel_eval, rel_train, rel_grad):
    z = np.load(GOLDEN)
    enc, dec = build(device)
    x = torch.from_numpy(z["small/x"]).float().div(255).to(device)
    enc.eval(); dec.eval()
    with torch.no_grad():
        feats = enc(x)
        disp = dec(feats)
    assert [tuple(f.shape[1:]) for f in feats] == [(64, 32, 64), (128, 16, 32), (216, 8, 16), (288, 4, 8), (288, 2, 4)]
    for i, f in enumerate(feats):
        _close(f[0, ::4] if i == 0 else f[0], z["small/eval/feat/%d" % i], rel_eval, "eval feature %d" % i)
    for s in range(4):
        assert disp[("disp", s)].shape == (2, 1, 64 >> s, 128 >> s)
        _close(disp[("disp", s)], z["small/eval/disp/%d" % s], 10 * rel_eval, "eval disp %d" % s)
    # train mode: BatchNorm batch statistics + stochastic depth (same draw order as the reference's
    # timm DropPath on the CPU generator; on the GPU the draws differ, so DropPath is checked on CPU only)
    enc.train(); dec.train()
    on_cpu = torch.device(device).type == "cpu"
    if not on_cpu:
        for m in enc.modules():
            if type(m).__name__ == "DropPath":
                m.drop_prob = 0.0
    torch.manual_seed(0)
    feats = enc(x)
    disp = dec(feats)
    w = loss_weights(2, 64, 128)
    loss = sum((disp[("disp", s)] * w[s].to(device)).mean() for s in range(4))
    loss.backward()
    def compare(prefix, gkey):
        for i, f in enumerate(feats):
            want = z["small/%s/featsum/%d" % (prefix, i)]
            assert abs(f.double().sum().item() - want[0]) <= rel_train * want[1], "train feature %d" % i
        for s in range(4):
            _close(disp[("disp", s)], z["small/%s/disp/%d" % (prefix, s)], 10 * rel_train, "train disp %d" % s)
        for tag, m in (("enc", enc), ("dec", dec)):
            names, rows = list(z["gradnames/" + tag]), z["%s/%s" % (gkey, tag)]
            params = dict(m.named_parameters())
            assert list(params.keys()) == names
            for name, (gsum, gabs) in zip(names, rows):
                g = params[name].grad
                if np.isnan(gsum):
                    assert g is None, "%s should receive no gradient" % name
                    continue
                assert g is not None, name
                # parameters whose gradient is zero in exact arithmetic (a per-channel shift in front of a
                # train-mode BatchNorm: fc2.bias of every path, InvRes.conv2.bn.bias) hold pure round-off
                # (|g| ~ 1e-9 and below); hence an absolute floor (median |g| mass of a parameter is 3e-3)
                floor = 1e-6 if on_cpu else 1e-5
                assert abs(g.double().sum().item() - gsum) <= rel_grad * gabs + floor, (tag, name)
                assert abs(g.double().abs().sum().item() - gabs) <= rel_grad * gabs + floor, (tag, name)

    if on_cpu:
        compare("train", "grad")
    # stochastic depth off: the assembled network's outputs AND every parameter's gradient against the reference's,
    # on whichever device this runs (the GPU tier's HIP token kernels included)
    for m in enc.modules():
        if type(m).__name__ == "DropPath":
            m.drop_prob = 0.0
    for m in (enc, dec):
        for q in m.parameters():
            q.grad = None
    feats = enc(x)
    disp = dec(feats)
    loss = sum((disp[("disp", s)] * w[s].to(device)).mean() for s in range(4))
    loss.backward()
    compare("train0", "grad0")
    # the decoder's never-used blocks (reference hr_decoder.py builds X_0j_Conv_0 and never calls them)
    free = sorted(n for n, p in dec.named_parameters() if p.grad is None)
    assert free == sorted("convs.X_0%d_Conv_0.conv.conv.%s" % (j, t) for j in range(4) for t in ("weight", "bias"))


def check_full_size(device, rel):
    z = np.load(GOLDEN)
    enc, dec = build(device)
    enc.train(); dec.train()
    for m in enc.modules():
        if type(m).__name__ == "DropPath":
            m.drop_prob = 0.0
    x = torch.from_numpy(z["full/x"]).float().div(255).to(device)
    with torch.no_grad():
        disp = dec(enc(x))
    for s in range(4):
        d = disp[("disp", s)]
        assert d.shape == (1, 1, 192 >> s, 640 >> s)
        _close(d[:, :, ::4] if s < 2 else d, z["full/disp/%d" % s], rel, "192x640 disp %d" % s)
