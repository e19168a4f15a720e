"""GPU tier of the fused BatchNorm(+residual)(+ReLU) kernels (csrc/bbd_nn.hip) against the stock PyTorch
ops they replace inside the encoders: forward, running statistics, all four gradients."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)


@pytest.mark.parametrize("shape,res,relu", [((12, 64, 96, 320), False, True), ((12, 64, 48, 160), True, True),
                                            ((4, 128, 24, 80), False, False), ((3, 7, 5, 9), True, True),
                                            ((12, 512, 6, 20), True, True), ((2, 5, 1, 3), False, True)])
def test_fused_bn_matches_torch(shape, res, relu):
    from baseboostdepth_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    N, C, H, W = shape
    x0 = (torch.randn(shape, generator=g) * 1.7 + 0.3).to(DEV)
    r0 = torch.randn(shape, generator=g).to(DEV) if res else None
    w0 = (torch.rand(C, generator=g) + 0.5).to(DEV)
    b0 = torch.randn(C, generator=g).to(DEV)
    gy = torch.randn(shape, generator=g).to(DEV)
    rm0, rv0 = torch.randn(C, generator=g).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)

    def run(fused):
        x = x0.clone().requires_grad_(True)
        r = r0.clone().requires_grad_(True) if res else None
        w, b = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        rm, rv = rm0.clone(), rv0.clone()
        if fused:
            y = ops.batch_norm_act(x, w, b, r, rm, rv, 0.1, 1e-5, relu)
        else:
            y = F.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)
            if res:
                y = y + r
            if relu:
                y = F.relu(y)
        y.backward(gy)
        return y.detach(), rm, rv, x.grad, w.grad, b.grad, (r.grad if res else None)

    got, want = run(True), run(False)
    names = ["y", "running_mean", "running_var", "grad_x", "grad_w", "grad_b", "grad_res"]
    tol = [1e-5, 1e-5, 1e-5, 2e-4, 2e-4, 2e-4, 1e-6]
    for n, a, b, t in zip(names, got, want, tol):
        if b is None:
            assert a is None
            continue
        assert _rel(a, b) < t, (n, _rel(a, b))
    # deterministic
    again = run(True)
    assert torch.equal(again[0], got[0]) and torch.equal(again[3], got[3]) and torch.equal(again[4], got[4])


def test_encoder_fused_vs_stock_and_state_dict():
    from baseboostdepth_amd import networks
    from baseboostdepth_amd.networks import encoder as enc_mod
    torch.manual_seed(0)
    enc = networks.ResnetEncoder(18, False).to(DEV).train()
    keys = set(enc.state_dict().keys())
    assert "encoder.layer1.0.bn1.running_var" in keys and "encoder.layer2.0.downsample.1.num_batches_tracked" in keys
    x = torch.rand(4, 3, 96, 320, device=DEV)
    state = {k: v.clone() for k, v in enc.state_dict().items()}

    def run(fused):
        enc.load_state_dict(state)
        enc.zero_grad(set_to_none=True)
        enc_mod.FusedBatchNorm2d.fused = fused
        feats = enc(x)
        loss = sum((f * f).mean() for f in feats)
        loss.backward()
        grads = {n: p.grad.clone() for n, p in enc.named_parameters() if p.grad is not None}
        return [f.detach().clone() for f in feats], grads, {k: v.clone() for k, v in enc.state_dict().items()}

    try:
        f1, g1, s1 = run(True)
        f0, g0, s0 = run(False)
    finally:
        enc_mod.FusedBatchNorm2d.fused = True
    for a, b in zip(f1, f0):
        assert _rel(a, b) < 2e-4
    # deep random-init net: ReLU masks flip at rounding level, so compare in aggregate and per tensor loosely
    num = sum(float((g1[n] - g0[n]).pow(2).sum()) for n in g0) ** 0.5
    den = sum(float(g0[n].pow(2).sum()) for n in g0) ** 0.5
    assert num / den < 2e-2, num / den
    worst = max((_rel(g1[n], g0[n]), n) for n in g0)
    assert worst[0] < 0.2, worst
    for k in s0:
        if s0[k].is_floating_point():
            assert _rel(s1[k], s0[k]) < 1e-4, k
        else:
            assert torch.equal(s1[k], s0[k]), k
    # eval mode goes through the stock path and still works
    enc.eval()
    with torch.no_grad():
        assert enc(x)[0].shape == (4, 64, 48, 160)


@pytest.mark.parametrize("shape", [(12, 16, 192, 640), (2, 3, 2, 2), (3, 5, 7, 9), (1, 1, 2, 5), (4, 32, 96, 320)])
def test_reflect_pad_matches_torch(shape):
    from baseboostdepth_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x0 = torch.randn(shape, generator=g).to(DEV)
    gy = torch.randn(shape[0], shape[1], shape[2] + 2, shape[3] + 2, generator=g).to(DEV)
    a = x0.clone().requires_grad_(True)
    b = x0.clone().requires_grad_(True)
    ya, yb = ops.reflect_pad1(a), F.pad(b, (1, 1, 1, 1), mode="reflect")
    assert torch.equal(ya, yb)
    ya.backward(gy)
    yb.backward(gy)
    assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("shape", [(12, 64, 96, 320), (2, 3, 7, 9), (1, 2, 1, 1), (3, 4, 8, 6), (2, 2, 5, 4)])
def test_maxpool_matches_torch(shape):
    from baseboostdepth_amd import ops
    g = torch.Generator().manual_seed(sum(shape) + 1)
    x0 = torch.randn(shape, generator=g)
    x0 = (x0 * 2).round() / 2                       # many ties: exercises the first-maximum rule
    if x0.numel() > 50:
        x0.view(-1)[7] = float("nan")
    x0 = x0.to(DEV)
    a = x0.clone().requires_grad_(True)
    b = x0.clone().requires_grad_(True)
    ya, yb = ops.maxpool3s2(a), F.max_pool2d(b, 3, 2, 1)
    assert torch.equal(torch.nan_to_num(ya, nan=123.0), torch.nan_to_num(yb, nan=123.0))
    gy = torch.randn(yb.shape, generator=g).to(DEV)
    ya.backward(gy)
    yb.backward(gy)
    assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("shape", [(12, 16, 192, 640), (2, 32, 96, 320), (3, 128, 24, 80), (2, 5, 2, 3), (1, 7, 9, 2)])
def test_dispconv_matches_pad_plus_conv(shape):
    from baseboostdepth_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    N, C, H, W = shape
    x0 = torch.randn(shape, generator=g).to(DEV)
    w0 = (torch.randn(1, C, 3, 3, generator=g) * 0.2).to(DEV)
    b0 = torch.randn(1, generator=g).to(DEV)
    gy = torch.randn(N, 1, H, W, generator=g).to(DEV)

    def run(fused):
        x, w, b = (t.clone().requires_grad_(True) for t in (x0, w0, b0))
        y = ops.dispconv(x, w, b) if fused else F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b)
        y.backward(gy)
        return y.detach(), x.grad, w.grad, b.grad

    got, want = run(True), run(False)
    for n, a, b, t in zip(["y", "grad_x", "grad_w", "grad_b"], got, want, [1e-5, 1e-5, 1e-4, 1e-4]):
        assert a.shape == b.shape and _rel(a, b) < t, (n, _rel(a, b))
    again = run(True)
    assert all(torch.equal(p, q) for p, q in zip(got, again))           # deterministic


def test_depth_decoder_fused_vs_stock():
    from baseboostdepth_amd import networks, ops
    torch.manual_seed(1)
    enc = networks.ResnetEncoder(18, False).to(DEV).eval()
    dec = networks.DepthDecoder(enc.num_ch_enc, [0, 1, 2, 3]).to(DEV).train()
    with torch.no_grad():
        feats = [f.clone() for f in enc(torch.rand(2, 3, 96, 320, device=DEV))]

    def run(flag):
        ops.FUSED_NN = flag
        dec.zero_grad(set_to_none=True)
        out = dec([f.clone().requires_grad_(True) for f in feats])
        sum((out[("disp", s)] ** 2).mean() for s in range(4)).backward()
        return [out[("disp", s)].detach() for s in range(4)], {n: p.grad.clone() for n, p in dec.named_parameters()}

    try:
        d1, g1 = run(True)
        d0, g0 = run(False)
    finally:
        ops.FUSED_NN = True
    for a, b in zip(d1, d0):
        assert _rel(a, b) < 1e-5
    for n in g0:
        assert _rel(g1[n], g0[n]) < 2e-3, n


@pytest.mark.parametrize("N,C1,C2,h,w", [(2, 16, 8, 6, 10), (1, 5, 0, 3, 5), (3, 32, 64, 24, 80), (2, 4, 4, 1, 2)])
def test_upsample_concat_pad_in_one_pass(N, C1, C2, h, w):
    """ReflectionPad2d(1)(cat(nearest_x2(x), skip)) against the three torch ops, forward and both gradients."""
    from baseboostdepth_amd import ops
    g = torch.Generator().manual_seed(N * 10 + C1)
    x0 = torch.randn(N, C1, h, w, generator=g).to(DEV)
    s0 = torch.randn(N, C2, 2 * h, 2 * w, generator=g).to(DEV) if C2 else None
    go = torch.randn(N, C1 + C2, 2 * h + 2, 2 * w + 2, generator=g).to(DEV)

    def run(fused):
        x = x0.clone().requires_grad_(True)
        s = s0.clone().requires_grad_(True) if s0 is not None else None
        if fused:
            y = ops.upcat_pad(x, s)
        else:
            u = F.interpolate(x, scale_factor=2, mode="nearest")
            y = F.pad(torch.cat([u, s], 1) if s is not None else u, (1, 1, 1, 1), mode="reflect")
        y.backward(go)
        return [y.detach(), x.grad] + ([s.grad] if s is not None else [])

    for a, b in zip(run(True), run(False)):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-6 * (1 + float(b.abs().max())), float((a - b).abs().max())


def test_bias_elu_in_place_and_its_backward():
    from baseboostdepth_amd import ops
    g = torch.Generator().manual_seed(4)
    for (N, C, H, W) in ((2, 16, 12, 20), (1, 3, 6, 10), (3, 64, 48, 160)):
        v0 = (2 * torch.randn(N, C, H, W, generator=g)).to(DEV)
        b0 = torch.randn(C, generator=g).to(DEV)
        gy = torch.randn(N, C, H, W, generator=g).to(DEV)

        def run(fused):
            v, b = v0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
            y = ops.bias_elu_(v * 1.0, b) if fused else F.elu(v + b.view(1, C, 1, 1))
            y.backward(gy)
            return y.detach(), v.grad, b.grad

        got, want = run(True), run(False)
        for name, a, b_, tol in zip(("y", "grad_x", "grad_bias"), got, want, (1e-6, 1e-6, 2e-5)):
            assert _rel(a, b_) < tol, (name, _rel(a, b_))
        assert all(torch.equal(p, q) for p, q in zip(got, run(True)))       # deterministic


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,smalls", [
    (2, 192, 640, [(96, 320), (48, 160), (24, 80)]),      # the reference's scales 1-3: separable tiled kernel
    (3, 50, 70, [(25, 35), (13, 18), (7, 9)]),            # ragged sizes, non-integer factors, partial tiles
    (1, 64, 96, [(4, 6)]),                                 # factor 16: the gather-form fallback
    (2, 64, 96, [(32, 48), (4, 6)]),                       # one launch mixing both -> fallback for all
])
def test_upsample_adjoint_matches_autograd_of_interpolate(B, H, W, smalls):
    """bbd_disp_upsample_adjoint (one launch for every reduced scale) = the autograd of
    F.interpolate(bilinear, align_corners=False) (trainer.py:455-458) applied to grad_up."""
    from baseboostdepth_amd import ops
    be = ops.default_backend()
    torch.manual_seed(5)
    grad_up = [torch.randn(B, H, W, device=DEV) for _ in smalls]
    outs = [torch.full((B, 1, h, w), float("nan"), device=DEV) for h, w in smalls]
    be.run("bbd_disp_upsample_adjoint", grad_up[0], ops._ptr_array(grad_up), ops._hw_array(outs), ops._ptr_array(outs),
           len(smalls), B, H, W)
    for g, o, (h, w) in zip(grad_up, outs, smalls):
        d = torch.zeros(B, 1, h, w, device=DEV, requires_grad=True)
        F.interpolate(d, [H, W], mode="bilinear", align_corners=False).backward(g.unsqueeze(1))
        assert torch.isfinite(o).all()
        assert float((o - d.grad).abs().max()) <= 2e-6 * float(d.grad.abs().max()), (h, w)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,res,relu,H,W", [([5, 3, 4], True, True, 12, 20), ([12, 12], False, True, 12, 20),
                                                ([1, 7, 2, 6], False, False, 12, 20),
                                                ([12, 2, 7], True, True, 48, 160),      # groups with different slice counts
                                                ([3, 12], False, True, 31, 47)])       # HW % 4 != 0: scalar path
def test_grouped_batch_norm_equals_separate_calls(rows, res, relu, H, W):
    """`ops.bn_call_groups`: one batched pass whose BatchNorm keeps its statistics per call group = the separate calls
    of the layer on the sub-batches (outputs, data gradients and running statistics bit for bit; parameter gradients
    to rounding - the separate calls' are added up by autograd in fp32, the grouped form sums in fp64)."""
    import copy
    from baseboostdepth_amd import ops
    from baseboostdepth_amd.networks.encoder import FusedBatchNorm2d
    torch.manual_seed(11)
    C = 16
    N = sum(rows)
    x = torch.randn(N, C, H, W, device=DEV) * 2 + 0.5
    r = torch.randn(N, C, H, W, device=DEV) if res else None
    wgt = torch.randn(N, C, H, W, device=DEV)
    sep = FusedBatchNorm2d(C).to(DEV).train()
    with torch.no_grad():
        sep.weight.uniform_(0.5, 1.5)
        sep.bias.uniform_(-0.5, 0.5)
    grp = copy.deepcopy(sep)

    xs, rs = x.clone().requires_grad_(True), (r.clone().requires_grad_(True) if res else None)
    ys, lo = [], 0
    for n in rows:
        ys.append(sep(xs[lo:lo + n], residual=rs[lo:lo + n] if res else None, relu=relu))
        lo += n
    y_sep = torch.cat(ys)
    (y_sep * wgt).sum().backward()

    xg, rg = x.clone().requires_grad_(True), (r.clone().requires_grad_(True) if res else None)
    with ops.bn_call_groups(rows):
        y_grp = grp(xg, residual=rg, relu=relu)
    (y_grp * wgt).sum().backward()

    assert torch.equal(y_grp, y_sep)
    assert torch.equal(xg.grad, xs.grad)
    if res:
        assert torch.equal(rg.grad, rs.grad)
    assert torch.equal(grp.running_mean, sep.running_mean) and torch.equal(grp.running_var, sep.running_var)
    assert int(grp.num_batches_tracked) == int(sep.num_batches_tracked) == len(rows)
    assert _rel(grp.weight.grad, sep.weight.grad) < 1e-5 and _rel(grp.bias.grad, sep.bias.grad) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("rows,pad,H,W", [([5, 3, 4], 4, 12, 20), ([12, 2, 7], 11, 48, 160)])
def test_padding_group_leaves_the_real_call_groups_untouched(rows, pad, H, W):
    """`ops.bn_call_groups(rows + [pad], padding_groups=1)`: a trailing group of zero rows (what `Trainer._pose_pairs`
    rounds the batched pose pass up with) - the real groups' outputs, data gradients, running statistics and
    num_batches_tracked are bit for bit those of the unpadded pass, parameter gradients too (the padding's terms are
    exact zeros)."""
    import copy
    from baseboostdepth_amd import ops
    from baseboostdepth_amd.networks.encoder import FusedBatchNorm2d
    torch.manual_seed(12)
    C, N = 16, sum(rows)
    x = torch.randn(N, C, H, W, device=DEV) * 2 + 0.5
    wgt = torch.randn(N, C, H, W, device=DEV)
    plain = FusedBatchNorm2d(C).to(DEV).train()
    with torch.no_grad():
        plain.weight.uniform_(0.5, 1.5)
        plain.bias.uniform_(-0.5, 0.5)
    padded = copy.deepcopy(plain)
    xa = x.clone().requires_grad_(True)
    with ops.bn_call_groups(rows):
        ya = plain(xa, relu=True)
    (ya * wgt).sum().backward()
    xb = torch.cat([x, x.new_zeros(pad, C, H, W)]).requires_grad_(True)
    with ops.bn_call_groups(rows + [pad], padding_groups=1):
        yb = padded(xb, relu=True)
    (yb[:N] * wgt).sum().backward()
    assert torch.equal(yb[:N], ya)
    assert torch.equal(xb.grad[:N], xa.grad) and not xb.grad[N:].any()
    assert torch.equal(padded.running_mean, plain.running_mean) and torch.equal(padded.running_var, plain.running_var)
    assert int(padded.num_batches_tracked) == int(plain.num_batches_tracked) == len(rows)
    assert torch.equal(padded.weight.grad, plain.weight.grad) and torch.equal(padded.bias.grad, plain.bias.grad)


@pytest.mark.gpu
def test_padded_pose_pass_equals_the_exact_one():
    """Trainer._pose_pairs with `pose_pad_rows`: poses, pose-network gradients and BatchNorm buffers of the exact pass."""
    from baseboostdepth_amd.trainer import Trainer
    import bench
    torch.manual_seed(3)
    opt = bench.make_options(4, 0, "md2")
    opt.height, opt.width = 64, 128
    tr = Trainer(opt)
    tr.set_train()
    frames = [torch.rand(n, 3, 64, 128, device=DEV) for n in (4, 4, 3, 2, 3, 2)]
    reqs = [(frames[0], frames[1], False), (frames[2], frames[4], True), (frames[3], frames[5], False)]
    params = [p for k in ("pose_encoder", "pose") for p in tr.models[k].parameters()]
    state = {k: {n: b.clone() for n, b in tr.models[k].named_buffers()} for k in ("pose_encoder", "pose")}

    def run(pad_rows):
        for k in state:
            for n, b in tr.models[k].named_buffers():
                b.copy_(state[k][n])
        for p in params:
            p.grad = None
        tr.pose_pad_rows = pad_rows
        Ts = tr._pose_pairs(reqs)
        sum((T * T).sum() for T in Ts).backward()
        return ([T.detach().clone() for T in Ts], [p.grad.clone() for p in params if p.grad is not None],
                [b.clone() for b in tr.models["pose_encoder"].buffers()])

    Ta, ga, ba = run(0)
    Tb, gb, bb = run(16)               # 9 rows -> 16
    for a, b in zip(Ta, Tb):
        assert float((a - b).abs().max()) < 1e-6, float((a - b).abs().max())
    for a, b in zip(ba, bb):           # (MIOpen picks its solvers per batch size: the activations agree to rounding, and so
        if a.is_floating_point():      #  do the running statistics; the update COUNT is exact - the padding group is not counted)
            assert _rel(a.float(), b.float()) < 1e-5
        else:
            assert torch.equal(a, b)
    for a, b in zip(ga, gb):           # MIOpen picks solvers per batch size: convolution weight gradients to rounding
        assert float((a - b).abs().max()) <= 2e-3 * float(a.abs().max()) + 1e-7


@pytest.mark.gpu
def test_batched_pose_pairs_equal_the_separate_calls():
    """Trainer._pose_pairs: all pose-network calls of a step as ONE pass (call groups in every BatchNorm) give the poses
    and the network gradients of the reference's separate calls."""
    import types
    from baseboostdepth_amd.trainer import Trainer
    import bench
    torch.manual_seed(3)
    opt = bench.make_options(4, 0, "md2")
    opt.height, opt.width = 64, 128
    tr = Trainer(opt)
    tr.set_train()
    frames = [torch.rand(n, 3, 64, 128, device=DEV) for n in (4, 4, 3, 2, 3, 2)]
    reqs = [(frames[0], frames[1], False), (frames[2], frames[4], True), (frames[3], frames[5], False)]
    params = [p for k in ("pose_encoder", "pose") for p in tr.models[k].parameters()]
    state = {k: {n: b.clone() for n, b in tr.models[k].named_buffers()} for k in ("pose_encoder", "pose")}

    def run(batched):
        for k in state:
            for n, b in tr.models[k].named_buffers():
                b.copy_(state[k][n])
        for p in params:
            p.grad = None
        tr.opt.batched_pose = batched
        Ts = tr._pose_pairs(reqs)
        sum((T * T).sum() for T in Ts).backward()
        return ([T.detach().clone() for T in Ts], [p.grad.clone() for p in params if p.grad is not None],
                [b.clone() for b in tr.models["pose_encoder"].buffers()])

    Ta, ga, ba = run(False)
    Tb, gb, bb = run(True)
    for a, b in zip(Ta, Tb):
        assert float((a - b).abs().max()) < 1e-5, float((a - b).abs().max())
    for a, b in zip(ba, bb):
        assert _rel(a.float(), b.float()) < 1e-5
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 2e-3 * float(a.abs().max()) + 1e-7
