"""GPU tier, BASELINE configs[4]: MonoViT (MPViT-small encoder + HR decoder) on the MI355X against the
reference-generated vectors, and `Trainer.process_batch` with `--ViT` feeding the SAME HIP loss kernels,
re-checked against the live oracle hot path on the networks' own outputs."""
import pytest
import torch

import vit_checks

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_encoder_decoder_match_reference_vectors_on_gpu():
    vit_checks.check_forward_and_gradients(DEV, rel_eval=1e-3, rel_train=1e-3, rel_grad=1e-2)


def test_full_size_forward_on_gpu():
    vit_checks.check_full_size(DEV, rel=5e-3)


def test_process_batch_vit_end_to_end():
    from test_gpu_trainer import make_opt, oracle_on_outputs
    from baseboostdepth_amd.trainer import Trainer
    from baseboostdepth_amd import networksvit
    from baseboostdepth_amd.synthetic import synthetic_batch
    H, W, ms, scales = 96, 160, [1, 1, 1], [0, 1, 2, 3]
    torch.manual_seed(0)
    opt = make_opt(H, W, len(ms), scales, False)
    opt.ViT = True
    tr = Trainer(opt)
    assert isinstance(tr.models["encoder"], networksvit.MPViT) and isinstance(tr.model_optimizer, torch.optim.AdamW)
    tr.set_train()
    inputs = synthetic_batch(ms, H, W, scales, device=DEV, seed=3)
    tr.opt.frame_ids = sorted(inputs["frames"], key=lambda it: float("inf") if isinstance(it, str) else abs(it))
    outputs, losses = tr.process_batch(inputs)
    assert torch.isfinite(losses["loss"])
    for s in scales:
        assert outputs[("disp", s)].shape == (len(ms), 1, H >> s, W >> s)
    ref, _ = oracle_on_outputs(tr, inputs, outputs, opt, ms)
    assert abs(float(losses["loss"].detach()) - float(ref["loss"].detach())) < 1e-5
    for i, s in enumerate(scales):
        got = outputs[("bbd", "to_optimise")][i].cpu()
        assert float((got - ref["min/%d" % s]).abs().max()) < 1e-4
        mism = outputs[("bbd", "argmin")][i].cpu() != ref["argmin/%d" % s]
        assert int((mism & (ref["margin/%d" % s] > 2e-4)).sum()) == 0
    losses["loss"].backward()
    free = {id(p) for p in tr.gradient_free_parameters()}
    for name, model in tr.models.items():
        for n, p in model.named_parameters():
            if id(p) in free:
                assert p.grad is None, (name, n)
            else:
                assert p.grad is not None and bool(torch.isfinite(p.grad).all()), (name, n)


def test_vit_train_steps_reduce_the_loss():
    from test_gpu_trainer import make_opt
    from baseboostdepth_amd.trainer import Trainer
    from baseboostdepth_amd.synthetic import synthetic_batch
    H, W, ms = 96, 160, [1, 1, 1, 1]
    torch.manual_seed(0)
    opt = make_opt(H, W, 4, [0, 1, 2, 3], False)
    opt.ViT = True
    tr = Trainer(opt)
    tr.set_train()
    for m in tr.models["encoder"].modules():          # deterministic objective for this check
        if type(m).__name__ == "DropPath":
            m.drop_prob = 0.0
    inputs = synthetic_batch(ms, H, W, [0, 1, 2, 3], device=DEV, seed=5)
    inputs.pop("noise")
    hist = []
    for _ in range(30):
        _, l = tr.train_step(dict(inputs))
        hist.append(float(l["loss"].detach()))
    assert all(h == h for h in hist)
    assert sum(hist[-5:]) / 5 < sum(hist[:5]) / 5, (hist[:5], hist[-5:])


def test_vit_boosted_recipe_in_pooled_form_with_step_graphs():
    """MonoViT (`--ViT`, trainer.py:52-58) under the boosted `--rand` recipe: the step runs in pooled form (the depth
    network is the only thing that differs from the ResNet path) and its bucket graph replays for new orderings."""
    import warnings
    from test_gpu_trainer import make_opt
    from baseboostdepth_amd.trainer import Trainer
    from baseboostdepth_amd.synthetic import synthetic_batch
    H, W, B = 96, 160, 4
    torch.manual_seed(0)
    opt = make_opt(H, W, B, [0, 1, 2, 3], True)
    opt.ViT, opt.rand, opt.step_graph = True, True, True
    tr = Trainer(opt)
    tr.opt.scales = [0]
    tr.set_train()
    losses = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i, ms in enumerate([[7, 7, 7, 1], [7, 7, 6, 1], [7, 7, 3, 2], [7, 6, 6, 1]]):      # 64 pose rows each
            b = synthetic_batch(ms, H, W, [0], device=DEV, seed=50 + i)
            b.pop("noise")
            b["cutt"] = torch.tensor(1.35)
            _, l = tr.train_step(b)
            losses.append(l["loss"].detach())
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(x)) for x in losses)
    assert tr.graph_stats == {"eager": 0, "captures": 1, "replays": 4} and tr._pooled.stats["fallbacks"] == 0


@pytest.mark.parametrize("B,H,W,C,splits,add", [
    (2, 12, 40, 216, (54, 81, 81), False),       # stage-2 ConvRelPosEnc: 27-channel heads, windows 3/5/7
    (3, 6, 20, 64, (16, 24, 24), False),
    (2, 9, 13, 64, (64,), True),                 # ConvPosEnc (residual folded in), odd sizes
    (1, 5, 3, 8, (2, 3, 3), False),              # image smaller than the 7x7 window
])
def test_depthwise_token_convolution_matches_torch(B, H, W, C, splits, add):
    """HIP depth-wise convolution on token-layout activations (forward, data / weight / bias gradients)
    against nn.Conv2d on the NCHW view, reading its input in place from a wider (qkv-like) row."""
    import torch.nn as nn
    from baseboostdepth_amd import ops
    g = torch.Generator().manual_seed(B * 100 + C)
    wide = torch.randn(B, H * W, 3 * C, generator=g).to(DEV).requires_grad_(True)
    convs = [nn.Conv2d(n, n, k, 1, k // 2, groups=n).to(DEV) for n, k in zip(splits, (3, 5, 7))]
    w = torch.randn(B, H * W, C, generator=g).to(DEV)
    x = wide[:, :, 2 * C:]                                                    # the "v" third, row stride 3C
    y = ops.dwconv_tokens(x, (H, W), convs, add_input=add)
    (y * w).sum().backward()
    got = [y.detach(), wide.grad.clone()] + [p.grad.clone() for c in convs for p in c.parameters()]
    wide.grad = None
    for c in convs:
        c.zero_grad()
    img = x.reshape(B, H, W, C).permute(0, 3, 1, 2)
    ref = torch.cat([c(p) for c, p in zip(convs, torch.split(img, list(splits), dim=1))], 1)
    if add:
        ref = ref + img
    ref = ref.permute(0, 2, 3, 1).reshape(B, H * W, C)
    (ref * w).sum().backward()
    want = [ref.detach(), wide.grad] + [p.grad for c in convs for p in c.parameters()]
    for a, b in zip(got, want):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 2e-5 * (float(b.abs().max()) + 1e-6) + 1e-6, (a.shape, float((a - b).abs().max()))


@pytest.mark.parametrize("B,H,W,C,heads", [
    (2, 12, 40, 216, 8),        # stage 2 of mpvit_small: Ch = 27
    (3, 6, 20, 288, 8),         # stage 3: Ch = 36 (114 KB of LDS in the backward)
    (2, 24, 80, 128, 8),        # stage 1: Ch = 16
    (1, 48, 160, 64, 8),        # stage 0 at full size: 7680 tokens, Ch = 8
    (2, 3, 5, 64, 8),           # fewer tokens than one staging step
])
def test_factorised_attention_matches_torch(B, H, W, C, heads):
    """Fused factorised attention (column softmax of k over the tokens, [Ch x Ch] contexts, output) and its
    backward against the reference's formulation in torch ops (mpvit.py:354-386) on the same packed qkv."""
    from baseboostdepth_amd import ops
    assert ops.factor_attention_supported(C, heads)
    N, Ch = H * W, C // heads
    scale = Ch ** -0.5
    g = torch.Generator().manual_seed(C + N)
    qkv = (1.5 * torch.randn(B, N, 3 * C, generator=g)).to(DEV).requires_grad_(True)
    convv = torch.randn(B, N, C, generator=g).to(DEV).requires_grad_(True)
    w = torch.randn(B, N, C, generator=g).to(DEV)
    out = ops.factor_attention(qkv, convv, heads, scale)
    (out * w).sum().backward()
    got = [out.detach(), qkv.grad.clone(), convv.grad.clone()]
    qkv.grad = convv.grad = None
    r = qkv.view(B, N, 3, heads, Ch).permute(2, 0, 3, 1, 4)            # [3, B, h, N, Ch] like the reference
    q, k, v = r[0], r[1], r[2]
    kv = torch.einsum("bhnk,bhnv->bhkv", k.softmax(dim=2), v)
    att = torch.einsum("bhnk,bhkv->bhnv", q, kv)
    ref = scale * att + q * convv.view(B, N, heads, Ch).transpose(1, 2)
    ref = ref.transpose(1, 2).reshape(B, N, C)
    (ref * w).sum().backward()
    want = [ref.detach(), qkv.grad, convv.grad]
    for name, a, b in zip(("out", "grad qkv", "grad convv"), got, want):
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)
        assert err < 2e-5, (name, err)
    # the three thirds separately: dk is the smallest and would hide behind dq / dv in a joint maximum
    for t, name in enumerate(("dq", "dk", "dv")):
        a, b = got[1][:, :, t * C:(t + 1) * C], want[1][:, :, t * C:(t + 1) * C]
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)
        assert err < 1e-4, (name, err)


@pytest.mark.parametrize("B,N,C,branch,drop", [(3, 37, 64, True, True), (2, 130, 128, True, False), (2, 61, 216, True, True),
                                               (5, 9, 288, True, True), (1, 7, 512, True, False), (3, 50, 64, False, False),
                                               (2, 33, 216, False, False), (1, 5, 8, True, True), (2, 3, 1024, True, True),
                                               (1, 1, 4, False, False), (12, 1920, 128, True, True)])
def test_residual_droppath_layernorm_matches_torch(B, N, C, branch, drop):
    """csrc/bbd_tokens.hip against the eager formulation of reference networksvit/mpvit.py:397-440
    (x + drop_path(branch), LayerNorm): both outputs and all five gradients, 2e-5 of each tensor's maximum."""
    from baseboostdepth_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + N + C)
    dev = "cuda:0"
    x = (torch.randn(B, N, C, generator=g) * 2 + 0.3).to(dev).requires_grad_(True)
    br = torch.randn(B, N, C, generator=g).to(dev).requires_grad_(True) if branch else None
    mask = None
    if drop:
        mask = (torch.rand(B, generator=g) < 0.6).float().div(0.6).to(dev)
        mask[0] = 0.0
    norm = torch.nn.LayerNorm(C, eps=1e-6).to(dev)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(C, generator=g) * 0.5 + 1.0)
        norm.bias.copy_(torch.randn(C, generator=g) * 0.2)
    wy, wz = torch.randn(B, N, C, generator=g).to(dev), torch.randn(B, N, C, generator=g).to(dev)

    def reference():
        y = x if br is None else x + br * (mask.view(B, 1, 1) if mask is not None else 1.0)
        z = norm(y)
        return y, z

    def fused():
        if br is None:           # C == 216: x routed through the node, so both of its gradients meet inside the kernel
            return ops.layernorm_tokens(x, norm, passthrough=True) if C == 216 else (x, ops.layernorm_tokens(x, norm))
        return ops.residual_layernorm(x, br, mask, norm)

    results = []
    for fn in (reference, fused):
        for t in (x, br, norm.weight, norm.bias):
            if t is not None:
                t.grad = None
        y, z = fn()
        ((y * wy).sum() + (z * wz).sum()).backward()
        results.append([y.detach().clone(), z.detach().clone(), x.grad.clone(), br.grad.clone() if br is not None else None,
                        norm.weight.grad.clone(), norm.bias.grad.clone()])
    names = ["y", "z", "grad x", "grad branch", "grad weight", "grad bias"]
    for name, want, got in zip(names, *results):
        if want is None:
            continue
        err = float((got - want).abs().max())
        assert err <= 2e-5 * float(want.abs().max()) + 1e-7, (name, err, float(want.abs().max()))
    if br is not None:          # residual only (the MLP branch's add): y and its two gradients
        x.grad = br.grad = None
        y = ops.residual_add(x, br, mask)
        (y * wy).sum().backward()
        want = x.detach() + br.detach() * (mask.view(B, 1, 1) if mask is not None else 1.0)
        assert float((y - want).abs().max()) <= 1e-6 * float(want.abs().max())
        assert torch.equal(x.grad, wy)
        assert torch.allclose(br.grad, wy * (mask.view(B, 1, 1) if mask is not None else 1.0), rtol=0, atol=0)


@pytest.mark.parametrize("B,H,W,C,heads,windows", [(2, 6, 10, 64, 8, {3: 2, 5: 3, 7: 3}), (1, 5, 7, 216, 8, {3: 2, 5: 3, 7: 3}),
                                                  (2, 4, 9, 128, 8, {3: 8})])
def test_attention_with_position_encoding_inside_equals_the_two_node_form(B, H, W, C, heads, windows):
    """`factor_attention_crpe` (ConvRelPosEnc of v inside the attention's autograd node, its data gradient added in place to
    gqkv's v third) against conv_v + factor_attention as two nodes: same output, same gqkv, same conv gradients."""
    from baseboostdepth_amd import ops
    from baseboostdepth_amd.networksvit.mpvit import ConvRelPosEnc
    N, Ch = H * W, C // heads
    g = torch.Generator().manual_seed(C + N)
    crpe = ConvRelPosEnc(Ch, heads, windows).to(DEV)
    qkv = (1.5 * torch.randn(B, N, 3 * C, generator=g)).to(DEV).requires_grad_(True)
    w = torch.randn(B, N, C, generator=g).to(DEV)
    res = []
    for fused in (False, True):
        qkv.grad = None
        for p in crpe.parameters():
            p.grad = None
        if fused:
            out = ops.factor_attention_crpe(qkv, (H, W), list(crpe.conv_list), heads, Ch ** -0.5)
        else:
            out = ops.factor_attention(qkv, crpe.conv_v(qkv[:, :, 2 * C:], (H, W)), heads, Ch ** -0.5)
        (out * w).sum().backward()
        res.append([out.detach().clone(), qkv.grad.clone()] + [p.grad.clone() for p in crpe.parameters()])
    for i, (a, b) in enumerate(zip(*res)):
        assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()) + 1e-9, i


def test_shipped_gemm_table_drives_tunableop_without_tuning():
    """`tuning.use_shipped_gemm_db()`: TunableOp on, tuning off, the in-tree table loaded (private copy); a recorded
    shape (stage-4 qkv Linear of mpvit_small at batch 12) computes the same Linear as the library default."""
    import torch.cuda.tunable as tunable
    from baseboostdepth_amd import tuning
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1440, 288, generator=g).to(DEV)
    lin = torch.nn.Linear(288, 864).to(DEV)
    tunable.enable(False)
    want = lin(x)
    path = tuning.use_shipped_gemm_db()
    assert path and "gemm_db_" in path and tunable.is_enabled() and not tunable.tuning_is_enabled()
    got = lin(x)
    torch.cuda.synchronize()
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert tunable.get_filename() == path
    # whether TunableOp will USE the table is reported (it drops a table with other versions in its header silently)
    assert tuning.STATUS["gemm_db"] == path and tuning.STATUS["gemm_db_accepted"] is True, tuning.STATUS


@pytest.mark.parametrize("B,N,cin,cout", [(3, 37, 64, 192), (2, 129, 216, 648), (1, 3, 288, 1152), (12, 480, 216, 216), (2, 50, 64, 62),
                                          (1, 1, 8, 4), (12, 7680, 64, 256)])
def test_linear_on_tokens_with_column_sum_bias_gradient(B, N, cin, cout):
    """ops.linear_tokens (bias gradient by bbd_colsum) against nn.Linear's own autograd; an output width that is not a
    multiple of 4 takes the module itself."""
    from baseboostdepth_amd import ops
    g = torch.Generator().manual_seed(cin + cout + N)
    lin = torch.nn.Linear(cin, cout).to(DEV)
    x = torch.randn(B, N, cin, generator=g).to(DEV).requires_grad_(True)
    w = torch.randn(B, N, cout, generator=g).to(DEV)
    res = []
    for fn in (lambda: lin(x), lambda: ops.linear_tokens(x, lin)):
        x.grad = lin.weight.grad = lin.bias.grad = None
        y = fn()
        (y * w).sum().backward()
        res.append([y.detach().clone(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()])
    for name, a, b in zip(("y", "grad x", "grad weight", "grad bias"), *res):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-6, name


@pytest.mark.parametrize("dim,layers,H,W", [(64, 3, 6, 10), (216, 2, 5, 7)])
def test_encoder_path_with_fused_glue_equals_the_eager_glue(dim, layers, H, W, monkeypatch):
    """One path of a stage (MHCAEncoder: `layers` blocks sharing their position encodings) with every round-3 fusion on -
    residual + LayerNorm passes, attention with the position encoding inside, column-sum bias gradients, grouped
    depth-wise launches, in-kernel accumulation of the shared parameters' gradients - against the same modules with
    BBD_FUSED_TOKEN_GLUE off (eager ATen glue, autograd's own accumulation): output and every parameter gradient."""
    from baseboostdepth_amd import ops
    from baseboostdepth_amd.networksvit.mpvit import MHCAEncoder
    torch.manual_seed(dim + layers)
    enc = MHCAEncoder(dim, num_layers=layers, num_heads=8, mlp_ratio=4, drop_path_list=[0.0] * layers).to(DEV)
    for p in enc.parameters():                      # non-trivial LayerNorm / bias values
        with torch.no_grad():
            p.add_(0.05 * torch.randn_like(p))
    x = torch.randn(2, H * W, dim, device=DEV, requires_grad=True)
    w = torch.randn(2, dim, H, W, device=DEV)
    res = []
    for fused in (False, True):
        monkeypatch.setattr(ops, "FUSED_TOKEN_GLUE", fused)
        x.grad = None
        for p in enc.parameters():
            p.grad = None
        y = enc(x, (H, W))
        (y * w).sum().backward()
        res.append([y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in enc.parameters()])
    names = ["output", "grad x"] + [n for n, _ in enc.named_parameters()]
    for name, a, b in zip(names, *res):
        assert float((a - b).abs().max()) <= 5e-5 * float(a.abs().max()) + 1e-7, name
