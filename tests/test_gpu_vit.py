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
