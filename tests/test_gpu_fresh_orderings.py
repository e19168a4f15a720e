"""GPU tier: the regime a real `--rand` epoch runs in - a NEW batch ordering every step (the reference redraws every
sample's frame set per item, mono_dataset.py:87-109, restacks per batch, trainer.py:867-886, and overwrites `frame_ids`
per batch, trainer.py:250).  Every integer table of a step must reach the GPU in ONE asynchronous upload, nothing in a
step may synchronise the training thread with the device, and the canonical sample order / the capture policy of the
step graphs must not change what is computed."""
import warnings

import pytest
import torch

from test_gpu_trainer import _deterministic_convolutions, make_opt

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _boosted_batch(ms, H, W, scales, seed, cutt):
    from baseboostdepth_amd.synthetic import synthetic_batch
    b = synthetic_batch(ms, H, W, scales, device=DEV, seed=seed)
    b.pop("noise")
    b["cutt"] = torch.tensor(cutt)
    return b


@pytest.mark.parametrize("scales,cutt,orderings", [
    ([0], 1.35, [[7, 5, 4, 3], [6, 6, 2, 1], [7, 7, 7, 1], [5, 4, 3, 3], [7, 3, 2, 2]]),        # epoch >= 10: incremental + partial
    ([0, 1, 2, 3], 0.3, [[2, 1, 1, 0], [2, 2, 1, 1], [1, 1, 1, 1], [2, 2, 2, 0], [2, 1, 0, 0]]),  # early curriculum
])
def test_five_orderings_five_steps_one_upload_each_and_no_synchronising_call(scales, cutt, orderings):
    """5 eager steps with 5 different orderings: after the first step (which may place shape-only constants) every
    step is ONE packed table upload, no single uploads, and torch's sync debug mode (which raises on a pageable
    host-to-device copy, `.item()` of a device tensor, ...) stays silent."""
    from baseboostdepth_amd import steptables
    from baseboostdepth_amd.trainer import Trainer
    H, W, B = 96, 160, 4
    torch.manual_seed(0)
    opt = make_opt(H, W, B, [0, 1, 2, 3], True)
    tr = Trainer(opt)
    tr.opt.scales = list(scales)
    tr.set_train()
    batches = [_boosted_batch(ms, H, W, scales, 40 + i, cutt) for i, ms in enumerate(orderings)]
    # a sixth ordering first: MIOpen solutions, Adam state, the shape-only constants of the loss nodes
    tr.train_step(_boosted_batch([7, 6, 5, 4] if cutt > 0.5 else [2, 1, 1, 1], H, W, scales, 39, cutt))
    torch.cuda.synchronize()
    losses = []
    for b in batches:
        steptables.reset_stats()
        torch.cuda.set_sync_debug_mode("error")
        try:
            _, l = tr.train_step(b)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        assert steptables.STATS["packed_uploads"] == 1, steptables.STATS
        assert steptables.STATS["single_uploads"] == 0, steptables.STATS
        losses.append(l["loss"])
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(x)) for x in losses)
    # the same signatures again: nothing to build, nothing to upload
    steptables.reset_stats()
    for b, ms in zip(batches, orderings):
        tr.train_step(_boosted_batch(ms, H, W, scales, 77, cutt))
    assert steptables.STATS["packed_uploads"] == 0 and steptables.STATS["builds"] == 0, steptables.STATS


def test_canonical_order_computes_the_callers_batch():
    """`train_step` takes the samples largest-offset-first; on the caller's (shuffled) order `process_batch` gives the
    same loss (sums in another order: 1e-5) and the same per-sample disparities, row for row through `batch_order`.
    (Without --partial_skip: the reference applies that rule by ROW NUMBER of the frame's stack, trainer.py:415-418, so
    its outcome depends on the batch order itself - for the reference too; on the canonical order, where a frame's owners
    are a prefix of the batch, the rule meets the sample it was computed for.  The pose fixtures pin the quirk.)"""
    from baseboostdepth_amd.trainer import Trainer
    from baseboostdepth_amd.synthetic import synthetic_batch
    H, W = 96, 160
    ms = [1, 3, 1, 2, 5, 2]          # (epoch >= 10 regime: every sample has temporal frames)
    torch.manual_seed(0)
    opt = make_opt(H, W, len(ms), [0, 1, 2, 3], True)
    opt.partial_skip = False
    tr = Trainer(opt)
    tr.opt.scales = [0]
    tr.set_eval()                       # BatchNorm on running statistics: per-sample results do not depend on the batch
    inputs = synthetic_batch(ms, H, W, [0], device=DEV, seed=3)
    inputs["cutt"] = torch.tensor(1.35)
    tr.opt.frame_ids = sorted(inputs["frames"], key=lambda it: float("inf") if isinstance(it, str) else abs(it))
    plain = dict(inputs)
    out_a, loss_a = tr.process_batch(plain)
    canon = dict(inputs)
    perm = tr.canonicalize(canon)
    assert perm == [4, 1, 3, 5, 0, 2] and canon["batch_order"] == perm
    assert canon["ordering"] == [[0, 5, -5], [0, 3, -3], [0, 2, -2], [0, 2, -2], [0, 1, -1], [0, 1, -1]]
    out_b, loss_b = tr.process_batch(canon)
    la, lb = float(loss_a["loss"].detach()), float(loss_b["loss"].detach())
    assert abs(la - lb) <= 1e-5 * abs(la), (la, lb)
    for b, p in enumerate(perm):
        assert torch.allclose(out_b[("disp", 0)][b], out_a[("disp", 0)][p], rtol=0, atol=1e-6)
        assert torch.equal(canon[("color", 0, 0)][b], inputs[("color", 0, 0)][p])
    # an already canonical batch is left alone
    assert tr.canonicalize(canon) is None


def test_signatures_are_captured_on_their_second_sighting(monkeypatch):
    """`graph_capture_after = 1`: a signature's first batch runs eagerly, its second is captured, later ones replay -
    and the parameters after the sequence EQUAL the all-eager loop's, bit for bit (deterministic convolution solvers; the
    identity noise handed in with the batch - drawn inside the step it comes from different generator offsets under
    capture, and this recipe's near-ties between identity candidates turn 1e-5 of noise into different arg-mins)."""
    from baseboostdepth_amd.trainer import Trainer
    _deterministic_convolutions(monkeypatch)
    H, W, B = 96, 160, 4
    A, Bm = [2, 1, 1, 0], [2, 2, 1, 1]
    seq = [A, Bm, A, A, Bm, A]

    def run(graph):
        opt = make_opt(H, W, B, [0, 1, 2, 3], True)
        opt.step_graph, opt.graph_capture_after = graph, 1
        torch.manual_seed(5)
        tr = Trainer(opt)
        tr.set_train()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for i, ms in enumerate(seq):
                batch = _boosted_batch(ms, H, W, [0, 1, 2, 3], 60 + i, 0.3)
                batch["noise"] = torch.randn(B, H, W, device=DEV, generator=torch.Generator(device=DEV).manual_seed(100 + i)) * 1e-5
                _, losses = tr.train_step(batch)
        torch.cuda.synchronize()
        return torch.cat([p.detach().flatten() for p in tr.parameters_to_train]), float(losses["loss"].detach()), tr

    pe, le, _ = run(False)
    pg, lg, trg = run(True)
    assert trg.graph_stats == {"eager": 2, "captures": 2, "replays": 4}, trg.graph_stats
    assert trg.step == len(seq) and len(trg._graphs) == 2
    assert le == lg and torch.equal(pe, pg)
