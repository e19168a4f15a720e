"""Command-line surface: the reference's flag names/defaults parse, out-of-scope zoos are refused, and the
parsed namespace is sufficient to construct the Trainer (CPU construct only - the step itself needs the GPU)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from baseboostdepth_amd.options import MonodepthOptions  # noqa: E402


def test_reference_defaults():
    o = MonodepthOptions().parse([])
    assert (o.height, o.width, o.batch_size, o.scales, o.frame_ids) == (192, 640, 12, [0, 1, 2, 3], [0, -1, 1])
    assert (o.min_depth, o.max_depth, o.disparity_smoothness, o.learning_rate) == (0.1, 100.0, 1e-3, 1e-4)
    assert (o.num_epochs, o.num_layers, o.pose_error, o.log_frequency, o.pytorch_random_seed) == (20, 18, 1, 250, 42)
    assert o.models_to_load == ["encoder", "depth", "pose_encoder", "pose"] and o.load_weights_folder == "None"


def test_boosted_command_line_and_refusals():
    o = MonodepthOptions().parse("--rand --trimin --decomp --incremental_skip --partial_skip --naive_mix --kt "
                                 "--pose_error 5.5 --weights_init scratch --batch_size 4".split())
    assert o.rand and o.trimin and o.decomp and o.incremental_skip and o.partial_skip and o.pose_error == 5.5
    for flag in ("--SQL", "--CA_depth", "--DIFFNet", "--SYNS_eval"):
        with pytest.raises(SystemExit):
            MonodepthOptions().parse([flag])


def test_namespace_constructs_trainer_on_cpu():
    from baseboostdepth_amd import Trainer
    o = MonodepthOptions().parse("--no_cuda --weights_init scratch --height 64 --width 128 --batch_size 2".split())
    tr = Trainer(o)
    assert set(tr.models) == {"encoder", "depth", "pose_encoder", "pose"}
    from baseboostdepth_amd._lib import BbdError
    from baseboostdepth_amd import synthetic
    batch = synthetic.synthetic_batch([1, 1], 64, 128, o.scales, device="cpu", seed=0)
    with pytest.raises(BbdError):                       # no CPU fallback for the hot path
        tr.process_batch(batch)


def test_vit_flag_builds_monovit_with_two_lr_groups():
    """--ViT (reference trainer.py:52-58, 106-109): MPViT-small + HR decoder, AdamW, encoder in its own
    5e-5 group and NOT in parameters_to_train."""
    import torch
    from baseboostdepth_amd import Trainer, networksvit
    o = MonodepthOptions().parse("--ViT --no_cuda --weights_init scratch --height 64 --width 128 --batch_size 2".split())
    tr = Trainer(o)
    assert isinstance(tr.models["encoder"], networksvit.MPViT) and isinstance(tr.models["depth"], networksvit.DepthDecoder)
    assert tr.models["encoder"].num_ch_enc == [64, 128, 216, 288, 288]
    assert isinstance(tr.model_optimizer, torch.optim.AdamW)
    g0, g1 = tr.model_optimizer.param_groups
    assert (g0["lr"], g1["lr"]) == (1e-4, 5e-5)
    enc_ids = {id(p) for p in tr.models["encoder"].parameters()}
    assert {id(p) for p in g1["params"]} == enc_ids and not (enc_ids & {id(p) for p in tr.parameters_to_train})
    assert len(tr.optimizer_parameters) == len(g0["params"]) + len(g1["params"])
    assert len(tr.gradient_free_parameters()) == 2 + 8          # pose encoder fc + the decoder's unused X_0j_Conv_0
