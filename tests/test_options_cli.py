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
    for flag in ("--ViT", "--SQL", "--CA_depth", "--DIFFNet", "--SYNS_eval"):
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
