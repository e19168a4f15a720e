"""GPU tier of the loader: batches collated on the device equal the reference's per-item Pillow pipeline
+ custom_collate bit for bit, and feed Trainer.train_step directly."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import image_checks  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("epoch,trimin,scales", [(3, True, [0, 1, 2, 3]), (12, True, [0])])
def test_collated_batches_equal_reference_pipeline(tmp_path, epoch, trimin, scales):
    image_checks.check_loader_batches(tmp_path, "cuda:0", None, epoch, trimin, scales)


def test_train_step_on_loader_batches(tmp_path):
    from test_gpu_trainer import make_opt
    from baseboostdepth_amd import datasets
    from baseboostdepth_amd.trainer import Trainer
    H, W, B = 96, 320, 4
    lines = image_checks.make_kitti_tree(str(tmp_path))
    opt = make_opt(H, W, B, [0, 1, 2, 3], True)
    tr = Trainer(opt)
    ds = datasets.KITTIRAWDataset(lines, 2, H, W, kt_path=str(tmp_path), rand=True, is_train=True, scales=opt.scales,
                                  kt=True, naive_mix=True, trimin=True, seed=5)
    loader = datasets.DeviceLoader(ds, B, datasets.DeviceCollate(H, W, opt.scales, "cuda:0"), num_workers=4, seed=2)
    losses = []
    for step, batch in enumerate(loader):
        assert batch[("color", 0, 0)].is_cuda
        _, l = tr.train_step(batch)
        losses.append(float(l["loss"].detach()))
        if step == 2:
            break
    assert all(torch.isfinite(torch.tensor(losses))) and len(losses) == 3
