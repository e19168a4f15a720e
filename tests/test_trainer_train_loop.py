"""CPU tier (host-port backend): `Trainer.train` writes the reference's checkpoint layout every
`save_frequency` epochs without any extra flag, and `--load_weights_folder weights_N` resumes at epoch N+1
with the LR schedule fast-forwarded (reference trainer.py:168-193, 774-829)."""
import json
import os

import torch

from host_port import HostPortBackend
from baseboostdepth_amd import Trainer, synthetic
from baseboostdepth_amd.options import MonodepthOptions

H, W, B = 64, 128, 2


def _opts(tmp, extra=""):
    return MonodepthOptions().parse(("--no_cuda --weights_init scratch --height %d --width %d --batch_size %d "
                                     "--log_dir %s --model_name t --num_epochs 1 %s" % (H, W, B, tmp, extra)).split())


def _loader(opts, steps=2):
    return lambda epoch: synthetic.synthetic_loader(B, steps, H, W, opts.scales, device="cpu", seed=5, epoch=epoch)


def test_train_saves_checkpoints_and_resumes(tmp_path):
    torch.manual_seed(0)
    torch.set_num_threads(4)
    opts = _opts(str(tmp_path))
    tr = Trainer(opts, backend=HostPortBackend())
    tr.train(_loader(opts))
    folder = os.path.join(str(tmp_path), "t", "models", "weights_0")
    for name in ("encoder", "depth", "pose_encoder", "pose", "adam"):
        assert os.path.isfile(os.path.join(folder, name + ".pth")), name
    assert json.load(open(os.path.join(str(tmp_path), "t", "models", "opt.json")))["height"] == H
    enc = torch.load(os.path.join(folder, "encoder.pth"))
    assert enc["height"] == H and enc["width"] == W           # consumers read the resolution from here
    assert tr.step == 2 and tr.epoch == 0

    # resume: weights_0 -> epoch 1; scheduler fast-forwarded by one step; weights identical to the saved ones
    opts2 = _opts(str(tmp_path), "--load_weights_folder %s" % folder)
    opts2.num_epochs = 2
    tr2 = Trainer(opts2, backend=HostPortBackend())
    assert tr2.resume_epoch() == 1
    for k, v in tr.models["depth"].state_dict().items():
        assert torch.equal(v, tr2.models["depth"].state_dict()[k]), k
    seen = []
    tr2.train(lambda epoch: (seen.append(epoch), _loader(opts2, 1)(epoch))[1], steps_per_epoch=2)
    assert seen == [1] and tr2.epoch == 1
    assert tr2.step == 1 * 2 + 1                                # fast-forwarded counter + the one step taken
    assert tr2.model_lr_scheduler.last_epoch == 2               # one fast-forward + run_epoch's own step
    assert os.path.isdir(os.path.join(str(tmp_path), "t", "models", "weights_1"))


def test_resume_epoch_parsing():
    tr = Trainer.__new__(Trainer)
    tr.opt = MonodepthOptions().parse([])
    assert tr.resume_epoch() == 0
    for folder, want in (("/x/weights_9", 10), ("/x/weights_best", 10), ("/x/weights_3_1200/", 1201), ("w_0", 1)):
        tr.opt.load_weights_folder = folder
        assert tr.resume_epoch() == want, folder


def test_vit_process_batch_on_cpu_host_port():
    """--ViT wiring on the CPU tier: MonoViT networks + the host port of the loss kernels, one step."""
    torch.manual_seed(0)
    torch.set_num_threads(4)
    o = MonodepthOptions().parse(("--ViT --no_cuda --weights_init scratch --height %d --width %d --batch_size %d"
                                  % (H, W, B)).split())
    tr = Trainer(o, backend=HostPortBackend())
    tr.set_train()
    batch = synthetic.synthetic_batch([1] * B, H, W, o.scales, device="cpu", seed=1)
    before = [p.detach().clone() for p in tr.models["encoder"].parameters()]
    outputs, losses = tr.train_step(batch)
    assert torch.isfinite(losses["loss"]) and outputs[("disp", 0)].shape == (B, 1, H, W)
    assert any(not torch.equal(a, b) for a, b in zip(before, tr.models["encoder"].parameters()))


def test_capture_warm_up_steps_issue_no_collective():
    """a graph-cache miss is a per-rank event, so the warm-up steps of a capture must not exchange
    gradients (a rank that warms up would otherwise issue 3 collectives more than a rank that replays)."""
    torch.manual_seed(0)
    torch.set_num_threads(4)
    o = MonodepthOptions().parse(("--no_cuda --weights_init scratch --height %d --width %d --batch_size %d"
                                  % (H, W, B)).split())
    tr = Trainer(o, backend=HostPortBackend())
    tr.set_train()
    calls = []
    tr.grad_sync = lambda: calls.append(1)
    batch = synthetic.synthetic_batch([1] * B, H, W, o.scales, device="cpu", seed=1)
    tr._local_only = True
    tr._eager_step(dict(batch))
    assert calls == []
    tr._local_only = False
    tr._eager_step(dict(batch))
    assert calls == [1]
    assert Trainer.max_graphs >= 1 and tr.max_graphs >= 1
