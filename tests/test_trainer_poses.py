"""CPU tier: Trainer.predict_poses (three pose modes), the network heads, and the whole
process_batch-style chain against golden vectors from the reference (host-port backend)."""
import os

import numpy as np
import pytest
import torch

from fake_nets import FakePoseEncoder, fill_deterministic
from fused_runner import bare_trainer, make_opt, compare_with_golden
from golden_io import Case, POSE_CASES, GOLDEN_DIR
from host_port import HostPortBackend
from pose_checks import check_pose_case
from baseboostdepth_amd import networks


@pytest.fixture(scope="module")
def backend():
    return HostPortBackend()


def _fkey(f):
    return "s" if f == "s" else str(int(f))


@pytest.mark.parametrize("name", POSE_CASES)
def test_predict_poses_and_step(name, backend):
    check_pose_case(name, backend, "cpu")


def test_decoders_match_reference_and_state_dict_keys():
    z = np.load(os.path.join(GOLDEN_DIR, "layers.npz"))
    num_ch_enc = np.array([64, 64, 128, 256, 512])
    dec = fill_deterministic(networks.DepthDecoder(num_ch_enc, [0, 1, 2, 3]), 0.3)
    assert sorted(dec.state_dict().keys()) == list(z["dec/keys"])
    feats = [torch.from_numpy(z["dec/feat/%d" % i]) for i in range(5)]
    out = dec(feats)
    for s in range(4):
        d = float((out[("disp", s)] - torch.from_numpy(z["dec/disp/%d" % s])).abs().max())
        assert d < 1e-4, (s, d)      # fp32 conv stacks with O(1) closed-form weights
    pose = fill_deterministic(networks.PoseDecoder(num_ch_enc, 1, 2), 0.4)
    assert sorted(pose.state_dict().keys()) == list(z["pose/keys"])
    aa, t = pose([feats])
    assert torch.allclose(aa, torch.from_numpy(z["pose/aa"]), atol=1e-7)
    assert torch.allclose(t, torch.from_numpy(z["pose/t"]), atol=1e-7)
    assert sum(p.numel() for p in dec.parameters()) == 3152724
    assert sum(p.numel() for p in pose.parameters()) == 1314572


def test_resnet_encoder_layout():
    """torchvision-identical keys / parameter count (the reference's encoder.pth must load)."""
    enc = networks.ResnetEncoder(18, False)
    sd = enc.state_dict()
    assert sum(p.numel() for p in enc.parameters()) == 11689512
    for k in ("encoder.conv1.weight", "encoder.bn1.running_mean", "encoder.layer1.0.conv1.weight",
              "encoder.layer2.0.downsample.0.weight", "encoder.layer2.0.downsample.1.num_batches_tracked",
              "encoder.layer4.1.bn2.bias", "encoder.fc.weight", "encoder.fc.bias"):
        assert k in sd, k
    assert sd["encoder.conv1.weight"].shape == (64, 3, 7, 7)
    feats = enc(torch.rand(1, 3, 64, 96))
    assert [f.shape[1] for f in feats] == [64, 64, 128, 256, 512]
    assert [f.shape[2] for f in feats] == [32, 16, 8, 4, 2]
    pe = networks.ResnetEncoder(18, False, num_input_images=2)
    assert pe.state_dict()["encoder.conv1.weight"].shape == (64, 6, 7, 7)
    e50 = networks.ResnetEncoder(50, False)
    assert list(e50.num_ch_enc) == [64, 256, 512, 1024, 2048]
    with pytest.raises(RuntimeError):
        networks.ResnetEncoder(18, True)


def test_transformation_from_parameters_matches_golden():
    from baseboostdepth_amd.layers import transformation_from_parameters as tfp, disp_to_depth
    z = np.load(os.path.join(GOLDEN_DIR, "layers.npz"))
    aa, t = torch.from_numpy(z["tfp/aa"]), torch.from_numpy(z["tfp/t"])
    assert torch.equal(tfp(aa, t), torch.from_numpy(z["tfp/M"]))
    assert torch.equal(tfp(aa, t, invert=True), torch.from_numpy(z["tfp/Minv"]))
    sd, dp = disp_to_depth(torch.from_numpy(z["d2d/disp"]), 0.1, 100.0)
    assert torch.equal(dp, torch.from_numpy(z["d2d/depth"]))


def test_custom_collate_matches_reference():
    """Same keys, shapes and values as the reference's Trainer.custom_collate on ragged items."""
    import refshim
    if not refshim.reference_available():
        pytest.skip("reference tree not present on this machine")
    import types
    from make_golden import make_item, make_ref_trainer, make_opt as ref_opt
    from baseboostdepth_amd.trainer import Trainer
    rt, rl, rn = refshim.import_reference()
    for ms in ([3, 1, 0, 5], [0, 0], [7, 2], [1, 1, 1]):
        gen = torch.Generator().manual_seed(1)
        items = [make_item(gen, m, 32, 64, [0, 1, 2, 3], max(max(ms), 1), 0.3) for m in ms]
        ref = make_ref_trainer(rt, rl, ref_opt(32, 64, len(ms), [0, 1, 2, 3], True, True)).custom_collate(items)
        mine = Trainer.__new__(Trainer)
        mine.opt = types.SimpleNamespace(scales=[0, 1, 2, 3])
        got = mine.custom_collate(items)
        assert set(got.keys()) == set(ref.keys())
        for k in ref:
            if torch.is_tensor(ref[k]):
                assert torch.equal(got[k], ref[k]), k
            else:
                assert got[k] == ref[k], k


def test_depth_metrics_oracle_matches_live_reference():
    """Validation metrics (resize, Garg crop, median scaling, 7 errors): oracle vs the live reference
    (the device kernel is checked against the same oracle and vectors in test_gpu_eval.py)."""
    import refshim
    if not refshim.reference_available():
        pytest.skip("reference tree not present on this machine")
    from oracle import eval_ref
    rt, rl, rn = refshim.import_reference()
    gen = torch.Generator().manual_seed(2)
    gt = (torch.rand(375, 1242, generator=gen) * 90).numpy()
    gt[gt < 20] = 0                                              # sparse LiDAR-like ground truth
    depth = 0.5 + 60 * torch.rand(1, 1, 192, 640, generator=gen)
    names = ["de/abs_rel", "de/sq_rel", "de/rms", "de/log_rms", "da/a1", "da/a2", "da/a3"]
    ref = rt.Trainer.__new__(rt.Trainer)
    ref.device, ref.depth_metric_names, ref.gt_depths = torch.device("cpu"), names, [gt]
    want = {}
    ref.compute_depth_losses({("depth", 0, 0): depth.clone()}, want, 0)
    got = eval_ref.compute_depth_losses_ref(depth.clone(), gt)["metrics"]
    for i, k in enumerate(names):
        assert abs(got[i] - float(want[k])) < 1e-6 * max(1.0, abs(float(want[k]))), (k, got[i], want[k])
