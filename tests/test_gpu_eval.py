"""GPU parity of the validation-metrics kernel (bbd_depth_metrics) through the C ABI:
against vectors captured from the live reference, and against the oracle on fresh inputs."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import eval_ref  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden", "eval_cases.npz")
CASES = ["kitti_375", "kitti_370", "full_res", "small_out", "even_count", "downsample"]


def _close(got, want, rtol=3e-5):
    np.testing.assert_allclose(np.asarray(got, np.float64), np.asarray(want, np.float64), rtol=rtol, atol=2e-6)


def test_metrics_match_reference_vectors_batched_launch():
    """All six golden cases; cases with equal prediction size go through ONE launch (ragged GT)."""
    from baseboostdepth_amd.evaluation import GroundTruthSet, depth_metrics
    v = np.load(GOLDEN)
    dev = torch.device("cuda:0")
    gts = GroundTruthSet([v[c + "/gt"] for c in CASES], dev)
    by_shape = {}
    for i, c in enumerate(CASES):
        by_shape.setdefault(v[c + "/pred"].shape, []).append(i)
    assert max(len(g) for g in by_shape.values()) >= 2
    for shape, idxs in by_shape.items():
        pred = torch.from_numpy(np.concatenate([v[CASES[i] + "/pred"] for i in idxs])).to(dev)
        rows = depth_metrics(pred, gts, idxs).cpu().numpy()
        for r, i in zip(rows, idxs):
            _close(r[:7], v[CASES[i] + "/metrics"])
            want = eval_ref.compute_depth_losses_ref(torch.from_numpy(v[CASES[i] + "/pred"]), v[CASES[i] + "/gt"])
            assert int(r[10]) == want["count"]
            # medians are order statistics of bit-exact resampled values: equal unless this host's
            # ATen differs from the kernel by an ulp - compare to float rounding
            _close([r[8], r[9], r[7]], [want["median_gt"], want["median_pred"], want["ratio"]], rtol=1e-6)


@pytest.mark.parametrize("seed,h,w,gh,gw", [(1, 192, 640, 375, 1242), (2, 96, 320, 376, 1241), (3, 48, 160, 40, 100)])
def test_metrics_match_oracle_both_modes(seed, h, w, gh, gw):
    from baseboostdepth_amd.evaluation import GroundTruthSet, depth_metrics
    g = torch.Generator().manual_seed(seed)
    depth = (2.0 + 50 * torch.rand(2, 1, h, w, generator=g)).float()
    gt = [(torch.rand(gh, gw, generator=g) * 95 * (torch.rand(gh, gw, generator=g) < 0.1)).numpy().astype(np.float32)
          for _ in range(2)]
    dev = torch.device("cuda:0")
    gts = GroundTruthSet(gt, dev)
    rows = depth_metrics(depth.to(dev), gts, [0, 1]).cpu().numpy()
    for b in range(2):
        want = eval_ref.compute_depth_losses_ref(depth[b:b + 1], gt[b])
        _close(rows[b, :7], want["metrics"])
        assert int(rows[b, 10]) == want["count"]
    # evaluate_depth.py mode: disparity in, cv2-style resize, np.median, optional stereo scale
    disp = (1.0 / depth).contiguous()
    rows = depth_metrics(disp.to(dev), gts, [0, 1], pred_is_disp=True, median="numpy").cpu().numpy()
    for b in range(2):
        want = eval_ref.evaluate_image_ref(disp[b, 0].numpy(), gt[b])
        _close(rows[b, :7], want["metrics"])
        _close(rows[b, 7], want["ratio"], rtol=2e-6)
    rows = depth_metrics(disp.to(dev), gts, [1, 0], pred_is_disp=True, median="numpy", median_scaling=False,
                         scale_factor=5.4).cpu().numpy()
    for b, gi in enumerate([1, 0]):
        want = eval_ref.evaluate_image_ref(disp[b, 0].numpy(), gt[gi], median_scaling=False, scale_factor=5.4)
        _close(rows[b, :7], want["metrics"])


def test_median_select_is_exact_and_empty_mask_gives_nan():
    """The radix select returns exactly torch.median / np.median of the masked values."""
    from baseboostdepth_amd.evaluation import GroundTruthSet, depth_metrics
    g = torch.Generator().manual_seed(9)
    dev = torch.device("cuda:0")
    for n_valid_parity in (0, 1):
        gh, gw = 64, 96
        gt = (torch.rand(gh, gw, generator=g) * 70 + 1).numpy().astype(np.float32)
        if n_valid_parity:
            gt[30, 40] = 0.0                                  # flips the parity of the valid count
        gts = GroundTruthSet([gt, np.zeros((gh, gw), np.float32)], dev, crop=False)
        pred = (torch.rand(2, gh, gw, generator=g) * 60 + 1).float()      # same size: resize is identity
        r = depth_metrics(pred.to(dev), gts, [0, 1]).cpu().numpy()
        m = gt > 1e-3
        assert r[0, 8] == torch.median(torch.from_numpy(gt[m])).item()
        assert r[0, 9] == torch.median(pred[0][torch.from_numpy(m)]).item()
        assert np.isnan(r[1, :7]).all() and r[1, 10] == 0
        r = depth_metrics((1 / pred).to(dev), gts, [0, 1], pred_is_disp=True, median="numpy").cpu().numpy()
        assert r[0, 8] == np.median(gt[m])
        assert r[0, 9] == np.median((1 / (1 / pred[0].numpy()))[m])


def test_trainer_val_loop_single_sync():
    """Trainer.val over a synthetic 'split': equals per-image oracle means; compute_depth_losses keeps
    the reference's (outputs, losses, idx, accumulate) calling convention."""
    from test_gpu_trainer import make_opt
    from baseboostdepth_amd.trainer import Trainer
    tr = Trainer(make_opt(64, 128, 2, [0, 1, 2, 3], False))
    g = torch.Generator().manual_seed(4)
    gts = [(torch.rand(90, 200, generator=g) * 60 * (torch.rand(90, 200, generator=g) < 0.3)).numpy().astype(np.float32)
           for _ in range(4)]
    tr.gt_depths = gts                                     # the reference's attribute, packed lazily
    batches = [{("color", 0, 0): torch.rand(2, 3, 64, 128, generator=g)} for _ in range(2)]
    result = tr.val(batches)
    tr.set_eval()
    want = np.zeros(7)
    with torch.no_grad():
        for bi, b in enumerate(batches):
            out, _ = tr.process_batch(dict(b), is_train=False)
            for r in range(2):
                want += eval_ref.compute_depth_losses_ref(out["depth", 0, 0][r:r + 1].cpu(), gts[2 * bi + r])["metrics"]
    _close([result[k] for k in tr.depth_metric_names], want / 4, rtol=1e-4)
    losses = {}
    tr.compute_depth_losses(out, losses, [2, 3])
    assert set(losses) == set(tr.depth_metric_names) and losses["de/abs_rel"].is_cuda


def test_evaluate_split_matches_per_image_oracle(tmp_path, capsys):
    """evaluation.evaluate (evaluate_depth.py drop-in) on a synthetic split: weights folder written by
    Trainer.save_model, frames from the device loader, metrics == oracle per image on the same disparities."""
    import types
    import image_checks
    from test_gpu_trainer import make_opt
    from baseboostdepth_amd import datasets, evaluation
    from baseboostdepth_amd.layers import disp_to_depth
    from baseboostdepth_amd.trainer import Trainer
    H, W = 96, 320
    lines = image_checks.make_kitti_tree(str(tmp_path / "kitti"), frames=20)
    test_files = [l.rsplit(" ", 2)[0] for l in lines][:10]
    split = tmp_path / "splits" / "eigen"
    split.mkdir(parents=True)
    (split / "test_files.txt").write_text("\n".join(test_files) + "\n")
    g = torch.Generator().manual_seed(6)
    gts = np.empty(10, dtype=object)
    for i in range(10):
        gh, gw = (375, 1242) if i % 2 else (370, 1226)
        gts[i] = (torch.rand(gh, gw, generator=g) * 85 * (torch.rand(gh, gw, generator=g) < 0.06)).numpy().astype(np.float32)
    np.savez_compressed(split / "gt_depths.npz", data=gts)
    opt = make_opt(H, W, 2, [0, 1, 2, 3], False)
    opt.log_dir, opt.model_name = str(tmp_path), "m"
    tr = Trainer(opt)
    tr.save_model("w")
    eopt = types.SimpleNamespace(eval_mono=True, eval_stereo=False, cuda=0, num_layers=18, kt_path=str(tmp_path / "kitti"),
                                 load_weights_folder=str(tmp_path / "m" / "models" / "weights_w"), splits_dir=str(tmp_path / "splits"),
                                 eval_split="eigen", disable_median_scaling=False, pred_depth_scale_factor=1, min_depth=0.1,
                                 max_depth=100.0, num_workers=2, height=H, width=W)
    mean_errors, ratios = evaluation.evaluate(eopt, batch_size=4)
    assert "abs_rel" in capsys.readouterr().out and ratios.shape == (10,)
    # oracle on the disparities the same weights produce
    ds = datasets.KITTIRAWDataset(test_files, 0, H, W, kt_path=str(tmp_path / "kitti"), is_train=False, kt=True, naive_mix=True)
    coll = datasets.DeviceCollate(H, W, [0], "cuda:0")
    tr.set_eval()
    want = []
    with torch.no_grad():
        for i in range(10):
            x = coll([ds[i]])[("color", 0, 0)]
            disp, _ = disp_to_depth(tr.models["depth"](tr.models["encoder"](x))[("disp", 0)], 0.1, 100.0)
            want.append(eval_ref.evaluate_image_ref(disp[0, 0].cpu().numpy(), gts[i]))
    _close(mean_errors, np.mean([w["metrics"] for w in want], 0), rtol=2e-4)
    _close(ratios, [w["ratio"] for w in want], rtol=2e-4)
    eopt.eval_mono, eopt.eval_stereo = False, True
    mean_s, none = evaluation.evaluate(eopt, batch_size=5)
    assert none is None
    # (the stereo arithmetic itself is checked in test_metrics_match_oracle_both_modes)
    assert np.isfinite(mean_s).all() and not np.allclose(mean_s, mean_errors)
