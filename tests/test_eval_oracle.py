"""Oracle for the validation metrics (oracle/eval_ref.py) against vectors captured from the live
reference's Trainer.compute_depth_losses (tools/make_golden_eval.py), CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import eval_ref  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "eval_cases.npz")
CASES = ["kitti_375", "kitti_370", "full_res", "small_out", "even_count", "downsample"]


@pytest.fixture(scope="module")
def vectors():
    return np.load(GOLDEN)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_vectors(vectors, name):
    pred = torch.from_numpy(vectors[name + "/pred"])
    got = eval_ref.compute_depth_losses_ref(pred, vectors[name + "/gt"])
    # same ops as the reference on the same host class; allow for cross-CPU differences of ATen kernels
    np.testing.assert_allclose(got["metrics"], vectors[name + "/metrics"], rtol=2e-5, atol=1e-7)
    assert got["count"] > 100


def test_cv2_resize_restatement_properties():
    """cv2.resize is absent here (parity unpinned); check the restatement's defining properties:
    identity at equal size, exact on constants, half-pixel symmetric, agrees with ATen's
    align_corners=False bilinear to float rounding on up-sampling."""
    g = np.random.default_rng(0)
    img = g.random((24, 40), dtype=np.float32)
    np.testing.assert_array_equal(eval_ref.cv2_resize_linear_ref(img, 40, 24), img)
    const = np.full((7, 9), 0.37, np.float32)
    np.testing.assert_allclose(eval_ref.cv2_resize_linear_ref(const, 31, 17), 0.37, rtol=3e-7)
    up = eval_ref.cv2_resize_linear_ref(img, 100, 60)
    np.testing.assert_allclose(up[:, ::-1], eval_ref.cv2_resize_linear_ref(img[:, ::-1].copy(), 100, 60), rtol=1e-5, atol=3e-6)
    aten = torch.nn.functional.interpolate(torch.from_numpy(img)[None, None], size=(60, 100), mode="bilinear",
                                           align_corners=False)[0, 0].numpy()
    np.testing.assert_allclose(up, aten, rtol=1e-4, atol=5e-6)


def test_evaluate_image_ref_runs_and_is_scale_invariant():
    v = np.load(GOLDEN)
    disp = (1.0 / v["kitti_375/pred"][0, 0]).astype(np.float32)
    a = eval_ref.evaluate_image_ref(disp, v["kitti_375/gt"])
    b = eval_ref.evaluate_image_ref(disp * np.float32(2.0), v["kitti_375/gt"])
    np.testing.assert_allclose(a["metrics"], b["metrics"], rtol=1e-5)     # median scaling removes scale
    np.testing.assert_allclose(b["ratio"], 2 * a["ratio"], rtol=1e-6)
