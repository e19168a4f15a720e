"""Pin oracle/ (the CPU restatement) to golden vectors captured from the live reference."""
import numpy as np
import pytest
import torch

from golden_io import Case, DIRECT_CASES
from oracle import hotpath_ref as O


@pytest.fixture(scope="module")
def layers_z():
    import os
    from golden_io import GOLDEN_DIR
    return np.load(os.path.join(GOLDEN_DIR, "layers.npz"))


def t(z, k):
    return torch.from_numpy(z[k])


def test_disp_to_depth_and_upsample(layers_z):
    z = layers_z
    sd, depth = O.disp_to_depth(t(z, "d2d/disp"))
    assert torch.equal(sd, t(z, "d2d/scaled")) and torch.equal(depth, t(z, "d2d/depth"))
    for s in (1, 2):
        H, W = z["d2d/disp"].shape[-2:]
        assert torch.equal(O.upsample_disp(t(z, "up/in/%d" % s), H, W), t(z, "up/out/%d" % s))


def test_pose_matrix(layers_z):
    z = layers_z
    assert torch.equal(O.pose_matrix(t(z, "tfp/aa"), t(z, "tfp/t")), t(z, "tfp/M"))
    assert torch.equal(O.pose_matrix(t(z, "tfp/aa"), t(z, "tfp/t"), invert=True), t(z, "tfp/Minv"))
    eye = torch.matmul(t(z, "tfp/M"), t(z, "tfp/Minv"))            # KAT K4
    assert torch.allclose(eye, torch.eye(4).expand_as(eye), atol=1e-5)


def test_backproject_project_sample(layers_z):
    z = layers_z
    depth, K, iK, T = t(z, "geo/depth"), t(z, "geo/K"), t(z, "geo/inv_K"), t(z, "geo/T")
    pts = O.backproject(depth, iK)
    assert torch.equal(pts, t(z, "geo/points"))
    H, W = depth.shape[-2:]
    assert torch.equal(O.project(pts, K, T, H, W), t(z, "geo/grid"))
    assert torch.equal(O.warp(t(z, "geo/img"), depth, K, iK, T), t(z, "geo/warped"))


def test_ssim_reproj_smooth(layers_z):
    z = layers_z
    x, y = t(z, "ssim/x"), t(z, "ssim/y")
    assert torch.equal(O.ssim_map(x, y), t(z, "ssim/out"))
    assert float(O.ssim_map(x, x).abs().max()) == 0.0                # KAT K2
    assert torch.equal(O.photometric_loss(x, y), t(z, "reproj/out"))
    assert torch.equal(O.smooth_loss(t(z, "smooth/disp"), x), t(z, "smooth/out"))


def test_identity_pose_is_identity_warp():
    """KAT K1: T = I reproduces the source up to K^-1 round-off (tolerance from SURVEY 4)."""
    from make_golden import kitti_intrinsics
    H, W = 192, 640
    K, iK = (torch.from_numpy(a)[None] for a in kitti_intrinsics(H, W))
    g = torch.Generator().manual_seed(0)
    img = torch.rand(1, 3, H, W, generator=g)
    depth = 0.1 + 99.9 * torch.rand(1, 1, H, W, generator=g)
    out = O.warp(img, depth, K, iK, torch.eye(4)[None])
    assert float((out - img).abs().max()) < 2e-3


def test_constant_depth_translation_is_uniform_shift():
    """KAT K3: depth d, T=[I|tx] shifts sampling by fx*tx/d pixels."""
    from make_golden import kitti_intrinsics
    H, W = 192, 640
    K, iK = (torch.from_numpy(a)[None] for a in kitti_intrinsics(H, W))
    T = torch.eye(4)[None].clone()
    T[0, 0, 3] = 0.1
    depth = torch.full((1, 1, H, W), 2.0)
    grid = O.project(O.backproject(depth, iK), K, T, H, W)
    xs = (grid[0, :, :, 0] / 2 + 0.5) * (W - 1)
    shift = xs - torch.arange(W, dtype=torch.float32)[None]
    assert float((shift - 0.58 * W * 0.1 / 2.0).abs().max()) < 1e-3


def _same_cpu_as_golden():
    """Bit equality with the vectors is only defined on the CPU family that generated them."""
    c = Case("md2_b2_32x64")
    out = O.hot_path(c.inputs, c.disp, c.poses, c.ms, c.scales, c.trimin, c.decomp, c.noise, c.H, c.W,
                     poses_error=c.poses_error())
    return torch.equal(out["min/0"], c.expected("out/min/0"))


@pytest.mark.parametrize("name", DIRECT_CASES)
def test_hot_path_matches_reference(name):
    c = Case(name)
    if not _same_cpu_as_golden():
        pytest.skip("PyTorch CPU kernels round differently on this host CPU; see test_hot_path_tolerance")
    out = O.hot_path(c.inputs, c.disp, c.poses, c.ms, c.scales, c.trimin, c.decomp, c.noise,
                     c.H, c.W, poses_error=c.poses_error(), keep=True)
    out["loss"].backward()
    for s in c.scales:
        assert torch.equal(out["min/%d" % s], c.expected("out/min/%d" % s)), "min map scale %d" % s
        assert torch.equal(out["argmin/%d" % s], c.expected("out/argmin/%d" % s))
        if c.has("out/depth/%d" % s):
            assert torch.equal(out[("depth", 0, s)].detach(), c.expected("out/depth/%d" % s))
        assert abs(float(out["loss/%d" % s]) - float(c.expected("out/loss/%d" % s))) < 2e-7
        g, ge = c.disp[s].grad, c.expected("grad/disp/%d" % s)
        assert torch.allclose(g, ge, rtol=1e-4, atol=2e-9), float((g - ge).abs().max())  # accumulation order over warp jobs differs
    assert abs(float(out["loss"]) - float(c.expected("out/loss"))) < 2e-7
    for f, T in c.poses.items():
        ge = c.expected("grad/T/%s" % f)
        g = T.grad if T.grad is not None else torch.zeros_like(T)
        assert torch.allclose(g, ge, rtol=1e-4, atol=1e-9), (f, float((g - ge).abs().max()))
    for k in c.z.files:                       # every stored warp, bit for bit
        if k.startswith("out/color"):
            _, kind, f, s = k.split("/")
            key = (kind, "s" if f == "s" else int(f), int(s))
            assert torch.equal(out[key].detach(), c.expected(k)), k


@pytest.mark.parametrize("name", DIRECT_CASES)
def test_hot_path_tolerance(name):
    """Host-independent form of the pin: 1e-5 on the loss, 1e-4 on maps, arg-min equal off ties."""
    c = Case(name)
    out = O.hot_path(c.inputs, c.disp, c.poses, c.ms, c.scales, c.trimin, c.decomp, c.noise,
                     c.H, c.W, poses_error=c.poses_error())
    for s in c.scales:
        assert float((out["min/%d" % s] - c.expected("out/min/%d" % s)).abs().max()) < 1e-4
        clear = c.expected("out/margin/%d" % s) > 2e-4
        assert int(((out["argmin/%d" % s] != c.expected("out/argmin/%d" % s)) & clear).sum()) == 0
    assert abs(float(out["loss"].detach()) - float(c.expected("out/loss"))) < 1e-5
