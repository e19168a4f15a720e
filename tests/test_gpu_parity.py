"""GPU tier (-m gpu): the HIP kernels, called through the C ABI, against the reference's golden
vectors and the oracle.  Forward results must match BIT FOR BIT (see bbd_math.h)."""
import os

import numpy as np
import pytest
import torch

from golden_io import Case, DIRECT_CASES, GOLDEN_DIR
from fused_runner import run_direct_case, compare_with_golden, compare_grads

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def backend():
    from baseboostdepth_amd import ops
    return ops.default_backend()


@pytest.fixture(scope="module")
def layers_z():
    return np.load(os.path.join(GOLDEN_DIR, "layers.npz"))


def g(z, k):
    return torch.from_numpy(z[k]).to(DEV)


def test_native_library_is_loaded_and_single_hip_runtime(backend):
    maps = open("/proc/self/maps").read()
    assert "libbbd_hip.so" in maps
    runtimes = {line.split()[-1] for line in maps.splitlines() if "libamdhip64" in line}
    assert len(runtimes) == 1, runtimes      # our .so must share torch's HIP runtime


def test_cheap_division_is_ieee_exact(backend):
    """2^28 random operand tuples: shared-reciprocal / constant-divisor divisions == hipcc's `/`."""
    from baseboostdepth_amd._lib import ptr
    bad = torch.zeros(1, dtype=torch.int32, device=DEV)
    for seed in (1, 2):
        backend.run("bbd_selftest_div", bad, 2048, 256, seed, ptr(bad))
    torch.cuda.synchronize()
    assert int(bad.item()) == 0


@pytest.mark.parametrize("name", DIRECT_CASES)
def test_fused_path_matches_reference_bit_for_bit(name, backend):
    case = Case(name, device=DEV)
    tr, inputs, outputs, losses = run_direct_case(case, backend, device=DEV)
    report = compare_with_golden(case, tr, outputs, losses, exact=True)
    losses["loss"].backward()
    torch.cuda.synchronize()
    compare_grads(case, report)


@pytest.mark.parametrize("name", ["md2_b2_32x64", "tri_3105_32x64"])
def test_without_materialised_warps_same_result(name, backend):
    a = Case(name, device=DEV)
    _, _, out_a, loss_a = run_direct_case(a, backend, device=DEV, materialize=True)
    b = Case(name, device=DEV)
    _, _, out_b, loss_b = run_direct_case(b, backend, device=DEV, materialize=False)
    assert ("color", 1, 0) in out_a and ("color", 1, 0) not in out_b
    assert torch.equal(out_a[("bbd", "to_optimise")], out_b[("bbd", "to_optimise")])
    assert torch.equal(out_a[("bbd", "argmin")], out_b[("bbd", "argmin")])
    assert float(loss_a["loss"]) == float(loss_b["loss"])


def test_standalone_layers(layers_z, backend):
    from baseboostdepth_amd import layers as L
    z = layers_z
    depth, K, iK, T = g(z, "geo/depth"), g(z, "geo/K"), g(z, "geo/inv_K"), g(z, "geo/T")
    n, _, H, W = depth.shape
    pts = L.BackprojectDepth(4, H, W)(depth, iK)
    assert torch.equal(pts.cpu(), torch.from_numpy(z["geo/points"]))
    grid = L.Project3D(4, H, W)(pts, K, T)
    assert torch.equal(grid.cpu(), torch.from_numpy(z["geo/grid"]))
    x, y = g(z, "ssim/x"), g(z, "ssim/y")
    assert torch.equal(L.SSIM()(x, y).cpu(), torch.from_numpy(z["ssim/out"]))
    assert float(L.SSIM()(x, x).abs().max()) == 0.0                      # KAT K2
    sd, dp = L.disp_to_depth(g(z, "d2d/disp"), 0.1, 100.0)
    assert torch.allclose(dp.cpu(), torch.from_numpy(z["d2d/depth"]), rtol=1e-6)
    M = L.transformation_from_parameters(g(z, "tfp/aa"), g(z, "tfp/t"))
    Mi = L.transformation_from_parameters(g(z, "tfp/aa"), g(z, "tfp/t"), invert=True)
    assert torch.allclose(M.cpu(), torch.from_numpy(z["tfp/M"]), atol=1e-6)
    assert torch.allclose(Mi.cpu(), torch.from_numpy(z["tfp/Minv"]), atol=1e-6)
    assert L.SSIM()(x.clone().requires_grad_(True), y).requires_grad    # differentiable: test_gpu_layers_autograd.py


def test_disp_to_depth_kernel(layers_z, backend):
    from baseboostdepth_amd import ops
    from oracle import hotpath_ref as O
    z = layers_z
    H, W = z["d2d/disp"].shape[-2:]
    d0 = g(z, "d2d/disp")
    assert torch.equal(ops.disp_to_depth_fullres(d0, H, W, 0.1, 100.0, backend).cpu(),
                       torch.from_numpy(z["d2d/depth"]))
    for s in (1, 2):
        small = torch.from_numpy(z["up/in/%d" % s])
        want = O.disp_to_depth(torch.from_numpy(z["up/out/%d" % s]))[1]
        got = ops.disp_to_depth_fullres(small.to(DEV), H, W, 0.1, 100.0, backend)
        assert torch.equal(got.cpu(), want)
    # adjoint against autograd of the oracle, at the real pyramid sizes (generic ATen kernel path)
    gen = torch.Generator().manual_seed(3)
    for s in (0, 1, 2, 3):
        disp = torch.rand(2, 1, 192 >> s, 640 >> s, generator=gen)
        up = torch.rand(2, 1, 192, 640, generator=gen)
        dc = disp.clone().requires_grad_(True)
        ref = O.disp_to_depth(O.upsample_disp(dc, 192, 640))[1]
        (ref * up).sum().backward()
        dg = disp.clone().to(DEV).requires_grad_(True)
        got = ops.disp_to_depth_fullres(dg, 192, 640, 0.1, 100.0, backend)
        assert torch.equal(got.detach().cpu(), ref.detach())
        (got * up.to(DEV)).sum().backward()
        err = float((dg.grad.cpu() - dc.grad).abs().max()) / float(dc.grad.abs().max())
        assert err < 1e-5, (s, err)


def _random_batch(B, H, W, ms, gen, trimin, decomp):
    """Seeded synthetic batch in the reference's post-collate layout (full resolution)."""
    from make_golden import kitti_intrinsics, synth_scene, u8_to_f32
    from oracle import hotpath_ref as O
    frames = sorted({f for m in ms for f in O.candidate_frames(m, trimin)} | {0}, key=lambda f: (f == "s", f))
    inputs = {}
    owners = {f: [b for b, m in enumerate(ms) if (m < 3 if f == "s" else m >= abs(f))] for f in frames}
    scenes = [synth_scene(gen, H, W, [f for f in frames if b in owners[f]]) for b in range(B)]
    for f in frames:
        inputs[("color", f, 0)] = torch.stack([u8_to_f32(scenes[b][f]) for b in owners[f]])
    K, iK = kitti_intrinsics(H, W)
    inputs[("K", 0)] = torch.from_numpy(K)[None].repeat(B, 1, 1)
    inputs[("inv_K", 0)] = torch.from_numpy(iK)[None].repeat(B, 1, 1)
    sT = torch.eye(4)[None].repeat(B, 1, 1)
    sT[:, 0, 3] = 0.1
    inputs["stereo_T"] = sT
    inputs["ordering"] = [[0, "s"] if m == 0 else [0, m, -m] for m in ms]
    inputs["frames"] = [f for f in frames]
    return inputs


def _note(text):
    """Observed error levels, kept next to the bars they justify (BBD_TEST_REPORT=<file> to collect)."""
    import os
    path = os.environ.get("BBD_TEST_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(text + "\n")


@pytest.mark.parametrize("ms,trimin,decomp,scales", [
    ([1, 1], False, False, [0, 1, 2, 3]),
    ([7, 3], True, True, [0]),
    ([1] * 12, False, False, [0, 1, 2, 3]),          # BASELINE configs[1] at its real size: B=12, 4 scales
])
def test_full_resolution_against_oracle(ms, trimin, decomp, scales, backend):
    """BASELINE sizes (192x640): HIP vs the oracle run live on this box's CPU.  PyTorch's CPU
    kernels round differently across host CPUs (oracle header), so this uses the tolerance
    protocol: maps to 1e-4, arg-min equal where the oracle's margin exceeds 2e-4, loss to 1e-5,
    gradients to 1e-4 of their max outside the 3x3 footprint of flipped pixels."""
    import types
    from make_golden import synth_disp
    from oracle import hotpath_ref as O
    from fused_runner import bare_trainer
    from baseboostdepth_amd.layers import transformation_from_parameters as tfp
    H, W, B = 192, 640, len(ms)
    gen = torch.Generator().manual_seed(11)
    inputs = _random_batch(B, H, W, ms, gen, trimin, decomp)
    for s in scales:
        if s:
            inputs[("color", 0, s)] = torch.nn.functional.interpolate(inputs[("color", 0, 0)], size=(H >> s, W >> s), mode="area")
    disp = synth_disp(gen, B, H, W, scales)
    jobs = O.warp_jobs(ms, trimin)
    poses = {}
    for f, rows in jobs.items():
        if f == "s":
            continue
        aa = 0.01 * torch.randn(len(rows), 1, 3, generator=gen)
        tt = 0.02 * abs(f) * torch.randn(len(rows), 1, 3, generator=gen)
        poses[f] = tfp(aa, tt, invert=(f < 0)).detach().requires_grad_(True)
    perr = {}
    for f, T in poses.items():
        Te = T.clone().detach()
        Te[:, :3, 3:] /= 5.5
        perr[f] = Te
    noise = torch.randn(B, H, W, generator=gen) * 0.00001
    ref = O.hot_path(inputs, disp, poses, ms, scales, trimin, decomp, noise, H, W, poses_error=perr)
    ref["loss"].backward()

    opt = types.SimpleNamespace(height=H, width=W, batch_size=B, scales=scales, frame_ids=[0], min_depth=0.1,
                                max_depth=100.0, disparity_smoothness=1e-3, no_ssim=False, trimin=trimin,
                                decomp=decomp, pose_error=5.5, incremental_skip=False, partial_skip=False,
                                materialize_warps=False)
    tr = bare_trainer(opt, backend, DEV)
    gin = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in inputs.items()}
    gin["noise"] = noise.to(DEV)
    tr.valid_frames_trimin(gin)
    gdisp = {s: disp[s].detach().clone().to(DEV).requires_grad_(True) for s in scales}
    gpose = {f: T.detach().clone().to(DEV).requires_grad_(True) for f, T in poses.items()}
    outputs = {("disp", s): gdisp[s] for s in scales}
    for f in gpose:
        outputs[("cam_T_cam", 0, f)] = gpose[f]
        outputs[("cam_T_cam_error", 0, f)] = perr[f].to(DEV)
    outputs.update(tr.generate_images_pred(gin, outputs))
    losses = tr.compute_losses(gin, outputs)
    losses["loss"].backward()
    total_flips = 0
    for i, s in enumerate(scales):
        got, arg = outputs[("bbd", "to_optimise")][i].cpu(), outputs[("bbd", "argmin")][i].cpu()
        assert float((got - ref["min/%d" % s]).abs().max()) < 1e-4
        mism = arg != ref["argmin/%d" % s]
        assert int((mism & (ref["margin/%d" % s] > 2e-4)).sum()) == 0
        flips = int(mism.sum())
        total_flips += flips
        ge = disp[s].grad
        gg = gdisp[s].grad.cpu()
        rel = (gg - ge).abs() / float(ge.abs().max())
        # a flipped near-tie pixel re-routes gradient inside its 3x3 window and, at coarse scales,
        # through the 2x2 bilinear footprints of those nine texels
        n_bad = int((rel > 1e-4).sum())
        _note("full_res ms=%s scale %d: flips %d, texels off (>1e-4 of max) %d, max rel %.3e, L2 rel %.3e" % (
            ms, s, flips, n_bad, float(rel.max()), float((gg - ge).norm() / ge.norm())))
        assert n_bad <= 25 * flips, ("disp grad", s, flips, n_bad, float(rel.max()))
        assert float((gg - ge).norm() / ge.norm()) < (1e-4 if flips == 0 else 5e-2), ("disp grad L2", s)
    assert abs(float(losses["loss"].detach()) - float(ref["loss"].detach())) < 1e-5
    for f, T in poses.items():
        if T.grad is None:
            continue
        err = float((gpose[f].grad.cpu() - T.grad).abs().max()) / (float(T.grad.abs().max()) + 1e-12)
        _note("full_res ms=%s pose %s: rel-to-max err %.3e (total flips %d)" % (ms, f, err, total_flips))
        assert err < (1e-4 if total_flips == 0 else 1e-1), ("pose grad", f, err)


def test_full_size_properties(backend):
    """Size-independent properties at BASELINE config 2 size (B=12, 4 scales, 192x640)."""
    import types
    from make_golden import synth_disp
    from fused_runner import bare_trainer
    H, W, B = 192, 640, 12
    gen = torch.Generator().manual_seed(5)
    inputs = _random_batch(B, H, W, [1] * B, gen, False, False)
    scales = [0, 1, 2, 3]
    gin = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in inputs.items()}
    for s in scales[1:]:
        gin[("color", 0, s)] = torch.nn.functional.interpolate(gin[("color", 0, 0)], size=(H >> s, W >> s), mode="area")
    gin["noise"] = (torch.randn(B, H, W, generator=gen) * 0.00001).to(DEV)
    opt = types.SimpleNamespace(height=H, width=W, batch_size=B, scales=scales, frame_ids=[0, -1, 1], min_depth=0.1,
                                max_depth=100.0, disparity_smoothness=1e-3, no_ssim=False, trimin=False,
                                decomp=False, pose_error=5.5, incremental_skip=False, partial_skip=False,
                                materialize_warps=True)
    tr = bare_trainer(opt, backend, DEV)
    tr.valid_frames_trimin(gin)
    disp = synth_disp(gen, B, H, W, scales)
    outputs = {("disp", s): disp[s].detach().to(DEV) for s in scales}
    # identity pose for +1, a pure x-translation giving an integer pixel shift for -1 (KAT K1/K3)
    eye = torch.eye(4, device=DEV)[None].repeat(B, 1, 1)
    outputs[("cam_T_cam", 0, 1)] = eye
    outputs[("cam_T_cam", 0, -1)] = eye.clone()
    res1 = tr.generate_images_pred(gin, dict(outputs))
    res2 = tr.generate_images_pred(gin, dict(outputs))
    # (1) determinism: two launches agree bit for bit
    for k in (("bbd", "to_optimise"), ("bbd", "argmin"), ("bbd", "loss_sum")):
        assert torch.equal(res1[k], res2[k])
    # (2) checksum of checksums: per-tile partial sums add up to the sum of the map
    tot = res1[("bbd", "to_optimise")].double().sum(dim=(1, 2, 3))
    assert torch.allclose(res1[("bbd", "loss_sum")].double(), tot, rtol=1e-5)
    # (3) identity pose reproduces the source image (KAT K1, tolerance from SURVEY 4)
    w = res1[("color", 1, 0)]
    assert float((w - gin[("color", 1, 0)]).abs().max()) < 2e-3
    # (4) the winning value is never above any identity candidate (min property)
    ident = res1[("bbd", "identity")]
    plan = tr.plan
    for b in range(B):
        for f in (1, -1):
            cand = ident[plan.ident_index[(b, f)]] + gin["noise"][b]
            assert bool((res1[("bbd", "to_optimise")][0, b] <= cand).all())
    # (5) arg-min ids stay inside the candidate list
    assert int(res1[("bbd", "argmin")].max()) < 4


def test_smoothness_kernel_against_oracle(backend):
    """layers.get_smooth_loss on the mean-normalised disparity, forward and backward, BASELINE sizes."""
    from baseboostdepth_amd import ops
    from oracle import hotpath_ref as O
    gen = torch.Generator().manual_seed(9)
    for s in (0, 2):
        h, w = 192 >> s, 640 >> s
        disp = torch.rand(3, 1, h, w, generator=gen) * 0.8 + 0.01
        img = torch.round(torch.rand(3, 3, h, w, generator=gen) * 255) / 255
        dc = disp.clone().requires_grad_(True)
        ref = O.smooth_loss(dc / (dc.mean(2, True).mean(3, True) + 1e-7), img)
        ref.backward()
        dg = disp.clone().to(DEV).requires_grad_(True)
        got = ops.normalised_smooth_loss(dg, img.to(DEV), backend)
        got.backward()
        assert abs(float(got) - float(ref)) < 2e-6 * max(1.0, abs(float(ref))), (float(got), float(ref))
        err = float((dg.grad.cpu() - dc.grad).abs().max()) / float(dc.grad.abs().max())
        _note("smoothness scale %d: grad rel-to-max err %.3e" % (s, err))
        assert err < 1e-4, (s, err)


@pytest.mark.parametrize("form", [0, 1, 2])
def test_every_forward_form_is_bit_exact(form):
    """The fused forward has three forms chosen per launch shape (three waves / double buffer; four waves with the target
    statistics re-derived per candidate; four waves with them held).  Each is valid for every input: the golden cases must
    come out bit for bit under each (BBD_FWD_FORM forces one; it is read once per process, hence the subprocess)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BBD_EXPERIMENT="1", BBD_FWD_FORM=str(form))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x",
                          "-k", "fused_path_matches_reference_bit_for_bit or disparity_mode_equals_depth_plane_mode"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and " passed" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])


def test_multi_scale_smoothness_launch_equals_the_single_scale_launches(backend):
    """bbd_smooth_loss_multi_* (all scales of a step in one launch pair each way) vs bbd_smooth_loss_* per scale: same
    kernels, same reduction order - values and gradients bit for bit."""
    from baseboostdepth_amd import ops
    gen = torch.Generator().manual_seed(19)
    B = 5
    disps = [(torch.rand(B, 1, 192 >> s, 640 >> s, generator=gen) * 0.8 + 0.01).to(DEV).requires_grad_(True) for s in range(4)]
    imgs = [(torch.round(torch.rand(B, 3, 192 >> s, 640 >> s, generator=gen) * 255) / 255).to(DEV) for s in range(4)]
    multi = ops.normalised_smooth_losses(disps, imgs, backend)
    wts = torch.tensor([1.0, 0.5, 0.25, 0.125], device=DEV)
    (multi * wts).sum().backward()
    for i in range(4):
        d2 = disps[i].detach().clone().requires_grad_(True)
        one = ops.normalised_smooth_loss(d2, imgs[i], backend)
        (one * wts[i]).backward()
        assert abs(float(one) - float(multi[i])) <= 3e-7 * abs(float(one))       # (1-2 ulp: the sum over chunks and the division are ATen ops in another order)
        assert torch.equal(d2.grad, disps[i].grad), i


def test_pose_matrix_rows_launch_equals_the_per_sign_launches(backend):
    """ops.pose_matrix(invert_rows=...) - the poses of both signs of a step in one launch each way - vs one call per sign."""
    from baseboostdepth_amd import ops
    gen = torch.Generator().manual_seed(23)
    n = 24
    aa = (0.3 * torch.randn(n, 1, 3, generator=gen)).to(DEV)
    tr = torch.randn(n, 1, 3, generator=gen).to(DEV)
    w = torch.randn(n, 4, 4, generator=gen).to(DEV)
    flags = torch.tensor([1] * 12 + [0] * 12, dtype=torch.int32, device=DEV)
    a1, t1 = aa.clone().requires_grad_(True), tr.clone().requires_grad_(True)
    M = ops.pose_matrix(a1, t1, backend=backend, invert_rows=flags)
    (M * w).sum().backward()
    a2, t2 = aa.clone().requires_grad_(True), tr.clone().requires_grad_(True)
    M2 = torch.cat([ops.pose_matrix(a2[:12], t2[:12], True, backend), ops.pose_matrix(a2[12:], t2[12:], False, backend)], 0)
    (M2 * w).sum().backward()
    assert torch.equal(M, M2) and torch.equal(a1.grad, a2.grad) and torch.equal(t1.grad, t2.grad)


def test_pose_matrix_kernel(backend):
    """Fused transformation_from_parameters vs the op-for-op torch form (values and gradients)."""
    from baseboostdepth_amd import layers as L
    gen = torch.Generator().manual_seed(4)
    for invert in (False, True):
        aa = (0.3 * torch.randn(7, 1, 3, generator=gen))
        tr = torch.randn(7, 1, 3, generator=gen)
        w = torch.randn(7, 4, 4, generator=gen)
        a1, t1 = aa.clone().requires_grad_(True), tr.clone().requires_grad_(True)
        ref = L._transformation_from_parameters_torch(a1, t1, invert)
        (ref * w).sum().backward()
        a2, t2 = aa.clone().to(DEV).requires_grad_(True), tr.clone().to(DEV).requires_grad_(True)
        got = L.transformation_from_parameters(a2, t2, invert)
        (got * w.to(DEV)).sum().backward()
        assert torch.allclose(got.detach().cpu(), ref.detach(), atol=2e-6), float((got.cpu() - ref).abs().max())
        assert torch.allclose(a2.grad.cpu(), a1.grad, atol=2e-5, rtol=1e-4), float((a2.grad.cpu() - a1.grad).abs().max())
        assert torch.allclose(t2.grad.cpu(), t1.grad, atol=2e-5, rtol=1e-4)
    eye = torch.matmul(L.transformation_from_parameters(a2.detach(), t2.detach(), False),
                       L.transformation_from_parameters(a2.detach(), t2.detach(), True))     # KAT K4
    assert torch.allclose(eye.cpu(), torch.eye(4).expand(7, 4, 4), atol=1e-5)


@pytest.mark.parametrize("name", ["md2_b2_32x64", "tri_3105_32x64"])
def test_no_ssim_option_on_gpu(name, backend):
    """L1-only loss (--no_ssim): HIP vs the live oracle (tolerance protocol) incl. gradients."""
    from oracle import hotpath_ref as O
    from fused_runner import make_opt, bare_trainer
    case, ref = Case(name, device=DEV), Case(name)
    out = O.hot_path(ref.inputs, ref.disp, ref.poses, ref.ms, ref.scales, ref.trimin, ref.decomp, ref.noise,
                     ref.H, ref.W, poses_error=ref.poses_error(), no_ssim=True)
    out["loss"].backward()
    opt = make_opt(case, materialize_warps=False, no_ssim=True)
    tr = bare_trainer(opt, backend, DEV)
    inputs = dict(case.inputs)
    inputs["noise"] = case.noise
    tr.valid_frames_trimin(inputs)
    outputs = {("disp", s): case.disp[s] for s in case.scales}
    perr = case.poses_error()
    for f, T in case.poses.items():
        outputs[("cam_T_cam", 0, f)] = T
        outputs[("cam_T_cam_error", 0, f)] = perr[f]
    outputs.update(tr.generate_images_pred(inputs, outputs))
    losses = tr.compute_losses(inputs, outputs)
    losses["loss"].backward()
    for i, s in enumerate(case.scales):
        got = outputs[("bbd", "to_optimise")][i].cpu()
        assert float((got - out["min/%d" % s]).abs().max()) < 1e-4
        mism = outputs[("bbd", "argmin")][i].cpu() != out["argmin/%d" % s]
        assert int((mism & (out["margin/%d" % s] > 2e-4)).sum()) == 0
        g, ge = case.disp[s].grad.cpu(), ref.disp[s].grad
        rel = (g - ge).abs() / float(ge.abs().max())
        _note("no_ssim/extreme scale %d: flips %d, texels off %d, max rel %.3e" % (s, int(mism.sum()), int((rel > 1e-4).sum()), float(rel.max())))
        assert int((rel > 1e-4).sum()) <= 25 * int(mism.sum())
    assert abs(float(losses["loss"].detach()) - float(out["loss"].detach())) < 1e-5


def test_edge_of_domain_poses_and_depths_on_gpu(backend):
    """Border clamps, z <= 0, extreme depths on the GPU vs the live oracle (tolerance protocol)."""
    from oracle import hotpath_ref as O
    from fused_runner import extreme_case
    case, ref = extreme_case(device=DEV), extreme_case()
    out = O.hot_path(ref.inputs, ref.disp, ref.poses, ref.ms, ref.scales, ref.trimin, ref.decomp, ref.noise,
                     ref.H, ref.W, poses_error=ref.poses_error())
    out["loss"].backward()
    tr, inputs, outputs, losses = run_direct_case(case, backend, device=DEV, materialize=False)
    losses["loss"].backward()
    flips = 0
    for i, s in enumerate(case.scales):
        got, want = outputs[("bbd", "to_optimise")][i].cpu(), out["min/%d" % s]
        assert float((got - want).abs().max()) < 1e-4
        mism = outputs[("bbd", "argmin")][i].cpu() != out["argmin/%d" % s]
        assert int((mism & (out["margin/%d" % s] > 2e-4)).sum()) == 0
        flips += int(mism.sum())
        g, ge = case.disp[s].grad.cpu(), ref.disp[s].grad
        rel = (g - ge).abs() / (float(ge.abs().max()) + 1e-12)
        _note("extreme scale %d: flips %d, texels off %d, max rel %.3e" % (s, int(mism.sum()), int((rel > 1e-4).sum()), float(rel.max())))
        assert int((rel > 1e-4).sum()) <= 25 * int(mism.sum()) + 4      # +4: clamp decisions at |ix - border| ~ ulp
    assert abs(float(losses["loss"].detach()) - float(out["loss"].detach())) < 1e-5


def test_backward_properties_full_size(backend):
    """Size-independent properties of the fused backward at BASELINE size (B=12, 4 scales, 192x640):
    bitwise determinism, exact linearity in the upstream scalar (power of two), pose rows that are
    constants (stereo / error-induced) receive exactly zero, and d loss/d depth is zero wherever an
    identity candidate won the whole 3x3 neighbourhood."""
    from baseboostdepth_amd import ops
    from baseboostdepth_amd.synthetic import synthetic_batch, synthetic_disp, synthetic_poses
    from baseboostdepth_amd.plan import get_plan
    H, W, B, scales = 192, 640, 12, [0, 1, 2, 3]
    ms = [2, 1, 0, 2, 1, 2, 1, 1, 2, 0, 1, 2]
    inputs = synthetic_batch(ms, H, W, scales, device=DEV, seed=8)
    plan = get_plan(inputs["ordering"], True, True)
    poses = synthetic_poses(plan, device=DEV, seed=3)
    job = {}
    for kind, f in plan.pose_jobs:
        if f == "s":
            job[(kind, f)] = inputs["stereo_T"][plan.jobs[f]]
        else:
            job[(kind, f)] = poses[("cam_T_cam" if kind == "T" else "cam_T_cam_error", 0, f)]
    table = ops.pose_table(plan, inputs[("K", 0)], inputs[("inv_K", 0)], job)
    disp = synthetic_disp(B, H, W, scales, device=DEV, seed=4)
    depth = ops.disp_pyramid_to_depth([disp[s] for s in scales], H, W, 0.1, 100.0, backend)
    frames = {f: inputs[("color", f, 0)] for f in plan.frames}
    target = inputs[("color", 0, 0)]
    ident = ops.identity_losses(plan, frames, target, False, backend)

    def grads(scale):
        d = depth.detach().clone().requires_grad_(True)
        t = table.detach().clone().requires_grad_(True)
        ls, mn, am, _ = ops.fused_reprojection_min(d, t, target, ident, inputs["noise"], plan, frames, False,
                                                   False, backend)
        (ls.sum() * scale).backward()
        return d.grad, t.grad, am

    g1, t1, am = grads(1.0 / 1024)
    g2, t2, _ = grads(1.0 / 1024)
    assert torch.equal(g1, g2) and torch.equal(t1, t2)                       # deterministic
    g4, t4, _ = grads(4.0 / 1024)
    assert torch.equal(g4, g1 * 4) and torch.allclose(t4, t1 * 4, rtol=1e-6)  # linear in the upstream scalar
    assert float(t1[:, :12].abs().max()) == 0 and float(t1[:, 28:].abs().max()) == 0   # only T gets gradient
    for (kind, f), off in plan.pose_offset.items():
        if kind == "E" or f == "s":
            assert float(t1[off:off + len(plan.jobs[f])].abs().max()) == 0, (kind, f)
    # pixels whose 3x3 neighbourhood was won entirely by identity candidates get no depth gradient
    n_warp = torch.tensor([sum(1 for k, _ in names if k != "I") for names in plan.cand_names], device=DEV)
    is_ident = (am >= n_warp.view(1, B, 1, 1)).float()
    all_ident = torch.nn.functional.avg_pool2d(is_ident, 3, 1, 1, count_include_pad=False) == 1.0
    assert float(g1[all_ident].abs().max()) == 0.0
    assert float(g1.abs().max()) > 0


@pytest.mark.parametrize("H,W", [(37, 70), (16, 64), (19, 130), (200, 646)])
def test_sizes_off_the_tile_grid_on_gpu(H, W, backend):
    """Partial tiles, widths not multiple of 4 (scalar load/store paths), borders inside tiles."""
    from oracle import hotpath_ref as O
    from fused_runner import odd_size_case
    case, ref = odd_size_case(H, W, device=DEV), odd_size_case(H, W)
    out = O.hot_path(ref.inputs, ref.disp, ref.poses, ref.ms, ref.scales, ref.trimin, ref.decomp, ref.noise,
                     H, W, poses_error=ref.poses_error())
    out["loss"].backward()
    tr, inputs, outputs, losses = run_direct_case(case, backend, device=DEV, materialize=True)
    losses["loss"].backward()
    got = outputs[("bbd", "to_optimise")][0].cpu()
    assert float((got - out["min/0"]).abs().max()) < 1e-4
    mism = outputs[("bbd", "argmin")][0].cpu() != out["argmin/0"]
    assert int((mism & (out["margin/0"] > 2e-4)).sum()) == 0
    g, ge = case.disp[0].grad.cpu(), ref.disp[0].grad
    rel = (g - ge).abs() / float(ge.abs().max())
    _note("odd size: flips %d, texels off %d, max rel %.3e" % (int(mism.sum()), int((rel > 1e-4).sum()), float(rel.max())))
    assert int((rel > 1e-4).sum()) <= 25 * int(mism.sum()) + 4
    assert abs(float(losses["loss"].detach()) - float(out["loss"].detach())) < 1e-5


@pytest.mark.parametrize("name", ["md2_b2_32x64", "tri_3105_32x64", "md2_b1_192x640"])
def test_disparity_mode_equals_depth_plane_mode(name, backend):
    """the fused launches fed with the low-resolution disparities (up-sampling + disp_to_depth per
    staged pixel, no depth buffer) give bit-identical maps / arg-min ids / depth to the two-kernel form, and
    the same disparity and pose gradients (the adjoint is regrouped: 1e-6 of max)."""
    from baseboostdepth_amd import ops
    res = {}
    for mode in (True, False):
        case = Case(name, device=DEV)
        from fused_runner import make_opt, bare_trainer
        opt = make_opt(case, materialize_warps=False)
        opt.fused_disp = bool(mode)
        tr = bare_trainer(opt, backend, DEV)
        inputs = dict(case.inputs)
        inputs["noise"] = case.noise
        tr.opt.frame_ids = sorted(inputs["frames"], key=lambda it: float("inf") if isinstance(it, str) else abs(it))
        tr.valid_frames_trimin(inputs)
        outputs = {}
        perr = case.poses_error()
        for f, T in case.poses.items():
            outputs[("cam_T_cam", 0, f)] = T
            if case.decomp:
                outputs[("cam_T_cam_error", 0, f)] = perr[f]
        for s in case.scales:
            outputs[("disp", s)] = case.disp[s]
        outputs.update(tr.generate_images_pred(inputs, outputs))
        losses = tr.compute_losses(inputs, outputs)
        losses["loss"].backward()
        res[mode] = (outputs, losses, case)
    (oa, la, ca), (ob, lb, cb) = res[True], res[False]
    assert torch.equal(oa[("bbd", "to_optimise")], ob[("bbd", "to_optimise")])
    assert torch.equal(oa[("bbd", "argmin")], ob[("bbd", "argmin")])
    assert float(la["loss"].detach()) == float(lb["loss"].detach())
    for s in ca.scales:
        assert torch.equal(oa[("depth", 0, s)], ob[("depth", 0, s)]), s
        ga, gb = ca.disp[s].grad, cb.disp[s].grad
        assert float((ga - gb).abs().max()) <= 1e-6 * float(gb.abs().max()), s
    for f in ca.poses:
        if cb.poses[f].grad is not None:
            assert float((ca.poses[f].grad - cb.poses[f].grad).abs().max()) <= 1e-6 * float(cb.poses[f].grad.abs().max()) + 1e-12
    # golden check of the by-product depth
    for s in ca.scales:
        if ca.has("out/depth/%d" % s):
            assert torch.equal(oa[("depth", 0, s)].cpu(), ca.expected("out/depth/%d" % s))


@pytest.mark.parametrize("H,W,no_ssim", [(37, 131, False), (16, 67, False), (5, 3, False), (9, 130, True), (192, 640, False)])
def test_identity_pass_forms_agree_on_ragged_sizes(backend, H, W, no_ssim):
    """The two entry points of the identity pre-pass - one workgroup per (item, tile), and the grouped form the training
    path uses (one workgroup per (target sample, tile) walking the sample's items) - bit for bit on odd widths, partial
    tiles, images smaller than a tile, --no_ssim."""
    from baseboostdepth_amd._lib import ptr
    from baseboostdepth_amd import ops
    from baseboostdepth_amd.plan import get_plan
    torch.manual_seed(H * 1000 + W)
    ms = [2, 1, 0, 3]
    plan = get_plan([[0, "s"] if m == 0 else [0, m, -m] for m in ms], True, True)
    frames = {f: (torch.round(torch.rand(len(plan.owners(f)), 3, H, W, device=DEV) * 255) / 255) for f in plan.frames}
    target = torch.round(torch.rand(len(ms), 3, H, W, device=DEV) * 255) / 255
    tb = plan.tables(target.device)
    fp = ops.frame_pointer_array(frames)
    a = torch.full((plan.NI, H, W), float("nan"), device=DEV)
    b = torch.full((plan.NI, H, W), float("nan"), device=DEV)
    backend.run("bbd_identity_loss_fwd", target, fp, ptr(target), ptr(tb["items"]), plan.NI, ptr(a), H, W, int(no_ssim))
    backend.run("bbd_identity_loss_grouped_fwd", target, fp, ptr(target), ptr(tb["items"]), ptr(tb["ident_off"]), plan.B,
                ptr(b), H, W, int(no_ssim))
    torch.cuda.synchronize()
    assert not torch.isnan(a).any() and torch.equal(a, b)
    assert torch.equal(ops.identity_losses(plan, frames, target, no_ssim, backend), a)
