"""CPU tier: the product's Python plumbing + the kernels' exact arithmetic (host build of
bbd_math.h) against the reference's golden vectors.  No GPU needed."""
import pytest
import torch

from golden_io import Case, DIRECT_CASES
from fused_runner import run_direct_case, compare_with_golden, compare_grads
from host_port import HostPortBackend


@pytest.fixture(scope="module")
def backend():
    return HostPortBackend()


def test_constant_divisions_are_correctly_rounded(backend):
    # every 97th float bit pattern across the positive range the kernels can see
    assert backend.check_div(0x2f000000, 8_000_000, 97) == 0
    assert backend.check_div(0xaf000000, 2_000_000, 389) == 0


@pytest.mark.parametrize("name", DIRECT_CASES)
def test_fused_path_host_port(name, backend):
    case = Case(name)
    tr, inputs, outputs, losses = run_direct_case(case, backend)
    report = compare_with_golden(case, tr, outputs, losses, exact=True)
    losses["loss"].backward()
    compare_grads(case, report)


@pytest.mark.parametrize("name", ["md2_b2_32x64", "tri_2102_32x64"])
def test_no_ssim_option_matches_oracle(name, backend):
    """--no_ssim (L1-only photometric loss, trainer.py:481-482): product path vs oracle, forward + grads."""
    from oracle import hotpath_ref as O
    from fused_runner import make_opt, bare_trainer
    case, ref = Case(name), Case(name)
    out = O.hot_path(ref.inputs, ref.disp, ref.poses, ref.ms, ref.scales, ref.trimin, ref.decomp, ref.noise,
                     ref.H, ref.W, poses_error=ref.poses_error(), no_ssim=True)
    out["loss"].backward()
    opt = make_opt(case, materialize_warps=False, no_ssim=True)
    tr = bare_trainer(opt, backend, "cpu")
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
        assert torch.equal(outputs[("bbd", "to_optimise")][i], out["min/%d" % s])
        assert torch.equal(outputs[("bbd", "argmin")][i], out["argmin/%d" % s])
        g, ge = case.disp[s].grad, ref.disp[s].grad
        assert float((g - ge).abs().max()) <= 1e-4 * float(ge.abs().max())
    assert abs(float(losses["loss"].detach()) - float(out["loss"].detach())) < 1e-6


def test_edge_of_domain_poses_and_depths(backend):
    """Border clamps, z <= 0, extreme depths: still bit-identical to the reference op sequence here."""
    from oracle import hotpath_ref as O
    from fused_runner import extreme_case, run_direct_case
    case, ref = extreme_case(), extreme_case()
    out = O.hot_path(ref.inputs, ref.disp, ref.poses, ref.ms, ref.scales, ref.trimin, ref.decomp, ref.noise,
                     ref.H, ref.W, poses_error=ref.poses_error())
    out["loss"].backward()
    tr, inputs, outputs, losses = run_direct_case(case, backend, materialize=False)
    losses["loss"].backward()
    for i, s in enumerate(case.scales):
        got, want = outputs[("bbd", "to_optimise")][i], out["min/%d" % s]
        nan = torch.isnan(want)
        assert torch.equal(torch.isnan(got), nan)
        assert torch.equal(got[~nan], want[~nan]), float((got[~nan] - want[~nan]).abs().max())
        assert torch.equal(outputs[("bbd", "argmin")][i], out["argmin/%d" % s])
        g, ge = case.disp[s].grad, ref.disp[s].grad
        assert float((g - ge).abs().max()) <= 1e-4 * float(ge.abs().max()) + 1e-12
    for f, T in case.poses.items():
        ge = ref.poses[f].grad
        if ge is None:
            continue
        assert float((T.grad - ge).abs().max()) <= 5e-3 * float(ge.abs().max()) + 1e-12, f


@pytest.mark.parametrize("H,W", [(37, 70), (16, 64), (19, 130)])
def test_sizes_off_the_tile_grid(H, W, backend):
    """Partial tiles / widths that are not multiples of 4: product plumbing + kernel math == oracle."""
    from oracle import hotpath_ref as O
    from fused_runner import odd_size_case, run_direct_case
    case, ref = odd_size_case(H, W), odd_size_case(H, W)
    out = O.hot_path(ref.inputs, ref.disp, ref.poses, ref.ms, ref.scales, ref.trimin, ref.decomp, ref.noise,
                     H, W, poses_error=ref.poses_error())
    out["loss"].backward()
    tr, inputs, outputs, losses = run_direct_case(case, backend, materialize=False)
    losses["loss"].backward()
    assert torch.equal(outputs[("bbd", "to_optimise")][0], out["min/0"])
    assert torch.equal(outputs[("bbd", "argmin")][0], out["argmin/0"])
    g, ge = case.disp[0].grad, ref.disp[0].grad
    assert float((g - ge).abs().max()) <= 1e-4 * float(ge.abs().max())
