"""Drive the product's hot path (Trainer.generate_images_pred + compute_losses) on a golden case.

Shared by the CPU tier (host-port backend) and the GPU tier (HIP backend through the C ABI)."""
import types

import torch

from baseboostdepth_amd.trainer import Trainer
from baseboostdepth_amd.plan import STEREO


def make_opt(case, **over):
    opt = types.SimpleNamespace(
        height=case.H, width=case.W, batch_size=case.B, scales=list(case.scales), frame_ids=[0, -1, 1],
        min_depth=0.1, max_depth=100.0, disparity_smoothness=1e-3, no_ssim=False,
        trimin=case.trimin, decomp=case.decomp, pose_error=5.5,
        incremental_skip=case.incremental, partial_skip=case.partial,
        materialize_warps=True, num_layers=18, weights_init="scratch", learning_rate=1e-4)
    opt.__dict__.update(over)
    return opt


def bare_trainer(opt, backend, device):
    """Trainer without networks/optimizer (the hot-path methods do not need them)."""
    tr = Trainer.__new__(Trainer)
    tr.opt = opt
    tr.device = torch.device(device)
    tr.num_scales = 4
    tr.backend = backend
    tr.models = {}
    tr.grad_sync = None
    tr.flat_grads = None
    tr.maxing_valid_frames = False
    from baseboostdepth_amd.layers import SSIM
    tr.ssim = SSIM()
    return tr


def run_direct_case(case, backend, device="cpu", materialize=True):
    """Poses given directly (fixture T/<f>).  Returns (trainer, inputs, outputs, losses)."""
    opt = make_opt(case, materialize_warps=materialize)
    tr = bare_trainer(opt, backend, device)
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
    return tr, inputs, outputs, losses


def compare_with_golden(case, tr, outputs, losses, loss_tol=1e-5, map_tol=2e-5, tie_margin=1e-4,
                        grad_rtol=2e-3, check_warps=True, exact=False):
    """Assertions shared by both tiers.

    exact=True (the bar both tiers are held to): per-pixel min-loss maps, arg-min ids, depth and
    warped images must equal the reference BIT FOR BIT - the kernels reproduce the reference CPU
    path's rounding order (bbd_math.h).  The scalar loss is a mean whose summation order differs
    (per-tile partial sums), so it is compared to north_star's 1e-5 (observed ~1e-8).
    exact=False keeps the looser protocol of SURVEY 7 (arg-min equal outside near-ties)."""
    report = {}
    if exact:
        map_tol, tie_margin = 0.0, -1.0
    for i, s in enumerate(case.scales):
        got = outputs[("bbd", "to_optimise")][i].detach().cpu()
        exp = case.expected("out/min/%d" % s)
        arg = outputs[("bbd", "argmin")][i].cpu()
        exp_arg = case.expected("out/argmin/%d" % s)
        margin = case.expected("out/margin/%d" % s)
        clear = margin > tie_margin
        flips = int((arg != exp_arg).sum())
        bad = int(((arg != exp_arg) & clear).sum())
        report["flips/%d" % s] = flips
        assert bad == 0, "scale %d: %d arg-min mismatches outside ties (total flips %d)" % (s, bad, flips)
        err = float((got - exp).abs().max())
        report["maxerr/%d" % s] = err
        assert err <= map_tol if exact else err < map_tol, "scale %d: min-loss map max err %.3e" % (s, err)
        le = float(case.expected("out/loss/%d" % s))
        assert abs(float(losses["loss/%d" % s].detach()) - le) < loss_tol
        if case.has("out/depth/%d" % s):
            d = outputs[("depth", 0, s)].detach().cpu()
            de = case.expected("out/depth/%d" % s)
            if exact:
                assert torch.equal(d, de), "depth scale %d not bit-exact" % s
            assert torch.allclose(d, de, rtol=2e-6, atol=1e-7), float((d - de).abs().max())
    assert abs(float(losses["loss"].detach()) - float(case.expected("out/loss"))) < loss_tol
    # identity photometric losses (un-warped source vs target), one map per (sample, frame)
    ident = outputs[("bbd", "identity")].detach().cpu()
    for k in case.z.files:
        if k.startswith("out/ident/"):
            f = k.split("/")[2]
            f = STEREO if f == "s" else int(f)
            want = case.expected(k)[:, 0]
            rows = [tr.plan.ident_index[(b, f)] for b in tr.plan.jobs[f]]
            got_i = ident[rows]
            if exact:
                assert torch.equal(got_i, want), "identity loss of frame %s not bit-exact" % (f,)
            assert float((got_i - want).abs().max()) < map_tol + 1e-6
    if check_warps:
        for k in case.z.files:
            if k.startswith("out/color"):
                _, kind, f, s = k.split("/")
                key = (kind, STEREO if f == "s" else int(f), int(s))
                w, we = outputs[key].detach().cpu(), case.expected(k)
                if exact:
                    assert torch.equal(w, we), "%s not bit-exact (max err %.3e)" % (k, float((w - we).abs().max()))
                assert float((w - we).abs().max()) < 2e-4, (k, float((w - we).abs().max()))
    return report


def compare_grads(case, report=None, grad_rtol=1e-4):
    """After losses['loss'].backward(): disp and pose gradients vs the reference's autograd.

    A pixel whose arg-min legitimately flipped at a near-tie (see compare_with_golden) routes its
    gradient to another candidate, which changes d loss/d disp inside that pixel's 3x3 SSIM
    window; those few texels are allowed to differ (<= 12 per flip), everything else must match."""
    report = report or {}
    total_flips = 0
    for s in case.scales:
        flips = report.get("flips/%d" % s, 0)
        total_flips += flips
        g, ge = case.disp[s].grad.detach().cpu(), case.expected("grad/disp/%d" % s)
        scale = float(ge.abs().max()) + 1e-12
        rel = (g - ge).abs() / scale
        n_bad = int((rel > grad_rtol).sum())
        assert n_bad <= 12 * flips, "grad disp scale %d: %d texels off (flips %d, max rel %.3e)" % (
            s, n_bad, flips, float(rel.max()))
    for f, T in case.poses.items():
        ge = case.expected("grad/T/%s" % f)
        g = (T.grad if T.grad is not None else torch.zeros_like(T)).detach().cpu()
        scale = float(ge.abs().max()) + 1e-12
        err = float((g - ge).abs().max()) / scale
        tol = grad_rtol if total_flips == 0 else 10 * grad_rtol
        assert err < tol, "grad T[%s]: rel-to-max err %.3e" % (f, err)


def extreme_case(name="tri_2102_32x64", device="cpu"):
    """A golden case with its poses/disparities replaced by edge-of-domain values: large rotations and
    translations (most samples land outside the source -> border clamp, zero coordinate gradient),
    points behind the camera (z <= 0), disparity at both ends of [0,1] (depth 100 and 0.1)."""
    from golden_io import Case
    case = Case(name, device=device)
    gen = torch.Generator().manual_seed(123)
    from baseboostdepth_amd.layers import _transformation_from_parameters_torch as tfp
    for f, T in list(case.poses.items()):
        n = T.shape[0]
        aa = 0.6 * torch.randn(n, 1, 3, generator=gen)
        tt = 1.5 * torch.randn(n, 1, 3, generator=gen)
        tt[0, 0, 2] = -3.0 if f > 0 else 3.0           # drives z through zero for near points
        case.poses[f] = tfp(aa, tt, invert=(f < 0)).to(device).requires_grad_(True)
    for s in case.scales:
        d = case.disp[s].detach().clone()
        d[:, :, : d.shape[2] // 2, : d.shape[3] // 2] = 0.0      # depth = max_depth
        d[:, :, d.shape[2] // 2:, d.shape[3] // 2:] = 1.0        # depth = min_depth
        case.disp[s] = d.requires_grad_(True)
    return case


def odd_size_case(H=37, W=70, ms=(2, 1, 0), trimin=True, decomp=True, device="cpu", seed=31):
    """Synthetic batch whose size is not a multiple of the 64x16 tile nor of 4: partial tiles, scalar
    (unaligned) loads/stores, image borders inside a tile.  Returns an object shaped like golden_io.Case."""
    import types
    from baseboostdepth_amd.synthetic import synthetic_batch
    from baseboostdepth_amd.plan import get_plan
    from baseboostdepth_amd.layers import _transformation_from_parameters_torch as tfp
    ms = list(ms)
    inputs = synthetic_batch(ms, H, W, [0], device="cpu", seed=seed)
    gen = torch.Generator().manual_seed(seed + 1)
    plan = get_plan(inputs["ordering"], trimin, decomp)
    poses = {}
    for f in plan.frames:
        if f == "s":
            continue
        n = len(plan.jobs[f])
        poses[f] = tfp(0.02 * torch.randn(n, 1, 3, generator=gen), 0.05 * torch.randn(n, 1, 3, generator=gen),
                       invert=(f < 0)).to(device).requires_grad_(True)
    case = types.SimpleNamespace()
    case.inputs = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in inputs.items()}
    case.noise = case.inputs.pop("noise")
    case.ms, case.scales, case.trimin, case.decomp = ms, [0], trimin, decomp
    case.incremental = case.partial = False
    case.B, case.H, case.W = len(ms), H, W
    case.disp = {0: (0.02 + 0.5 * torch.rand(len(ms), 1, H, W, generator=gen)).to(device).requires_grad_(True)}
    case.poses = poses

    def poses_error(pose_error=5.5):
        out = {}
        for f, T in case.poses.items():
            Te = T.clone().detach().cpu()
            Te[:, :3, 3:] /= pose_error
            out[f] = Te.to(T.device)
        return out
    case.poses_error = poses_error
    return case
