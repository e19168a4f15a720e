"""CPU oracle for the photometric-reprojection hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain PyTorch (fp32, eager) restatement of what the upstream reference
computes on the path BASELINE.json names - it is the checker the HIP kernels are compared
against, and the thing `bench.py` times as `cpu_baseline`.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it; the
product package (`baseboostdepth_amd/`) never does.

Pinning: every function here is checked against golden vectors captured from the live
reference (tests/golden/*.npz, made by tools/make_golden.py) in tests/test_oracle_golden.py.
On the machine that generated the vectors, per-pixel maps and arg-min ids agree bit for bit and
scalar losses to fp32 round-off.  NOTE: PyTorch's CPU kernels (MKL bmm, vectorised ATen loops)
round differently on different host CPUs - on the GPU box's EPYC this same file differs from
the golden vectors by ~1e-5 per pixel (tools/diag_cpu_repro.py) - so tests that run the oracle
live use a tolerance and an off-tie arg-min protocol, while tests against the committed
vectors demand bit equality (the HIP kernels reproduce the vectors' rounding order exactly).

Reference citations are `file:line` into /root/reference.  The third-party arithmetic the
reference delegates to PyTorch ATen (grid_sampler_2d, avg_pool2d, reflection_pad2d,
upsample_bilinear2d, bmm, min.dim; reference pins pytorch=1.8.0, environment.yml:38) is
delegated to the same ATen ops here, so this oracle is "the reference's op sequence on the
installed torch", not an independent re-derivation; SURVEY.md Appendix A holds the formulas.
"""
import torch
import torch.nn.functional as F

MIN_DEPTH, MAX_DEPTH = 0.1, 100.0


# ------------------------------------------------------------------ a1  layers.py:13-22
def disp_to_depth(disp, min_depth=MIN_DEPTH, max_depth=MAX_DEPTH):
    lo, hi = 1 / max_depth, 1 / min_depth
    scaled = lo + (hi - lo) * disp
    return scaled, 1 / scaled


def upsample_disp(disp, H, W):
    """trainer.py:456 - bilinear, align_corners=False, to full resolution."""
    return F.interpolate(disp, [H, W], mode="bilinear", align_corners=False)


# ------------------------------------------------------------------ A8  layers.py:25-100
def pose_matrix(axisangle, translation, invert=False):
    """axisangle, translation: [n,1,3] -> [n,4,4]; M = T*R, or R^T*T(-t) when inverted."""
    n = axisangle.shape[0]
    theta = torch.norm(axisangle, 2, 2, True)
    axis = axisangle / (theta + 1e-7)
    ca, sa = torch.cos(theta), torch.sin(theta)
    C = 1 - ca
    x, y, z = (axis[..., i].unsqueeze(1) for i in range(3))
    xs, ys, zs = x * sa, y * sa, z * sa
    xC, yC, zC = x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    R = torch.zeros(n, 4, 4, device=axisangle.device)
    entries = {(0, 0): x * xC + ca, (0, 1): xyC - zs, (0, 2): zxC + ys,
               (1, 0): xyC + zs, (1, 1): y * yC + ca, (1, 2): yzC - xs,
               (2, 0): zxC - ys, (2, 1): yzC + xs, (2, 2): z * zC + ca}
    for (i, j), v in entries.items():
        R[:, i, j] = torch.squeeze(v)
    R[:, 3, 3] = 1
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t *= -1
    Tm = torch.zeros(n, 4, 4, device=axisangle.device)
    for i in range(4):
        Tm[:, i, i] = 1
    Tm[:, :3, 3, None] = t.contiguous().view(-1, 3, 1)
    return torch.matmul(R, Tm) if invert else torch.matmul(Tm, R)


# ------------------------------------------------------------------ a2/a3/a4  layers.py:136-195, trainer.py:434-442
def pixel_rays(H, W, device=None):
    """[1,3,H*W] homogeneous pixel coords, x fastest (layers.py:146-158)."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=device),
                            torch.arange(W, dtype=torch.float32, device=device), indexing="ij")
    return torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, device=device)], 0)[None]


def backproject(depth, inv_K):
    n, _, H, W = depth.shape
    pix = pixel_rays(H, W, depth.device).repeat(n, 1, 1)
    cam = torch.matmul(inv_K[:, :3, :3], pix)
    cam = depth.view(n, 1, -1) * cam
    return torch.cat([cam, torch.ones(n, 1, H * W, device=depth.device)], 1)


def project(points, K, T, H, W, eps=1e-7):
    P = torch.matmul(K, T)[:, :3, :]
    cam = torch.matmul(P, points)
    pix = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + eps)
    pix = pix.view(len(K), 2, H, W).permute(0, 2, 3, 1)
    pix[..., 0] /= W - 1
    pix[..., 1] /= H - 1
    return (pix - 0.5) * 2


def warp(images, depth, K, inv_K, T):
    H, W = depth.shape[-2:]
    grid = project(backproject(depth, inv_K), K, T, H, W)
    return F.grid_sample(images, grid, align_corners=True, padding_mode="border")


# ------------------------------------------------------------------ a5/a6  layers.py:219-249, trainer.py:477-486
def ssim_map(x, y):
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    y = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x, mu_y = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
    sigma_x = F.avg_pool2d(x ** 2, 3, 1) - mu_x ** 2
    sigma_y = F.avg_pool2d(y ** 2, 3, 1) - mu_y ** 2
    sigma_xy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + C1) * (2 * sigma_xy + C2)
    d = (mu_x ** 2 + mu_y ** 2 + C1) * (sigma_x + sigma_y + C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def photometric_loss(pred, target, no_ssim=False):
    l1 = torch.abs(target - pred).mean(1, True)
    if no_ssim:
        return l1
    return 0.85 * ssim_map(pred, target).mean(1, True) + 0.15 * l1


# ------------------------------------------------------------------ a12  layers.py:203-216
def smooth_loss(disp, img):
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


# ------------------------------------------------------------------ a10/a11 candidate sets (trainer.py:888-1100)
def candidate_frames(m, trimin):
    """Source frames whose reprojection competes at a target pixel of a sample with max offset m.

    trainer.py:987 (stereo only), :993-995 (m=1), :1006-1010 (m=2), :1025-1030 (m>=3);
    MD2 path :549-554.  Order is the reference's arg-min id order.
    """
    if m == 0:
        return ["s"]
    if not trimin:
        return [m, -m]
    if m == 1:
        return [1, -1, "s"]
    if m == 2:
        return [2, -2, 1, -1, "s"]
    return [m, -m, m - 1, -(m - 1), m - 2, -(m - 2)]


def candidate_list(m, trimin, decomp):
    """[(kind, frame)] with kind in 'T' (reprojection), 'E' (error-induced), 'I' (identity+noise)."""
    fr = candidate_frames(m, trimin)
    out = [("T", f) for f in fr]
    if trimin and decomp:
        out += [("E", f) for f in fr if f != "s"]
    out += [("I", f) for f in fr]
    return out


def warp_jobs(ms, trimin):
    """{frame: [sample ids]} - which target samples each source frame is warped for.

    trainer.py:900 (valid_mask_dict) / :912-918 (valid_tri_mask_dict) plus the
    valid_frames extension :961-981.
    """
    jobs = {}
    for b, m in enumerate(ms):
        for f in candidate_frames(m, trimin):
            jobs.setdefault(f, []).append(b)
    return jobs


def source_row(ms, f, b):
    """Row of sample b inside inputs[("color", f, 0)] (custom_collate stacks only samples that
    own frame f: trainer.py:882; 's' exists for m < 3: mono_dataset.py:107-108)."""
    if f == "s":
        owners = [i for i, m in enumerate(ms) if m < 3]
    else:
        owners = [i for i, m in enumerate(ms) if m >= abs(f)]
    return owners.index(b)


# ------------------------------------------------------------------ a7-a10 the whole hot path
def hot_path(inputs, disp, poses, ms, scales, trimin, decomp, noise, H, W,
             poses_error=None, smoothness=1e-3, num_scales=4, no_ssim=False, keep=False):
    """generate_images_pred + compute_losses (trainer.py:444-570) for one batch.

    inputs : dict with ("color", f, 0) [n_f,3,H,W], ("color", 0, s), ("K",0), ("inv_K",0), "stereo_T"
    disp   : {s: [B,1,H>>s,W>>s]}       poses: {f: [n_job,4,4]} rows in warp_jobs order
    noise  : [B,H,W] identity noise per sample (already scaled by 1e-5)
    Returns dict: loss, loss/s, min/s [B,H,W], argmin/s [B,H,W] (+ warps, depth when keep).
    """
    B = len(ms)
    jobs = warp_jobs(ms, trimin)
    target = inputs[("color", 0, 0)]
    K, inv_K = inputs[("K", 0)], inputs[("inv_K", 0)]
    out = {}

    def src_images(f, rows):
        idx = [source_row(ms, f, b) for b in rows]
        return inputs[("color", f, 0)][idx]

    # identity losses once per step (trainer.py:501-508)
    ident = {}
    for f, rows in jobs.items():
        ident[f] = photometric_loss(src_images(f, rows), target[rows], no_ssim)

    total = 0
    for s in scales:
        depth = disp_to_depth(upsample_disp(disp[s], H, W))[1]
        reproj, reproj_e = {}, {}
        for f, rows in jobs.items():
            n = len(rows)
            Kn, iKn = K[:n], inv_K[:n]            # count-sliced, trainer.py:431-432
            T = inputs["stereo_T"][rows] if f == "s" else poses[f]
            src, tgt, dep = src_images(f, rows), target[rows], depth[rows]
            w = warp(src, dep, Kn, iKn, T)
            reproj[f] = photometric_loss(w, tgt, no_ssim)
            if keep:
                out[("color", f, s)] = w
            if trimin and decomp and f != "s":
                we = warp(src, dep, Kn, iKn, poses_error[f])
                reproj_e[f] = photometric_loss(we, tgt, no_ssim)
                if keep:
                    out[("color_D", f, s)] = we
        dev = target.device
        mins = torch.zeros(B, H, W, device=dev)
        args = torch.zeros(B, H, W, dtype=torch.uint8, device=dev)
        margin = torch.zeros(B, H, W, device=dev)      # runner-up minus winner: how decisive the arg-min is
        parts = []
        for m in sorted(set(ms)):                      # one min per group, as x_min_opt does
            rows = [b for b in range(B) if ms[b] == m]
            stack = []
            for kind, f in candidate_list(m, trimin, decomp):
                pos = [jobs[f].index(b) for b in rows]
                if kind == "T":
                    stack.append(reproj[f][pos])
                elif kind == "E":
                    stack.append(reproj_e[f][pos])
                else:
                    stack.append(ident[f][pos] + noise[rows][:, None])
            stacked = torch.cat(stack, dim=1)
            val, idx = torch.min(stacked, dim=1)
            parts.append(val)
            mins[rows] = val.detach()
            args[rows] = idx.to(torch.uint8)
            if stacked.shape[1] > 1:
                two = torch.topk(stacked.detach(), 2, dim=1, largest=False).values
                margin[rows] = two[:, 1] - two[:, 0]
        to_optimise = torch.cat(parts, dim=0)
        loss = to_optimise.mean()
        d = disp[s]
        norm = d / (d.mean(2, True).mean(3, True) + 1e-7)
        loss = loss + smoothness * smooth_loss(norm, inputs[("color", 0, s)]) / (2 ** s)
        total = total + loss
        out["loss/%d" % s] = loss
        out["min/%d" % s] = mins
        out["argmin/%d" % s] = args
        out["margin/%d" % s] = margin
        if keep:
            out[("depth", 0, s)] = depth
    out["loss"] = total / num_scales                    # trainer.py:568 (frozen 4)
    return out
