"""torch.autograd wrappers around the C-ABI kernels (include/bbd_hip.h).

PyTorch is plumbing here (device memory, streams, autograd tape); all arithmetic of the hot
path runs in baseboostdepth_amd/csrc/bbd_kernels.hip.  `HipBackend` is the only backend the
product has: it refuses non-GPU tensors and raises if the extension is missing.  (The CPU test
tier injects a host build of the same arithmetic through the `backend=` seam - see
tests/host_port.py - but nothing in this package can fall back to it.)
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import COMPOSE_STRIDE, COMPOSE_ERROR, COMPOSE_REPLACE, POSE_STRIDE, PROJ_STRIDE, MAX_FRAME_SLOTS, ptr
from .plan import frame_slot


# BBD_FUSED_NN=0 sends the encoder / decoder glue (pad, max-pool) back to the stock ATen kernels (A/B runs)
def _experiment(name, default):
    """Experiment knobs of the A/B tooling (tools/*.sh) are read only under BBD_EXPERIMENT=1: a stray variable in a
    user's environment cannot put the shipped library into a configuration no test covers."""
    if os.environ.get("BBD_EXPERIMENT") != "1":
        return default
    return os.environ.get(name, default)


FUSED_NN = os.environ.get("BBD_FUSED_NN", "1") != "0"
FUSED_TOKEN_GLUE = os.environ.get("BBD_FUSED_TOKEN_GLUE", "1") != "0"   # MonoViT: residual + DropPath + LayerNorm passes


class KernelTimer:
    """HIP-event timing of individual C-ABI launches, on the stream they are launched on.

    Used by bench.py to measure the fused kernels' own duration inside the full training step
    (torch.cuda.Event records on torch's current stream, which is the stream `HipBackend` launches
    on).  Off by default: `backend.timer = KernelTimer()` turns it on."""

    def __init__(self):
        self.events = {}

    def launch(self, name, fn):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        # (the grouped identity launch is reported under the plain entry point's name: same work, same byte model)
        self.events.setdefault(name.replace("_grouped_", "_"), []).append((e0, e1))

    def reset(self):
        self.events = {}

    def summary(self):
        """{kernel: (launches, mean ms)} - call after a device synchronise."""
        out = {}
        for name, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = (len(ms), sum(ms) / max(len(ms), 1))
        return out


class HipBackend:
    """Launches on the current torch HIP stream of the tensors' device."""

    name = "hip"

    def __init__(self):
        self.lib = _lib.get_lib()
        self.timer = None
        self._shared_work = {}

    def num_tiles(self, H, W):
        return self.lib.num_tiles(H, W)

    def num_tiles_fwd(self, H, W):
        return self.lib.num_tiles_fwd(H, W)

    def num_tiles_bwd(self, H, W):
        return self.lib.num_tiles_bwd(H, W)

    def smooth_chunks(self):
        return self.lib.smooth_chunks

    def fused_work_items(self, plan, S, H, W, backward, device):
        """Device table [S*B*ntiles, 2] of bbd_fused_work_items (slab order per XCD, the plan's samples with the most
        candidates first).  Batch order - every batch whose samples arrive sorted by candidate count, `Trainer`'s
        canonical order - is ONE table per launch shape, uploaded once and shared by all plans; any other order is part
        of the step's table upload (`steptables.StepTables`) or, for callers without one, a single asynchronous upload."""
        if _experiment("BBD_WORK_TABLE", "1") == "0":      # A/B switch: grid order, decoded in the kernel
            return None
        if plan.sample_order is None:
            tb = self._shared_work
            key = (str(device), plan.B, S, H, W, int(backward))
        else:
            tb = plan.tables(device)
            key = ("work", S, H, W, int(backward))
        if key not in tb:
            from . import steptables
            n = S * plan.B * (self.num_tiles_bwd(H, W) if backward else self.num_tiles_fwd(H, W))
            host = torch.empty(n, 2, dtype=torch.int32, pin_memory=torch.device(device).type == "cuda")
            order = None if plan.sample_order is None else (ctypes.c_int32 * plan.B)(*plan.sample_order)
            try:
                self.lib.call("bbd_fused_work_items", plan.B, S, H, W, int(backward), order, host.data_ptr())
                tb[key] = host.to(device, non_blocking=True)
                steptables.STATS["single_uploads"] += 1
            except _lib.BbdError:
                # outside the table's packing limits (more than 4 096 samples, 8 scales or 2^17 tiles): the launches take
                # the grid order and decode it themselves - a speed choice only
                tb[key] = None
        return tb[key]

    @staticmethod
    def _check(*tensors):
        for t in tensors:
            if t is not None and not t.is_cuda:
                raise _lib.BbdError("the HIP hot path needs GPU tensors (got %s); there is no CPU path"
                                    % t.device)

    def run(self, name, anchor, *args):
        if self.timer is not None:
            self.timer.launch(name, lambda: self.lib.call(name, *args, self.lib.stream_for(anchor)))
        else:
            self.lib.call(name, *args, self.lib.stream_for(anchor))


_BACKEND = None


def _frame_list(frame_tensors):
    return [frame_tensors] if torch.is_tensor(frame_tensors) else list(frame_tensors.values())


def default_backend():
    global _BACKEND
    if _BACKEND is None:
        _BACKEND = HipBackend()
    return _BACKEND


def frame_pointer_array(frame_tensors):
    """Host array[MAX_FRAME_SLOTS] of base addresses; `frame_tensors` maps frame id -> [n,3,H,W] - or IS one pool tensor
    [F,3,H,W] holding every frame of the step (`pooled.PooledStep`): all slots then point at it and the tables address
    it by row."""
    arr = (ctypes.c_void_p * MAX_FRAME_SLOTS)()
    if torch.is_tensor(frame_tensors):
        assert frame_tensors.is_contiguous() and frame_tensors.dtype == torch.float32
        for i in range(MAX_FRAME_SLOTS):
            arr[i] = frame_tensors.data_ptr()
        return arr
    for f, t in frame_tensors.items():
        assert t.is_contiguous() and t.dtype == torch.float32
        arr[frame_slot(f)] = t.data_ptr()
    return arr


# ---------------------------------------------------------------------------- disp -> depth
class _DispToDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, H, W, min_depth, max_depth, backend):
        B, one, h, w = disp.shape
        assert one == 1
        disp = disp.contiguous()
        depth = torch.empty(B, 1, H, W, device=disp.device, dtype=torch.float32)
        backend._check(disp)
        backend.run("bbd_disp_to_depth_fwd", disp, ptr(disp), ptr(depth), B, h, w, H, W,
                    float(min_depth), float(max_depth))
        ctx.save_for_backward(disp, depth)
        ctx.meta = (H, W, float(min_depth), float(max_depth), backend)
        return depth

    @staticmethod
    def backward(ctx, grad_depth):
        disp, depth = ctx.saved_tensors
        H, W, lo, hi, backend = ctx.meta
        B, _, h, w = disp.shape
        grad_depth = grad_depth.contiguous()
        grad_disp = torch.empty_like(disp)
        backend.run("bbd_disp_to_depth_bwd", disp, ptr(disp), ptr(depth), ptr(grad_depth), ptr(grad_disp), B, h, w,
                    H, W, lo, hi)
        return grad_disp, None, None, None, None, None


def disp_to_depth_fullres(disp, H, W, min_depth, max_depth, backend=None):
    """F.interpolate(bilinear, align_corners=False) + layers.disp_to_depth (trainer.py:455-461)."""
    return _DispToDepth.apply(disp, H, W, min_depth, max_depth, backend or default_backend())


class _DispPyramidToDepth(torch.autograd.Function):
    """All scales at once: S low-res disparity maps -> one [S,B,H,W] depth buffer (no cat copy)."""

    @staticmethod
    def forward(ctx, H, W, min_depth, max_depth, backend, *disps):
        disps = [d.contiguous() for d in disps]
        B = disps[0].shape[0]
        depth = torch.empty(len(disps), B, H, W, device=disps[0].device, dtype=torch.float32)
        backend._check(*disps)
        for i, d in enumerate(disps):
            backend.run("bbd_disp_to_depth_fwd", d, ptr(d), ptr(depth[i]), B, d.shape[2], d.shape[3], H, W,
                        float(min_depth), float(max_depth))
        ctx.save_for_backward(depth, *disps)
        ctx.meta = (H, W, float(min_depth), float(max_depth), backend)
        return depth

    @staticmethod
    def backward(ctx, grad_depth):
        depth, disps = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        H, W, lo, hi, backend = ctx.meta
        grad_depth = grad_depth.contiguous()
        grads = []
        for i, d in enumerate(disps):
            g = torch.empty_like(d)
            backend.run("bbd_disp_to_depth_bwd", d, ptr(d), ptr(depth[i]), ptr(grad_depth[i]), ptr(g), d.shape[0],
                        d.shape[2], d.shape[3], H, W, lo, hi)
            grads.append(g)
        return (None, None, None, None, None) + tuple(grads)


def disp_pyramid_to_depth(disps, H, W, min_depth, max_depth, backend=None):
    """[("disp", s)] list -> depth [S,B,H,W]; depth[i] is the reference's outputs[("depth",0,s_i)]."""
    return _DispPyramidToDepth.apply(H, W, min_depth, max_depth, backend or default_backend(), *disps)


# ---------------------------------------------------------------------------- pose matrix
class _PoseMatrix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, axisangle, translation, invert, backend, invert_rows):
        aa = axisangle.reshape(-1, 3).contiguous()
        tr = translation.reshape(-1, 3).contiguous()
        backend._check(aa, tr)
        n = aa.shape[0]
        assert invert_rows is None or (invert_rows.dtype == torch.int32 and invert_rows.numel() == n)
        M = torch.empty(n, 4, 4, device=aa.device, dtype=torch.float32)
        backend.run("bbd_pose_matrix_fwd", aa, ptr(aa), ptr(tr), ptr(M), n, int(invert), ptr(invert_rows))
        ctx.save_for_backward(aa, tr)
        ctx.meta = (bool(invert), backend, axisangle.shape, translation.shape, invert_rows)
        return M

    @staticmethod
    def backward(ctx, gM):
        aa, tr = ctx.saved_tensors
        invert, backend, shp_a, shp_t, invert_rows = ctx.meta
        gM = gM.contiguous()
        ga, gt = torch.empty_like(aa), torch.empty_like(tr)
        backend.run("bbd_pose_matrix_bwd", aa, ptr(aa), ptr(tr), ptr(gM), ptr(ga), ptr(gt), aa.shape[0], int(invert),
                    ptr(invert_rows))
        return ga.view(shp_a), gt.view(shp_t), None, None, None


def pose_matrix(axisangle, translation, invert=False, backend=None, invert_rows=None):
    """layers.transformation_from_parameters as one kernel (forward) + one (backward).  `invert_rows` (device int32 [n])
    replaces the scalar flag row by row: the poses of both signs of a step in one launch each way."""
    return _PoseMatrix.apply(axisangle, translation, invert, backend or default_backend(), invert_rows)


# ---------------------------------------------------------------------------- smoothness
class _SmoothLoss(torch.autograd.Function):
    """get_smooth_loss(disp / (mean(disp)+1e-7), img) as two launches forward, two backward."""

    @staticmethod
    def forward(ctx, disp, img, backend):
        B, _, h, w = disp.shape
        disp, img = disp.contiguous(), img.contiguous()
        backend._check(disp, img)
        chunks = backend.smooth_chunks()
        mean = torch.empty(B, chunks, device=disp.device, dtype=torch.float32)   # partial sums of disp
        sums = torch.empty(B, chunks, 2, device=disp.device, dtype=torch.float32)
        backend.run("bbd_smooth_loss_fwd", disp, ptr(disp), ptr(img), ptr(mean), ptr(sums), B, h, w)
        ctx.save_for_backward(disp, img, mean)
        ctx.meta = (backend, chunks)
        tot = sums.sum(dim=(0, 1))
        return tot[0] / (B * h * (w - 1)) + tot[1] / (B * (h - 1) * w)

    @staticmethod
    def backward(ctx, g):
        disp, img, mean = ctx.saved_tensors
        backend, chunks = ctx.meta
        B, _, h, w = disp.shape
        gscale = g.reshape(1).contiguous().to(torch.float32)
        grad = torch.empty_like(disp)
        dots = torch.empty(B, chunks, device=disp.device, dtype=torch.float32)
        backend.run("bbd_smooth_loss_bwd", disp, ptr(disp), ptr(img), ptr(mean), ptr(gscale), ptr(grad), ptr(dots),
                    B, h, w)
        return grad, None, None


class _SmoothLossMulti(torch.autograd.Function):
    """The smoothness terms of ALL scales of a step (trainer.py:560-564 inside the scale loop) as one launch pair each way
    (bbd_smooth_loss_multi_*): MD2's four scales took 16 launches.  Returns the [S] vector of smooth values."""
    _den = {}

    @staticmethod
    def forward(ctx, backend, n, *tensors):
        disps = [t.contiguous() for t in tensors[:n]]
        imgs = [t.contiguous() for t in tensors[n:]]
        backend._check(*disps, *imgs)
        B, dev = disps[0].shape[0], disps[0].device
        chunks = backend.smooth_chunks()
        mean = torch.empty(n, B, chunks, device=dev, dtype=torch.float32)
        sums = torch.empty(n, B, chunks, 2, device=dev, dtype=torch.float32)
        backend.run("bbd_smooth_loss_multi_fwd", disps[0], _ptr_array(disps), _ptr_array(imgs), _hw_array(disps), ptr(mean),
                    ptr(sums), n, B)
        ctx.save_for_backward(mean, *disps, *imgs)
        ctx.meta = (backend, chunks, n)
        key = (str(dev), B) + tuple(tuple(d.shape[-2:]) for d in disps)
        den = _SmoothLossMulti._den.get(key)
        if den is None:       # mean over the x-terms' B*h*(w-1) and the y-terms' B*(h-1)*w elements (layers.py:213-216)
            from .steptables import upload_single      # (pinned + asynchronous: legal wherever this node first runs)
            den = upload_single([[B * d.shape[-2] * (d.shape[-1] - 1), B * (d.shape[-2] - 1) * d.shape[-1]] for d in disps],
                                dev, torch.float32)
            _SmoothLossMulti._den[key] = den
        return (sums.sum(dim=(1, 2)) / den).sum(dim=1)

    @staticmethod
    def backward(ctx, g):
        backend, chunks, n = ctx.meta
        mean = ctx.saved_tensors[0]
        disps, imgs = ctx.saved_tensors[1:1 + n], ctx.saved_tensors[1 + n:]
        B = disps[0].shape[0]
        gscale = g.contiguous().to(torch.float32)
        grads = [torch.empty_like(d) for d in disps]
        dots = torch.empty(n, B, chunks, device=mean.device, dtype=torch.float32)
        backend.run("bbd_smooth_loss_multi_bwd", disps[0], _ptr_array(disps), _ptr_array(imgs), _hw_array(disps), ptr(mean),
                    ptr(gscale), _ptr_array(grads), ptr(dots), n, B)
        return (None, None) + tuple(grads) + (None,) * n


def normalised_smooth_losses(disps, imgs, backend=None):
    """[S] tensor: normalised_smooth_loss of every (disp, img) pair, one launch pair each way (at most 4 scales)."""
    return _SmoothLossMulti.apply(backend or default_backend(), len(disps), *disps, *imgs)


def normalised_smooth_loss(disp, img, backend=None):
    """layers.get_smooth_loss(disp / (disp.mean(2,True).mean(3,True) + 1e-7), img)  (trainer.py:560-563)."""
    return _SmoothLoss.apply(disp, img, backend or default_backend())


# ---------------------------------------------------------------------------- per-scale losses -> total (trainer.py:557-568)
class _CombineLosses(torch.autograd.Function):
    """per[s] = loss_sum[s] / n_px + w[s] * smooth[s];  total = sum(per) / num_scales  (trainer.py:557, :563-564, :566-568).
    As scalar torch ops this was ~25 launches forward and ~45 backward per step (divisions, multiplications, additions and
    their autograd nodes on one-element tensors - the eager hot path is launch-bound, profiles/r04/hot_path_launches.txt);
    here it is 4 + 3 on [S]-vectors, the backward being two constants times the incoming gradient."""
    _w = {}

    @staticmethod
    def forward(ctx, loss_sum, smooths, n_px, smoothness, scales, num_scales):
        key = (str(loss_sum.device), float(smoothness), tuple(scales))
        w = _CombineLosses._w.get(key)
        if w is None:       # disparity_smoothness / 2**s in fp32: scaling by a power of two commutes with the rounding
            from .steptables import upload_single
            w = upload_single((torch.tensor([float(smoothness)] * len(scales), dtype=torch.float32) /
                               torch.tensor([2.0 ** s for s in scales], dtype=torch.float32)), loss_sum.device, torch.float32)
            _CombineLosses._w[key] = w
        per = torch.addcmul(loss_sum / n_px, smooths, w)
        total = per.sum() / num_scales
        ctx.save_for_backward(w)
        ctx.meta = (float(n_px), float(num_scales))
        return total, per

    @staticmethod
    def backward(ctx, g_total, g_per):
        (w,) = ctx.saved_tensors
        n_px, num_scales = ctx.meta
        g = (g_total / num_scales).expand_as(w)
        if g_per is not None:
            g = g + g_per
        return g / n_px, g * w, None, None, None, None


def combine_losses(loss_sum, smooths, n_px, smoothness, scales, num_scales):
    """(total, per-scale [S]) of the step's loss from the fused launch's per-scale sums and the smoothness terms."""
    return _CombineLosses.apply(loss_sum, smooths, n_px, smoothness, scales, num_scales)


# ---------------------------------------------------------------------------- pose composition
FUSED_POSE_COMPOSE = os.environ.get("BBD_FUSED_POSE_COMPOSE", "1") != "0"


class ComposeTable:
    """Host-built description of every composed 4x4 of a step (include/bbd_hip.h, bbd_pose_compose_fwd): rows =
    [(chain of step-row indices, direct row | -1, flags)], plus the inverse table the backward walks."""

    def __init__(self, rows, n_steps):
        import numpy as np
        self.NO, self.R = len(rows), n_steps
        tab = np.zeros((max(self.NO, 1), COMPOSE_STRIDE), dtype=np.int32)
        refs = [[] for _ in range(n_steps)]
        for o, (chain, direct, flags) in enumerate(rows):
            assert len(chain) <= 7
            tab[o, 0] = len(chain)
            tab[o, 1:1 + len(chain)] = chain
            tab[o, 8], tab[o, 9] = direct, flags
            for k, r in enumerate(chain):
                refs[r].append((o, k))
            if direct >= 0 and (flags & COMPOSE_REPLACE):
                refs[direct].append((o, -1))
        off = np.zeros(n_steps + 1, dtype=np.int32)
        for r in range(n_steps):
            off[r + 1] = off[r] + len(refs[r])
        flat = np.array([e for rr in refs for e in rr], dtype=np.int32).reshape(-1, 2)
        if flat.shape[0] == 0:
            flat = np.zeros((1, 2), dtype=np.int32)
        self.np = (tab, off, flat)
        self._dev = {}

    def device(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(torch.from_numpy(a).to(device).contiguous() for a in self.np)
        return self._dev[key]


class _PoseCompose(torch.autograd.Function):
    @staticmethod
    def forward(ctx, steps, table, pose_error, backend):
        steps = steps.contiguous()
        tab, off, refs = table.device(steps.device)
        out = torch.empty(table.NO, 4, 4, device=steps.device, dtype=torch.float32)
        backend._check(steps)
        backend.run("bbd_pose_compose_fwd", steps, ptr(steps), ptr(tab), ptr(out), table.NO, float(pose_error))
        ctx.save_for_backward(steps)
        ctx.meta = (table, backend)
        return out

    @staticmethod
    def backward(ctx, gout):
        (steps,) = ctx.saved_tensors
        table, backend = ctx.meta
        tab, off, refs = table.device(steps.device)
        gout = gout.contiguous()
        gsteps = torch.empty_like(steps)
        backend.run("bbd_pose_compose_bwd", steps, ptr(steps), ptr(tab), ptr(off), ptr(refs), ptr(gout), ptr(gsteps),
                    table.R)
        return gsteps, None, None, None


def pose_compose(steps, table, pose_error, backend=None):
    """steps [R,4,4] -> [NO,4,4]: every chained / error-induced / partially swapped pose of the step, one launch."""
    return _PoseCompose.apply(steps, table, pose_error, backend or default_backend())


class _PoseComposeStatic(torch.autograd.Function):
    """`_PoseCompose` on tables that are views of the step's static table buffer (`pooled.PooledStep`): NO rows of which
    the trailing ones are constant no-op rows, `off` = R + 1 entries."""

    @staticmethod
    def forward(ctx, steps, tab, off, refs, NO, pose_error, backend):
        steps = steps.contiguous()
        out = torch.empty(NO, 4, 4, device=steps.device, dtype=torch.float32)
        backend._check(steps, tab, off, refs)
        assert off.numel() == steps.shape[0] + 1 and tab.shape[0] >= NO
        backend.run("bbd_pose_compose_fwd", steps, ptr(steps), ptr(tab), ptr(out), NO, float(pose_error))
        ctx.save_for_backward(steps, tab, off, refs)
        ctx.backend = backend
        return out

    @staticmethod
    def backward(ctx, gout):
        steps, tab, off, refs = ctx.saved_tensors
        gout = gout.contiguous()
        gsteps = torch.empty_like(steps)
        ctx.backend.run("bbd_pose_compose_bwd", steps, ptr(steps), ptr(tab), ptr(off), ptr(refs), ptr(gout), ptr(gsteps),
                        steps.shape[0])
        return gsteps, None, None, None, None, None, None


def pose_compose_static(steps, tab, off, refs, NO, pose_error, backend=None):
    return _PoseComposeStatic.apply(steps, tab, off, refs, NO, pose_error, backend or default_backend())


# ---------------------------------------------------------------------------- identity pre-pass
def identity_losses(plan, frame_tensors, target, no_ssim=False, backend=None):
    """[NI,H,W] identity photometric losses (no gradient: inputs are images)."""
    backend = backend or default_backend()
    B, _, H, W = target.shape
    tb = plan.tables(target.device)
    ident = torch.empty(plan.NI, H, W, device=target.device, dtype=torch.float32)
    backend._check(target, *_frame_list(frame_tensors))
    frames = frame_pointer_array(frame_tensors)
    # one workgroup per (target sample, tile) walks the sample's identity candidates (round 3).  Round 5 measured a streaming,
    # LDS-free form against it - slower (tools/experiments/streaming_identity.hip.txt, profiles/r05/identity_forms.txt), round 6
    # the form with aligned 8-byte loads and the halo columns over DPP - slower too (profiles/r06/identity_stream_ab.txt)
    backend.run("bbd_identity_loss_grouped_fwd", target, frames, ptr(target), ptr(tb["items"]), ptr(tb["ident_off"]),
                plan.B, ptr(ident), H, W, int(no_ssim))
    return ident


def gather_pairs(pool, idx_a, idx_b, normalize=None, backend=None):
    """[R, 6, H, W]: row r = cat(pool[idx_a[r]], pool[idx_b[r]]) of the frame pool [F, 3, H, W] - the batched pose pass's input
    in the pooled form - in one launch; `normalize = (mean, std)` folds the encoder's `(x - mean) / std` in (evaluated like
    PyTorch-ROCm does: subtract, then multiply by the float reciprocal of std).  No gradient: the pool holds images."""
    import numpy as np
    backend = backend or default_backend()
    F, C, H, W = pool.shape
    R = idx_a.numel()
    assert pool.is_contiguous() and idx_a.dtype == torch.int32 and idx_b.dtype == torch.int32 and idx_b.numel() == R
    backend._check(pool, idx_a, idx_b)
    out = torch.empty(R, 2 * C, H, W, device=pool.device, dtype=torch.float32)
    sub, mul = (0.0, 1.0) if normalize is None else (float(np.float32(normalize[0])),
                                                     float(np.float32(1.0) / np.float32(normalize[1])))
    backend.run("bbd_gather_pairs", pool, ptr(pool), ptr(idx_a.contiguous()), ptr(idx_b.contiguous()), ptr(out), R, C * H * W, sub, mul)
    return out


# ---------------------------------------------------------------------------- pose table
def pose_table(plan, K, inv_K, poses):
    """[NP,40] rows  K[:3,:] | T | inv_K[:3,:3] | pad, differentiable w.r.t. the poses.

    `poses[(kind, f)]` is the [n_job,4,4] transform of warp job (kind, f) with rows in
    plan.jobs[f] order.  K/inv_K rows are taken by count like the reference (trainer.py:431-432).
    The kernels form P = (K@T)[:3,:] themselves (reference CPU rounding order, bbd_math.h).
    """
    tb = plan.tables(K.device)
    T_all = torch.cat([poses[job] for job in plan.pose_jobs], dim=0)
    return pose_table_rows(T_all, K.index_select(0, tb["k_rows"]), inv_K.index_select(0, tb["k_rows"]))


def pose_table_rows(T_all, K_all, iK_all):
    """[NP,40] pose-table rows from the already gathered per-row matrices ([NP,4,4] each)."""
    NP = T_all.shape[0]
    pad = torch.zeros(NP, POSE_STRIDE - 37, device=T_all.device, dtype=torch.float32)
    return torch.cat([K_all[:, :3, :].reshape(NP, 12), T_all.reshape(NP, 16),
                      iK_all[:, :3, :3].reshape(NP, 9), pad], dim=1).contiguous()


# ---------------------------------------------------------------------------- fused warp+SSIM+min
def _kt_times(K3, gP):
    """K[:3,:]^T @ dL/dP for every pose row ([NP,3,4] x [NP,3,4] -> [NP,4,4]) as a broadcast multiply + a three-term sum:
    no BLAS call on the hot path.  The batch count NP changes with every batch signature of the boosted recipe, and a batched
    GEMM of a new shape can make rocBLAS load another kernel library on the spot (1.6 s on the training thread, measured)."""
    prod = K3.unsqueeze(3) * gP.unsqueeze(2)            # [NP,3,4,1] * [NP,3,1,4] -> [NP,3,4,4]
    return (prod[:, 0] + prod[:, 1]) + prod[:, 2]


class _FusedReprojectionMin(torch.autograd.Function):
    """depth [S,B,H,W], pose table [NP,40] -> per-scale sum of the per-pixel minimum loss."""

    @staticmethod
    def forward(ctx, depth, proj, target, ident, noise, plan, frame_tensors, no_ssim, materialize, backend):
        S, B, H, W = depth.shape
        dev = depth.device
        tb = plan.tables(dev)
        ntiles = backend.num_tiles_fwd(H, W)
        depth, proj = depth.contiguous(), proj.contiguous()
        min_loss = torch.empty(S, B, H, W, device=dev, dtype=torch.float32)
        argmin = torch.empty(S, B, H, W, device=dev, dtype=torch.uint8)
        partial = torch.empty(S, B, ntiles, device=dev, dtype=torch.float32)
        warped = torch.empty(S, plan.NP, 3, H, W, device=dev, dtype=torch.float32) if materialize else None
        backend._check(depth, proj, target, ident, noise, *_frame_list(frame_tensors))
        frames = frame_pointer_array(frame_tensors)
        # pose table (K | T | inv_K) -> projection table (P | inv_K), once per step
        ptab = torch.empty(plan.NP, PROJ_STRIDE, device=dev, dtype=torch.float32)
        backend.run("bbd_pose_expand", proj, ptr(proj), ptr(ptab), plan.NP)
        backend.run("bbd_warp_ssim_min_fwd", depth, frames, ptr(target), ptr(depth), ptr(ptab), ptr(ident),
                    ptr(noise), ptr(tb["cand"]), ptr(tb["ncand"]), ptr(min_loss), ptr(argmin), ptr(partial),
                    ptr(warped), S, B, plan.NP, H, W, int(no_ssim))
        ctx.save_for_backward(depth, proj, target, argmin, ptab)
        ctx.meta = (plan, frame_tensors, frames, int(no_ssim), backend, backend.num_tiles_bwd(H, W))
        ctx.mark_non_differentiable(min_loss, argmin)
        outs = (partial.view(S, -1).sum(dim=1), min_loss, argmin)
        if materialize:
            ctx.mark_non_differentiable(warped)
            return outs + (warped,)
        return outs + (None,)

    @staticmethod
    def backward(ctx, g_sum, *_unused):
        depth, proj, target, argmin, ptab = ctx.saved_tensors
        plan, frame_tensors, frames, no_ssim, backend, ntiles = ctx.meta
        S, B, H, W = depth.shape
        dev = depth.device
        tb = plan.tables(dev)
        gscale = g_sum.contiguous().to(torch.float32)
        grad_depth = torch.empty_like(depth)
        gp_partial = torch.empty(S, plan.NP, ntiles, 12, device=dev, dtype=torch.float32)
        backend.run("bbd_warp_ssim_min_bwd", depth, frames, ptr(target), ptr(depth), ptr(ptab), ptr(tb["cand"]),
                    ptr(tb["ncand"]), ptr(argmin), ptr(gscale), ptr(grad_depth), ptr(gp_partial),
                    S, B, plan.NP, H, W, no_ssim)
        # dL/dT = K[:3,:]^T @ dL/dP  (P = (K@T)[:3,:]); K and inv_K columns get no gradient
        gP = gp_partial.sum(dim=(0, 2)).view(plan.NP, 3, 4)
        gT = _kt_times(proj[:, :12].view(plan.NP, 3, 4), gP)
        grad_pose = torch.zeros_like(proj)
        grad_pose[:, 12:28] = gT.reshape(plan.NP, 16)
        return grad_depth, grad_pose, None, None, None, None, None, None, None, None


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        assert t.is_contiguous() and t.dtype == torch.float32
        arr[i] = t.data_ptr()
    return arr


def _hw_array(tensors):
    arr = (ctypes.c_int32 * (2 * len(tensors)))()
    for i, t in enumerate(tensors):
        arr[2 * i], arr[2 * i + 1] = t.shape[-2], t.shape[-1]
    return arr


class _FusedReprojectionMinDisp(torch.autograd.Function):
    """The fused launch fed with the decoder's disparity maps themselves: bilinear up-sampling +
    disp_to_depth happen per staged pixel inside the kernels, no depth buffer is read, and the backward folds
    d depth / d disparity in; only the bilinear adjoint of the reduced scales is a second (single) launch.
    Returns (loss_sum [S], min_loss, argmin, warped | None, depth [S,B,H,W] | None)."""

    @staticmethod
    def forward(ctx, proj, target, ident, noise, plan, frame_tensors, no_ssim, materialize, want_depth, min_depth,
                max_depth, backend, *disps):
        disps = [d.contiguous() for d in disps]
        S, B = len(disps), disps[0].shape[0]
        H, W = target.shape[-2:]
        dev = target.device
        tb = plan.tables(dev)
        ntiles = backend.num_tiles_fwd(H, W)
        proj = proj.contiguous()
        min_loss = torch.empty(S, B, H, W, device=dev, dtype=torch.float32)
        argmin = torch.empty(S, B, H, W, device=dev, dtype=torch.uint8)
        partial = torch.empty(S, B, ntiles, device=dev, dtype=torch.float32)
        warped = torch.empty(S, plan.NP, 3, H, W, device=dev, dtype=torch.float32) if materialize else None
        depth = torch.empty(S, B, H, W, device=dev, dtype=torch.float32) if want_depth else None
        backend._check(proj, target, ident, noise, *disps, *_frame_list(frame_tensors))
        frames = frame_pointer_array(frame_tensors)
        ptab = torch.empty(plan.NP, PROJ_STRIDE, device=dev, dtype=torch.float32)
        backend.run("bbd_pose_expand", proj, ptr(proj), ptr(ptab), plan.NP)
        backend.run("bbd_warp_ssim_min_disp_fwd", target, frames, ptr(target), _ptr_array(disps), _hw_array(disps),
                    float(min_depth), float(max_depth), ptr(ptab), ptr(ident), ptr(noise), ptr(tb["cand"]),
                    ptr(tb["ncand"]), ptr(backend.fused_work_items(plan, S, H, W, False, dev)), ptr(min_loss), ptr(argmin),
                    ptr(partial), ptr(warped), ptr(depth), S, B, plan.NP, H, W, int(no_ssim))
        # the depth by-product is handed to the caller (outputs[("depth",0,s)], NOT differentiable: a depth-based
        # regulariser must use disp_to_depth / opt.fused_disp=False) AND read by the backward: saving it
        # through autograd makes an in-place edit by the caller an error instead of a silently corrupted gradient
        ctx.has_depth = depth is not None
        ctx.save_for_backward(proj, target, argmin, ptab, *([depth] if depth is not None else []), *disps)
        ctx.meta = (plan, frame_tensors, frames, int(no_ssim), backend, float(min_depth), float(max_depth))
        ctx.mark_non_differentiable(min_loss, argmin)
        if materialize:
            ctx.mark_non_differentiable(warped)
        if want_depth:
            ctx.mark_non_differentiable(depth)
        return partial.view(S, -1).sum(dim=1), min_loss, argmin, warped, depth

    @staticmethod
    def backward(ctx, g_sum, *_unused):
        proj, target, argmin, ptab = ctx.saved_tensors[:4]
        depth = ctx.saved_tensors[4] if ctx.has_depth else None
        disps = ctx.saved_tensors[5:] if ctx.has_depth else ctx.saved_tensors[4:]
        plan, frame_tensors, frames, no_ssim, backend, lo, hi = ctx.meta
        S, B = len(disps), disps[0].shape[0]
        H, W = target.shape[-2:]
        dev = target.device
        tb = plan.tables(dev)
        gscale = g_sum.contiguous().to(torch.float32)
        grad_up = torch.empty(S, B, H, W, device=dev, dtype=torch.float32)
        ntb = backend.num_tiles_bwd(H, W)
        # (a plan view over static tables has more pose-table rows than the step uses: the launch never writes those)
        make = torch.zeros if getattr(plan, "zero_partials", False) else torch.empty
        gp_partial = make(S, plan.NP, ntb, 12, device=dev, dtype=torch.float32)
        backend.run("bbd_warp_ssim_min_disp_bwd", target, frames, ptr(target), _ptr_array(disps), _hw_array(disps), lo, hi,
                    ptr(depth), ptr(ptab), ptr(tb["cand"]), ptr(tb["ncand"]), ptr(backend.fused_work_items(plan, S, H, W, True, dev)),
                    ptr(argmin), ptr(gscale), ptr(grad_up), ptr(gp_partial), S, B, plan.NP, H, W, no_ssim)
        # a scale at full resolution: grad_up IS its disparity gradient; the reduced ones share one adjoint launch
        grads, small, small_g, small_up = [None] * S, [], [], []
        for i, d in enumerate(disps):
            if tuple(d.shape[-2:]) == (H, W):
                grads[i] = grad_up[i].unsqueeze(1)
            else:
                grads[i] = torch.empty_like(d)
                small.append(d)
                small_g.append(grads[i])
                small_up.append(grad_up[i])
        if small:
            backend.run("bbd_disp_upsample_adjoint", grad_up, _ptr_array(small_up), _hw_array(small), _ptr_array(small_g),
                        len(small), B, H, W)
        gP = gp_partial.sum(dim=(0, 2)).view(plan.NP, 3, 4)
        gT = _kt_times(proj[:, :12].view(plan.NP, 3, 4), gP)
        grad_pose = torch.zeros_like(proj)
        grad_pose[:, 12:28] = gT.reshape(plan.NP, 16)
        return (grad_pose,) + (None,) * 11 + tuple(grads)


def fused_reprojection_min_disp(disps, proj, target, ident, noise, plan, frame_tensors, min_depth, max_depth,
                                no_ssim=False, materialize=False, want_depth=True, backend=None):
    """[("disp", s)] maps in, (loss_sum [S], min_loss, argmin, warped | None, depth [S,B,H,W] | None) out."""
    return _FusedReprojectionMinDisp.apply(proj, target, ident, noise, plan, frame_tensors, bool(no_ssim),
                                           bool(materialize), bool(want_depth), min_depth, max_depth,
                                           backend or default_backend(), *disps)


def fused_reprojection_min(depth, proj, target, ident, noise, plan, frame_tensors, no_ssim=False,
                           materialize=False, backend=None):
    """Returns (loss_sum [S], min_loss [S,B,H,W], argmin u8 [S,B,H,W], warped [S,NP,3,H,W] | None)."""
    return _FusedReprojectionMin.apply(depth, proj, target, ident, noise, plan, frame_tensors, bool(no_ssim),
                                       bool(materialize), backend or default_backend())


# ---------------------------------------------------------------------------- stand-alone layers (autograd)
class _Backproject(torch.autograd.Function):
    """layers.BackprojectDepth.forward (layers.py:160-167) with its closed-form backward."""

    @staticmethod
    def forward(ctx, depth, inv_K, H, W, backend):
        n = len(inv_K)
        depth, inv_K = depth.contiguous(), inv_K.contiguous()
        backend._check(depth, inv_K)
        pts = torch.empty(n, 4, H * W, device=depth.device, dtype=torch.float32)
        backend.run("bbd_backproject_fwd", depth, ptr(depth), ptr(inv_K), ptr(pts), n, H, W)
        ctx.save_for_backward(inv_K)
        ctx.meta = (n, H, W, depth.shape, backend)
        return pts

    @staticmethod
    def backward(ctx, grad_points):
        (inv_K,) = ctx.saved_tensors
        n, H, W, shape, backend = ctx.meta
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("BackprojectDepth: no gradient w.r.t. inv_K (camera intrinsics are constants "
                                      "of the reference's training path)")
        grad_points = grad_points.contiguous()
        grad_depth = torch.empty(n, H, W, device=grad_points.device, dtype=torch.float32)
        backend.run("bbd_backproject_bwd", grad_points, ptr(grad_points), ptr(inv_K), ptr(grad_depth), n, H, W)
        return grad_depth.view(shape), None, None, None, None


def backproject(depth, inv_K, H, W, backend=None):
    return _Backproject.apply(depth, inv_K, H, W, backend or default_backend())


class _Project3D(torch.autograd.Function):
    """layers.Project3D.forward (layers.py:181-195); backward gives d points, d T and d K."""

    @staticmethod
    def forward(ctx, points, K, T, H, W, eps, backend):
        n = len(K)
        points, K, T = points.contiguous(), K.contiguous(), T.contiguous()
        backend._check(points, K, T)
        grid = torch.empty(n, H, W, 2, device=points.device, dtype=torch.float32)
        backend.run("bbd_project3d_fwd", points, ptr(points), ptr(K), ptr(T), ptr(grid), n, H, W, float(eps))
        ctx.save_for_backward(points, K, T)
        ctx.meta = (n, H, W, float(eps), backend)
        return grid

    @staticmethod
    def backward(ctx, grad_grid):
        points, K, T = ctx.saved_tensors
        n, H, W, eps, backend = ctx.meta
        grad_grid = grad_grid.contiguous()
        grad_points = torch.empty_like(points)
        blocks = backend.lib.project3d_bwd_blocks
        gp_partial = torch.empty(n, blocks, 12, device=points.device, dtype=torch.float32)
        backend.run("bbd_project3d_bwd", points, ptr(points), ptr(K), ptr(T), ptr(grad_grid), ptr(grad_points),
                    ptr(gp_partial), n, H, W, eps)
        gP = gp_partial.sum(dim=1).view(n, 3, 4)                    # dL/dP,  P = (K @ T)[:3, :]
        grad_K = grad_T = None
        if ctx.needs_input_grad[1]:
            grad_K = torch.zeros_like(K)
            grad_K[:, :3, :] = torch.matmul(gP, T.transpose(1, 2))
        if ctx.needs_input_grad[2]:
            grad_T = torch.matmul(K[:, :3, :].transpose(1, 2), gP)
        return grad_points, grad_K, grad_T, None, None, None, None


def project3d(points, K, T, H, W, eps=1e-7, backend=None):
    return _Project3D.apply(points, K, T, H, W, eps, backend or default_backend())


class _SSIMMap(torch.autograd.Function):
    """layers.SSIM.forward (layers.py:235-249), [n,3k,H,W] x2 -> same shape; differentiable in both."""

    @staticmethod
    def forward(ctx, x, y, backend):
        x, y = x.contiguous(), y.contiguous()
        backend._check(x, y)
        n, c, H, W = x.shape
        assert (n * c) % 3 == 0, "SSIM kernels work on groups of three colour planes"
        out = torch.empty_like(x)
        backend.run("bbd_ssim_fwd", x, ptr(x), ptr(y), ptr(out), (n * c) // 3, H, W)
        ctx.save_for_backward(x, y)
        ctx.backend = backend
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, y = ctx.saved_tensors
        backend = ctx.backend
        n, c, H, W = x.shape
        grad_out = grad_out.contiguous()
        gx = gy = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            backend.run("bbd_ssim_bwd", x, ptr(x), ptr(y), ptr(grad_out), ptr(gx), (n * c) // 3, H, W)
        if ctx.needs_input_grad[1]:       # the SSIM expression is symmetric in its two arguments
            gy = torch.empty_like(y)
            backend.run("bbd_ssim_bwd", x, ptr(y), ptr(x), ptr(grad_out), ptr(gy), (n * c) // 3, H, W)
        return gx, gy, None


def ssim_map(x, y, backend=None):
    return _SSIMMap.apply(x, y, backend or default_backend())


# ---------------------------------------------------------------------------- MonoViT: depth-wise conv on tokens
def _slice_ptr(t, c0):
    return ctypes.c_void_p(t.data_ptr() + 4 * c0)


class SharedGrads:
    """Gradient accumulation for parameters that several layers of one sequential chain share (MPViT's MHCAEncoder: the
    blocks of a path share one ConvPosEnc and one ConvRelPosEnc).  As plain autograd every extra use costs an
    AccumulateGrad add per parameter - ~270 tiny launches per MonoViT step, each a cross-stream fork/join inside the
    step graph.  Here the layers' weight-gradient launches add into ONE buffer per parameter (the kernel's `accumulate`
    mode; backward runs the layers last to first, on one stream) and only the chain's first layer hands it to autograd."""

    def __init__(self):
        self.buf = {}
        self.last = {}

    def target(self, key, like, index, count):
        """(buffer to write the gradient into, accumulate flag) for layer `index` of `count`.  One sink belongs to ONE
        forward pass of the chain (MHCAEncoder.forward makes a fresh pair per call), so two forwards before a backward, or
        a second backward through a retained graph, cannot meet in one buffer; inside a pass the layers must arrive
        last to first - anything else (nodes of a partial autograd.grad pass interleaved out of order) is refused rather
        than summed wrongly."""
        if key in self.buf and index >= self.last[key]:
            raise RuntimeError("shared position-encoding gradients: layer %d after layer %d in one backward pass "
                               "(set BBD_FUSED_TOKEN_GLUE=0 for per-layer gradients)" % (index, self.last[key]))
        start = key not in self.buf                          # the layer whose backward runs first starts the sum
        if start:
            self.buf[key] = torch.empty_like(like)
        self.last[key] = index
        return self.buf[key], 0 if start else 1

    def result(self, key, index):
        """What the layer returns to autograd for this parameter: the finished sum from layer 0, nothing from the others."""
        if index != 0:
            return None
        self.last.pop(key, None)
        return self.buf.pop(key)


def _wgrad_target(share, tag, like):
    if share is None:
        return torch.empty_like(like), 0
    sink, index, count = share
    return sink.target(tag, like, index, count)


def _wgrad_result(share, tag, tensor):
    if share is None:
        return tensor
    sink, index, _ = share
    return sink.result(tag, index)


def _i32_array(values):
    return (ctypes.c_int32 * len(values))(*[int(v) for v in values])


def _addr_array(tensors):
    """Host array of device addresses (0 for None)."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        if t is not None:
            assert t.is_contiguous() and t.dtype == torch.float32
            arr[i] = t.data_ptr()
    return arr


def _dw_group_args(splits, weights):
    c0, at = [], 0
    for n in splits:
        c0.append(at)
        at += n
    return len(splits), _i32_array(c0), _i32_array(splits), _i32_array([w.shape[-1] for w in weights])


def _dwconv_groups_fwd(backend, x, x_base, x_row, y, y_base, y_row, splits, weights, biases, B, H, W, add_input, flip):
    """One launch for all channel groups (several window sizes); `*_base`: first channel of the slice inside the rows."""
    n, c0, cn, k = _dw_group_args(splits, weights)
    backend.run("bbd_dwconv_tokens_groups_fwd", x, _slice_ptr(x, x_base), x_row, _slice_ptr(y, y_base), y_row, n, c0, cn, k,
                _addr_array(weights), _addr_array(biases), B, H, W, int(add_input), int(flip))


def _dwconv_groups_wgrad(backend, x, x_base, x_row, gy, gy_base, gy_row, splits, weights, gws, gbs, B, H, W, accumulate):
    n, c0, cn, k = _dw_group_args(splits, weights)
    scratch = torch.empty(backend.lib.dwconv_groups_wgrad_scratch_floats(B, H, W, n, cn, k), device=x.device, dtype=torch.float32)
    backend.run("bbd_dwconv_tokens_groups_wgrad", x, _slice_ptr(x, x_base), x_row, _slice_ptr(gy, gy_base), gy_row, ptr(scratch),
                n, c0, cn, k, _addr_array(gws), _addr_array(gbs), B, H, W, int(accumulate))


class _DepthwiseTokens(torch.autograd.Function):
    """Depth-wise k x k convolutions over token-layout activations [B, H*W, C] (csrc/bbd_vit.hip): channel
    groups `splits` use their own (weight [n,1,k,k], bias [n]) - MPViT's ConvRelPosEnc gives head groups
    windows 3 / 5 / 7, ConvPosEnc is one group with the residual folded in (`add_input`).  `x` may be a
    channel-slice view of wider rows (v inside the packed qkv activation)."""

    @staticmethod
    def forward(ctx, x, H, W, add_input, backend, splits, share, *params):
        B, N, C = x.shape
        assert N == H * W and sum(splits) == C and x.dtype == torch.float32
        if x.stride(2) != 1 or x.stride(0) != N * x.stride(1):
            x = x.contiguous()
        backend._check(x, *params)
        y = torch.empty(B, N, C, device=x.device, dtype=torch.float32)
        if len(splits) > 1:
            _dwconv_groups_fwd(backend, x, 0, x.stride(1), y, 0, C, splits, [params[2 * i].contiguous() for i in range(len(splits))],
                               [params[2 * i + 1] for i in range(len(splits))], B, H, W, add_input, 0)
        else:
            w, b = params[0].contiguous(), params[1]
            backend.run("bbd_dwconv_tokens_fwd", x, ptr(x) if x.is_contiguous() else _slice_ptr(x, 0), x.stride(1), ptr(w), ptr(b),
                        ptr(y), C, B, H, W, C, w.shape[-1], int(add_input), 0)
        ctx.save_for_backward(x, *params)
        ctx.meta = (H, W, bool(add_input), backend, tuple(splits), share)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, params = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        H, W, add_input, backend, splits, share = ctx.meta
        B, N, C = x.shape
        gy = gy.contiguous()
        gx = torch.empty(B, N, C, device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        ws = [params[2 * i].contiguous() for i in range(len(splits))]
        bs = [params[2 * i + 1] for i in range(len(splits))]
        gws, gbs, acc = [], [], 0
        for i, (w, b) in enumerate(zip(ws, bs)):
            gw, acc = _wgrad_target(share, ("w", i), w)
            gws.append(gw)
            gbs.append(_wgrad_target(share, ("b", i), b)[0] if b is not None else None)
        if len(splits) > 1:
            if gx is not None:
                _dwconv_groups_fwd(backend, gy, 0, C, gx, 0, C, splits, ws, [None] * len(ws), B, H, W, add_input, 1)
            _dwconv_groups_wgrad(backend, x, 0, x.stride(1), gy, 0, C, splits, ws, gws, gbs, B, H, W, acc)
        else:
            k = ws[0].shape[-1]
            if gx is not None:
                backend.run("bbd_dwconv_tokens_fwd", gy, ptr(gy), C, ptr(ws[0]), ptr(None), ptr(gx), C, B, H, W, C, k, int(add_input), 1)
            scratch = torch.empty(backend.lib.dwconv_wgrad_scratch_floats(B, H, W, C, k), device=x.device, dtype=torch.float32)
            backend.run("bbd_dwconv_tokens_wgrad", x, _slice_ptr(x, 0), x.stride(1), ptr(gy), C, ptr(scratch), ptr(gws[0]), ptr(gbs[0]),
                        B, H, W, C, k, acc)
        grads = []
        for i, b in enumerate(bs):
            grads += [_wgrad_result(share, ("w", i), gws[i]), _wgrad_result(share, ("b", i), gbs[i]) if b is not None else None]
        return (gx, None, None, None, None, None, None) + tuple(grads)


def dwconv_tokens(x, size, convs, add_input=False, backend=None, share=None):
    """`convs`: list of nn.Conv2d (depth-wise, stride 1, padding k//2) covering consecutive channel groups.
    `share` = (SharedGrads, layer index, layer count) when the convs' parameters are shared by a chain of layers."""
    params, splits = [], []
    for conv in convs:
        params += [conv.weight, conv.bias]
        splits.append(conv.weight.shape[0])
    return _DepthwiseTokens.apply(x, size[0], size[1], add_input, backend or default_backend(), tuple(splits), share, *params)


class _FactorAttention(torch.autograd.Function):
    """MPViT's factorised attention on the packed qkv activation (csrc/bbd_vit.hip):
    out = q (scale * softmax_N(k)^T v) + q * convv, per head.  qkv [B,N,3C] is the qkv Linear's output,
    convv [B,N,C] the ConvRelPosEnc convolution of v."""

    @staticmethod
    def forward(ctx, qkv, convv, heads, scale, backend):
        B, N, C3 = qkv.shape
        C = C3 // 3
        Ch = C // heads
        qkv, convv = qkv.contiguous(), convv.contiguous()
        backend._check(qkv, convv)
        dev = qkv.device
        kmax = torch.empty(B, C, device=dev, dtype=torch.float32)
        krsum = torch.empty(B, C, device=dev, dtype=torch.float32)
        ctxs = torch.empty(B, C * Ch, device=dev, dtype=torch.float32)
        scratch = torch.empty(backend.lib.factor_att_scratch_floats(B, N, C, Ch), device=dev, dtype=torch.float32)
        out = torch.empty(B, N, C, device=dev, dtype=torch.float32)
        backend.run("bbd_factor_att_fwd", qkv, ptr(qkv), ptr(convv), ptr(kmax), ptr(krsum), ptr(ctxs), ptr(scratch),
                    ptr(out), B, N, C, Ch, float(scale))
        ctx.save_for_backward(qkv, convv, kmax, krsum, ctxs)
        ctx.meta = (heads, float(scale), backend)
        return out

    @staticmethod
    def backward(ctx, gout):
        qkv, convv, kmax, krsum, ctxs = ctx.saved_tensors
        heads, scale, backend = ctx.meta
        B, N, C3 = qkv.shape
        C = C3 // 3
        Ch = C // heads
        gout = gout.contiguous()
        dev = qkv.device
        dctx = torch.empty_like(ctxs)
        scratch = torch.empty(backend.lib.factor_att_scratch_floats(B, N, C, Ch), device=dev, dtype=torch.float32)
        gqkv = torch.empty_like(qkv)
        gconvv = torch.empty_like(convv)
        backend.run("bbd_factor_att_bwd", qkv, ptr(qkv), ptr(convv), ptr(kmax), ptr(krsum), ptr(ctxs), ptr(gout),
                    ptr(dctx), ptr(scratch), ptr(gqkv), ptr(gconvv), B, N, C, Ch, scale)
        return gqkv, gconvv, None, None, None


class _FactorAttentionCRPE(torch.autograd.Function):
    """`_FactorAttention` with the ConvRelPosEnc convolution of v inside the node: v is read in place from the packed qkv
    activation and - the point - its data gradient is ADDED in place to the v third of the attention's gqkv.  As two
    nodes autograd materialises the slice's gradient (a zero fill of [B,N,3C], a strided copy) and adds the two
    [B,N,3C] tensors: three passes over the widest activation of the block, per block."""

    @staticmethod
    def forward(ctx, qkv, heads, scale, H, W, backend, splits, share, *params):
        B, N, C3 = qkv.shape
        C = C3 // 3
        Ch = C // heads
        assert N == H * W and sum(splits) == C
        qkv = qkv.contiguous()
        backend._check(qkv, *params)
        dev = qkv.device
        convv = torch.empty(B, N, C, device=dev, dtype=torch.float32)
        _dwconv_groups_fwd(backend, qkv, 2 * C, C3, convv, 0, C, splits, [params[2 * i].contiguous() for i in range(len(splits))],
                           [params[2 * i + 1] for i in range(len(splits))], B, H, W, 0, 0)
        kmax = torch.empty(B, C, device=dev, dtype=torch.float32)
        krsum = torch.empty(B, C, device=dev, dtype=torch.float32)
        ctxs = torch.empty(B, C * Ch, device=dev, dtype=torch.float32)
        scratch = torch.empty(backend.lib.factor_att_scratch_floats(B, N, C, Ch), device=dev, dtype=torch.float32)
        out = torch.empty(B, N, C, device=dev, dtype=torch.float32)
        backend.run("bbd_factor_att_fwd", qkv, ptr(qkv), ptr(convv), ptr(kmax), ptr(krsum), ptr(ctxs), ptr(scratch),
                    ptr(out), B, N, C, Ch, float(scale))
        ctx.save_for_backward(qkv, convv, kmax, krsum, ctxs, *params)
        ctx.meta = (heads, float(scale), H, W, backend, tuple(splits), share)
        return out

    @staticmethod
    def backward(ctx, gout):
        qkv, convv, kmax, krsum, ctxs = ctx.saved_tensors[:5]
        params = ctx.saved_tensors[5:]
        heads, scale, H, W, backend, splits, share = ctx.meta
        B, N, C3 = qkv.shape
        C = C3 // 3
        Ch = C // heads
        gout = gout.contiguous()
        dev = qkv.device
        dctx = torch.empty_like(ctxs)
        scratch = torch.empty(backend.lib.factor_att_scratch_floats(B, N, C, Ch), device=dev, dtype=torch.float32)
        gqkv = torch.empty_like(qkv)
        gconvv = torch.empty_like(convv)
        backend.run("bbd_factor_att_bwd", qkv, ptr(qkv), ptr(convv), ptr(kmax), ptr(krsum), ptr(ctxs), ptr(gout),
                    ptr(dctx), ptr(scratch), ptr(gqkv), ptr(gconvv), B, N, C, Ch, scale)
        ws = [params[2 * i].contiguous() for i in range(len(splits))]
        bs = [params[2 * i + 1] for i in range(len(splits))]
        gws, gbs, acc = [], [], 0
        for i, (w, b) in enumerate(zip(ws, bs)):
            gw, acc = _wgrad_target(share, ("w", i), w)
            gws.append(gw)
            gbs.append(_wgrad_target(share, ("b", i), b)[0] if b is not None else None)
        # data gradient of the convolutions, accumulated into the v third of gqkv; then the weight gradients - one launch each
        _dwconv_groups_fwd(backend, gconvv, 0, C, gqkv, 2 * C, C3, splits, ws, [None] * len(ws), B, H, W, 2, 1)
        _dwconv_groups_wgrad(backend, qkv, 2 * C, C3, gconvv, 0, C, splits, ws, gws, gbs, B, H, W, acc)
        grads = []
        for i, b in enumerate(bs):
            grads += [_wgrad_result(share, ("w", i), gws[i]), _wgrad_result(share, ("b", i), gbs[i]) if b is not None else None]
        return (gqkv, None, None, None, None, None, None, None) + tuple(grads)


def factor_attention_crpe(qkv, size, convs, heads, scale, backend=None, share=None):
    """Factorised attention with MPViT's ConvRelPosEnc of v computed inside (`convs`: its depth-wise nn.Conv2d list;
    `share`: see dwconv_tokens)."""
    params, splits = [], []
    for conv in convs:
        params += [conv.weight, conv.bias]
        splits.append(conv.weight.shape[0])
    return _FactorAttentionCRPE.apply(qkv, heads, scale, size[0], size[1], backend or default_backend(), tuple(splits), share,
                                      *params)


def factor_attention(qkv, convv, heads, scale, backend=None):
    return _FactorAttention.apply(qkv, convv, heads, scale, backend or default_backend())


def factor_attention_supported(C, heads, backend=None):
    return (backend or default_backend()).lib.factor_att_supported(C, C // heads)


class _ResidualLayerNorm(torch.autograd.Function):
    """y = x + branch * mask[b];  z = LayerNorm(y)  on [B, N, C] tokens in one pass each way (csrc/bbd_tokens.hip).
    `branch is None`: plain LayerNorm of x (returns z only).  `mask`: [B] stochastic-depth scale (0 or 1/keep) or None."""

    @staticmethod
    def forward(ctx, x, branch, mask, weight, bias, eps, backend, passthrough=False):
        B, N, C = x.shape
        x = x.contiguous()
        backend._check(x, branch, mask, weight, bias)
        if branch is not None:
            branch = branch.contiguous()
        y = torch.empty_like(x) if branch is not None else x
        z = torch.empty_like(x)
        stats = torch.empty(B * N, 2, device=x.device, dtype=torch.float32)
        backend.run("bbd_token_ln_fwd", x, ptr(x), ptr(branch), ptr(mask), ptr(weight), ptr(bias),
                    ptr(y) if branch is not None else ptr(None), ptr(z), ptr(stats), B * N, N, C, float(eps))
        ctx.save_for_backward(y, stats, weight, mask)
        ctx.meta = (backend, branch is not None, bool(passthrough))
        if branch is not None:
            return y, z
        # `passthrough`: x is handed back as an OUTPUT of this node (a view), so a caller that goes on using it (the
        # residual) sends that gradient here and the kernel adds it to LayerNorm's - no separate add
        return (x.view_as(x), z) if passthrough else z

    @staticmethod
    def backward(ctx, *grads):
        y, stats, weight, mask = ctx.saved_tensors
        backend, has_branch, passthrough = ctx.meta
        gy, gz = (grads if (has_branch or passthrough) else (None, grads[0]))
        B, N, C = y.shape
        if gz is None:                     # z unused downstream: only the residual carries gradient
            gz = torch.zeros_like(y)
        gz = gz.contiguous()
        gy = gy.contiguous() if gy is not None else None
        gx = torch.empty_like(y)
        gbranch = torch.empty_like(y) if has_branch else None
        gw, gb = torch.empty_like(weight), torch.empty_like(weight)
        scratch = torch.empty(backend.lib.token_ln_scratch_floats(B * N, C), device=y.device, dtype=torch.float32)
        backend.run("bbd_token_ln_bwd", gz, ptr(gz), ptr(gy), ptr(y), ptr(stats), ptr(weight), ptr(mask), ptr(gx), ptr(gbranch),
                    ptr(scratch), ptr(gw), ptr(gb), B * N, N, C)
        return gx, gbranch, None, gw, gb, None, None, None


class _ResidualAdd(torch.autograd.Function):
    """y = x + branch * mask[b] (one launch instead of a broadcast multiply and an add)."""

    @staticmethod
    def forward(ctx, x, branch, mask, backend):
        B, N, C = x.shape
        x, branch = x.contiguous(), branch.contiguous()
        backend._check(x, branch, mask)
        y = torch.empty_like(x)
        backend.run("bbd_token_ln_fwd", x, ptr(x), ptr(branch), ptr(mask), ptr(None), ptr(None), ptr(y), ptr(None), ptr(None),
                    B * N, N, C, 0.0)
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, gy):
        (mask,) = ctx.saved_tensors
        return gy, (gy if mask is None else gy * mask.view(-1, 1, 1)), None, None


class _LinearTokens(torch.autograd.Function):
    """nn.Linear on [..., C_in] tokens with the bias gradient from `bbd_colsum` (two short launches) instead of ATen's
    generic reduction of the tall-skinny [tokens, C_out] gradient; the three GEMMs stay with hipBLASLt / rocBLAS."""

    @staticmethod
    def forward(ctx, x, weight, bias, backend):
        ctx.save_for_backward(x, weight)
        ctx.meta = (backend, bias is not None)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        backend, has_bias = ctx.meta
        g2 = g.reshape(-1, g.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        gx = g2.mm(weight).view(x.shape) if ctx.needs_input_grad[0] else None
        gw = g2.t().mm(x.reshape(-1, x.shape[-1])) if ctx.needs_input_grad[1] else None
        gb = None
        if has_bias and ctx.needs_input_grad[2]:
            rows, C = g2.shape
            gb = torch.empty(C, device=g2.device, dtype=torch.float32)
            scratch = torch.empty(backend.lib.colsum_scratch_floats(rows, C), device=g2.device, dtype=torch.float32)
            backend.run("bbd_colsum", g2, ptr(g2), ptr(scratch), ptr(gb), rows, C)
        return gx, gw, gb, None


def linear_tokens(x, linear, backend=None):
    """`linear(x)` for an nn.Linear on GPU fp32 tokens whose output width is a multiple of 4 (else the module itself)."""
    if not (x.is_cuda and x.dtype == torch.float32 and FUSED_TOKEN_GLUE and linear.out_features % 4 == 0):
        return linear(x)
    return _LinearTokens.apply(x, linear.weight, linear.bias, backend or default_backend())


def token_glue_supported(x, backend=None):
    return x.dim() == 3 and (backend or default_backend()).lib.token_ln_supported(x.shape[-1])


def layernorm_tokens(x, norm, backend=None, passthrough=False):
    """nn.LayerNorm `norm` over the channel axis of [B, N, C] tokens; `passthrough`: returns (x, norm(x)) with x routed
    through the node (see _ResidualLayerNorm.forward)."""
    return _ResidualLayerNorm.apply(x, None, None, norm.weight, norm.bias, norm.eps, backend or default_backend(), passthrough)


def residual_layernorm(x, branch, mask, norm, backend=None):
    """(x + branch * mask[b], norm(of that)) - `mask` [B] or None."""
    return _ResidualLayerNorm.apply(x, branch, mask, norm.weight, norm.bias, norm.eps, backend or default_backend())


def residual_add(x, branch, mask, backend=None):
    return _ResidualAdd.apply(x, branch, mask, backend or default_backend())


# ---------------------------------------------------------------------------- BatchNorm (+add) (+ReLU)
BN_MAX_GROUPS = 32          # BBD_BN_MAX_GROUPS of include/bbd_hip.h
_bn_groups = None           # row counts of the call groups of the batched pass being run (None = one group)
_bn_untracked = 0           # trailing groups of it that are padding (no running-statistics update)
_bn_device = None           # (device group table, G, largest group) of `bn_call_groups_device`, or None


class bn_call_groups:
    """`with ops.bn_call_groups([n_0, n_1, ...]):` - every fused BatchNorm inside treats its batch as consecutive
    call groups of n_g samples with their own batch statistics: one batched pass of a network computes what the
    reference's separate calls on the sub-batches compute (the pose network: trainer.py:348-418)."""

    def __init__(self, rows, padding_groups=0):
        """`padding_groups`: the last that many groups are padding rows - normalised, but the running statistics and
        num_batches_tracked see only the real calls (bbd_bn_act_grouped_fwd, `untracked_groups`)."""
        self.rows = [int(r) for r in rows] if rows is not None and len(rows) > 1 else None
        self.untracked = int(padding_groups) if self.rows is not None else 0
        assert self.rows is None or (len(self.rows) <= BN_MAX_GROUPS and min(self.rows) > 0)
        assert 0 <= self.untracked < max(len(self.rows or [0]), 1)

    def __enter__(self):
        global _bn_groups, _bn_untracked
        self.prev, _bn_groups = (_bn_groups, _bn_untracked), self.rows
        _bn_untracked = self.untracked
        return self

    def __exit__(self, *exc):
        global _bn_groups, _bn_untracked
        _bn_groups, _bn_untracked = self.prev


class bn_call_groups_device:
    """`with ops.bn_call_groups_device(table, G, max_rows):` - the call groups of `bn_call_groups`, read from a DEVICE table
    (int32 [BN_MAX_GROUPS + 2]: rows[0..G], number of tracked groups at index BN_MAX_GROUPS + 1; groups may be empty) by
    `bbd_bn_act_grouped_dev_*`: nothing of the batch signature is in the launch arguments, so one captured step graph
    serves every ordering with the same padded row count (`pooled.PooledStep`).  `max_rows` bounds every group."""

    def __init__(self, table, groups, max_rows):
        assert table.dtype == torch.int32 and table.numel() >= BN_MAX_GROUPS + 2 and 1 <= groups <= BN_MAX_GROUPS
        self.spec = (table, int(groups), int(max_rows))

    def __enter__(self):
        global _bn_device
        self.prev, _bn_device = _bn_device, self.spec
        return self

    def __exit__(self, *exc):
        global _bn_device
        _bn_device = self.prev


def _group_table(rows):
    arr = (ctypes.c_int32 * (len(rows) + 1))()
    for i, r in enumerate(rows):
        arr[i + 1] = arr[i] + r
    return arr


class _BatchNormAct(torch.autograd.Function):
    """Training-mode BatchNorm2d, optional residual add and ReLU in two launches each way
    (csrc/bbd_nn.hip) - the tail of every ResNet block of the encoders."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, batches, momentum, eps, relu, backend):
        x = x.contiguous()
        N, C, H, W = x.shape
        if residual is not None:
            residual = residual.contiguous()
            assert residual.shape == x.shape
        backend._check(x, weight, bias, residual, running_mean, running_var)
        y = torch.empty_like(x)
        if _bn_device is not None:
            # group table on the device: the launch arguments hold nothing of the batch signature
            table, G, biggest = _bn_device
            assert _bn_groups is None and biggest <= N
            rows = (table, G, biggest)
            mean = torch.empty(G, C, device=x.device, dtype=torch.float32)
            invstd = torch.empty(G, C, device=x.device, dtype=torch.float32)
            scratch = torch.empty(backend.lib.bn_grouped_scratch_doubles(biggest, G, C, H * W), device=x.device,
                                  dtype=torch.float64)
            backend.run("bbd_bn_act_grouped_dev_fwd", x, ptr(x), ptr(residual), ptr(weight), ptr(bias), ptr(y), ptr(mean),
                        ptr(invstd), ptr(running_mean), ptr(running_var), ptr(batches), ptr(scratch), ptr(table), G,
                        biggest, N, C, H * W, float(eps), float(momentum), int(relu))
            ctx.save_for_backward(x, y if (relu and residual is not None) else None, weight, bias, mean, invstd)
            ctx.meta = (bool(relu), residual is not None, backend, rows)
            return y
        rows = _bn_groups if _bn_groups is not None else [N]
        assert sum(rows) == N, "ops.bn_call_groups does not describe this batch (%s vs %d rows)" % (rows, N)
        G = len(rows)
        mean = torch.empty(G, C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(G, C, device=x.device, dtype=torch.float32)
        scratch = torch.empty(backend.lib.bn_grouped_scratch_doubles(max(rows), G, C, H * W), device=x.device,
                              dtype=torch.float64)
        backend.run("bbd_bn_act_grouped_fwd", x, ptr(x), ptr(residual), ptr(weight), ptr(bias), ptr(y), ptr(mean),
                    ptr(invstd), ptr(running_mean), ptr(running_var), ptr(batches), ptr(scratch), _group_table(rows), G,
                    _bn_untracked if _bn_groups is not None else 0, N, C, H * W, float(eps), float(momentum), int(relu))
        # without a residual the backward re-derives the ReLU mask from x (one activation read less per launch)
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, weight, bias, mean, invstd)
        ctx.meta = (bool(relu), residual is not None, backend, rows)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, y, weight, bias, mean, invstd = ctx.saved_tensors
        relu, has_res, backend, rows = ctx.meta
        N, C, H, W = x.shape
        grad_y = grad_y.contiguous()
        grad_x = torch.empty_like(x)
        grad_res = torch.empty_like(x) if has_res else None
        grad_w = torch.empty(C, device=x.device, dtype=torch.float32)
        grad_b = torch.empty(C, device=x.device, dtype=torch.float32)
        if isinstance(rows, tuple):                     # device group table (bn_call_groups_device)
            table, G, biggest = rows
            scratch = torch.empty(backend.lib.bn_grouped_scratch_doubles(biggest, G, C, H * W), device=x.device,
                                  dtype=torch.float64)
            backend.run("bbd_bn_act_grouped_dev_bwd", x, ptr(x), ptr(y), ptr(grad_y), ptr(weight), ptr(bias), ptr(mean),
                        ptr(invstd), ptr(grad_x), ptr(grad_res), ptr(grad_w), ptr(grad_b), ptr(scratch), ptr(table), G,
                        biggest, N, C, H * W, int(relu))
            return grad_x, grad_w, grad_b, grad_res, None, None, None, None, None, None, None
        G = len(rows)
        scratch = torch.empty(backend.lib.bn_grouped_scratch_doubles(max(rows), G, C, H * W), device=x.device,
                              dtype=torch.float64)
        backend.run("bbd_bn_act_grouped_bwd", x, ptr(x), ptr(y), ptr(grad_y), ptr(weight), ptr(bias), ptr(mean),
                    ptr(invstd), ptr(grad_x), ptr(grad_res), ptr(grad_w), ptr(grad_b), ptr(scratch), _group_table(rows),
                    G, N, C, H * W, int(relu))
        return grad_x, grad_w, grad_b, grad_res, None, None, None, None, None, None, None


def batch_norm_act(x, weight, bias, residual, running_mean, running_var, momentum, eps, relu, backend=None,
                   num_batches_tracked=None):
    return _BatchNormAct.apply(x, weight, bias, residual, running_mean, running_var, num_batches_tracked, momentum,
                               eps, relu, backend or default_backend())


# ---------------------------------------------------------------------------- ReflectionPad2d(1), MaxPool2d(3,2,1)
class _ReflectPad1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, backend):
        x = x.contiguous()
        N, C, H, W = x.shape
        backend._check(x)
        out = torch.empty(N, C, H + 2, W + 2, device=x.device, dtype=torch.float32)
        backend.run("bbd_reflect_pad1_fwd", x, ptr(x), ptr(out), N * C, H, W)
        ctx.meta = (N, C, H, W, backend)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        N, C, H, W, backend = ctx.meta
        grad_out = grad_out.contiguous()
        grad_in = torch.empty(N, C, H, W, device=grad_out.device, dtype=torch.float32)
        backend.run("bbd_reflect_pad1_bwd", grad_out, ptr(grad_out), ptr(grad_in), N * C, H, W)
        return grad_in, None


def reflect_pad1(x, backend=None):
    """nn.ReflectionPad2d(1) (layers.py:124); gather-form backward instead of ATen's atomics."""
    return _ReflectPad1.apply(x, backend or default_backend())


class _MaxPool3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, backend):
        x = x.contiguous()
        N, C, H, W = x.shape
        backend._check(x)
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty(N, C, OH, OW, device=x.device, dtype=torch.float32)
        code = torch.empty(N, C, OH, OW, device=x.device, dtype=torch.uint8)
        backend.run("bbd_maxpool3s2_fwd", x, ptr(x), ptr(out), ptr(code), N * C, H, W)
        ctx.save_for_backward(code)
        ctx.meta = (N, C, H, W, backend)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (code,) = ctx.saved_tensors
        N, C, H, W, backend = ctx.meta
        grad_out = grad_out.contiguous()
        grad_in = torch.empty(N, C, H, W, device=grad_out.device, dtype=torch.float32)
        backend.run("bbd_maxpool3s2_bwd", grad_out, ptr(grad_out), ptr(code), ptr(grad_in), N * C, H, W)
        return grad_in, None


def maxpool3s2(x, backend=None):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the ResNet stem."""
    return _MaxPool3s2.apply(x, backend or default_backend())


# ---------------------------------------------------------------------------- decoder glue
class _UpCatPad(torch.autograd.Function):
    """ReflectionPad2d(1)(cat(nearest_x2(x), skip)) in one pass each way (decoder glue, csrc/bbd_nn.hip)."""

    @staticmethod
    def forward(ctx, x, skip, backend):
        x = x.contiguous()
        N, C1, h, w = x.shape
        C2 = 0
        if skip is not None:
            skip = skip.contiguous()
            C2 = skip.shape[1]
            assert skip.shape == (N, C2, 2 * h, 2 * w)
        backend._check(x, skip)
        out = torch.empty(N, C1 + C2, 2 * h + 2, 2 * w + 2, device=x.device, dtype=torch.float32)
        backend.run("bbd_upcat_pad1_fwd", x, ptr(x), ptr(skip), ptr(out), N, C1, C2, h, w)
        ctx.meta = (N, C1, C2, h, w, backend)
        return out

    @staticmethod
    def backward(ctx, gout):
        N, C1, C2, h, w, backend = ctx.meta
        gout = gout.contiguous()
        gx = torch.empty(N, C1, h, w, device=gout.device, dtype=torch.float32)
        gs = torch.empty(N, C2, 2 * h, 2 * w, device=gout.device, dtype=torch.float32) if C2 else None
        backend.run("bbd_upcat_pad1_bwd", gout, ptr(gout), ptr(gx), ptr(gs), N, C1, C2, h, w)
        return gx, gs, None


def upcat_pad(x, skip=None, backend=None):
    return _UpCatPad.apply(x, skip, backend or default_backend())


def upcat_pad_supported(x, skip):
    n_planes = x.shape[0] * (x.shape[1] + (skip.shape[1] if skip is not None else 0))
    return (FUSED_NN and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and n_planes <= 65535
            and (skip is None or (skip.is_cuda and skip.dtype == torch.float32)))


class _BiasELU(torch.autograd.Function):
    """ELU(conv_out + bias) in place on the bias-free convolution's output; backward from the saved OUTPUT
    (ELU' = y > 0 ? 1 : y + 1) with the bias gradient in the same pass."""

    @staticmethod
    def forward(ctx, y, bias, backend):
        assert y.is_contiguous()
        N, C, H, W = y.shape
        backend._check(y, bias)
        backend.run("bbd_bias_elu_fwd", y, ptr(y), ptr(bias), N, C, H * W)
        ctx.mark_dirty(y)
        ctx.save_for_backward(y)
        ctx.backend = backend
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        backend = ctx.backend
        N, C, H, W = y.shape
        gy = gy.contiguous()
        gx = torch.empty_like(y)
        gb = torch.empty(C, device=y.device, dtype=torch.float32)
        scratch = torch.empty(backend.lib.bias_elu_scratch_doubles(N, C, H * W), device=y.device, dtype=torch.float64)
        backend.run("bbd_bias_elu_bwd", y, ptr(y), ptr(gy), ptr(gx), ptr(gb), ptr(scratch), N, C, H * W)
        return gx, gb, None


def bias_elu_(conv_out, bias, backend=None):
    return _BiasELU.apply(conv_out, bias, backend or default_backend())


# ---------------------------------------------------------------------------- disparity head Conv3x3(C -> 1)
class _DispConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, backend):
        x = x.contiguous()
        N, C, H, W = x.shape
        w = weight.contiguous()
        backend._check(x, w, bias)
        y = torch.empty(N, 1, H, W, device=x.device, dtype=torch.float32)
        backend.run("bbd_dispconv_fwd", x, ptr(x), ptr(w), ptr(bias), ptr(y), N, C, H, W)
        ctx.save_for_backward(x, w)
        ctx.meta = (bias is not None, backend)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, w = ctx.saved_tensors
        has_bias, backend = ctx.meta
        N, C, H, W = x.shape
        grad_y = grad_y.contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])
        grad_x = torch.empty_like(x) if need_x else None
        grad_w = torch.empty_like(w) if need_w else None
        grad_b = torch.empty(1, device=x.device, dtype=torch.float32) if (need_w and has_bias) else None
        scratch = torch.empty(backend.lib.dispconv_scratch_doubles(C), device=x.device, dtype=torch.float64)
        backend.run("bbd_dispconv_bwd", x, ptr(x), ptr(w), ptr(grad_y), ptr(grad_x), ptr(grad_w), ptr(grad_b),
                    ptr(scratch), N, C, H, W)
        return grad_x, grad_w, grad_b, None


def dispconv(x, weight, bias, backend=None):
    """layers.Conv3x3(C, 1): reflection pad + 3x3 convolution to ONE channel + bias (the decoder's disparity heads)."""
    return _DispConv.apply(x, weight, bias, backend or default_backend())
