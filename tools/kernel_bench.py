#!/usr/bin/env python3
"""Micro-benchmark of the hot-path kernels alone (no networks): times the fused forward, its
backward and the identity pre-pass with HIP events and prints achieved algorithmic GB/s
(SURVEY.md 8d byte model).  bench.py is the contract benchmark; this is the tuning loop."""
import argparse
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from baseboostdepth_amd import ops  # noqa: E402
from baseboostdepth_amd.synthetic import synthetic_batch, synthetic_disp, synthetic_poses  # noqa: E402
from baseboostdepth_amd.trainer import Trainer  # noqa: E402


def algorithmic_bytes(plan, S, H, W):
    """forward / backward / identity compulsory bytes for one step."""
    P = H * W
    fwd = bwd = 0
    for names in plan.cand_names:
        c_src = len({f for k, f in names if k in ("T", "E")})
        c_id = sum(1 for k, _ in names if k == "I")
        fwd += P * (25 + 12 * c_src + 4 * c_id)
        bwd += P * (21 + 12 * c_src)
    return fwd * S, bwd * S, 28 * P * plan.NI


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="md2", choices=["md2", "boost7", "boost_e15"])
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--smooth", action="store_true",
                    help="spatially smooth disparities (17x17 box filter of the random maps), like the training step's "
                         "network outputs: per-pixel random disparities scatter the gathers and overstate the times")
    args = ap.parse_args()
    dev = "cuda:0"
    H, W, B = 192, 640, args.batch
    if args.config == "md2":
        ms, trimin, decomp, scales = [1] * B, False, False, [0, 1, 2, 3]
    elif args.config == "boost7":
        ms, trimin, decomp, scales = [7] * B, True, True, [0]
    else:
        import random
        rnd = random.Random(15)
        probs = [.050, .050, .077, .094, .139, .142, .448]
        ms = [rnd.choices(range(1, 8), probs)[0] for _ in range(B)]
        trimin, decomp, scales = True, True, [0]
    inputs = synthetic_batch(ms, H, W, scales, device=dev, seed=42)
    opt = types.SimpleNamespace(height=H, width=W, batch_size=B, scales=scales, frame_ids=[0], min_depth=0.1,
                                max_depth=100.0, disparity_smoothness=1e-3, no_ssim=False, trimin=trimin,
                                decomp=decomp, pose_error=5.5, incremental_skip=False, partial_skip=False,
                                materialize_warps=False)
    tr = Trainer.__new__(Trainer)
    tr.opt, tr.device, tr.num_scales, tr.backend, tr.maxing_valid_frames = opt, torch.device(dev), 4, None, False
    be = tr._backend()
    plan = tr.valid_frames_trimin(inputs)
    disp = synthetic_disp(B, H, W, scales, device=dev, seed=1)
    if args.smooth:
        import torch.nn.functional as F
        disp = {s: F.avg_pool2d(F.pad(d, (8, 8, 8, 8), mode="replicate"), 17, 1) for s, d in disp.items()}
    poses = synthetic_poses(plan, device=dev, seed=2, pose_error=5.5)
    outputs = {("disp", s): disp[s].requires_grad_(True) for s in scales}
    outputs.update(poses)
    S = len(scales)

    def fwd():
        o = tr.generate_images_pred(inputs, dict(outputs))
        return o[("bbd", "loss_sum")]

    def time_it(fn, n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    # isolate the three kernels
    target = inputs[("color", 0, 0)]
    frame_tensors = {f: inputs[("color", f, 0)] for f in plan.frames}
    depth = ops.disp_pyramid_to_depth([disp[s] for s in scales], H, W, 0.1, 100.0, be).detach()
    table = ops.pose_table(plan, inputs[("K", 0)], inputs[("inv_K", 0)], tr._job_poses(inputs, outputs)).detach()
    ident = ops.identity_losses(plan, frame_tensors, target, False, be)
    noise = inputs["noise"]
    fb, bb, ib = algorithmic_bytes(plan, S, H, W)

    def k_ident():
        ops.identity_losses(plan, frame_tensors, target, False, be)

    # the training path: disparity-mode launches (low-resolution disparities in, depth + coordinates by-products)
    dd = [disp[s].detach().clone().requires_grad_(True) for s in scales]
    t2 = table.clone().requires_grad_(True)

    def k_fwd():
        return ops.fused_reprojection_min_disp(dd, t2, target, ident, noise, plan, frame_tensors, 0.1, 100.0, False, False,
                                               True, be)

    ls = k_fwd()[0]
    gsum = torch.full_like(ls, 1.0 / (B * H * W))

    def k_bwd():
        torch.autograd.grad(ls, dd + [t2], gsum, retain_graph=True)

    res = {"config": args.config, "smooth": bool(args.smooth), "batch": B, "scales": scales, "NP": plan.NP, "NI": plan.NI,
           "cands_per_sample": [len(n) for n in plan.cand_names]}
    # HIP events directly around each C-ABI launch (python overhead excluded)
    timer = ops.KernelTimer()
    for name, fn, nbytes, key in (("identity", k_ident, ib, "bbd_identity_loss_fwd"),
                                  ("fwd", k_fwd, fb, "bbd_warp_ssim_min_disp_fwd"),
                                  ("bwd", k_bwd, bb, "bbd_warp_ssim_min_disp_bwd")):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        timer.reset()
        be.timer = timer
        for _ in range(args.iters):
            fn()
        torch.cuda.synchronize()
        be.timer = None
        ms_ = timer.summary()[key][1]
        res[name] = {"ms": round(ms_, 4), "alg_MB": round(nbytes / 1e6, 1), "GBps": round(nbytes / ms_ / 1e6, 1),
                     "frac_of_8TBps": round(nbytes / ms_ / 1e6 / 8000, 4)}
    for _ in range(args.warmup):
        fwd()
    res["generate_images_pred_total_ms"] = round(time_it(fwd, args.iters), 4)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
