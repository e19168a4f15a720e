#!/usr/bin/env python3
"""Contract benchmark: training images/sec of the MD2 ResNet-18 step at 640x192 on N MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks ITSELF (a child
`python -m torch.distributed.run`, one rank per GPU over RCCL) before anything touches the GPU and
relays the child's output and exit status.

A "step" is the reference's own definition (trainer.py:233,260-264): process_batch + zero_grad
+ backward + optimizer.step on one pre-resident synthetic KITTI-shaped batch (BASELINE.json
configs[1]: MD2 ResNet-18 encoder + DepthDecoder + pose nets, 640x192, batch 12 per GPU, frames
[0,-1,1], 4 scales, fp32), bracketed by barrier + device synchronise; value = global images / max
over ranks of the wall time.  Rank 0 prints ONE JSON line carrying, besides the contract keys:

  roofline      the hot path's most expensive HIP kernel inside the timed steps: algorithmic bytes
                per launch (SURVEY.md 8d byte model) / its mean HIP-event duration, vs 8 TB/s HBM
  kernels       the same for each of the three hot-path kernels
  cpu_baseline  the oracle (PyTorch-CPU restatement of the reference step, BASELINE.json
                configs[0]: batch 4) timed on this box's host cores, bounded to a few steps
  hot_path_ab   BASELINE configs[1] A/B on this GPU: the hot path alone as fused HIP kernels vs the
                same arithmetic as eager PyTorch-ROCm ops (grid_sample / avg_pool2d / cat / min)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from baseboostdepth_amd.benchmodes import (H, W, SCALES, cpu_model, make_options, run_fresh, run_loader_fed,  # noqa: E402
                                           stream_copy_ceiling, workload_name)


def algorithmic_bytes(plan, S):
    """Compulsory HBM bytes per launch of each hot-path kernel."""
    P = H * W
    fwd = bwd = 0
    for names in plan.cand_names:
        c_src = len({f for k, f in names if k in ("T", "E")})
        c_id = sum(1 for k, _ in names if k == "I")
        fwd += P * (25 + 12 * c_src + 4 * c_id)
        bwd += P * (21 + 12 * c_src)
    # the disparity-mode launches are the same kernels fed with the low-resolution disparities:
    # they are priced with the SAME 8d figure (which still counts 4 B/px of depth they no longer read)
    return {"bbd_warp_ssim_min_fwd": fwd * S, "bbd_warp_ssim_min_bwd": bwd * S,
            "bbd_warp_ssim_min_disp_fwd": fwd * S, "bbd_warp_ssim_min_disp_bwd": bwd * S,
            "bbd_identity_loss_fwd": 28 * P * plan.NI}


def cpu_baseline(batch=4, budget_s=25.0):
    """The oracle's full MD2 step (networks + hot path + Adam) on the host CPU, all cores, then
    one step at the reference's own setting of ONE thread (train.py:23) if the budget allows."""
    from baseboostdepth_amd import networks
    from baseboostdepth_amd.synthetic import synthetic_batch
    from oracle.step_ref import md2_step
    ms = [1] * batch
    torch.manual_seed(42)
    models = {"encoder": networks.ResnetEncoder(18, False), "pose_encoder": networks.ResnetEncoder(18, False, 2)}
    models["depth"] = networks.DepthDecoder(models["encoder"].num_ch_enc, SCALES)
    models["pose"] = networks.PoseDecoder(models["pose_encoder"].num_ch_enc, 1, 2)
    params = [p for m in models.values() for p in m.parameters()]
    opt = torch.optim.Adam(params, 1e-4)
    inputs = synthetic_batch(ms, H, W, SCALES, device="cpu", seed=42)
    noise = inputs.pop("noise")
    # pick the thread count: PyTorch defaults to every hardware thread it can see, which can be far
    # more than this process is allowed to run on (measured here: 128 threads 0.64 images/s vs 16
    # threads 9.3 images/s) - try the scheduler-affinity count and smaller powers of two, keep the best
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else torch.get_num_threads()
    cands = sorted({c for c in (8, 16, 32, 64, avail) if c <= max(avail, 8)})
    md2_step(models, opt, inputs, ms, SCALES, H, W, noise)          # warm-up (allocator, oneDNN primitives)
    t_all, best, probe = time.perf_counter(), None, {}
    for c in cands:
        torch.set_num_threads(c)
        md2_step(models, opt, inputs, ms, SCALES, H, W, noise)
        t0 = time.perf_counter()
        md2_step(models, opt, inputs, ms, SCALES, H, W, noise)
        probe[c] = time.perf_counter() - t0
        if best is None or probe[c] < probe[best]:
            best = c
        elif probe[c] > 1.5 * probe[best] or (time.perf_counter() - t_all) > budget_s * 0.5:
            break
    cores = best
    torch.set_num_threads(cores)
    times = []
    while len(times) < 5 and (time.perf_counter() - t_all) < budget_s * 0.8:
        t0 = time.perf_counter()
        md2_step(models, opt, inputs, ms, SCALES, H, W, noise)
        times.append(time.perf_counter() - t0)
    times = times or [probe[best]]
    med = sorted(times)[len(times) // 2]
    out = {"value": round(batch / med, 3), "unit": "images/sec", "cores": cores, "kind": "port",
           "sample": "%d full MD2 steps (ResNet-18 nets + oracle hot path + Adam), batch %d, 640x192, 4 scales, "
                     "median step %.3f s; threads probed %s of %d schedulable" % (
                         len(times), batch, med, {k: round(v, 2) for k, v in probe.items()}, avail)}
    # hot-path-only split: generate_images_pred + compute_losses + backward on fixed disp/poses
    try:
        from oracle import hotpath_ref as O
        g = torch.Generator().manual_seed(3)
        hd = {s: torch.rand(batch, 1, H >> s, W >> s, generator=g).requires_grad_(True) for s in SCALES}
        hp = {f: O.pose_matrix(0.01 * torch.randn(batch, 1, 3, generator=g), 0.05 * torch.randn(batch, 1, 3, generator=g),
                               invert=(f < 0)).requires_grad_(True) for f in (1, -1)}
        O.hot_path(inputs, hd, hp, ms, SCALES, False, False, noise, H, W)["loss"].backward()
        t0 = time.perf_counter()
        O.hot_path(inputs, hd, hp, ms, SCALES, False, False, noise, H, W)["loss"].backward()
        out["hot_path_only_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    except Exception as e:      # the baseline is informational; never fail the benchmark on it
        out["hot_path_only_ms"] = "n/a (%s)" % type(e).__name__
    if (time.perf_counter() - t_all) < budget_s:
        torch.set_num_threads(1)                                     # the reference's own setting (train.py:23)
        t0 = time.perf_counter()
        md2_step(models, opt, inputs, ms, SCALES, H, W, noise)
        out["one_thread_images_per_sec"] = round(batch / (time.perf_counter() - t0), 3)
        torch.set_num_threads(cores)
    return out


def eager_hot_path_ab(trainer, inputs, opt, iters=10):
    """BASELINE configs[1] A/B: the hot path only (generate_images_pred + compute_losses + backward
    w.r.t. disp and poses) as (a) the fused HIP kernels and (b) the oracle's eager PyTorch-ROCm op
    sequence (F.grid_sample / avg_pool2d / cat / min - what the reference would launch on this GPU),
    on identical device-resident inputs.  Measurement only; the product never runs (b)."""
    from oracle import hotpath_ref as O
    plan = trainer.plan
    dev = inputs[("color", 0, 0)].device
    gen = torch.Generator(device=dev).manual_seed(7)
    disp = {s: torch.rand(plan.B, 1, H >> s, W >> s, generator=gen, device=dev).requires_grad_(True)
            for s in opt.scales}
    from baseboostdepth_amd.synthetic import synthetic_poses
    pp = synthetic_poses(plan, device=dev, seed=2)
    poses = {f: pp[("cam_T_cam", 0, f)].clone().requires_grad_(True) for f in plan.frames if f != "s"}

    def fused():
        out = {("disp", s): disp[s] for s in opt.scales}
        out.update({("cam_T_cam", 0, f): T for f, T in poses.items()})
        out.update(trainer.generate_images_pred(inputs, out))
        trainer.compute_losses(inputs, out)["loss"].backward()

    def eager():
        noise = torch.randn(plan.B, H, W, device=dev) * 0.00001     # drawn per step, like the fused path does
        O.hot_path(inputs, disp, poses, plan.ms, opt.scales, False, False, noise, H, W)["loss"].backward()

    res = {}
    for name, fn in (("fused_hip_ms", fused), ("eager_rocm_ms", eager)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        res[name] = round((time.perf_counter() - t0) / iters * 1e3, 3)
    res["speedup"] = round(res["eager_rocm_ms"] / res["fused_hip_ms"], 2)
    res["what"] = "hot path only: warp+SSIM+min over 4 scales + smoothness, forward+backward, batch %d" % plan.B
    return res


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`--gpus N` without a torchrun environment: run the N ranks as a CHILD process group (never exec -
    this process must not have touched the GPU yet, and does not) and hand its exit status back.  The child
    is its own session: if it has not finished after `--launch-timeout` seconds (a hung rendezvous or
    collective) the whole group is terminated and the status is 124 - a hang cannot outlive the caller."""
    import signal
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return child.wait(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        sys.stderr.write("bench.py: the %d-rank child did not finish within %d s - terminating its process group\n"
                         % (args.gpus, args.launch_timeout))
        for sig, grace in ((signal.SIGTERM, 15), (signal.SIGKILL, 15)):
            try:
                os.killpg(child.pid, sig)
            except ProcessLookupError:
                break
            try:
                child.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        return 124


def device_identity(local):
    """A string that is different for every physical GPU of the node: uuid + PCI address where PyTorch exposes them."""
    pr = torch.cuda.get_device_properties(local)
    parts = [str(getattr(pr, "uuid", "")), pr.name]
    if hasattr(pr, "pci_bus_id"):
        parts.append("%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0)))
    return "|".join(parts)


def dry_launch(args):
    """Rendezvous check of the multi-rank path without the workload: every rank joins the process group,
    one all-reduce and one broadcast run on the backend the benchmark would use (RCCL on GPUs; gloo when
    BBD_DIST_BACKEND=gloo or there is no GPU), rank 0 prints one JSON line."""
    import torch.distributed as dist
    from baseboostdepth_amd import distributed as bdist
    rank, local, world = bdist.init_from_env()
    assert world == args.gpus, "world size %d != --gpus %d" % (world, args.gpus)
    backend = dist.get_backend() if world > 1 else "none"
    dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    t = torch.full((4,), float(rank + 1), device=dev)
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
    ok = bool((t == world * (world + 1) / 2).all()) if world > 1 else True
    if rank == 0:
        print(json.dumps({"dry_launch": True, "ranks": world, "n_gpus": world,
                          "collective": {"nccl": "rccl"}.get(backend, backend), "all_reduce_ok": ok}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def eager_full_step(trainer, inputs, opt, iters=10):
    """BASELINE configs[1] A/B, whole step: the SAME networks and optimizer on this GPU, but the hot path
    as the oracle's eager PyTorch-ROCm op sequence (F.grid_sample / avg_pool2d / cat / min, one stream) -
    what the reference's trainer would launch here.  Runs after the timed region; measurement only."""
    from oracle.step_ref import md2_step
    plan = trainer.plan
    noise = torch.randn(plan.B, H, W, device=inputs[("color", 0, 0)].device) * 0.00001
    fn = lambda: md2_step(trainer.models, trainer.model_optimizer, inputs, plan.ms, opt.scales, H, W, noise)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    return {"ms_per_step": round(ms, 3), "images_per_sec": round(plan.B / ms * 1e3, 2),
            "what": "same nets + Adam on this GPU, hot path as eager PyTorch-ROCm ops (oracle), batch %d" % plan.B}


ISSUE_NOTE = ("stall (the launch runs at frac_of_issue_bound of its vector-issue bound, kernels.*; HBM traffic is at or below "
              "the algorithmic bytes - DESIGN.md 3, findings 14 and 26)")


def committed_constants(config):
    """Counter-derived constants bench.py quotes next to its own measurements: per-kernel HBM traffic of a committed
    rocprofv3 PMC pass of this workload and the static instruction mix, with the hash of the kernel source they were
    taken from (tools/pmc_summary.py, tools/isa_mix.py).  `stale` = the shipped source is not that source."""
    from baseboostdepth_amd.csrc.build import source_sha16
    now = source_sha16()
    out = {"traffic": None, "traffic_path": None, "isa_mix": {}, "isa_mix_path": None, "source_sha16": now, "stale": []}
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        cand = os.path.join(ROOT, "profiles", rnd, "traffic_%s.json" % config)
        if out["traffic"] is None and os.path.isfile(cand):
            out["traffic"], out["traffic_path"] = json.load(open(cand)), os.path.relpath(cand, ROOT)
        cand = os.path.join(ROOT, "profiles", rnd, "isa_mix.json")
        if not out["isa_mix"] and os.path.isfile(cand):
            out["isa_mix"], out["isa_mix_path"] = json.load(open(cand)), os.path.relpath(cand, ROOT)
    for name, blob, path in (("traffic", out["traffic"], out["traffic_path"]), ("isa_mix", out["isa_mix"], out["isa_mix_path"])):
        if blob and blob.get("kernel_source_sha16") != now:
            out["stale"].append("%s (%s: taken from source %s, shipped %s)" % (name, path, blob.get("kernel_source_sha16"), now))
    return out


def kernel_table(timer, plan, S, config, batch):
    """Per hot-path kernel: this run's HIP-event mean, the algorithmic GB/s it implies, and (batch 12 only) the committed
    counter constants.  Returns (kernels, dominant kernel, roofline)."""
    nbytes = algorithmic_bytes(plan, S)
    kernels = {}
    for name, (count, mean_ms) in timer.summary().items():
        if name in nbytes and mean_ms > 0:
            gbps = nbytes[name] / (mean_ms * 1e-3) / 1e9
            kernels[name] = {"launches": count, "mean_ms": round(mean_ms, 4), "alg_MB_per_launch": round(nbytes[name] / 1e6, 2),
                             "achieved_GBps": round(gbps, 1), "frac": round(gbps / 8000.0, 4)}
    if not kernels:
        return {}, None, None, None
    dom = max(kernels, key=lambda k: kernels[k]["mean_ms"])
    cc = committed_constants(config)
    traffic = None
    alias = lambda k: k.replace("_disp_", "_")           # PMC files name the kernels, not the entry points
    if batch == 12 and cc["traffic"] is not None:
        tj = cc["traffic"]
        traffic = tj.get(alias(dom), {}).get("traffic_bytes")
        for k in kernels:
            e = tj.get(alias(k))
            if not e:
                continue
            kernels[k]["pmc_traffic_MB_per_launch"] = round(e["traffic_bytes"] / 1e6, 1)
            # (the forward has two instantiations by candidate count, chosen like launch_fused_fwd does)
            # (the forward has three forms by launch shape, chosen like fused_fwd_form() of the library does)
            suffix = ""
            if alias(k).endswith("_fwd") and plan.NP > 4 * plan.B:
                suffix = "_held" if S * plan.B * ((H + 15) // 16) * ((W + 63) // 64) >= 4096 else "_many"
            mix = cc["isa_mix"].get(alias(k) + suffix)
            if "valu_wave_instructions" in e and mix:
                # vector-issue bound (finding 14): the launch's counter-measured vector instructions (a
                # committed PMC pass, not this run) x the kernel's mean cycles per instruction by static issue class at
                # 2.4 GHz over 1024 SIMDs = the time the launch would take if two waves could always issue on every SIMD.
                # (No entry for a kernel the instruction-mix file does not list - no default.)
                bound_ms = e["valu_wave_instructions"] * mix["cycles_per_valu_instruction"] / 1024 / 2.4e9 * 1e3
                kernels[k]["issue_bound_ms"] = round(bound_ms, 4)
                kernels[k]["frac_of_issue_bound"] = round(bound_ms / kernels[k]["mean_ms"], 3)
    fib = kernels[dom].get("frac_of_issue_bound")
    roofline = {"bound": "hbm", "limiter": ISSUE_NOTE if fib is None else ISSUE_NOTE.replace("frac_of_issue_bound", "%.2f" % fib, 1),
                "kernel": dom, "achieved": kernels[dom]["achieved_GBps"], "peak": 8000.0,
                "unit": "GB/s", "frac": kernels[dom]["frac"], "traffic": traffic,
                "traffic_source": cc["traffic_path"] if traffic is not None else None,
                # the fraction the builder argues from: measured time vs the launch's own vector-issue bound
                "frac_of_issue_bound": fib}
    return kernels, dom, roofline, cc


def batch_for(config, batch, rank):
    import random as _random
    draw = _random.Random(1234 + rank)
    if config.endswith("_coherent"):          # the same draw as the base configuration, stacked canonically
        return sorted(batch_for(config.replace("_coherent", ""), batch, rank), reverse=True)
    if config in ("md2", "vit"):
        return [1] * batch
    if config == "boosted":
        return [7] * batch
    if config == "boosted15":                # P(m = 1..7) at epoch 15
        return draw.choices(range(1, 8), [.050, .050, .077, .094, .139, .142, .448], k=batch)
    return draw.choices(range(0, 3), [.062, .573, .366], k=batch)     # epoch 5: P(m = 0,1,2) = .062 .573 .366


def run_workload(args, ctx, config, steps, warmup, want_graph, dp_mode):
    """Build a fresh Trainer for `config`, warm up, time EXACTLY `steps` steps between barrier + synchronise, then the
    per-kernel HIP-event pass.  Returns a dict (trainer / opt / inputs included: the caller releases them)."""
    from baseboostdepth_amd import distributed as bdist
    from baseboostdepth_amd import ops
    from baseboostdepth_amd.synthetic import synthetic_batch
    from baseboostdepth_amd.trainer import Trainer
    rank, local, world, dev = ctx["rank"], ctx["local"], ctx["world"], ctx["dev"]

    def build_trainer(step_graph):
        torch.manual_seed(42)
        opt = make_options(args.batch, local, config)
        opt.fused_adam = not args.no_fused_adam
        opt.step_graph = bool(step_graph)
        opt.dp_capture = bool(step_graph) and dp_mode == "graph-overlap"
        run_scales = list(opt.scales)
        opt.scales = list(SCALES)      # networks + num_scales are built for 4 scales (trainer.py:44); the epoch>=10
        tr = Trainer(opt)              # curriculum then trains on scale 0 only (run_epoch, trainer.py:209-212)
        tr.opt.scales = run_scales
        if args.channels_last:
            for m in tr.models.values():
                m.to(memory_format=torch.channels_last)
        tr.set_train()
        bdist.attach(tr)
        return tr, opt

    if world > 1 and dp_mode == "overlap":
        want_graph = False
    trainer, opt = build_trainer(want_graph)
    ms = batch_for(config, args.batch, rank)
    coherent = config.endswith("_coherent")
    if coherent:
        # the regime a TRAINED network produces (coherent arg-min maps, a few live candidates per tile) - constructed:
        # a planar scene whose frames are true shifts, networks whose heads predict exactly that scene
        # (synthetic.structured_batch / constant_heads).  Learning rate 0: the optimizer runs the same kernels, the
        # constructed registration does not drift over the timed steps
        from baseboostdepth_amd.synthetic import constant_heads, structured_batch
        constant_heads(trainer)
        for g in trainer.model_optimizer.param_groups:
            g["lr"] = 0.0
        inputs = structured_batch(ms, H, W, opt.scales, device=dev, seed=42 + rank)
    else:
        inputs = synthetic_batch(ms, H, W, opt.scales, device=dev, seed=42 + rank)
    # the identity-candidate noise is drawn INSIDE every step, as the reference does (trainer.py:518-523) - except for the
    # random-initialised boosted secondaries, whose near-ties between identity candidates turn 1e-5 of noise into other
    # arg-min maps (and backward times): they take ONE seeded draw with the batch, so that two runs time the same maps
    if config not in ("boosted", "boosted15", "trimin5"):
        inputs.pop("noise")
    if config in ("boosted", "boosted15", "boosted15_coherent"):
        inputs["cutt"] = torch.tensor(1.35)      # epoch >= 10 regime: incremental + partial pose modes
    backend = ops.default_backend()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graph_note = None
    if trainer.use_graph:
        failure = None
        try:                                   # capture happens inside the first step of a batch signature
            trainer.train_step(inputs)
            torch.cuda.synchronize()
        except Exception as e:                 # never lose the benchmark to the capture: fall back to the eager loop
            if args.step_graph == "on":
                raise
            failure = "%s: %s" % (type(e).__name__, str(e)[:120])
            torch.cuda.synchronize()
        if world > 1:                          # the fallback re-attaches (collectives): every rank takes it, or none
            ok = torch.tensor([0.0 if failure else 1.0], device=dev)
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
            if float(ok) < 0.5 and failure is None:
                failure = "capture failed on another rank"
        if failure:
            graph_note = "capture failed (%s), eager loop used" % failure
            trainer, opt = build_trainer(False)
    for _ in range(warmup):
        trainer.train_step(inputs)
    if trainer.grad_sync is not None:
        trainer.grad_sync.timing = True
    timer = ops.KernelTimer()
    backend.timer = None if trainer.use_graph else timer
    # per-step durations for the median: one event at every step boundary on the main stream (no host sync)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    sync_all()
    t0 = time.perf_counter()
    for i in range(steps):
        marks[i].record()
        trainer.train_step(inputs)
    marks[steps].record()
    sync_all()
    elapsed = time.perf_counter() - t0
    backend.timer = None
    exchange_ms, reduce_op, overlapped = 0.0, None, None
    if trainer.grad_sync is not None:
        trainer.grad_sync.timing = False
        exchange_ms, reduce_op = trainer.grad_sync.exchange_ms(), trainer.grad_sync.reduce_op
        overlapped = getattr(trainer.grad_sync, "launched_in_backward", None)
    kernel_timing = "HIP events around each C-ABI launch inside the timed steps"
    if trainer.use_graph:
        # a replayed graph's nodes cannot be bracketed by events: the SAME kernels on the same inputs are timed over
        # eager steps of the same trainer right after the timed region (rocprofv3 stats under profiles/ agree)
        n_evt = max(5, min(steps, 20))
        backend.timer = timer
        for _ in range(n_evt):
            trainer._eager_step(dict(inputs))
        torch.cuda.synchronize()
        backend.timer = None
        kernel_timing = ("HIP events around each C-ABI launch over %d eager steps run right after the timed region "
                         "(the timed region replays one hipGraph per step)" % n_evt)
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    median_ms = per_step[len(per_step) // 2] if per_step else 0.0
    rank_ms = [elapsed / steps * 1e3]
    if world > 1:
        t = torch.tensor([elapsed, median_ms, exchange_ms], device=dev, dtype=torch.float64)
        mine = t[:1].clone()
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        rank_ms = [float(e.item()) / steps * 1e3 for e in every]
        elapsed, median_ms, exchange_ms = float(t[0].item()), float(t[1].item()), float(t[2].item())
    S = len(opt.scales)
    live = None
    if config.startswith("boosted") or config.startswith("trimin"):
        # how many candidates win at least one pixel of a backward tile (32x16): what the backward's candidate loop sees
        from baseboostdepth_amd.synthetic import live_candidates_per_tile
        with torch.no_grad():
            out_l, _ = trainer.process_batch(dict(inputs))
        live = live_candidates_per_tile(out_l[("bbd", "argmin")][0])
        del out_l
    kernels, dom, roofline, cc = kernel_table(timer, trainer.plan, S, config.replace("_coherent", ""), args.batch)
    dp = "single" if world == 1 else (("graph-overlap" if trainer.dp_capture else "graph") if trainer.use_graph else "overlap")
    return {"config": config, "trainer": trainer, "opt": opt, "inputs": inputs, "ms": ms, "S": S, "steps": steps, "warmup": warmup,
            "elapsed": elapsed, "median_ms": median_ms, "rank_ms": rank_ms, "kernels": kernels, "dominant": dom, "roofline": roofline,
            "constants": cc, "graph_note": graph_note, "kernel_timing": kernel_timing, "exchange_ms": exchange_ms,
            "reduce_op": reduce_op, "overlapped": overlapped, "dp_mode": dp, "live_candidates_per_tile": live,
            "value": args.batch * world * steps / elapsed, "ms_per_step": elapsed / steps * 1e3}


def release(res):
    """Drop a workload's trainer (its captured graphs keep private memory pools) before the next one is built."""
    import gc
    for k in ("trainer", "opt", "inputs"):
        res.pop(k, None)
    gc.collect()
    torch.cuda.empty_cache()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=12, help="per-GPU batch (BASELINE config 2: 12)")
    ap.add_argument("--config", default="md2", choices=["md2", "boosted", "boosted15", "trimin5", "vit", "boosted15_fresh",
                                                        "trimin5_fresh", "md2_loader", "boosted15_coherent"],
                    help="md2 = BASELINE configs[1]/[3] (the headline); configs[2] (SURVEY 8d config 3): boosted = "
                         "worst case m=7 for every sample, boosted15 = the epoch-15 offset distribution (standard "
                         "draw, fixed seed), trimin5 = early curriculum (epoch 5: m in {0,1,2}, 4 scales); "
                         "vit = configs[4]: MonoViT (mpvit_small) encoder + HR decoder, MD2 frame set")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block: after the headline's measurements a default one-GPU md2 run also times "
                         "BASELINE configs[2] (boosted, boosted15) and configs[4] (vit) for --secondary-steps steps each, in this "
                         "process, fresh Trainer each, and reports them under `secondary` of the same JSON line")
    ap.add_argument("--secondary-steps", type=int, default=10)
    ap.add_argument("--secondary-budget", type=float, default=240.0, help="seconds; configs not started within it are listed as skipped")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--miopen-benchmark", action="store_true", help="torch.backends.cudnn.benchmark=True (MIOpen find)")
    ap.add_argument("--no-fused-adam", action="store_true")
    ap.add_argument("--no-eager-ab", action="store_true",
                    help="skip timing the hot path / the whole step as eager PyTorch-ROCm ops (oracle) vs the fused kernels")
    ap.add_argument("--step-graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the whole step (both HIP streams, MIOpen, the C-ABI launches, fused Adam) as one hipGraph: "
                         "the ~1 200 launches of a step cost ~16 ms of host time, which a slow or busy host CPU turns into "
                         "the bottleneck (measured: 480 vs 575 images/s on two boxes with identical GPU time).  auto = on for "
                         "every run, falling back to the eager loop if capture fails; multi-rank runs replay two graphs "
                         "per step (forward+backward+gradient pack | eager RCCL all-reduce | optimizer)")
    ap.add_argument("--dp-mode", default="graph", choices=["graph", "overlap", "graph-overlap", "all"],
                    help="multi-rank loop: graph = two hipGraphs around ONE eager all-reduce of the flat gradient buffer (host "
                         "out of the loop, exchange exposed); overlap = eager loop, 32 MB buckets all-reduced from autograd "
                         "hooks while backward still runs; graph-overlap = ONE hipGraph per step with the bucketed RCCL "
                         "all-reduces captured into it (host out of the loop AND exchange overlapped; opt-in until it has run on "
                         "a multi-GPU node).  graph / overlap print `exchange_ms` (exposed wait) so one node can compare them.  "
                         "all = the headline in graph mode, then overlap and graph-overlap back to back in the same launch, "
                         "reported under `dp_modes`; graph-overlap runs under a watchdog that prints the line and ends the "
                         "rank if it hangs, so it cannot take the headline down")
    ap.add_argument("--dp-extra-timeout", type=float, default=60.0,
                    help="--dp-mode all: seconds each extra mode may take before the watchdog prints the line without it")
    ap.add_argument("--launch-timeout", type=int, default=1500,
                    help="`--gpus N` self-launch: seconds after which the child process group is terminated (status 124)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="only prove that --gpus N ranks start and rendezvous (one all-reduce), then exit")
    argv = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(argv)

    # ---- multi-rank launch: before ANY GPU call (torch.cuda.* initialises HIP; a process that has done so
    #      must never be replaced, and the children must find the devices untouched)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, argv)
    if args.dry_launch:
        return dry_launch(args)

    from baseboostdepth_amd import distributed as bdist
    rank, local, world = bdist.init_from_env()
    assert world == args.gpus, "world size %d != --gpus %d (launch with torchrun --nproc-per-node == --gpus, " \
                               "or plain `python bench.py --gpus N`)" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU path in the product)"
    collective = "none"
    if world > 1:
        backend_name = torch.distributed.get_backend()
        assert backend_name == "nccl" or os.environ.get("BBD_DIST_BACKEND"), \
            "multi-GPU runs exchange gradients over RCCL (backend 'nccl'), got %r" % backend_name
        collective = {"nccl": "rccl"}.get(backend_name, backend_name)
    if os.environ.get("BBD_SHARE_GPU0"):       # test hook: all ranks on GPU 0 (with BBD_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ctx = {"rank": rank, "local": local, "world": world, "dev": dev}
    # proof that the ranks sit on DISTINCT physical devices (a mis-set LOCAL_RANK / visible-devices mask would otherwise
    # print an N-GPU line from N ranks sharing one GPU)
    identities = [device_identity(local)]
    if world > 1:
        identities = [None] * world
        torch.distributed.all_gather_object(identities, device_identity(local))

    from baseboostdepth_amd import tuning

    tuning.use_shipped_db()            # explicit (the Trainer would do it too): before the first convolution
    torch.manual_seed(42)
    if args.miopen_benchmark:
        torch.backends.cudnn.benchmark = True
    want_graph = args.step_graph == "on" or (args.step_graph == "auto" and not args.no_fused_adam)
    head_mode = "graph" if args.dp_mode == "all" else args.dp_mode
    if args.config.endswith("_fresh") or args.config == "md2_loader":
        # stand-alone form of a secondary line (profiling runs): one JSON line, no headline
        assert world == 1, "the fresh-ordering / loader-fed lines are one-GPU measurements"
        r = run_fresh(args, ctx, args.config, want_graph) if args.config.endswith("_fresh") else run_loader_fed(args, ctx, want_graph)
        print(json.dumps(r))
        return 0
    res = run_workload(args, ctx, args.config, args.steps, args.warmup, want_graph, head_mode)
    trainer, opt, inputs = res["trainer"], res["opt"], res["inputs"]

    line = None
    if rank == 0:
        S, kernels, roofline, cc = res["S"], res["kernels"], res["roofline"], res["constants"]
        global_batch = args.batch * world
        line = {
            "metric": "training images/sec at 640x192, MD2 ResNet18" if args.config != "vit" else
                      "training images/sec at 640x192, MonoViT",
            "value": round(res["value"], 2), "unit": "images/sec",
            "n_gpus": world, "ranks": world, "collective": collective,
            # distinct physical devices behind the ranks (uuid / PCI address all-gathered from every rank)
            "devices": len(set(identities)), "device_names": sorted({i.split("|")[1] for i in identities if i and "|" in i}),
            "dp_mode": res["dp_mode"],
            # (captured collectives cannot be bracketed by events: null there)
            "exchange_ms": None if (trainer.use_graph and trainer.dp_capture and world > 1) else round(res["exchange_ms"], 4),
            "reduce_op": res["reduce_op"],
            "buckets_launched_in_backward": res["overlapped"],
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(res["ms_per_step"], 3), "ms_per_step_median": round(res["median_ms"], 3),
            # every rank's own wall time per step over the timed region (value uses the max): load imbalance between ranks
            "ms_per_step_rank_min": round(min(res["rank_ms"]), 3), "ms_per_step_rank_max": round(max(res["rank_ms"]), 3),
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_name(args.config, args.batch, S, res["ms"]),
                       "global_batch": global_batch, "parallelism": "dp%d" % world,
                       "miopen": {"user_db": ("shipped (baseboostdepth_amd/miopen_db, tools/miopen_tune.sh)"
                                              if os.path.basename(os.environ.get("MIOPEN_USER_DB_PATH", "")).startswith("miopen_db") else
                                              os.environ.get("MIOPEN_USER_DB_PATH")),
                                  # MIOpen appends to its private copy: a copy that was already there also holds the find
                                  # results of earlier runs on this machine (BBD_MIOPEN_CACHE=<fresh dir> for a cold one)
                                  "user_db_prewarmed": tuning.STATUS["miopen_db_prewarmed"],
                                  # MIOpen reads a find database only if its file name carries the running build's version
                                  "db_accepted": tuning.STATUS["miopen_db_accepted"], "db_why_not": tuning.STATUS["miopen_db_why"],
                                  "find": bool(torch.backends.cudnn.benchmark)},
                       # TunableOp is process-wide and only the MonoViT trainer turns it on (secondary.vit below)
                       "gemm": {"tunableop_enabled": bool(torch.cuda.tunable.is_enabled()),
                                "tuning_at_run_time": bool(torch.cuda.tunable.tuning_is_enabled()
                                                           and torch.cuda.tunable.is_enabled())}},
            "roofline": roofline, "kernels": kernels, "kernel_timing": res["kernel_timing"],
            **({"live_candidates_per_backward_tile": res["live_candidates_per_tile"]} if res.get("live_candidates_per_tile") else {}),
            # what in `kernels` is measured by THIS run (mean_ms, achieved_GBps, frac) and what is read from committed files
            "kernels_constants": "pmc_traffic_MB_per_launch and the instruction count behind issue_bound_ms come from the "
                                 "committed rocprofv3 PMC pass of this workload (roofline.traffic_source); the cycles per "
                                 "instruction from %s (issue classes of profiles/r03/valu_rate.txt)" % (cc["isa_mix_path"] if cc else None),
            # true when the committed counter / instruction-mix files were taken from other kernel source than the shipped one
            "kernels_constants_stale": bool(cc and cc["stale"]),
            "kernels_constants_stale_what": (cc["stale"] if cc else None) or None,
            "kernel_source_sha16": cc["source_sha16"] if cc else None,
            "step_graph": (res["graph_note"] if res["graph_note"] is not None else
                           (("one graph per step, bucketed RCCL all-reduces captured inside" if trainer.dp_capture else
                             "split: forward+backward+pack graph | eager all-reduce | optimizer graph")
                            if (trainer.use_graph and world > 1) else bool(trainer.use_graph))),
        }
        if roofline is not None:
            # the on-box stream-copy ceiling beside the 8 TB/s specification (outside the timed region, < 1 s)
            try:
                copy = stream_copy_ceiling(dev)
                roofline["peak_measured"] = copy["GBps"]
                roofline["frac_of_measured"] = round(roofline["achieved"] / copy["GBps"], 4)
                roofline["peak_measured_how"] = copy
            except Exception as e:    # informational
                roofline["peak_measured"] = "n/a (%s)" % type(e).__name__
        if world == 1 and not args.no_eager_ab and args.config == "md2":
            line["hot_path_ab"] = eager_hot_path_ab(trainer, inputs, opt)
            try:
                line["eager_step"] = eager_full_step(trainer, inputs, opt)
                line["eager_step_images_per_sec"] = line["eager_step"]["images_per_sec"]
            except Exception as e:    # informational, never fail the benchmark on it
                line["eager_step"] = "n/a (%s: %s)" % (type(e).__name__, e)
    del trainer, opt, inputs
    release(res)

    # ---- BASELINE configs[2] and configs[4] in the same command (one GPU, default flags): fresh Trainer each, graphs
    #      released in between, never a re-exec
    if world == 1 and args.config == "md2" and not args.no_secondary:
        secondary, t_sec = [], time.perf_counter()
        frozen, frozen_rows = {"md2": res["value"]}, {}
        for cfg in ("boosted", "boosted15", "boosted15_coherent", "trimin5", "vit", "boosted15_fresh", "trimin5_fresh", "md2_loader"):
            if time.perf_counter() - t_sec > args.secondary_budget:
                secondary.append({"config": cfg, "skipped": "secondary budget of %.0f s used up" % args.secondary_budget})
                continue
            try:
                if cfg.endswith("_fresh") or cfg == "md2_loader":
                    # the regimes a real run is in: a new ordering every step / the loader in the loop
                    r = run_fresh(args, ctx, cfg, want_graph) if cfg.endswith("_fresh") else run_loader_fed(args, ctx, want_graph)
                    ref = frozen.get(cfg.replace("_fresh", "").replace("_loader", ""))
                    if ref:
                        r["frozen_batch_images_per_sec"] = round(ref, 2)
                        r["frozen_batch_pose_rows"] = frozen_rows.get(cfg.replace("_fresh", ""))
                        # the frozen batch is ONE ordering; the fresh draws ask for other (on average more) pose rows: the same
                        # comparison per row of the pose pass - per row the batches asked for (padding counts against the fresh
                        # line) and per row that ran
                        fr, steady = r["frozen_batch_pose_rows"], (r["passes"][1] if len(r.get("passes", [])) == 4 else (r.get("passes") or [None])[0])
                        if fr and steady and steady.get("pose_rows_mean"):
                            per_row_frozen = (args.batch / ref * 1e3) / fr
                            r["vs_frozen_batch_per_pose_row_asked"] = round(per_row_frozen / (steady["ms_per_step"] / steady["pose_rows_mean"]), 4)
                            r["vs_frozen_batch_per_pose_row_run"] = round(per_row_frozen / (steady["ms_per_step"] / steady["pose_rows_run_mean"]), 4)
                        r["vs_frozen_batch"] = round(r["value"] / ref, 4)
                        if "passes" in r:
                            r["vs_frozen_batch_seen_signatures"] = round(r["passes"][-1]["images_per_sec"] / ref, 4)
                            r["vs_frozen_batch_cold_start"] = round(r["passes"][0]["images_per_sec"] / ref, 4)
                    secondary.append(r)
                    continue
                r = run_workload(args, ctx, cfg, max(10, args.secondary_steps), 3, want_graph, "graph")
                frozen[cfg] = r["value"]
                sched = getattr(getattr(r["trainer"], "tables", None), "schedule", None)
                frozen_rows[cfg] = sched.total_rows if sched is not None else None      # (exact: a frozen batch is not padded)
                rf = r["roofline"] or {}
                secondary.append({
                    "config": cfg, "workload": workload_name(cfg, args.batch, r["S"], r["ms"]),
                    "value": round(r["value"], 2), "unit": "images/sec", "ms_per_step": round(r["ms_per_step"], 3),
                    "ms_per_step_median": round(r["median_ms"], 3), "steps": r["steps"], "warmup": r["warmup"],
                    "step_graph": r["graph_note"] if r["graph_note"] is not None else bool(r["trainer"].use_graph),
                    **({"live_candidates_per_backward_tile": r["live_candidates_per_tile"]} if r.get("live_candidates_per_tile") else {}),
                    "roofline": {k: rf.get(k) for k in ("bound", "limiter", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                        "traffic_source", "frac_of_issue_bound")},
                    "kernels": {k: {f: v[f] for f in ("mean_ms", "alg_MB_per_launch", "frac") if f in v} for k, v in r["kernels"].items()},
                    "kernels_constants_stale": bool(r["constants"] and r["constants"]["stale"]),
                    **({"gemm": {"tunableop_table": "shipped (baseboostdepth_amd/gemm_db, tools/gemm_tune.sh)",
                                 "accepted_by_tunableop": tuning.STATUS["gemm_db_accepted"],
                                 "why_not": tuning.STATUS["gemm_db_why"]}} if cfg == "vit" else {})})
                release(r)
            except Exception as e:    # a secondary configuration never takes the headline down
                secondary.append({"config": cfg, "error": "%s: %s" % (type(e).__name__, str(e)[:200])})
                torch.cuda.set_sync_debug_mode("default")
                torch.cuda.synchronize()
        if line is not None:
            line["secondary"] = secondary
            line["secondary_seconds"] = round(time.perf_counter() - t_sec, 1)

    # ---- --dp-mode all: the other two multi-rank loops in the same launch, under a watchdog
    if world > 1 and args.dp_mode == "all":
        import threading
        print_lock = threading.Lock()
        modes = {"graph": {"value": line["value"], "ms_per_step": line["ms_per_step"], "exchange_ms": line["exchange_ms"]}} if line else {}
        if line is not None:
            line["dp_modes"] = modes
        for mode in ("overlap", "graph-overlap"):
            fired = threading.Event()

            def bail(mode=mode):
                # a hung collective cannot be cancelled from Python: print what has been measured and end the rank with the
                # status of a timeout (124, like `timeout(1)` and --launch-timeout) - the launcher and CI must see that a
                # mode hung, the line on stdout carries the headline and says which one
                fired.set()
                if line is not None:
                    with print_lock:
                        line["dp_modes"][mode] = "timeout after %.0f s (rank watchdog)" % args.dp_extra_timeout
                        line["partial"] = "dp mode %s timed out; exit status 124" % mode
                        print(json.dumps(line))
                        sys.stdout.flush()
                os._exit(124)
            dog = threading.Timer(args.dp_extra_timeout, bail)
            dog.daemon = True
            dog.start()
            try:
                r = run_workload(args, ctx, args.config, args.steps, min(args.warmup, 5), want_graph, mode)
                entry = {"value": round(r["value"], 2), "ms_per_step": round(r["ms_per_step"], 3), "dp_mode": r["dp_mode"],
                         "exchange_ms": None if r["dp_mode"] == "graph-overlap" else round(r["exchange_ms"], 4),
                         "buckets_launched_in_backward": r["overlapped"],
                         "step_graph": r["graph_note"] if r["graph_note"] is not None else bool(r["trainer"].use_graph)}
                release(r)
            except Exception as e:
                entry = "failed: %s: %s" % (type(e).__name__, str(e)[:160])
            dog.cancel()
            if fired.is_set():
                time.sleep(3600)       # the watchdog is printing / exiting
            if line is not None:
                with print_lock:       # (the watchdog thread may be printing `line`)
                    line["dp_modes"][mode] = entry

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["cpu_baseline"]["cpu_model"] = cpu_model()
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
