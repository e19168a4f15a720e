"""Secondary measurement modes of bench.py (kept out of the contract benchmark's file): the regimes a real run is in -
a NEW batch ordering every step (`run_fresh`: the `--rand` recipe redraws every sample's frame set per item,
mono_dataset.py:87-109, trainer.py:250, 867-886) and the loader in the loop (`run_loader_fed`) - plus what
they share with the headline (`make_options`, `workload_name`) and the on-box ceilings the bench line quotes
(`stream_copy_ceiling`, `cpu_model`).  Measurement only; nothing here is on the product path, and the legs of the benchmark that run the CPU checker
(the CPU baseline, the eager A/B) stay in bench.py."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 192, 640
SCALES = [0, 1, 2, 3]


def make_options(batch, device_index, config):
    import types
    config = config.replace("_fresh", "").replace("_coherent", "")
    boosted = config not in ("md2", "vit")
    one_scale = config in ("boosted", "boosted15")
    return types.SimpleNamespace(
        height=H, width=W, batch_size=batch, scales=([0] if one_scale else list(SCALES)), frame_ids=[0, -1, 1],
        min_depth=0.1, max_depth=100.0, disparity_smoothness=1e-3, no_ssim=False,
        trimin=boosted, decomp=boosted, pose_error=5.5, incremental_skip=boosted, partial_skip=boosted,
        materialize_warps=False, num_layers=18, weights_init="scratch", learning_rate=1e-4,
        no_cuda=False, cuda=device_index, load_weights_folder="None", log_dir="/tmp", model_name="bench",
        ViT=(config == "vit"))


def workload_name(config, batch, S, ms):
    if config.endswith("_coherent"):
        return workload_name(config.replace("_coherent", ""), batch, S, ms) + (
            "; COHERENT arg-min regime: planar scene, frames = true shifts, heads predicting it (synthetic.structured_batch), lr 0")
    net = "MonoViT (mpvit_small) encoder + HR DepthDecoder" if config == "vit" else "MD2 ResNet-18 encoder+DepthDecoder"
    if config in ("md2", "vit"):
        return "%s+PoseNet training step, 640x192, per-GPU batch %d, frames [0,-1,1], %d scales, HIP fused warp+SSIM+min" % (net, batch, S)
    return ("BaseBoostDepth boosted step (trimin+decomp+incremental+partial, config %s, per-sample max offsets %s), "
            "ResNet-18, 640x192, per-GPU batch %d, %d scale(s)" % (config, ms, batch, S))


def fresh_offsets(config, batch, n, seed=2025):
    """`n` draws of a batch's per-sample frame offsets, as the reference loader redraws them per item
    (mono_dataset.py:87-109; synthetic.draw_offsets) - stacked in canonical order like the device collate does."""
    import random as _random
    from baseboostdepth_amd.synthetic import draw_offsets
    rnd = _random.Random(seed)
    epoch = 15 if config.startswith("boosted15") else 5
    return [sorted(draw_offsets(rnd, batch, epoch, True), reverse=True) for _ in range(n)]


def run_fresh(args, ctx, config, want_graph, n_batches=30, pass_budget=20.0):
    """The regime a real `--rand` epoch runs in: EVERY step brings a new batch signature (the loader redraws each
    sample's frame set, trainer.py:250, 867-886).  Pre-resident batches with different orderings (90 for the boosted recipe,
    30 for the early curriculum, whose 91 possible signatures would repeat), every per-signature cache cold at the first
    step, device synchronise on both sides of each pass:
      pass 1  new signatures: every step builds + uploads its tables and runs eagerly (a signature that comes back within
              the pass is captured where the signature space is small).  With 90 orderings it is reported in two parts:
              steps 1-30 (which also hold the process's allocator growth) and steps 31-90 = the steady state of a regime
              in which no signature ever comes back
      pass 2  the first 30 batches again (second sighting: the early curriculum captures a step graph now; the boosted
              recipe stays eager, tables cached)
      pass 3  third sighting (replays where pass 2 captured)
    Reported per pass: ms/step, the train_step calls one by one (median, slowest three), table uploads and bytes per step
    (steptables.STATS), host table-build time, and the synchronising calls torch itself flags
    (torch.cuda.set_sync_debug_mode("warn"): a pageable host-to-device copy is one)."""
    import warnings
    n_batches = 90 if config.startswith("boosted") else n_batches          # (boosted: 30 cold-start + 60 steady-state orderings)
    from baseboostdepth_amd import ops, plan as plan_mod, steptables
    from baseboostdepth_amd.synthetic import synthetic_batch
    from baseboostdepth_amd.trainer import Trainer
    base = config.replace("_fresh", "")
    local, dev = ctx["local"], ctx["dev"]
    torch.manual_seed(42)
    opt = make_options(args.batch, local, base)
    opt.fused_adam = not args.no_fused_adam
    opt.step_graph = bool(want_graph)
    opt.rand = True
    small_space = base == "trimin5"          # 91 signatures: they come back; 18 564 (epoch >= 10): they do not
    opt.graph_capture_after = 1 if small_space else 1 << 30
    run_scales = list(opt.scales)
    opt.scales = list(SCALES)
    tr = Trainer(opt)
    tr.opt.scales = run_scales
    tr.set_train()
    draws = fresh_offsets(base, args.batch, n_batches)
    batches = []
    for i, ms in enumerate(draws):
        b = synthetic_batch(ms, H, W, run_scales, device=dev, seed=1000 + i)
        b.pop("noise")
        b["cutt"] = torch.tensor(1.35 if base.startswith("boosted") else 0.3)
        batches.append(b)
    signatures = len({tuple(ms) for ms in draws})
    # what a training run does before the first step of a curriculum phase (Trainer.run_epoch): the step graphs of the phase's
    # pose-row buckets are captured on synthetic batches (nothing trains) - allocator growth, MIOpen's first use of a row
    # count and the captures themselves happen HERE, not in the first minute of training.  `pooled_step = False` in the
    # options (BBD_POOLED_STEP=0) gives round 5's per-signature loop for A/B: it warms the same things on orderings outside
    # the draw.  Then every per-signature cache is dropped: the first timed step of every signature is cold
    torch.cuda.reset_peak_memory_stats(dev)
    t_pre = time.perf_counter()
    if tr.pooled_step:
        prewarm = tr.prewarm(epoch=15 if base.startswith("boosted") else 5, seed=7)
    else:
        def offsets_for_rows(rows):
            # epoch >= 10 recipe: the pass has 24 + 4 * sum(m - 1) rows (incremental + partial calls, batch 12)
            ms, want = [1] * args.batch, (rows - 24) // 4
            i = 0
            while sum(m - 1 for m in ms) < want and min(ms) < 7:       # (a small --batch cannot reach the larger row counts)
                if ms[i % args.batch] < 7:
                    ms[i % args.batch] += 1
                i += 1
            return sorted(ms, reverse=True)
        warm_sets = ([offsets_for_rows(r) for r in (160, 192, 208, 224, 240, 256, 272, 288, 320)] if base.startswith("boosted")
                     else [[2] * args.batch, [1] * args.batch, [2] * (args.batch // 2) + [1] * (args.batch - args.batch // 2)])
        keep = tr.capture_after
        tr.capture_after = 1 << 30
        for k, wms in enumerate(warm_sets):
            warm = synthetic_batch(wms, H, W, run_scales, device=dev, seed=7 + k)
            warm.pop("noise")
            warm["cutt"] = batches[0]["cutt"].clone()
            for _ in range(2):
                tr.train_step(dict(warm))
        del warm
        tr.capture_after = keep
        prewarm = {"buckets": len(warm_sets), "seconds": None}
    torch.cuda.synchronize()
    prewarm["seconds"] = round(time.perf_counter() - t_pre, 2)
    prewarm["graphs"] = len(tr._graphs)
    steptables._STEP_CACHE.clear()
    plan_mod._PLAN_CACHE.clear()
    tr.__dict__.pop("_index_cache", None)
    tr._sightings.clear()
    if tr._pooled is not None:
        tr._pooled.cache.clear()
    passes = []
    pad_rows = tr.pose_pad_rows
    # pass 1: new signatures, per-signature caches cold.  With 90 orderings it is reported in two parts: the first 30 steps
    # (which also contain the process's allocator growth: two or three calls of 0.3-3 s in which PyTorch's caching allocator
    # meets a sequence of activation sizes it has no free block for - profiles/r05/fresh_cold_step_profile.txt) and the
    # following 60 = the steady state of a regime in which no signature ever comes back.  Passes 2, 3: the first 30 again.
    segments = [("1 (steps 1-30: cold start)", batches[:30]), ("1 (steps 31-90: every signature new, steady state)", batches[30:]),
                ("2", batches[:30]), ("3", batches[:30])] if len(batches) > 30 else [("1", batches), ("2", batches), ("3", batches)]
    for p, (label, seg) in enumerate(segments):
        steptables.reset_stats()
        g0, launch0 = dict(tr.graph_stats), tr.launch_wait_s
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            torch.cuda.set_sync_debug_mode("warn")
            torch.cuda.synchronize()
            t0, cpu0 = time.perf_counter(), time.thread_time()
            done, per_call, rows_real, rows_run = 0, [], [], []
            prof_step = int(os.environ.get("BBD_BENCH_PROFILE_STEP", "-1")) if p == 0 else -1      # (diagnosis: cProfile one call)
            n_seg = len(seg)
            for b in seg:
                c0 = time.perf_counter()
                if done == prof_step:
                    import cProfile, pstats
                    pr = cProfile.Profile()
                    pr.enable()
                    tr.train_step(dict(b))
                    pr.disable()
                    pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(14)
                else:
                    tr.train_step(dict(b))
                per_call.append(time.perf_counter() - c0)
                if tr.last_pooled is not None:
                    rows_real.append(tr.last_pooled.n_real)
                    rows_run.append(tr.last_pooled.R)
                done += 1
                if done in (3, 10) and time.perf_counter() - t0 > pass_budget * done / 30 * 3:
                    torch.cuda.synchronize()      # far over budget (e.g. MIOpen compiling solvers for unseen row counts)
                    break
            t_host = time.perf_counter() - t0          # the training thread is done enqueueing here
            t_cpu = time.thread_time() - cpu0          # ... and this is how much of that it spent ON the CPU (not waiting)
            torch.cuda.set_sync_debug_mode("default")
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        sync_w = [w for w in caught if "synchroniz" in str(w.message).lower()]
        syncs = len(sync_w)
        sync_where = sorted({"%s:%d" % (os.path.relpath(w.filename, ROOT) if w.filename.startswith(ROOT) else os.path.basename(w.filename),
                                        w.lineno) for w in sync_w})
        st = dict(steptables.STATS)
        g1 = dict(tr.graph_stats)
        passes.append({"pass": label, "steps": done, "ms_per_step": round(dt / done * 1e3, 3),
                       "images_per_sec": round(args.batch * done / dt, 2),
                       # how long the training thread took to ENQUEUE the steps (if this is well below ms_per_step the loop is
                       # bound by the GPU, not by the host)
                       "host_enqueue_ms_per_step": round(t_host / done * 1e3, 3),
                       # CPU time of the training thread per step (time.thread_time): with graph replays the wall time above is
                       # mostly the thread WAITING in a launch call for room in the queue - it runs two steps ahead of the GPU
                       "host_cpu_ms_per_step": round(t_cpu / done * 1e3, 3),
                       # the training thread's time OUTSIDE hipGraphLaunch (table look-up / build, the copies into the pool,
                       # Python): what the step costs the host; the launch call itself waits (busily) for room in the queue
                       "host_ms_outside_graph_launch_per_step": round((t_host - (tr.launch_wait_s - launch0)) / done * 1e3, 3),
                       # the train_step CALLS one by one: a few slow ones (a first use of something) or all of them?
                       "host_call_ms_median": round(sorted(per_call)[len(per_call) // 2] * 1e3, 2),
                       "host_call_ms_slowest3": [round(v * 1e3, 1) for v in sorted(per_call)[-3:]],
                       "host_call_slowest_step": int(max(range(len(per_call)), key=per_call.__getitem__)),
                       "table_uploads_per_step": round((st["packed_uploads"] + st["single_uploads"]) / done, 3),
                       "table_bytes_per_step": int(st["packed_words"] * 4 / done),
                       "table_build_ms_per_step": round(st.get("build_ms", 0.0) / done, 3),
                       "synchronising_calls_per_step": round(syncs / done, 3), "synchronising_calls_at": sync_where,
                       "eager_steps": g1["eager"] - g0["eager"], "captures": g1["captures"] - g0["captures"],
                       "replays": g1["replays"] - g0["replays"],
                       # rows of the batched pose pass (the step's dominant cost under the boosted recipe): what the batches'
                       # frame sets ask for, and what ran after rounding up to a measured row count
                       "pose_rows_mean": round(sum(rows_real) / len(rows_real), 1) if rows_real else None,
                       "pose_rows_run_mean": round(sum(rows_run) / len(rows_run), 1) if rows_run else None})
    steady = passes[1] if len(segments) == 4 else passes[0]
    out = {"config": config, "workload": "%s with a NEW ordering every step: %d pre-resident batches, %d distinct signatures, "
                                          "offsets drawn per sample like mono_dataset.py:87-109, stacked largest offset first"
                                          % (workload_name(base, args.batch, len(run_scales), "redrawn per step"), n_batches, signatures),
           "value": steady["images_per_sec"], "unit": "images/sec", "ms_per_step": steady["ms_per_step"],
           "steps": steady["steps"], "passes": passes, "step_graph": bool(tr.use_graph),
           "cold_start_images_per_sec": passes[0]["images_per_sec"],
           "graph_capture_after": None if opt.graph_capture_after >= 1 << 30 else opt.graph_capture_after,
           "pose_pad_rows": pad_rows, "pooled_step": bool(tr.pooled_step), "prewarm": prewarm,
           "step_graphs_in_use": len(tr._graphs),
           "pooled_fallbacks": tr._pooled.stats["fallbacks"] if tr._pooled is not None else None,
           # device memory with every bucket graph of the phase captured (the graphs share one pool)
           "max_memory_allocated_GB": round(torch.cuda.max_memory_allocated(dev) / 1e9, 2),
           "memory_reserved_GB": round(torch.cuda.memory_reserved(dev) / 1e9, 2),
           "what": ("value = every signature new, per-signature caches cold: steps 31-90 of pass 1; cold_start_images_per_sec = "
                    "its first 30 steps, right after Trainer.prewarm() (pooled form: the step graphs of the phase's pose-row "
                    "buckets, captured before the first step - `prewarm`); last pass = every signature seen before"
                    if len(segments) == 4 else
                    "value = pass 1 (every per-signature cache cold, right after Trainer.prewarm(): ONE step graph serves every "
                    "ordering of the early curriculum); pass 3 = every signature seen before")}
    del tr, batches
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def stream_copy_ceiling(device, gib=1.0, reps=5):
    """On-box stream-copy rate: a float4 device copy of `gib` GiB (bbd_stream_copy with 1, 4 and 8 independent
    loads per thread in flight; one launch pair to warm, then `reps` timed with HIP events on the launch stream each), GB/s of
    read + written bytes of the best.  < 1 s, outside any timed region."""
    from . import _lib
    lib = _lib.get_lib()
    n = int(gib * (1 << 30)) // 4 // 4 * 4
    src = torch.empty(n, device=device, dtype=torch.float32).normal_()
    dst = torch.empty_like(src)
    best, per = None, {}
    for unroll in (1, 4, 8):
        launch = lambda: lib.call("bbd_stream_copy", _lib.ptr(src), _lib.ptr(dst), n, unroll, lib.stream_for(src))
        launch(), launch()
        torch.cuda.synchronize(device)
        evs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize(device)
        ms = min(a.elapsed_time(b) for a, b in evs)
        per[unroll] = round(2 * 4 * n / (ms * 1e-3) / 1e9, 1)
        best = ms if best is None or ms < best else best
    del src, dst
    return {"GBps": round(2 * 4 * n / (best * 1e-3) / 1e9, 1), "GiB": gib, "best_ms": round(best, 4), "GBps_by_loads_in_flight": per,
            "what": "float4 grid-stride device copy (bbd_stream_copy), read + written bytes / best of %d launches" % reps}


def cpu_model():
    """Model name of the host CPU (/proc/cpuinfo) for the `cpu_baseline` block."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_quota():
    """CPUs' worth of time this process's cgroup may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited -
    `sched_getaffinity` says which CPUs are schedulable, not how many can run at once."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else round(int(quota) / int(period), 2)
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = int(f.read())
        return None if q <= 0 else round(q / p, 2)
    except Exception:
        return None


def run_loader_fed(args, ctx, want_graph, steps=60, warmup=20, workers=None):
    """SURVEY 8f-3's purpose - real-data images/sec: the MD2 step fed by the loader instead of a pre-resident batch.
    A synthetic KITTI-raw tree of JPEGs (KITTI's sizes; synthetic.synthetic_kitti_tree) -> `datasets.KITTIRAWDataset`
    (frame-set selection + JPEG decode in worker processes, mono_dataset.py:76-146) -> shared pinned ring -> `DeviceCollate`
    (resize / pyramid / colour jitter / ToTensor / stacking as HIP kernels) -> `Trainer.train_step`
    (trainer.py:214-220, 232-264).  Also times the loader alone (no training) = the host decode ceiling."""
    import shutil
    import tempfile
    from baseboostdepth_amd import datasets
    from baseboostdepth_amd.synthetic import synthetic_kitti_tree
    from baseboostdepth_amd.trainer import Trainer
    local, dev = ctx["local"], ctx["dev"]
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    quota = cpu_quota()
    cores = int(min(avail, quota)) if quota else avail
    # decode workers: the cores this process may actually use, minus two for the training thread and the collate thread
    workers = workers or max(4, min(24, cores - 2))
    tmp = tempfile.mkdtemp(prefix="bbd_kitti_")
    try:
        lines = synthetic_kitti_tree(tmp, frames=40) * 40          # 3 840 split lines over 160 JPEG files
        torch.manual_seed(42)
        opt = make_options(args.batch, local, "md2")
        opt.fused_adam = not args.no_fused_adam
        opt.step_graph = bool(want_graph)
        tr = Trainer(opt)
        tr.set_train()

        cache = datasets.FrameCache(dev, 2 << 30)          # (the synthetic tree is 160 files = 0.22 GB decoded)

        def make_loader(use_cache):
            ds = datasets.KITTIRAWDataset(lines, 0, H, W, kt_path=tmp, rand=False, is_train=True, scales=opt.scales, kt=True,
                                          naive_mix=True, trimin=False, seed=1)
            collate = datasets.DeviceCollate(H, W, opt.scales, dev, cache=cache if use_cache else None)
            return datasets.DeviceLoader(ds, args.batch, collate, num_workers=workers, prefetch=3, seed=0, workers="process")

        def drain(loader, n_warm, n_steps, step):
            """-> (images/s, images, last loss, last batch, {ms per batch in steady state: waiting for the loader's next batch,
            inside train_step (host side of the step: table look-ups, copies into the graph's static inputs, the graph
            launch), and the loader's own producer-side times})."""
            n, t0, last, batch, t_get, t_step, snap = 0, None, None, None, 0.0, 0.0, None
            it = iter(loader)
            i = 0
            while True:
                a = time.perf_counter()
                batch = next(it, None)
                b = time.perf_counter()
                if batch is None:
                    break
                if i == n_warm:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    t_get = t_step = 0.0
                    snap = dict(loader.stats)
                elif i > n_warm:
                    t_get += b - a
                if step is not None:
                    c = time.perf_counter()
                    last = step(batch)
                    t_step += time.perf_counter() - c
                if i >= n_warm:
                    n += args.batch
                if i == n_warm + n_steps - 1:
                    break
                i += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            st, k = loader.stats, max(n // args.batch, 1)
            loop = {"consumer_waited_for_batch_ms": round(t_get / k * 1e3, 2), "consumer_in_train_step_ms": round(t_step / k * 1e3, 2),
                    "producer_fetch_ms": round((st["fetch_s"] - snap["fetch_s"]) / max(st["batches"] - snap["batches"], 1) * 1e3, 2),
                    "producer_collate_ms": round((st["collate_s"] - snap["collate_s"]) / max(st["batches"] - snap["batches"], 1) * 1e3, 2)}
            it.close() if hasattr(it, "close") else None
            return n / dt, n, last, batch, loop

        # (0) one host thread decoding JPEGs (Pillow, what the reference's loader does per frame): frames per second per core
        from PIL import Image
        import glob as _glob
        files = sorted(_glob.glob(os.path.join(tmp, "**", "*.jpg"), recursive=True))[:40]
        t0 = time.perf_counter()
        for fpath in files:
            with Image.open(fpath) as im:
                im.convert("RGB").load()
        one_thread_decode = len(files) / (time.perf_counter() - t0)
        # (a) the loader alone, every frame decoded at every use (the reference's behaviour): the host's decode ceiling
        loader_alone, _, _, batch, _ = drain(make_loader(False), 3 * workers + 8, 200, None)     # (past the prefetched backlog)
        frames_per_sample = sum(1 for k in batch if isinstance(k, tuple) and k[0] == "color" and k[2] == 0)
        train = lambda b: tr.train_step(b)[1]["loss"]
        # (b) the step fed by it
        fed_decode, _, _, _, loop_decode = drain(make_loader(False), warmup, steps, train)
        # (c) the step fed through the HBM-resident frame cache: a frame is decoded once, later uses are table entries.  The
        #     DataLoader hands `prefetch x workers` batches to the decode workers before the first one is collated - those
        #     were planned against an empty cache; a pass over them first (a real epoch has 3 317 batches, these are its
        #     first forty), then the measurement
        drain(make_loader(True), 0, 3 * workers + 8, train)
        fed_cached, n, last, _, loop_ms = drain(make_loader(True), warmup, steps, train)
        finite = bool(torch.isfinite(last.detach()).item())
        cache_stats = cache.stats()
        del cache
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {"config": "md2_loader", "workload": "MD2 step fed by datasets.KITTIRAWDataset + DeviceCollate on a synthetic KITTI-raw JPEG "
                                                "tree (1242x375 -> 640x192, %d JPEG frames per sample: 0, -1, +1, stereo), batch %d"
                                                % (frames_per_sample, args.batch),
           "value": round(fed_cached, 2), "unit": "images/sec", "ms_per_step": round(args.batch / fed_cached * 1e3, 3), "steps": n // args.batch,
           "warmup": warmup, "step_graph": bool(tr.use_graph), "loss_finite": finite,
           "what": "value = the step fed by the loader with decoded frames resident in HBM (datasets.FrameCache: each JPEG is "
                   "decoded once); decode_every_use_images_per_sec = the same loop decoding every frame at every use, as the "
                   "reference's loader does",
           "decode_every_use_images_per_sec": round(fed_decode, 2),
           "loader_alone_images_per_sec": round(loader_alone, 1),
           "host_decode_frames_per_sec": round(loader_alone * frames_per_sample, 1),
           "frame_cache": cache_stats, "loop_ms_per_batch": loop_ms, "loop_ms_per_batch_decode_every_use": loop_decode,
           "decode_workers": workers, "schedulable_cpus": avail, "cgroup_cpu_quota": quota,
           "one_thread_decode_frames_per_sec": round(one_thread_decode, 1),
           # KITTI (Eigen-Zhou): 39 810 samples name ~45 000 distinct frames -> 1.13 first-time decodes per sample in epoch 1
           "kitti_epoch1_decodes_per_sample": 1.13,
           "data": "synthetic JPEG tree (decoded, resized, jittered for real)"}
    out["limiter_without_cache"] = ("host JPEG decode: the loader alone delivers %.0f images/s (%d worker processes, %.0f frames/s "
                                    "per thread measured here, %s CPUs usable)" % (loader_alone, workers, one_thread_decode,
                                                                                   quota if quota else avail))
    del tr
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out



