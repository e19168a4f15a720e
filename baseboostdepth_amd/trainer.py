"""`Trainer` with the reference's step interface (trainer.py:29-570), MI355X-native inside.

Kept from the reference (so train/eval scripts are drop-in): `Trainer(opts)`, `.models`,
`.process_batch(inputs, batch_idx=None, is_train=True) -> (outputs, losses)`, `.predict_poses`,
`.generate_images_pred`, `.compute_losses`, `.compute_reprojection_loss`, `.valid_frames_trimin`,
`.set_train/.set_eval`, `.save_model/.load_model`, the `outputs` / `losses` key conventions and the
checkpoint layout.  Different by design: the sub-batch masks become one index table
(`plan.ReprojectionPlan`), and warping + SSIM/L1 + per-pixel min over all candidates run as ONE
fused HIP launch per step (forward) and one for its backward; warped images are materialised
only on request (`opt.materialize_warps`) because nothing in the loss needs them.
"""
import json
import os
import time

import torch
import torch.optim as optim

from . import networks, ops, steptables
from .layers import SSIM, BackprojectDepth, Project3D, disp_to_depth, transformation_from_parameters
from .plan import STEREO, canonical_permutation, get_plan, owners_of, sample_max_offsets


def _frame_sort_key(item):
    return float("inf") if isinstance(item, str) else abs(item)


class Trainer:
    _local_only = False      # True only inside a graph capture's warm-up steps: gradients are not exchanged
    dp_capture = False       # data parallel under a step graph: RCCL all-reduces captured into the graph (opt-in)
    _main_stream = None
    max_graphs = 8
    pose_pad_rows = 0
    pooled_step = False      # `--rand`: the step in pooled form (`pooled.PooledStep`), one step graph per pose-row bucket
    _pooled = None
    last_pooled = None       # `pooled.PooledTables` of the last batch that ran in pooled form
    launch_wait_s = 0.0      # seconds spent inside hipGraphLaunch (graph.replay) so far

    def __init__(self, options, backend=None):
        self.opt = options
        opt = self.opt
        if not getattr(opt, "no_cuda", False):
            from . import tuning
            tuning.use_shipped_db()      # MIOpen reads its database path at the first convolution (explicit, not at import)
            if getattr(opt, "ViT", False) or os.environ.get("BBD_GEMM_DB") == "1":
                # hipBLASLt / rocBLAS solutions of MonoViT's token GEMMs (tuning itself stays off).  Only the MonoViT
                # configuration turns TunableOp on - it is process-wide, and the ResNet configurations have no GEMMs
                tuning.use_shipped_gemm_db()
        self.log_path = os.path.join(getattr(opt, "log_dir", "."), getattr(opt, "model_name", "mdp"))
        assert opt.height % 32 == 0, "'height' must be a multiple of 32"
        assert opt.width % 32 == 0, "'width' must be a multiple of 32"
        self.device = torch.device("cpu" if getattr(opt, "no_cuda", False) else "cuda:%d" % getattr(opt, "cuda", 0))
        self.num_scales = len(opt.scales)          # frozen here; compute_losses divides by it (trainer.py:44,568)
        self.num_pose_frames = 2
        self.backend = backend

        self.models = {}
        self.parameters_to_train = []
        self.use_vit = bool(getattr(opt, "ViT", False))
        if self.use_vit:
            # MonoViT (BASELINE configs[4], trainer.py:52-58): MPViT-small encoder + HR decoder.  The
            # encoder's parameters go into their own optimizer group below, not into parameters_to_train.
            from . import networksvit
            ckpt = getattr(opt, "mpvit_checkpoint", "./ckpt/mpvit_small.pth") if opt.weights_init == "pretrained" else None
            self.models["encoder"] = networksvit.mpvit_small(checkpoint=ckpt)
            self.models["encoder"].num_ch_enc = [64, 128, 216, 288, 288]
            self.models["depth"] = networksvit.DepthDecoder()
        else:
            self.models["encoder"] = networks.ResnetEncoder(opt.num_layers, opt.weights_init == "pretrained")
            self.models["depth"] = networks.DepthDecoder(self.models["encoder"].num_ch_enc, opt.scales)
        self.models["pose_encoder"] = networks.ResnetEncoder(18, opt.weights_init == "pretrained",
                                                             num_input_images=self.num_pose_frames)
        self.models["pose"] = networks.PoseDecoder(self.models["pose_encoder"].num_ch_enc,
                                                   num_input_features=1, num_frames_to_predict_for=2)
        for name in ("encoder", "depth", "pose_encoder", "pose"):
            self.models[name].to(self.device)
            if not (self.use_vit and name == "encoder"):
                self.parameters_to_train += list(self.models[name].parameters())
        # same optimizer hyper-parameters as the reference (trainer.py:106-111): Adam(lr) for the ResNet
        # path; AdamW with two groups for MonoViT - everything but the encoder at 1e-4, the encoder at 5e-5.
        # On the GPU the single-kernel "fused" implementation replaces ~20 multi-tensor launches per step
        fused = self.device.type == "cuda" and getattr(opt, "fused_adam", True)
        self.use_graph = bool(fused and (getattr(opt, "step_graph", False) or os.environ.get("BBD_STEP_GRAPH") == "1"))
        if self.use_vit:
            self.params = [{"params": self.parameters_to_train, "lr": 1e-4},
                           {"params": list(self.models["encoder"].parameters()), "lr": 5e-5}]
            self.model_optimizer = optim.AdamW(self.params, fused=fused, capturable=self.use_graph)
        else:
            self.model_optimizer = optim.Adam(self.parameters_to_train, opt.learning_rate, fused=fused,
                                              capturable=self.use_graph)
        # every parameter the optimizer steps, in group order (the gradient exchange packs exactly these)
        self.optimizer_parameters = [p for g in self.model_optimizer.param_groups for p in g["params"]]
        self._graphs = {}
        # step graphs share ONE memory pool (a step's graph is never replayed concurrently with another's and takes fresh
        # static inputs), so a cached signature costs its static input copies, not a private pool: 128 of them by default
        self.max_graphs = max(1, int(os.environ.get("BBD_MAX_GRAPHS", "128")))      # (0 / negative would empty an empty LRU)
        self._graph_pool = None
        # a batch signature is captured when it comes back for the (capture_after + 1)-th time; until then its steps run
        # eagerly.  Fixed-frame-set training (MD2) has one signature: 0.  The boosted recipe (--rand) redraws every
        # sample's frame set per item (mono_dataset.py:87-109): from epoch 10 on a batch of 12 has 18 564 possible
        # signatures, hardly any comes back and capturing each (a warm-up step + the capture) would cost more than it
        # saves; the early curriculum's 91 signatures all come back within a few hundred steps
        self.capture_after = int(getattr(opt, "graph_capture_after", 2 if getattr(opt, "rand", False) else 0))
        self._sightings = steptables.LRU(8192)
        # rows the batched pose pass is rounded up to (0 = exact): see `_pose_pairs`.  On for the recipes whose frame sets
        # change per batch; fixed-frame-set training has ONE row count and pays nothing
        self.pose_pad_rows = int(getattr(opt, "pose_pad_rows", 32 if getattr(opt, "rand", False) else 0))
        self._capture_checked = False
        self.graph_stats = {"replays": 0, "captures": 0, "eager": 0}
        # `--rand` recipes: the step in pooled form (`pooled.PooledStep`: one frame pool, static step tables, shapes that
        # depend on the padded pose rows only), so that ONE step graph serves every ordering of a row-count bucket
        want_pooled = getattr(opt, "pooled_step", None)
        want_pooled = getattr(opt, "rand", False) if want_pooled is None else want_pooled
        self.pooled_step = bool(want_pooled) and self.device.type == "cuda" and os.environ.get("BBD_POOLED_STEP", "1") != "0"
        self._pooled = None
        # data parallel + step graph: capture the bucketed RCCL all-reduces INTO the graph (one graph per step, exchange
        # overlapped with backward inside the replay) instead of two graphs around one exposed all-reduce.  Opt-in:
        # multi-rank RCCL capture cannot be exercised on the one-GPU boxes this was built on
        self.dp_capture = bool(getattr(opt, "dp_capture", False) or os.environ.get("BBD_DP_CAPTURE") == "1")
        self.model_lr_scheduler = optim.lr_scheduler.MultiStepLR(
            self.model_optimizer, milestones=[11, 13, 15, 16, 17, 18, 19], gamma=0.4)
        if getattr(opt, "load_weights_folder", "None") not in (None, "None"):
            self.load_model()

        self.ssim = SSIM()
        self.backproject_depth = {0: BackprojectDepth(opt.batch_size, opt.height, opt.width)}
        self.project_3d = {0: Project3D(opt.batch_size, opt.height, opt.width)}
        self.depth_metric_names = ["de/abs_rel", "de/sq_rel", "de/rms", "de/log_rms", "da/a1", "da/a2", "da/a3"]
        self.grad_sync = None      # set by distributed.attach(): called between backward and optimizer.step
        self.flat_grads = None
        self.epoch, self.step = 0, 0

    def gradient_free_parameters(self):
        """Trainable parameters no loss ever reaches: torchvision-layout ResNets carry an `fc` head the
        encoders never call - they stay in `parameters_to_train` (Adam state layout as the
        reference's `adam.pth`) but the gradient exchange must not wait for them."""
        out = []
        for name in ("encoder", "pose_encoder"):
            enc = getattr(self.models.get(name), "encoder", None)
            fc = getattr(enc, "fc", None)
            if fc is not None:
                out += list(fc.parameters())
        if getattr(self, "use_vit", False):
            # the HR decoder builds X_0j_Conv_0 (j = 0..3) and never calls them (reference
            # networksvit/hr_decoder.py:36-48 vs its forward)
            for j in range(4):
                out += list(self.models["depth"].convs["X_0%d_Conv_0" % j].parameters())
        return out

    # ------------------------------------------------------------------ mode switches
    def set_train(self):
        for m in self.models.values():
            m.train()

    def set_eval(self):
        for m in self.models.values():
            m.eval()

    def _pose_stream(self):
        """Second HIP stream for the pose network (BBD_POSE_STREAM=0 disables it)."""
        if self.device.type != "cuda" or os.environ.get("BBD_POSE_STREAM", "1") == "0":
            return None
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        return self._side_stream

    def _backend(self):
        if self.backend is None:
            self.backend = ops.default_backend()
        return self.backend

    # ------------------------------------------------------------------ the step (trainer.py:250-263)
    # ------------------------------------------------------------------ whole-step hipGraph (opt-in)
    def _graph_key(self, inputs):
        """Batches that replay the same graph: same candidate plan, same tensor shapes, same lr."""
        shapes = tuple(sorted((str(k), tuple(v.shape)) for k, v in inputs.items() if torch.is_tensor(v) and v.is_cuda))
        # (train_step has made `cutt` a host value: float() of a device tensor here would be a device -> host sync per step)
        return (str(inputs["ordering"]), str(inputs.get("frames")), float(inputs["cutt"]), tuple(self.opt.scales), shapes,
                tuple(g["lr"] for g in self.model_optimizer.param_groups))

    def _graph_signature(self, inputs):
        """(pooled tables | None, graph key, noise handed in?) of a batch.  In pooled form (`--rand`) the step's launches
        depend on the batch through the padded pose rows, the group grid of the pose pass's BatchNorms and its row bound only:
        ONE graph per row-count bucket, whatever the ordering (14-19 per epoch from epoch 10 on, seven for the early curriculum);
        otherwise the key is the whole signature."""
        tab = None
        if self.pooled_step:
            self.valid_frames = list(set([el for sub in inputs["ordering"] for el in sub if el != 0]))
            self.valid_frames_trimin(inputs)
            tab = self._pooled_tables(inputs)
        has_noise = inputs.get("noise") is not None
        if tab is not None:
            key = ("pooled", tab.R, tab.G, tab.bound, has_noise, bool(self.maxing_valid_frames), tuple(self.opt.scales),
                   tuple(g["lr"] for g in self.model_optimizer.param_groups))
        else:
            key = self._graph_key(inputs)
        return tab, key, has_noise

    def _snapshot_state(self):
        params = [p for g in self.model_optimizer.param_groups for p in g["params"]]
        buffers = [b for m in self.models.values() for b in m.buffers()]
        had_state = len(self.model_optimizer.state) > 0
        return (params, buffers, [p.detach().clone() for p in params], [b.detach().clone() for b in buffers],
                {p: {k: v.clone() for k, v in st.items() if torch.is_tensor(v)}
                 for p, st in self.model_optimizer.state.items()} if had_state else None, self.step)

    def _restore_state(self, snap):
        params, buffers, snap_p, snap_b, snap_s, step0 = snap
        with torch.no_grad():
            for p, v in zip(params, snap_p):
                p.copy_(v)
            for b, v in zip(buffers, snap_b):
                b.copy_(v)
            for p, st in self.model_optimizer.state.items():
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if snap_s is not None and p in snap_s and k in snap_s[p]:
                            v.copy_(snap_s[p][k])
                        else:
                            v.zero_()          # fresh Adam state: exp_avg = exp_avg_sq = step = 0
        self.step = step0

    def _capture(self, key, inputs, tab, has_noise, first_checked_capture=False):
        """Warm-up (that must not train) + capture of `process_batch + backward + optimizer.step` for a graph key."""
        # graphs of another learning rate can never replay again (MultiStepLR only moves forward): they go first - a phase of
        # the boosted recipe holds ~35 bucket graphs (42 GB allocated / 59 GB reserved at batch 12), a few generations of
        # them would not fit the device
        lrs = tuple(g["lr"] for g in self.model_optimizer.param_groups)
        for k in [k for k in self._graphs if k[-1] != lrs]:
            self._graphs.pop(k)
        # bound the cache: the least recently replayed signature goes first
        while len(self._graphs) >= self.max_graphs:
            self._graphs.pop(next(iter(self._graphs)))
        if self._graph_pool is None:
            self._graph_pool = torch.cuda.graph_pool_handle()
        pool = self._graph_pool
        self.graph_stats["captures"] += 1
        # a signature that has already run eagerly has warmed MIOpen / the allocator for its shapes: one warm-up step
        warm_steps = 1 if ((self._sightings.get(key) or 0) > 0 or (tab is not None and self._graphs)) else 3
        if tab is not None:
            static = dict(inputs)           # the pooled step owns its static buffers: `load` copies the batch into them
            scales = list(self.opt.scales)

            def run_forward():
                return self._pooled.forward(tab.R, tab.G, tab.bound, has_noise)
        else:
            static = {k: (v.clone() if torch.is_tensor(v) and v.is_cuda else v) for k, v in inputs.items()}

            def run_forward():
                return self.process_batch(dict(static))
        # eager warm-up (allocator, MIOpen solutions, Adam state tensors) must not TRAIN: parameters,
        # BatchNorm buffers, the optimizer state and the step counter are restored afterwards, so the
        # first batch of a signature is counted once (by the replay that follows), like on the eager path
        snap = self._snapshot_state()
        warm = torch.cuda.Stream(device=self.device)
        warm.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(warm):
            # NO collective in the warm-up: a cache miss is a per-rank event (ranks draw different frame sets
            # under --rand / the boosted recipe), so a rank that warms up must issue exactly the collectives of a
            # rank that replays - the ONE exchange after the replay - or the ranks' all-reduces mis-pair
            self._local_only = True
            if hasattr(self.grad_sync, "paused"):
                self.grad_sync.paused = True
            try:
                for _ in range(warm_steps):
                    self._eager_step(dict(static))
            finally:
                self._local_only = False
                if hasattr(self.grad_sync, "paused"):
                    self.grad_sync.paused = False
            self._restore_state(snap)
        torch.cuda.current_stream(self.device).wait_stream(warm)
        graph, tail = torch.cuda.CUDAGraph(), None
        if tab is not None:
            self._pooled.load(static, tab, scales)
        if self.grad_sync is None:
            self.model_optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(graph, pool=pool):
                outputs, losses = run_forward()
                losses["loss"].backward()
                self.model_optimizer.step()
        elif self.dp_capture:
            # Captured collectives have only ever run with the ONE rank a single-GPU box allows, so the
            # first captured signature is checked against the eager data-parallel step on the same batch: one eager
            # step WITH its exchange gives the reference gradients, the state is restored, and after the first replay
            # the flat gradient buffer must agree (every rank takes this path on its first signature: same collectives on
            # all of them).  BBD_DP_CAPTURE_CHECK=0 skips it.
            # (a one-shot flag, not "the cache is empty": LRU eviction empties the cache on a per-rank event, and a
            # rank that re-ran the check alone would issue an exchange its peers do not - ADVICE r4)
            check = first_checked_capture and os.environ.get("BBD_DP_CAPTURE_CHECK", "1") != "0"
            self._capture_checked = True
            if check:
                snap2 = self._snapshot_state()
                self._eager_step(dict(static))
                self._capture_reference = self.flat_grads.flat.detach().clone()      # the exchanged (averaged) gradients
                self._restore_state(snap2)
                if tab is not None:
                    self._pooled.load(static, tab, scales)
            # data parallel, collectives captured: ONE graph.  The post-accumulate hooks fire while backward is being
            # captured, so every bucket's pack + RCCL all-reduce becomes a node on RCCL's stream, forked from and
            # joined to the capturing stream by the work handles' waits - the replay overlaps the exchange with
            # the rest of backward like the eager overlapped loop does, with no host in between
            self.flat_grads.zero()
            with torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local"):
                outputs, losses = run_forward()
                losses["loss"].backward()
                self.grad_sync()
                self.model_optimizer.step()
        else:
            # data parallel: the step is split in two graphs around ONE eager exchange of the flat gradient
            # buffer - forward + backward + pack | all-reduce (RCCL, outside any capture) | optimizer.  The
            # collective's ~1 ms is not overlapped with backward, in exchange for ~1 350 launches per step
            # leaving the host (an eager multi-rank loop is host-bound on a busy node).  thread_local capture
            # mode: the process group's watchdog thread may query events while this thread captures.
            self.flat_grads.zero()
            with torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local"):
                outputs, losses = run_forward()
                losses["loss"].backward()
                self.flat_grads.pack()
            tail = torch.cuda.CUDAGraph()
            with torch.cuda.graph(tail, pool=pool, capture_error_mode="thread_local"):
                self.model_optimizer.step()
        # the graph's launches hold raw device addresses of the step tables / plan buffers: the entry keeps them alive
        # itself (not only through the retained autograd nodes of `losses`)
        keep = (getattr(self, "tables", None), getattr(self, "plan", None), self._pooled)
        entry = self._graphs[key] = (graph, tail, None if tab is not None else static, outputs, losses, keep)
        return entry

    def _graph_step(self, inputs):
        """Capture `process_batch + backward + optimizer.step` for this batch's graph key once, then replay it:
        ~1 340 launches per step become one hipGraphLaunch, which takes the training thread's 16 ms of
        launch work off the host.  Fixed-frame-set training (the MD2 baseline) has one key; BaseBoostDepth's `--rand`
        batches change their frame sets per batch and run in pooled form, where the key is the row-count bucket."""
        tab, key, has_noise = self._graph_signature(inputs)
        entry = self._graphs.get(key)
        if entry is not None:
            self._graphs[key] = self._graphs.pop(key)       # most recently used last
        # data parallel with captured collectives: the first capture is checked against the eager exchange, which takes one
        # extra exchange - so it happens on every rank's FIRST call (the same global step), never on a later, per-rank miss
        first_checked_capture = bool(self.grad_sync is not None and self.dp_capture and not self._capture_checked)
        if entry is None and not first_checked_capture:
            seen = self._sightings.get(key) or 0
            if seen < (0 if tab is not None else self.capture_after):      # (a bucket's graph is captured at first sight)
                # not captured (yet): an eager step issues exactly the collectives of a replaying rank (one exchange of
                # the flat buffer in the split-graph loop, the same buckets in the same order with captured collectives)
                self._sightings.put(key, seen + 1)
                self.graph_stats["eager"] += 1
                return self._eager_step_off_default_stream(inputs)
        if entry is None:
            entry = self._capture(key, inputs, tab, has_noise, first_checked_capture)
        graph, tail, static, outputs, losses, _ = entry
        self.graph_stats["replays"] += 1
        if tab is not None:
            self._pooled.load(inputs, tab, list(self.opt.scales))
            outputs = self._pooled.with_pose_views(outputs, tab)
        else:
            for k, v in inputs.items():
                if torch.is_tensor(v) and v.is_cuda:
                    static[k].copy_(v, non_blocking=True)
        t_launch = time.perf_counter()
        graph.replay()
        # (hipGraphLaunch returns when the runtime has room for the graph's ~1 300 nodes: with the thread two steps ahead of the
        # GPU that is a busy wait inside the call, not host work - kept apart so that a benchmark can tell the two)
        self.launch_wait_s += time.perf_counter() - t_launch
        if tail is not None:
            self.grad_sync.exchange()
            tail.replay()
        self._verify_captured_exchange()
        self.step += 1
        return outputs, losses

    def _verify_captured_exchange(self):
        """First replay of the first captured-collective graph vs the eager data-parallel step on the same batch (see
        `_capture`); the bar is the run-to-run spread of MIOpen's atomics-based weight gradients, far below a wrong exchange."""
        ref = getattr(self, "_capture_reference", None)
        if ref is None:
            return
        self._capture_reference = None
        num = float((self.flat_grads.flat - ref).abs().sum())
        den = float(ref.abs().sum()) + 1e-30
        ok = num / den < 1e-3
        world_ok = ok
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # the verdict is collective: a rank that raised alone would leave its peers waiting in their next all-reduce
            flag = torch.tensor([1.0 if ok else 0.0], device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            world_ok = bool(float(flag) > 0.5)
        if not world_ok:
            raise RuntimeError("step graph with captured collectives disagrees with the eager data-parallel step on its "
                               "first batch (relative L1 difference of the exchanged gradients on this rank %.3e%s): "
                               "refusing to train on it; use dp_capture=False (split graphs around one exposed all-reduce)"
                               % (num / den, "" if not ok else "; another rank failed the check"))

    def prewarm(self, epoch=None, seed=0):
        """Before the first step of a `--rand` curriculum phase: capture the step graph of EVERY pose-row bucket the phase
        can meet (the pooled form's graph keys: a few dozen for epochs >= 10, seven for the early curriculum) on synthetic batches,
        so that no training step pays for a capture, for the allocator meeting a new activation size or for MIOpen loading
        a row count's solvers (round 5: a fresh process ran its first 30 steps at half speed).  The warm-up steps inside a
        capture restore parameters, buffers, optimizer state and the step counter: nothing is trained.  Without step
        graphs the buckets get one restored eager step each.  Returns {"buckets": n, "seconds": t}."""
        import time
        if not self.pooled_step:
            return {"buckets": 0, "seconds": 0.0}
        from . import pooled, synthetic
        t0 = time.perf_counter()
        opt = self.opt
        epoch = self.epoch if epoch is None else epoch
        early = bool(getattr(opt, "rand", True)) and epoch < 10
        scales = list(opt.scales)
        cutt = 0.1 + 0.04 * epoch if epoch < 10 else 0.15 * epoch - 0.9
        if self._pooled is None:
            self._pooled = pooled.PooledStep(self)
        keep_frames = list(opt.frame_ids)
        done = 0
        for ms in self._pooled.bucket_orderings(early):
            batch = synthetic.synthetic_batch(ms, opt.height, opt.width, scales, device=self.device, seed=seed + done)
            batch.pop("noise")
            batch["cutt"] = torch.tensor(cutt)
            opt.frame_ids = sorted(batch["frames"], key=_frame_sort_key)
            if self.use_graph:
                tab, key, has_noise = self._graph_signature(batch)
                if tab is None or key in self._graphs:
                    continue
                first = bool(self.grad_sync is not None and self.dp_capture and not self._capture_checked)
                snap = self._snapshot_state() if first else None
                entry = self._capture(key, batch, tab, has_noise, first)
                if first:
                    # captured collectives: the one-shot check against the eager exchange needs the graph's first replay
                    self._pooled.load(batch, tab, scales)
                    entry[0].replay()
                    self._verify_captured_exchange()
                    self._restore_state(snap)
            else:
                snap = self._snapshot_state()
                self._local_only = True
                if hasattr(self.grad_sync, "paused"):
                    self.grad_sync.paused = True
                try:
                    self._eager_step(batch)
                finally:
                    self._local_only = False
                    if hasattr(self.grad_sync, "paused"):
                        self.grad_sync.paused = False
                self._restore_state(snap)
            done += 1
        opt.frame_ids = keep_frames
        torch.cuda.synchronize(self.device)
        return {"buckets": done, "seconds": round(time.perf_counter() - t0, 3)}

    def _eager_step_off_default_stream(self, inputs):
        """An eager step of a trainer that ALSO captures step graphs runs on a side stream of its own, never on the
        default stream: autograd binds every parameter's AccumulateGrad node to the stream of the step that first used it
        and keeps it while any loss / output of that step is alive - a later capture would then find gradient accumulation
        queued on the legacy default stream in the middle of a capturing stream (hipGraph capture dies in capture_end)."""
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, "_eager_stream", None) is None:
            self._eager_stream = torch.cuda.Stream(device=self.device)
        side = self._eager_stream
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            result = self._eager_step(inputs)
        cur.wait_stream(side)
        return result

    def train_step(self, inputs):
        """One optimisation step on a collated batch, as the body of `run_epoch` does it.  The samples are taken in
        canonical order (`canonicalize`; `opt.canonical_order = False` keeps the caller's)."""
        if "frames" in inputs:
            self.opt.frame_ids = sorted(inputs["frames"], key=_frame_sort_key)
        if self.device.type == "cuda":
            cutt = inputs.get("cutt")
            if torch.is_tensor(cutt) and cutt.is_cuda:
                # the pose-mode threshold is part of the graph key: fetch it ONCE per batch dict (the reference's collate
                # keeps it on the host, trainer.py:867-886; a caller that moved it pays one sync here, not one per step)
                inputs["cutt"] = cutt.detach().cpu()
            for key, ipt in inputs.items():
                if key not in ["frames", "ordering", "cutt"] and torch.is_tensor(ipt):
                    inputs[key] = ipt.to(self.device, non_blocking=True)
        if getattr(self.opt, "canonical_order", True):
            self.canonicalize(inputs)
        if self.use_graph and self.device.type == "cuda":
            return self._graph_step(inputs)
        return self._eager_step(inputs)

    def canonicalize(self, inputs):
        """Reorders the samples of a collated batch IN PLACE to `plan.canonical_permutation` (largest frame offset first)
        and returns the permutation (None: already canonical; `inputs["batch_order"][b]` = the caller's position of sample
        b).  Every tensor is gathered by its own rows: B-row tensors by the permutation, a frame's stack
        `("color" | "color_aug", f, s)` by where its owners (custom_collate, trainer.py:882) went.  The device loader and
        `synthetic_loader` collate in this order already, so this costs nothing there."""
        ordering = inputs["ordering"]
        ms = sample_max_offsets(ordering)
        perm = canonical_permutation(ms)
        B = len(ms)
        if perm == list(range(B)):
            return None
        new_ms = [ms[p] for p in perm]
        lists, todo = {}, []
        for key, t in inputs.items():
            if not torch.is_tensor(t) or t.dim() == 0 or key in ("cutt", "to_use"):
                continue
            if isinstance(key, tuple) and len(key) >= 2 and key[0] in ("color", "color_aug") and key[1] != 0:
                old_own, new_own = owners_of(ms, key[1]), owners_of(new_ms, key[1])
                if t.shape[0] != len(old_own):
                    # permuting everything else would pair this stack's frames with the wrong samples
                    raise ValueError("canonicalize: %r has %d rows, the batch's ordering gives frame %r %d owners"
                                     % (key, t.shape[0], key[1], len(old_own)))
                rows = tuple(old_own.index(perm[b]) for b in new_own)
            elif t.shape[0] == B:
                rows = tuple(perm)
            else:
                continue
            if list(rows) != list(range(len(rows))):
                lists.setdefault((rows, str(t.device)), None)
                todo.append((key, rows))
        # the row lists of one device travel as one pinned, asynchronous upload
        by_dev = {}
        for rows, dev in lists:
            by_dev.setdefault(dev, []).append(rows)
        for dev, rows_list in by_dev.items():
            pk = steptables.Packer()
            for rows in rows_list:
                pk.add(rows, list(rows), (len(rows),))
            views = pk.upload(dev)
            for rows in rows_list:
                lists[(rows, dev)] = views[rows]
        for key, rows in todo:
            inputs[key] = inputs[key].index_select(0, lists[(rows, str(inputs[key].device))])
        inputs["ordering"] = [ordering[p] for p in perm]
        inputs["batch_order"] = [p if "batch_order" not in inputs else inputs["batch_order"][p] for p in perm]
        return perm

    def _eager_step(self, inputs):
        outputs, losses = self.process_batch(inputs)
        if self.flat_grads is not None:
            self.flat_grads.zero()
        else:
            self.model_optimizer.zero_grad(set_to_none=True)
        losses["loss"].backward()
        if self.grad_sync is not None and not self._local_only:
            self.grad_sync()
        elif self.flat_grads is not None:
            self.flat_grads.pack()            # capture warm-up: same memory traffic, no exchange
        self.model_optimizer.step()
        self.step += 1
        return outputs, losses

    def run_epoch(self, loader):
        """One epoch over an iterable of collated batches, with the reference's per-epoch curriculum
        (trainer.py:196-232): LR schedule step, and - under --rand - all four scales before epoch 10,
        scale 0 only afterwards.  The loader itself (frame-set selection by baseline, mono_dataset.py)
        is the caller's: `synthetic.synthetic_loader` here, a KITTI loader in the reference."""
        self.model_lr_scheduler.step()
        self.set_train()
        if getattr(self.opt, "rand", False):
            self.opt.scales = [0, 1, 2, 3] if self.epoch < 10 else [0]
            if self.pooled_step and getattr(self.opt, "prewarm", True):
                self.last_prewarm = self.prewarm(self.epoch)      # (graphs already captured for this phase / lr are skipped)
        last = None
        log_frequency = getattr(self.opt, "log_frequency", 0)
        for self.batch_idx, inputs in enumerate(loader):
            last = self.train_step(inputs)
            # the reference validates every `log_frequency` steps (trainer.py:266-283)
            if log_frequency and self.batch_idx > 0 and self.batch_idx % log_frequency == 0:
                val_loader = self.kitti_val_loader()
                if val_loader is not None:
                    self.last_val = self.val(val_loader)
        return last

    def kitti_loader(self, epoch):
        """The reference's per-epoch training loader (trainer.py:206-220) in index-table form: a
        `KITTIRAWDataset` rebuilt every epoch (curriculum), decoded on host threads, collated on the GPU."""
        from . import datasets
        opt = self.opt
        if not hasattr(self, "train_filenames"):
            self.train_filenames = datasets.readlines(os.path.join(
                getattr(opt, "splits_dir", "splits"), "eigen_zhou", "{}.txt".format(opt.training_file)))
        if getattr(opt, "rand", False):
            opt.scales = [0, 1, 2, 3] if epoch < 10 else [0]
        ds = datasets.KITTIRAWDataset(self.train_filenames, epoch, opt.height, opt.width, kt_path=opt.kt_path,
                                      rand=getattr(opt, "rand", False), is_train=True, scales=opt.scales, kt=True,
                                      naive_mix=True, trimin=opt.trimin,
                                      seed=getattr(opt, "pytorch_random_seed", 0) + 7919 * self._rank_world()[0])
        collate = datasets.DeviceCollate(opt.height, opt.width, opt.scales, self.device, self.backend, cache=self.frame_cache())
        # data parallel: every rank shuffles with the SAME seed and takes every world-th index of that
        # order (disjoint shards of one epoch, equal length); augmentation draws are per-rank
        rank, world = self._rank_world()
        return datasets.DeviceLoader(ds, opt.batch_size, collate, shuffle=True, drop_last=True,
                                     num_workers=getattr(opt, "num_workers", 8),
                                     seed=getattr(opt, "pytorch_random_seed", 0),
                                     workers=getattr(opt, "loader_workers", "process"),
                                     rank=rank, world=world)

    def frame_cache(self):
        """Decoded frames resident in HBM for the life of the trainer (`datasets.FrameCache`): a KITTI frame is decoded once,
        not once per use per epoch.  `opt.frame_cache_gb` (default 64: the Eigen-Zhou split's ~45 000 distinct frames are
        63 GB decoded; capped at half of the free device memory; 0 = off, the reference's behaviour)."""
        if getattr(self, "_frame_cache", None) is None:
            gb = float(getattr(self.opt, "frame_cache_gb", 64.0))
            if self.device.type != "cuda" or gb <= 0:
                return None
            from . import datasets
            free, _ = torch.cuda.mem_get_info(self.device)
            # scratch area: one boosted batch with nothing resident yet (resume at a late epoch, or a cache smaller than the
            # data set) carries up to 16 full-resolution KITTI frames per sample (1242 x 375 x 3 bytes each)
            scratch = max(192 << 20, int(self.opt.batch_size * 17 * 1242 * 375 * 3 * 1.1))
            self._frame_cache = datasets.FrameCache(self.device, int(min(gb * (1 << 30), free // 2)), scratch_bytes=scratch)
        return self._frame_cache

    def kitti_val_loader(self):
        """Validation split of trainer.py:127-131 + ground truth of :150-151, built once: `val_files.txt`
        through the device loader (batch 16 instead of 1 - metrics are per image either way) and
        `gt_depths.npz` packed into HBM.  None when the split files are not there (synthetic runs)."""
        if getattr(self, "_val_loader", None) is not None or getattr(self, "_val_missing", False):
            return getattr(self, "_val_loader", None)
        from . import datasets
        opt = self.opt
        split = os.path.join(getattr(opt, "splits_dir", "splits"), "eigen_zhou")
        files, gt = os.path.join(split, "val_files.txt"), os.path.join(split, "gt_depths.npz")
        if not (getattr(opt, "kt_path", None) and os.path.isfile(files) and os.path.isfile(gt)):
            self._val_missing = True
            return None
        import numpy as np
        self.set_ground_truth(np.load(gt, fix_imports=True, encoding="latin1", allow_pickle=True)["data"])
        ds = datasets.KITTIRAWDataset(datasets.readlines(files), 0, opt.height, opt.width, kt_path=opt.kt_path,
                                      is_train=False, kt=True, naive_mix=True)
        # no FrameCache here: validation runs in the middle of an epoch (log_frequency) while the training loader's producer
        # thread is collating on its own stream, and its 4 424 frames are read once per pass anyway (the reference decodes them
        # every time too); the cache's scratch areas are per collate and its index is locked, should a caller share one
        self._val_loader = datasets.DeviceLoader(ds, 16, datasets.DeviceCollate(opt.height, opt.width, [0], self.device,
                                                                               self.backend, cache=None),
                                                 shuffle=False, drop_last=False, num_workers=getattr(opt, "num_workers", 8))
        return self._val_loader

    def _rank_world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def resume_epoch(self):
        """Start epoch implied by `--load_weights_folder` (trainer.py:169-183): `weights_<N>` resumes at
        epoch N+1, `weights_best` (or any non-numeric suffix) at epoch 10; `None` starts at 0."""
        folder = getattr(self.opt, "load_weights_folder", "None")
        if folder in (None, "None"):
            return 0
        suffix = os.path.basename(os.path.normpath(folder)).split("_")[-1]
        try:
            return int(suffix) + 1
        except ValueError:
            return 10

    def train(self, loader_factory=None, num_epochs=None, steps_per_epoch=None):
        """`loader_factory(epoch)` -> iterable of batches (default: `kitti_loader`).  As the reference
        (trainer.py:168-193): resume at the epoch the loaded weights folder names, with the LR schedule and
        the step counter fast-forwarded; a checkpoint every `save_frequency` epochs, unconditionally
        (written by rank 0 only when several ranks train)."""
        loader_factory = loader_factory or self.kitti_loader
        rank, world = self._rank_world()
        start = self.resume_epoch()
        if start:
            if steps_per_epoch is None:
                steps_per_epoch = (len(getattr(self, "train_filenames", [])) // self.opt.batch_size) if hasattr(
                    self, "train_filenames") else 0
            self.step = start * steps_per_epoch
            for _ in range(start):
                self.model_lr_scheduler.step()
        self.epoch = start
        if rank == 0:
            self.save_opts()
        for self.epoch in range(start, num_epochs or self.opt.num_epochs):
            self.run_epoch(loader_factory(self.epoch))
            if (self.epoch + 1) % getattr(self.opt, "save_frequency", 1) == 0:
                if rank == 0:
                    self.save_model()
                if world > 1:
                    import torch.distributed as dist
                    dist.barrier()

    def process_batch(self, inputs, batch_idx=None, is_train=True):
        for key, ipt in inputs.items():
            if key not in ["frames", "ordering", "cutt"] and torch.is_tensor(ipt):
                inputs[key] = ipt.to(self.device, non_blocking=True)
        if is_train:
            self.valid_frames = list(set([el for sub in inputs["ordering"] for el in sub if el != 0]))
            self.valid_frames_trimin(inputs)
            tab = self._pooled_tables(inputs)
            if tab is not None:
                # pooled form (`--rand`): the batch goes into the static frame pool / table buffer (ONE table upload), the
                # step runs on them with shapes that depend on the padded pose rows only
                ps = self._pooled
                ps.load(inputs, tab, list(self.opt.scales))
                outputs, losses = ps.forward(tab.R, tab.G, tab.bound, inputs.get("noise") is not None)
                return ps.with_pose_views(outputs, tab), losses
            self._step_tables(inputs)          # the step's ONE table upload goes out before its first launch
            side = self._pose_stream()
            if side is None:
                outputs = self.predict_poses(inputs)
            else:
                # the pose network and the depth network are independent until the warp: run the pose
                # passes on a second HIP stream so their small-grid layers overlap the depth network's
                # (autograd replays each backward node on its forward stream, so the backward overlaps too)
                main = self._main_stream = torch.cuda.current_stream(self.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    outputs = self.predict_poses(inputs)
            feats = self.models["encoder"](inputs["color_aug", 0, 0])
            outputs.update(self.models["depth"](feats))
            if side is not None:
                main.wait_stream(side)
                if not torch.cuda.is_current_stream_capturing():
                    for v in outputs.values():          # produced on `side`, consumed on `main` from here on
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(main)
            outputs.update(self.generate_images_pred(inputs, outputs))
            losses = self.compute_losses(inputs, outputs)
        else:
            outputs = {}
            feats = self.models["encoder"](inputs["color", 0, 0])
            outputs.update(self.models["depth"](feats))
            _, outputs["depth", 0, 0] = disp_to_depth(outputs["disp", 0], self.opt.min_depth, self.opt.max_depth)
            losses = None
        return outputs, losses

    # ------------------------------------------------------------------ collate (trainer.py:867-886)
    def custom_collate(self, batch):
        """Per-item dicts from the loader (mono_dataset.py:76-146) -> the batch dict `process_batch` takes.
        Tensors of a key are stacked over the items that HAVE that key, so `("color", f, 0)` has one
        row per sample whose own frame set reaches |f| (variable n_f)."""
        out = {}
        max_frames = [int(torch.max(item["frames"]).item()) for item in batch]
        out["ordering"] = [[0, STEREO] if m == 0 else [0, m, -m] for m in max_frames]
        top = max(max_frames)
        if top == 0:
            frame_ids = [0, STEREO]
        else:
            frame_ids = list(range(-top, top + 1))
            if any(m in (0, 1, 2) for m in max_frames):
                frame_ids.append(STEREO)
        keys = [("color_aug", f, 0) for f in frame_ids if f != STEREO]
        keys += [("color", f, s) if f == 0 else ("color", f, 0) for f in frame_ids for s in self.opt.scales]
        keys += [("K", 0), ("inv_K", 0), "stereo_T"]
        for key in keys:
            out[key] = torch.stack([item[key] for item in batch if key in item], dim=0)
        out["frames"] = frame_ids
        out["cutt"] = batch[0]["cutt_off"]
        out["to_use"] = batch[0]["to_use"]
        return out

    # ------------------------------------------------------------------ index table (a11)
    def valid_frames_trimin(self, inputs):
        """Builds the candidate/index table for this batch's `ordering` (replaces the mask dicts of
        trainer.py:888-981) and extends `valid_frames` like :961-981."""
        self.plan = get_plan(inputs["ordering"], self.opt.trimin, self.opt.decomp)
        self.valid_frames = list(self.plan.valid_frames)
        self.tables = None
        return self.plan

    def _step_tables(self, inputs):
        """Every integer table this step needs (plan, work order, pose schedule + its row lists, composition table,
        invert flags) resident on the device after ONE asynchronous upload per batch signature (`steptables`).  The pose
        mode follows the reference's threshold on the batch's `cutt` (trainer.py:312), which the collate keeps on the host."""
        opt = self.opt
        cutt = inputs["cutt"]
        if torch.is_tensor(cutt) and cutt.is_cuda:      # a caller that moved it pays one sync here, once per batch dict
            cutt = inputs["cutt"] = cutt.detach().cpu()
        self.maxing_valid_frames = float(cutt) > 0.5
        incremental = bool(opt.incremental_skip and self.maxing_valid_frames)
        partial = bool(opt.partial_skip and self.maxing_valid_frames)
        lib = None
        if self.device.type == "cuda":
            lib = getattr(self._backend(), "lib", None)
        self.tables = steptables.get_step_tables(self.plan, opt.frame_ids, incremental, partial, bool(opt.decomp),
                                                 len(opt.scales), opt.height, opt.width, self.device, lib, self._pose_chunk())
        return self.tables

    def _pooled_tables(self, inputs):
        """`pooled.PooledTables` of this batch when the step can run in pooled form, else None (the per-signature path):
        needs the fused disparity-mode launches, the batched pose pass, the composition kernel and a batch of the trainer's
        batch size."""
        if not self.pooled_step:
            return None
        opt, plan = self.opt, self.plan
        if (getattr(opt, "materialize_warps", False) or not getattr(opt, "fused_disp", True) or len(opt.scales) > 4
                or not ops.FUSED_POSE_COMPOSE or not self._batched_pose_pairs() or plan.B != opt.batch_size):
            return None
        cutt = inputs["cutt"]
        if torch.is_tensor(cutt) and cutt.is_cuda:
            cutt = inputs["cutt"] = cutt.detach().cpu()
        self.maxing_valid_frames = float(cutt) > 0.5
        if self._pooled is None:
            from . import pooled
            self._pooled = pooled.PooledStep(self)
        self.tables = None
        self.last_pooled = self._pooled.tables_for(plan, inputs)
        return self.last_pooled

    def _rows(self, tensor, rows):
        if rows is None or (len(rows) == tensor.shape[0] and list(rows) == list(range(tensor.shape[0]))):
            return tensor
        return tensor.index_select(0, self._index(rows, tensor.device))

    def _index(self, rows, device, dtype=torch.int32):
        """Row-index tensor of a sub-batch selection.  Inside a step these are views of the step's ONE table upload
        (`steptables.StepTables.index`); a list the step did not announce (callers outside process_batch, the CPU
        loops) takes a single pinned, asynchronous upload kept in a bounded LRU."""
        tables = getattr(self, "tables", None)
        if tables is not None and dtype == torch.int32 and tables.device == torch.device(device):
            hit = tables.index(rows)
            if hit is not None:
                return hit
        key = (tuple(rows), str(device), dtype)
        cache = self.__dict__.setdefault("_index_cache", steptables.LRU(1024))
        hit = cache.get(key)
        if hit is None:
            hit = cache.put(key, steptables.upload_single(list(rows), device, dtype))
        return hit

    # ------------------------------------------------------------------ poses (trainer.py:310-419)
    def _pose_pair(self, first, second, invert):
        feats = [self.models["pose_encoder"](torch.cat([first, second], 1))]
        axisangle, translation = self.models["pose"](feats)
        return transformation_from_parameters(axisangle[:, 0], translation[:, 0], invert=invert)

    def _batched_pose_pairs(self):
        """One batched pass for all pose-network calls of a step?  The reference calls the pose network once per
        frame pair (trainer.py:348-418: 2 calls for MD2, up to 26 on <= 12 samples each for the boosted recipe); no
        call depends on another's output.  On the GPU the calls run as ONE pass over the concatenated pairs, with
        every fused BatchNorm keeping separate batch statistics per call group (`ops.bn_call_groups`,
        `bbd_bn_act_grouped_*`) - the numbers of the separate calls, fewer and larger launches."""
        if getattr(self.opt, "batched_pose", True) is False or os.environ.get("BBD_BATCHED_POSE", "1") == "0":
            return False
        if self.device.type != "cuda":
            return False
        from .networks.encoder import FusedBatchNorm2d
        enc = self.models["pose_encoder"]
        return (not enc.training) or FusedBatchNorm2d.fused

    def _pose_pairs(self, requests):
        """requests: [(first, second, invert)] in the reference's call order -> [T] (4x4 per row of the pair)."""
        if len(requests) <= 1 or not self._batched_pose_pairs():
            return [self._pose_pair(a, b, inv) for a, b, inv in requests]
        out = []
        tables = getattr(self, "tables", None)
        chunk_size = self._pose_chunk()
        for c, lo in enumerate(range(0, len(requests), chunk_size)):
            chunk = requests[lo:lo + chunk_size]
            rows = [a.shape[0] for a, _, _ in chunk]
            n_real = sum(rows)
            parts = [torch.cat([a, b], 1) for a, b, _ in chunk]
            # boosted batches change the pass's row count almost every step, and every new row count is a new problem for
            # every convolution of the pose network (MIOpen compiles solvers for tens of seconds at first sight): round the
            # pass up to the next row count the shipped MIOpen database has find results for (`tuning.POSE_ROW_COUNTS`; beyond
            # them to a multiple of `pose_pad_rows`) with one trailing call group of zero rows.  Its BatchNorm statistics
            # are its own and leave the running statistics alone (`untracked_groups`), nobody reads its outputs and its
            # gradient contributions are exact zeros - the real calls compute what they compute without it
            pad = 0
            if self.pose_pad_rows > 0 and self.models["pose_encoder"].training:
                from . import tuning
                pad = tuning.padded_pose_rows(n_real, self.pose_pad_rows) - n_real
            if pad:
                parts.append(parts[0].new_zeros((pad,) + tuple(parts[0].shape[1:])))
            if self.pose_pad_rows > 0 and self.models["pose_encoder"].training:
                from . import tuning
                tuning.note_pose_rows(n_real + pad)        # (warns once per row count MIOpen has no find results for)
            x = torch.cat(parts, 0)
            with ops.bn_call_groups(rows + ([pad] if pad else []), padding_groups=1 if pad else 0):
                feats = [self.models["pose_encoder"](x)]
            axisangle, translation = self.models["pose"](feats)
            if pad:
                axisangle, translation = axisangle[:n_real], translation[:n_real]
            # the chunk's pose matrices in ONE launch each way: the `invert` flag (negative frame ids, trainer.py:360,384,402)
            # goes row by row as a small device table - part of the step's one table upload
            flags = [int(bool(inv)) for n, (_, _, inv) in zip(rows, chunk) for _ in range(n)]
            if tables is not None and c < len(tables.invert) and tables.schedule.chunks[c][3] == flags:
                invert_rows = tables.invert[c]
            else:
                invert_rows = self._index(flags, self.device)
            M = ops.pose_matrix(axisangle[:, 0], translation[:, 0], backend=self._backend(), invert_rows=invert_rows)
            out.extend(torch.split(M, rows, dim=0))
        return out

    def _pose_chunk(self):
        """Calls per batched pose pass: the grouped BatchNorm takes BBD_BN_MAX_GROUPS groups, one of which is the padding's."""
        return ops.BN_MAX_GROUPS - (1 if self.pose_pad_rows > 0 else 0)

    def _error_pose(self, T):
        Te = T.clone().detach()                       # no pose gradient through the error-induced warp
        # tensor / tensor: a Python-scalar divisor would be turned into a multiply by 1/pose_error on
        # the GPU, which rounds differently from the reference's CPU division (trainer.py:377)
        Te[:, :3, 3:] = Te[:, :3, 3:] / torch.full((), float(self.opt.pose_error), device=Te.device)
        return Te

    def predict_poses(self, inputs):
        plan, opt = self.plan, self.opt
        outputs = {}
        tables = self._step_tables(inputs)
        sched = tables.schedule
        self.valid_frames_pose = list(sched.valid_frames_pose)
        temporal, incremental, partial = sched.temporal, sched.incremental, sched.partial
        # every pose-network call of the step, in the reference's call order (trainer.py:348-418: one call per adjacent
        # pair in incremental mode, one per warp job otherwise, plus the direct 0->f calls of --partial_skip); none
        # depends on another's result, so the host-side schedule lists them first (`steptables.PoseSchedule`) and they
        # run as one batched pass (`_pose_pairs`)
        slot = sched.slot
        requests = [(self._rows(inputs["color_aug", a[0], 0], a[1]), self._rows(inputs["color_aug", b[0], 0], b[1]), inv)
                    for _, a, b, inv, _ in sched.requests]
        Ts = self._pose_pairs(requests)

        if ops.FUSED_POSE_COMPOSE and Ts and Ts[0].is_cuda and tables.compose_table is not None:
            return self._compose_poses(Ts, tables)

        if incremental:
            for f in temporal:
                step = Ts[slot[("step", f)]]
                if abs(f) == 1:
                    outputs[("cam_T_cam", 0, f)] = step
                    outputs[("cam_T_cam_step", 0, f)] = step.clone()
                else:
                    nb = f + 1 if f < 0 else f - 1
                    own_f = plan.owners(f)
                    outputs[("cam_T_cam_step", nb, f)] = step
                    if f not in self.valid_frames_pose:
                        continue
                    T = torch.eye(4, device=self.device).unsqueeze(0).expand(step.shape[0], -1, -1)
                    # the reference chains with range(f, 0, -1), which is EMPTY for negative f:
                    # negative offsets beyond -1 keep the identity pose (reference behaviour, kept)
                    for k in range(f, 0, -1):
                        S_k = outputs[("cam_T_cam_step", k - 1, k)]
                        S_k = self._rows(S_k, [plan.owners(k).index(b) for b in own_f])
                        T = torch.matmul(T, S_k)
                    outputs[("cam_T_cam", 0, f)] = T
                if opt.decomp:
                    outputs[("cam_T_cam_error", 0, f)] = self._error_pose(outputs[("cam_T_cam", 0, f)])
        else:
            for f in self.valid_frames:
                if f == STEREO:
                    continue
                T = Ts[slot[("job", f)]]
                outputs[("cam_T_cam", 0, f)] = T
                if opt.decomp:
                    outputs[("cam_T_cam_error", 0, f)] = self._error_pose(T)

        if partial:
            nonstereo = [m for m in plan.ms if m != 0]
            for f in self.valid_frames:
                if f == STEREO or abs(f) <= 1:
                    continue
                direct = Ts[slot[("direct", f)]]
                chained = outputs[("cam_T_cam", 0, f)]
                replaced = torch.cat([chained[:, :, :3], direct[:, :, 3:]], dim=2)
                # the reference indexes its all-sample list by ROW number of the n_f-row tensor
                keep = self._index([abs(f) == nonstereo[r] - 2 for r in range(chained.shape[0])], self.device,
                                   torch.bool).view(-1, 1, 1)
                outputs[("cam_T_cam", 0, f)] = torch.where(keep, chained, replaced)
        return outputs

    def _compose_poses(self, Ts, tables):
        """GPU form of the loops above: the incremental chain, the error-induced poses and the partial
        swap of the whole step as ONE launch each way (`ops.pose_compose`).  The integer table is a function of the
        batch signature only (`steptables.PoseSchedule._compose_rows`) and reached the device with the step's one table
        upload; the composed matrices are views of one [NO,4,4] buffer."""
        table = tables.compose_table
        outputs = {}
        for okey, i in tables.passthrough:
            outputs[okey] = Ts[i]
        if table.NO:
            out = ops.pose_compose(torch.cat(list(Ts), 0), table, float(self.opt.pose_error), self._backend())
            for okey, o0, n, const in tables.compose_views:
                view = out[o0:o0 + n]
                # the reference's T_error is a detached clone (trainer.py:376): the kernel's backward skips those rows
                outputs[okey] = view.detach() if const else view
        return outputs

    # ------------------------------------------------------------------ warp + loss (trainer.py:444-570)
    def _job_poses(self, inputs, outputs):
        plan = self.plan
        per_source_rows = bool(self.opt.incremental_skip and self.maxing_valid_frames)
        poses = {}
        for kind, f in plan.pose_jobs:
            if f == STEREO:
                poses[(kind, f)] = self._rows(inputs["stereo_T"], plan.jobs[f])
                continue
            T = outputs[("cam_T_cam" if kind == "T" else "cam_T_cam_error", 0, f)]
            if per_source_rows:                        # poses have n_f rows: select the job's (trainer.py:468)
                T = self._rows(T, plan.job_rows_in_source(f))
            poses[(kind, f)] = T
        return poses

    def generate_images_pred(self, inputs, outputs):
        """Depth per scale, pose table, identity pre-pass and the fused warp+SSIM+min launch.
        Fills `("depth",0,s)`; `("color"/"color_D",f,s)` only with `opt.materialize_warps`."""
        opt, plan, be = self.opt, self.plan, self._backend()
        H, W = opt.height, opt.width
        scales = list(opt.scales)
        target = inputs[("color", 0, 0)]
        new = {}
        proj = ops.pose_table(plan, inputs[("K", 0)], inputs[("inv_K", 0)], self._job_poses(inputs, outputs))
        frame_tensors = {f: inputs[("color", f, 0)] for f in plan.frames}
        noise = inputs.get("noise")
        if noise is None:   # trainer.py:518-523 draws randn*1e-5 per positive group; one draw covers the batch
            noise = torch.randn(plan.B, H, W, device=target.device) * 0.00001
        ident = ops.identity_losses(plan, frame_tensors, target, opt.no_ssim, be)
        materialize = bool(getattr(opt, "materialize_warps", False))
        disps = [outputs[("disp", s)] for s in scales]
        if getattr(opt, "fused_disp", True) and len(scales) <= 4:
            # the kernels read the low-resolution disparities themselves (up-sampling + disp_to_depth
            # per staged pixel); outputs[("depth",0,s)] is a by-product of the forward launch
            loss_sum, min_loss, argmin, warped, depth = ops.fused_reprojection_min_disp(
                disps, proj, target, ident, noise, plan, frame_tensors, opt.min_depth, opt.max_depth, opt.no_ssim,
                materialize, bool(getattr(opt, "materialize_depth", True)), be)
        else:
            depth = ops.disp_pyramid_to_depth(disps, H, W, opt.min_depth, opt.max_depth, be)          # [S,B,H,W]
            loss_sum, min_loss, argmin, warped = ops.fused_reprojection_min(
                depth, proj, target, ident, noise, plan, frame_tensors, opt.no_ssim, materialize, be)
        if depth is not None:
            for i, s in enumerate(scales):
                new[("depth", 0, s)] = depth[i].unsqueeze(1)
        new[("bbd", "loss_sum")] = loss_sum
        new[("bbd", "to_optimise")] = min_loss
        new[("bbd", "argmin")] = argmin
        new[("bbd", "identity")] = ident
        if materialize:
            for kind, f in plan.pose_jobs:
                off, n = plan.pose_offset[(kind, f)], len(plan.jobs[f])
                for i, s in enumerate(scales):
                    new[("color" if kind == "T" else "color_D", f, s)] = warped[i, off:off + n]
        return new

    def compute_reprojection_loss(self, pred, target):
        """0.85*SSIM + 0.15*L1 map, [n,3,H,W] x2 -> [n,1,H,W]  (trainer.py:477-486); differentiable (the
        training step itself uses the fused launch, this is the reference's stand-alone method)."""
        l1 = torch.abs(target - pred).mean(1, True)
        if self.opt.no_ssim:
            return l1
        return 0.85 * self.ssim(pred, target, backend=self._backend()).mean(1, True) + 0.15 * l1

    def compute_losses(self, inputs, outputs):
        opt = self.opt
        if ("bbd", "loss_sum") not in outputs:
            raise RuntimeError("compute_losses needs the outputs of generate_images_pred (fused launch)")
        losses = {}
        n_px = self.plan.B * opt.height * opt.width
        # the edge-aware smoothness of every scale (:560-563, layers.py:203-216) in one launch pair each way
        smooths = ops.normalised_smooth_losses([outputs[("disp", s)] for s in opt.scales],
                                               [inputs[("color", 0, s)] for s in opt.scales], self._backend())
        # loss/s = to_optimise.mean() + disparity_smoothness * smooth / 2**s (:557, :563-564); loss = sum / num_scales (:568,
        # the frozen 4) - one small node on [S]-vectors instead of ~70 one-element launches per step
        loss_sum = outputs[("bbd", "loss_sum")]
        total, per = ops.combine_losses(loss_sum, smooths, n_px, opt.disparity_smoothness, list(opt.scales), self.num_scales)
        for i, s in enumerate(opt.scales):
            losses["loss/{}".format(s)] = per[i]
        losses["loss"] = total
        return losses

    # ------------------------------------------------------------------ validation metrics (trainer.py:572-617)
    def set_ground_truth(self, gt_depths):
        """Packs a split's ground-truth depth maps (the reference's `self.gt_depths`, loaded from
        `splits/<split>/gt_depths.npz`, trainer.py:138-143) into device memory once."""
        from .evaluation import GroundTruthSet
        self.gt_depths = gt_depths if isinstance(gt_depths, GroundTruthSet) else GroundTruthSet(gt_depths, self.device)
        return self.gt_depths

    def compute_depth_losses(self, outputs, losses, idx, SYNS=False, accumulate=False):
        """KITTI depth metrics of `outputs[("depth",0,0)]` against ground-truth map(s) `idx` of
        `self.gt_depths` (trainer.py:572-617, KITTI branch), in ONE kernel launch on the device
        (`bbd_depth_metrics`): bilinear resize to the GT size, clamp to [1e-3, 80], Garg crop, median
        scaling, the seven metrics of `layers.compute_depth_errors`.  `idx` is an int (the reference's
        batch-1 validation loop) or one index per row of a batched prediction; metrics of a batch are
        summed, as `accumulate=True` does over the reference's loop.  Values stay on the device
        (0-dim tensors) so the validation loop never synchronises per image."""
        if SYNS:
            raise NotImplementedError("SYNS edge metrics (cv2/scipy host code, trainer.py:577-593) are out of scope")
        from .evaluation import depth_metrics, GroundTruthSet
        if not isinstance(self.gt_depths, GroundTruthSet):
            self.set_ground_truth(self.gt_depths)
        indices = [int(idx)] if not hasattr(idx, "__len__") else [int(i) for i in idx]
        rows = depth_metrics(outputs["depth", 0, 0], self.gt_depths, indices, backend=self._backend())
        sums = rows[:, :7].sum(0)
        for k, name in enumerate(self.depth_metric_names):
            losses[name] = losses[name] + sums[k] if accumulate and name in losses else sums[k]
        return losses

    def val(self, val_loader, is_init=None):
        """Validation pass (trainer.py:623-665): mean of the seven metrics over the loader's images;
        keeps `self.best` (abs_rel).  One host synchronisation, at the end."""
        self.set_eval()
        losses, images = {}, 0
        with torch.no_grad():
            for batch_idx, inputs in enumerate(val_loader):
                outputs, _ = self.process_batch(inputs, batch_idx, is_train=False)
                n = outputs["depth", 0, 0].shape[0]
                self.compute_depth_losses(outputs, losses, list(range(images, images + n)), accumulate=True)
                images += n
        result = {name: float(losses[name]) / max(images, 1) for name in self.depth_metric_names}
        if result["de/abs_rel"] < getattr(self, "best", float("inf")):
            self.best = result["de/abs_rel"]
        self.set_train()
        return result

    def argmin_masks(self, outputs, scale_index=0, reference_dicts=False):
        """The reference's `self.ident` bookkeeping (trainer.py:547; x_min_opt :1002-1003, :1021-1022, :1044-1045 and the
        non-decomp branch): per sample, where a true-pose reprojection ('norm') or an error-induced one ('guide') won the
        per-pixel minimum.  The fused launch's arg-min ids follow the reference's `torch.cat` order (true-pose warps,
        error-induced warps, identity maps), so 'norm' = id < n_T and 'guide' = n_T <= id < n_T + n_E.
        Returns (norm [B,H,W] bool, guide [B,H,W] bool) in batch order; `reference_dicts=True` gives the reference's own
        shape instead - (dictor_norm, dictor_guide), keys `(m | 's', 'norm' | 'guide')` for every sample group of the
        batch, each holding one [n_group,H,W] tensor (samples of the group in batch order; 'guide' only under --decomp
        and never for the stereo group, as in the reference)."""
        arg = outputs[("bbd", "argmin")][scale_index]
        norm, guide = [], []
        for b, names in enumerate(self.plan.cand_names):
            n_t = sum(1 for k, _ in names if k == "T")
            n_e = sum(1 for k, _ in names if k == "E")
            norm.append(arg[b] < n_t)
            guide.append((arg[b] >= n_t) & (arg[b] < n_t + n_e))
        norm, guide = torch.stack(norm), torch.stack(guide)
        if not reference_dicts:
            return norm, guide
        ms = self.plan.ms
        key = lambda m: STEREO if m == 0 else m
        dictor_norm = {(key(m), "norm"): [] for m in ms}
        dictor_guide = {(key(m), "guide"): [] for m in ms}
        # the reference visits the groups in `temp_positive` order; every group appends exactly one tensor to its own key
        for m in sorted(set(ms)):
            rows = [b for b, mb in enumerate(ms) if mb == m]
            dictor_norm[(key(m), "norm")].append(norm[rows])
            if self.plan.decomp and m != 0:
                dictor_guide[(key(m), "guide")].append(guide[rows])
        return dictor_norm, dictor_guide

    # ------------------------------------------------------------------ checkpoints (trainer.py:774-829)
    def save_opts(self):
        folder = os.path.join(self.log_path, "models")
        os.makedirs(folder, exist_ok=True)
        with open(os.path.join(folder, "opt.json"), "w") as f:
            json.dump({k: v for k, v in vars(self.opt).items()}, f, indent=2, default=str)

    def save_model(self, name=None):
        folder = os.path.join(self.log_path, "models", "weights_{}".format(self.epoch if name is None else name))
        os.makedirs(folder, exist_ok=True)
        for model_name, model in self.models.items():
            to_save = model.state_dict()
            if model_name == "encoder":               # consumers read the training resolution from here
                to_save["height"] = self.opt.height
                to_save["width"] = self.opt.width
            torch.save(to_save, os.path.join(folder, "{}.pth".format(model_name)))
        torch.save(self.model_optimizer.state_dict(), os.path.join(folder, "adam.pth"))
        return folder

    def load_model(self):
        folder = os.path.expanduser(self.opt.load_weights_folder)
        assert os.path.isdir(folder), "Cannot find folder {}".format(folder)
        for n in getattr(self.opt, "models_to_load", ["encoder", "depth", "pose_encoder", "pose"]):
            path = os.path.join(folder, "{}.pth".format(n))
            model_dict = self.models[n].state_dict()
            pretrained = torch.load(path, map_location=self.device)
            model_dict.update({k: v for k, v in pretrained.items() if k in model_dict})
            self.models[n].load_state_dict(model_dict)
        adam = os.path.join(folder, "adam.pth")
        if os.path.isfile(adam):
            try:
                self.model_optimizer.load_state_dict(torch.load(adam, map_location=self.device))
            except ValueError:
                pass    # parameter grouping differs: start Adam fresh, like the reference's fallback
