"""Data-parallel gradient exchange: one process per GPU, ONE collective per step.

The reference is single-GPU (no torch.distributed anywhere).  The hot path shards by batch
sample with no data-path exchange: per-rank batches are independent and the only
communication is the fp32 gradient average.  MI355X-first layout: all gradients of the four
networks live in ONE contiguous HBM buffer (`p.grad` are views into it), so the step needs no
flatten/unflatten copies, `zero_grad` is one memset, and the exchange is a single RCCL
all-reduce over xGMI (107 MB; a ring moves 2*(N-1)/N of that per link, ~1.2 ms at 8 GPUs, which
is small against the conv backward - see DESIGN.md for why it is not bucketed further).
"""
import datetime
import os

import torch
import torch.distributed as dist


class FlatGradients:
    """One contiguous fp32 buffer holding every parameter's gradient.

    Autograd is left to ALLOCATE each `.grad` itself (grads are reset to None every step): if
    `.grad` pre-exists, AccumulateGrad launches one `add` kernel per parameter (~190 launches,
    ~0.8 ms per step measured), whereas packing the fresh gradients into the flat buffer is one
    multi-tensor copy (111 MB, ~0.1 ms).  After `pack()` every `p.grad` is a view of the buffer, so the
    collective and the optimizer see the same memory with no unpack step."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views, self.offsets, off = [], [], 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[off:off + n].view_as(p))
            self.offsets.append(off)
            off += n

    def zero(self):
        """Start of a step: drop the views so that backward assigns fresh gradients."""
        for p in self.params:
            p.grad = None

    def pack(self):
        """After backward: gather the gradients into the flat buffer and re-point `.grad` at it.
        Parameters that received no gradient (the ResNet `fc` layers) contribute zeros."""
        have = [(v, p.grad) for v, p in zip(self.views, self.params) if p.grad is not None]
        missing = [v for v, p in zip(self.views, self.params) if p.grad is None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v in missing:
            v.zero_()
        for v, p in zip(self.views, self.params):
            p.grad = v

    @property
    def nbytes(self):
        return self.flat.numel() * 4


class GradientAverager:
    """Callable placed between backward() and optimizer.step() (Trainer.grad_sync): pack + one all-reduce."""

    def __init__(self, flat, group=None):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.use_avg = True
        # diagnostics for the first multi-GPU runs (bench.py prints them): which reduction ran, and how long the step
        # waited for the exchange - HIP events around the exposed part (the whole all-reduce in the split-graph loop, the
        # flush + waits of the overlapped loop), enabled by `timing = True`
        self.timing = False
        self._events = []

    def _reduce(self, tensor, async_op=False):
        if self.backend == "nccl" and self.use_avg:      # RCCL: average in the collective
            try:
                return dist.all_reduce(tensor, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
            except RuntimeError:                          # a build without ncclAvg: sum, divide afterwards
                self.use_avg = False                      # (a synchronous failure only; `reduce_op` reports what ran)
        return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    @property
    def reduce_op(self):
        return "AVG" if (self.backend == "nccl" and self.use_avg) else "SUM+div"

    def _mark(self):
        if self.timing and self.flat.flat.is_cuda and not torch.cuda.is_current_stream_capturing():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        return None

    def exchange_ms(self):
        """Mean exposed exchange time per step (ms) over the steps taken since `timing` was switched on."""
        if not self._events:
            return 0.0
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._events) / len(self._events)

    @property
    def divide_after(self):
        return not (self.backend == "nccl" and self.use_avg)

    def exchange(self):
        """Average the (already packed) flat buffer over the ranks.  Also the eager middle of the split-graph step
        (Trainer._graph_step_dp): forward+backward+pack and the optimizer are hipGraphs, this stays a plain call."""
        if self.world == 1:
            return
        t0 = self._mark()
        self._reduce(self.flat.flat)
        if self.divide_after:               # gloo (CPU tests) has no AVG
            self.flat.flat.div_(self.world)
        t1 = self._mark()
        if t0 is not None:
            self._events.append((t0, t1))

    def __call__(self):
        self.flat.pack()
        self.exchange()


class OverlappedGradientAverager(GradientAverager):
    """Same result, but the exchange overlaps the backward pass.

    The flat buffer is cut into buckets in REVERSE parameter order (the order gradients become
    ready).  A post-accumulate hook per parameter counts its bucket down; a complete bucket is packed
    (one multi-tensor copy) and its all-reduce is issued asynchronously - strictly in bucket order,
    so every rank issues the same collective sequence even when their autograd graphs differ (boosted
    batches have rank-specific pose-net call patterns).  `__call__` flushes what is left (parameters
    that never get a gradient, e.g. the ResNet `fc` layers, contribute zeros) and waits.
    xGMI is point-to-point, so a ring all-reduce is per-link bound: buckets are sized (32 MB) to be
    well past the latency regime while leaving 3-4 of them to pipeline against the conv backward."""

    def __init__(self, flat, group=None, bucket_bytes=32 << 20, never=()):
        """`never`: parameters that are trainable but can never receive a gradient (the ResNet encoders'
        unused `fc` layers, reference quirk SURVEY 5c-5).  Their hooks never fire, so they are not counted
        in a bucket's pending set - otherwise the first buckets (reverse parameter order puts the `fc`
        layers there) would never complete during backward and nothing would overlap."""
        super().__init__(flat, group)
        self.never = {id(p) for p in never}
        order = list(range(len(flat.params)))[::-1]
        self.buckets, cur, cur_bytes = [], [], 0
        for idx in order:
            cur.append(idx)
            cur_bytes += flat.params[idx].numel() * 4
            if cur_bytes >= bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self.bucket_of = {}
        self.slices = []
        for b, idxs in enumerate(self.buckets):
            lo = min(flat.offsets[i] for i in idxs)
            hi = max(flat.offsets[i] + flat.params[i].numel() for i in idxs)
            self.slices.append(flat.flat[lo:hi])          # reverse order keeps each bucket contiguous
            for i in idxs:
                self.bucket_of[i] = b
        self.streams = []          # HIP streams gradients may be produced on (set by attach())
        self.paused = False        # True during a graph capture's warm-up steps: hooks do nothing, no collective goes out
        self._reset()
        for i, p in enumerate(flat.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _reset(self):
        self.pending = [sum(1 for i in b if id(self.flat.params[i]) not in self.never) for b in self.buckets]
        self.launched = 0
        self.works = []

    def _make_hook(self, i):
        def hook(param):
            if self.paused:
                return
            b = self.bucket_of[i]
            self.pending[b] -= 1
            self._launch_ready()
        if id(self.flat.params[i]) in self.never:
            def unexpected(param):
                raise RuntimeError("parameter declared gradient-free received a gradient (bucket accounting)")
            return unexpected
        return hook

    def _pack_bucket(self, b):
        flat = self.flat
        have_v, have_g = [], []
        for i in self.buckets[b]:
            p, v = flat.params[i], flat.views[i]
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                have_v.append(v)
                have_g.append(p.grad)
        if have_v:
            torch._foreach_copy_(have_v, have_g)
        for i in self.buckets[b]:
            flat.params[i].grad = flat.views[i]

    def _join_streams(self):
        """The trainer runs the pose network on a second HIP stream, and autograd replays every backward
        node (and fires this hook) on its forward stream: a bucket may hold gradients produced on the
        other stream, so the stream that packs and hands the bucket to RCCL first waits for both."""
        streams = self.streams() if callable(self.streams) else self.streams
        if not streams:
            return
        cur = torch.cuda.current_stream(self.flat.flat.device)
        for st in streams:
            if st != cur:
                cur.wait_stream(st)

    def _launch_ready(self, force=False):
        # (a bucket holding only gradient-free parameters has nothing pending: it goes out, as zeros, with
        # the first hook that fires - still in bucket order on every rank)
        while self.launched < len(self.buckets) and (force or self.pending[self.launched] == 0):
            b = self.launched
            self._join_streams()
            self._pack_bucket(b)
            self.works.append(self._reduce(self.slices[b], async_op=True))
            self.launched += 1

    def __call__(self):
        self.launched_in_backward = self.launched      # diagnostics: buckets whose exchange overlapped backward
        t0 = self._mark()
        self._launch_ready(force=True)
        for w in self.works:
            w.wait()
        if self.divide_after:
            self.flat.flat.div_(self.world)
        t1 = self._mark()
        if t0 is not None:
            self._events.append((t0, t1))
        self._reset()


def init_from_env(backend=None):
    """torchrun-style environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) -> (rank, local, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # BBD_DIST_BACKEND=gloo lets a single-GPU box exercise the multi-rank code path (tests)
            backend = os.environ.get("BBD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        # One timeout covers the rendezvous and every collective (torch.distributed has no separate knob that survives the
        # launcher's own store).  It is RCCL's default, 600 s, not shorter: rank skew is legitimate - a rank-0-only
        # checkpoint before a barrier, a new-signature graph capture, MIOpen compiling a solver on one rank only, a slow
        # first loader epoch - and a 120 s bound (round 4) would abort a healthy job.  BBD_DIST_TIMEOUT_S
        # overrides; `bench.py --gpus N` additionally bounds the whole launch (--launch-timeout) from outside
        timeout = datetime.timedelta(seconds=float(os.environ.get("BBD_DIST_TIMEOUT_S", "600")))
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local),
                                    timeout=timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
    return rank, local, world


def attach(trainer, group=None):
    """Multi-rank runs get a flat gradient buffer + one all-reduce per step; a single rank keeps
    PyTorch's own per-parameter gradients (nothing to exchange, nothing to pack)."""
    forced = os.environ.get("BBD_DP_FORCE_ATTACH") == "1"     # test hook: a ONE-rank group still packs and all-reduces
    if not (dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or forced)):
        return None
    flat = FlatGradients(getattr(trainer, "optimizer_parameters", None) or trainer.parameters_to_train)
    trainer.flat_grads = flat
    # identical initial weights on every rank
    for p in flat.params:
        dist.broadcast(p.data, src=0, group=group)
    for m in trainer.models.values():
        for b in m.buffers():
            if b.is_floating_point():
                dist.broadcast(b.data, src=0, group=group)
    # a step replayed as hipGraphs cannot launch collectives from autograd hooks: the split-graph step packs inside
    # its forward+backward graph and exchanges the whole buffer between its two graphs
    # ... unless the collectives themselves are captured (`dp_capture`: RCCL only - gloo cannot be captured): then the
    # bucketed all-reduces launched from the autograd hooks become nodes of the ONE step graph and overlap backward
    # inside the replay
    capture = bool(getattr(trainer, "dp_capture", False)) and dist.get_backend(group) == "nccl"
    trainer.dp_capture = capture
    overlap = os.environ.get("BBD_NO_OVERLAP", "0") != "1" and (capture or not getattr(trainer, "use_graph", False))
    bucket = int(os.environ.get("BBD_BUCKET_BYTES", str(32 << 20)))
    never = trainer.gradient_free_parameters() if hasattr(trainer, "gradient_free_parameters") else ()
    trainer.grad_sync = (OverlappedGradientAverager(flat, group, bucket, never=never) if overlap
                         else GradientAverager(flat, group))
    if capture and overlap:
        # RCCL sets its channels / connections up lazily, per message size, by talking to its peers: let that happen
        # here, where every rank is, not inside one rank's graph capture (a cache miss is a per-rank event)
        for sl in trainer.grad_sync.slices:
            trainer.grad_sync._reduce(sl)
        torch.cuda.synchronize()
        flat.flat.zero_()
    side = trainer._pose_stream() if hasattr(trainer, "_pose_stream") else None
    if overlap and side is not None:
        # the stream process_batch ran on (the default stream in the eager loop, the capturing stream while a step
        # graph is being captured - a capturing stream must not wait on a stream outside the capture) + the pose stream
        default = torch.cuda.default_stream(trainer.device)
        trainer.grad_sync.streams = lambda: [getattr(trainer, "_main_stream", None) or default, side]
    return flat
