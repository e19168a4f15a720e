"""Data-parallel gradient exchange: one process per GPU, ONE collective per step.

The reference is single-GPU (no torch.distributed anywhere).  The hot path shards by batch
sample with no data-path exchange: per-rank batches are independent and the only
communication is the fp32 gradient average.  MI355X-first layout: all gradients of the four
networks live in ONE contiguous HBM buffer (`p.grad` are views into it), so the step needs no
flatten/unflatten copies, `zero_grad` is one memset, and the exchange is a single RCCL
all-reduce over xGMI (107 MB; a ring moves 2*(N-1)/N of that per link, ~1.2 ms at 8 GPUs, which
is small against the conv backward - see DESIGN.md for why it is not bucketed further).
"""
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
        self.views, off = [], 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[off:off + n].view_as(p))
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
    """Callable placed between backward() and optimizer.step() (Trainer.grad_sync)."""

    def __init__(self, flat, group=None):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)

    def __call__(self):
        self.flat.pack()
        if self.world == 1:
            return
        if self.backend == "nccl":          # RCCL: average in the collective
            dist.all_reduce(self.flat.flat, op=dist.ReduceOp.AVG, group=self.group)
        else:                               # gloo (CPU tests) has no AVG
            dist.all_reduce(self.flat.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.flat.div_(self.world)


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
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def attach(trainer, group=None):
    """Multi-rank runs get a flat gradient buffer + one all-reduce per step; a single rank keeps
    PyTorch's own per-parameter gradients (nothing to exchange, nothing to pack)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return None
    flat = FlatGradients(trainer.parameters_to_train)
    trainer.flat_grads = flat
    if True:
        # identical initial weights on every rank
        for p in trainer.parameters_to_train:
            dist.broadcast(p.data, src=0, group=group)
        for m in trainer.models.values():
            for b in m.buffers():
                if b.is_floating_point():
                    dist.broadcast(b.data, src=0, group=group)
        trainer.grad_sync = GradientAverager(flat, group)
    return flat
