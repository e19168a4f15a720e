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
    """Re-homes every parameter's .grad into one flat fp32 buffer."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero(self):
        self.flat.zero_()

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
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def attach(trainer, group=None):
    """Give `trainer` flat gradients (always) and a cross-rank average (when world > 1)."""
    flat = FlatGradients(trainer.parameters_to_train)
    trainer.flat_grads = flat
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        # identical initial weights on every rank
        for p in trainer.parameters_to_train:
            dist.broadcast(p.data, src=0, group=group)
        for m in trainer.models.values():
            for b in m.buffers():
                if b.is_floating_point():
                    dist.broadcast(b.data, src=0, group=group)
        trainer.grad_sync = GradientAverager(flat, group)
    return flat
