#!/usr/bin/env python3
"""Multi-rank correctness check that runs on ONE GPU (ranks share cuda:0, gloo backend):

    BBD_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29511 tools/ddp_check.py

Every rank runs the production step on its own synthetic batch (second HIP stream for the pose network,
bucketed overlapped all-reduce).  Rank 0 recomputes every rank's gradient alone on the same weights and
compares their mean with the exchanged flat buffer; then two full train_steps must leave all ranks with
bit-identical parameters.  Prints DDP_CHECK_OK on success."""
import copy
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from baseboostdepth_amd import Trainer, distributed as bdist  # noqa: E402
from baseboostdepth_amd.synthetic import synthetic_batch  # noqa: E402

H, W, B, STEPS = 96, 320, 4, 2


def make_trainer():
    opt = bench.make_options(B, 0, "md2")
    opt.height, opt.width = H, W
    torch.manual_seed(7)
    return Trainer(opt), opt


def main():
    rank, _, world = bdist.init_from_env()
    torch.cuda.set_device(0)
    tr, opt = make_trainer()
    tr.set_train()
    start = copy.deepcopy({k: m.state_dict() for k, m in tr.models.items()})
    bdist.attach(trainer=tr)
    batches = [[synthetic_batch([1] * B, H, W, opt.scales, device="cuda:0", seed=100 * s + r) for r in range(world)]
               for s in range(STEPS)]
    # one step of Trainer.train_step, stopped before optimizer.step: the exchanged gradient is the evidence
    _, losses = tr.process_batch(dict(batches[0][rank]))
    tr.flat_grads.zero()
    losses["loss"].backward()
    tr.grad_sync()
    torch.cuda.synchronize()
    if hasattr(tr.grad_sync, "buckets"):      # the exchange must have started DURING backward, not after it
        n_b, early = len(tr.grad_sync.buckets), tr.grad_sync.launched_in_backward
        print("rank %d: %d of %d buckets launched inside backward" % (rank, early, n_b))
        assert early >= n_b - 1, "gradient buckets did not overlap backward (%d of %d)" % (early, n_b)
    mine = tr.flat_grads.flat.clone()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if rank == 0:
        for g in gathered[1:]:
            assert torch.equal(g, gathered[0]), "ranks hold different averaged gradients"
        ref, _ = make_trainer()
        ref.set_train()
        total = None
        for r in range(world):
            for k, m in ref.models.items():
                m.load_state_dict(start[k])
            ref.model_optimizer.zero_grad(set_to_none=True)
            _, l = ref.process_batch(dict(batches[0][r]))
            l["loss"].backward()
            g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten()
                           for p in ref.parameters_to_train])
            total = g if total is None else total + g
        want = total / world
        err = float((gathered[0] - want).abs().max()) / float(want.abs().max())
        print("max rel-to-max difference of the averaged gradient: %.3e" % err)
        assert err < 1e-4, err
        # and a full train_step keeps the ranks in lock-step
    for s in range(STEPS):
        tr.train_step(dict(batches[s][rank]))
    torch.cuda.synchronize()
    mine = torch.cat([p.detach().flatten() for p in tr.parameters_to_train])
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if rank == 0:
        for g in gathered[1:]:
            assert torch.equal(g, gathered[0]), "ranks diverged"
        print("DDP_CHECK_OK")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
