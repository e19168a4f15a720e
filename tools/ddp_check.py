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


def make_trainer(step_graph=False):
    opt = bench.make_options(B, 0, "md2")
    opt.height, opt.width = H, W
    opt.step_graph, opt.fused_adam = step_graph, True
    torch.manual_seed(7)
    return Trainer(opt), opt


def graph_mode():
    """--graph: the split-graph data-parallel step (forward+backward+pack graph | eager all-reduce | optimizer
    graph) against the eager data-parallel step on the same batches: same parameters after 4 steps (to the
    run-to-run tolerance of MIOpen's atomics), ranks bit-identical to each other."""
    rank, _, world = bdist.init_from_env()
    torch.cuda.set_device(0)
    res = {}
    for mode in (False, True):
        tr, opt = make_trainer(step_graph=mode)
        tr.set_train()
        bdist.attach(trainer=tr)
        assert tr.use_graph == mode and tr.grad_sync is not None
        assert hasattr(tr.grad_sync, "buckets") != mode      # hooks-free averager under graphs
        batch = synthetic_batch([1] * B, H, W, opt.scales, device="cuda:0", seed=100 + rank)
        for _ in range(4):
            _, losses = tr.train_step(dict(batch))
        torch.cuda.synchronize()
        assert tr.step == 4
        if mode:
            assert len(tr._graphs) == 1 and list(tr._graphs.values())[0][1] is not None
        res[mode] = (torch.cat([p.detach().flatten() for p in tr.parameters_to_train]), float(losses["loss"]))
    mine = res[True][0]
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    for g in gathered[1:]:
        assert torch.equal(g, gathered[0]), "ranks diverged under the split-graph step"
    init = torch.cat([p.detach().flatten() for p in make_trainer()[0].parameters_to_train])
    moved = float((res[True][0] - init).abs().max())
    diff = float((res[True][0] - res[False][0]).abs().max())
    print("rank %d: graph vs eager max parameter difference %.3e (graph path moved them by up to %.3e); loss %.6f vs %.6f"
          % (rank, diff, moved, res[True][1], res[False][1]))
    # 4 Adam steps from a random initialisation amplify MIOpen's atomically accumulated (run-to-run different) weight
    # gradients - a parameter whose gradient is round-off moves by +-lr per step either way: same bars as the
    # single-rank graph test (tests/test_gpu_trainer.py::test_step_graph_replay_matches_eager)
    assert moved > 1e-4, moved
    assert abs(res[True][1] - res[False][1]) < 5e-2 * abs(res[False][1])
    assert diff < 5e-3 * float(res[False][0].abs().max()), diff
    dist.barrier()
    if rank == 0:
        print("DDP_GRAPH_OK")
    dist.destroy_process_group()


def capture_mode():
    """--capture: ONE rank over RCCL (`BBD_DP_FORCE_ATTACH=1`; two ranks cannot share a GPU under RCCL): the step graph
    with the bucketed all-reduces captured inside (Trainer.dp_capture) against the eager overlapped loop.  Proves the
    mechanics - autograd hooks firing under capture, RCCL nodes in the graph, joins of the pose stream, replay - not the
    exchange itself (an average over one rank is the identity)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    os.environ["BBD_DP_FORCE_ATTACH"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    torch.backends.cudnn.deterministic = True          # MIOpen: no atomics-based weight gradients, runs comparable to 1e-5
    res = {}
    for mode in (False, True):
        tr, opt = make_trainer(step_graph=mode)
        tr.dp_capture = mode
        tr.set_train()
        bdist.attach(trainer=tr)
        assert tr.grad_sync is not None and hasattr(tr.grad_sync, "buckets") and tr.dp_capture == mode
        batch = synthetic_batch([1] * B, H, W, opt.scales, device="cuda:0", seed=100)
        for _ in range(4):
            _, losses = tr.train_step(dict(batch))
        torch.cuda.synchronize()
        assert tr.step == 4
        if mode:
            assert len(tr._graphs) == 1 and list(tr._graphs.values())[0][1] is None      # one graph, no tail
            print("buckets %d, launched inside the captured backward: %d" % (len(tr.grad_sync.buckets), tr.grad_sync.launched_in_backward))
            assert tr.grad_sync.launched_in_backward >= len(tr.grad_sync.buckets) - 1
        res[mode] = (torch.cat([p.detach().flatten() for p in tr.parameters_to_train]), float(losses["loss"]))
    init = torch.cat([p.detach().flatten() for p in make_trainer()[0].parameters_to_train])
    moved = float((res[True][0] - init).abs().max())
    diff = float((res[True][0] - res[False][0]).abs().max())
    print("captured-collective graph vs eager overlapped loop: max parameter difference %.3e (moved by up to %.3e); loss %.6f vs %.6f"
          % (diff, moved, res[True][1], res[False][1]))
    assert moved > 1e-4, moved
    assert diff <= 1e-5 * max(1.0, float(res[False][0].abs().max())), diff
    print("DDP_CAPTURE_OK")
    dist.destroy_process_group()


def pooled_mode():
    """--pooled: ONE rank over RCCL (`BBD_DP_FORCE_ATTACH=1`): the bucketed step graphs of the `--rand` recipe (pooled form:
    ONE graph per pose-row bucket, `Trainer._graph_step`) under data parallelism - the flat-gradient pack inside the graph
    (split-graph loop: graph | all-reduce | optimizer graph) and the captured-collective form (bucketed RCCL all-reduces as
    nodes of the one graph) - each over seven different orderings of one bucket against the eager data-parallel loop of the
    same trainer: one capture, seven replays, same parameters.  So the first multi-GPU run does not also debut a new
    capture path."""
    import warnings
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29518")
    os.environ["BBD_DP_FORCE_ATTACH"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    torch.backends.cudnn.deterministic = True
    orderings = [[7, 7, 7, 1], [7, 7, 6, 1], [7, 7, 3, 2], [7, 7, 3, 1], [7, 7, 2, 2], [7, 7, 2, 1], [7, 6, 6, 1]]     # 64 pose rows each
    Hs, Ws = 96, 160

    def run(step_graph, capture):
        opt = bench.make_options(B, 0, "boosted15")
        opt.height, opt.width, opt.rand = Hs, Ws, True
        opt.step_graph, opt.fused_adam = step_graph, True
        opt.scales = [0, 1, 2, 3]
        torch.manual_seed(7)
        tr = Trainer(opt)
        tr.opt.scales = [0]
        tr.dp_capture = capture
        tr.set_train()
        bdist.attach(trainer=tr)
        assert tr.grad_sync is not None and tr.flat_grads is not None and tr.pooled_step
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for i, ms in enumerate(orderings):
                b = synthetic_batch(ms, Hs, Ws, [0], device="cuda:0", seed=300 + i)
                b["cutt"] = torch.tensor(1.35)
                b["noise"] = torch.randn(B, Hs, Ws, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(i)) * 1e-5
                _, losses = tr.train_step(b)
        torch.cuda.synchronize()
        assert tr.step == len(orderings) and tr._pooled.stats["fallbacks"] == 0
        return torch.cat([p.detach().flatten() for p in tr.parameters_to_train]), float(losses["loss"]), tr

    pe, le, _ = run(False, False)
    for capture in (False, True):
        pg, lg, tr = run(True, capture)
        assert tr.dp_capture == capture
        assert tr.graph_stats == {"eager": 0, "captures": 1, "replays": len(orderings)}, tr.graph_stats
        graph, tail = list(tr._graphs.values())[0][:2]
        assert (tail is None) == capture                         # split graphs keep an optimizer graph behind the exchange
        diff = float((pg - pe).abs().max())
        print("pooled step graph (%s) vs eager data-parallel loop: max parameter difference %.3e; loss %.6f vs %.6f"
              % ("captured collectives" if capture else "split around the exchange", diff, lg, le))
        assert diff <= 1e-5 * max(1.0, float(pe.abs().max())), diff
    print("DDP_POOLED_OK")
    dist.destroy_process_group()


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
        assert err < 5e-4, err        # run-to-run level of MIOpen's atomically accumulated weight gradients: ~1e-4
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
    if "--capture" in sys.argv:
        capture_mode()
    elif "--pooled" in sys.argv:
        pooled_mode()
    elif "--graph" in sys.argv:
        graph_mode()
    else:
        main()
