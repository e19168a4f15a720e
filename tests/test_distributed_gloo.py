"""CPU tier: the N>1 path with world_size 2 over gloo (127.0.0.1).

(1) FlatGradients/GradientAverager: flat views, one all-reduce, average == single-process
    gradient on the concatenated batch.
(2) The hot path shards by sample with no data-path collective: the mean of the per-rank losses
    equals the reference's full-batch loss and per-rank gradients are the full-batch ones x world.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup(rank, world, port):
    for p in (os.path.dirname(HERE), HERE, os.path.join(os.path.dirname(HERE), "tools")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from baseboostdepth_amd import distributed as bdist
    r, l, w = bdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    return bdist


class _Holder:
    pass


def _worker_flat(rank, world, port, q, overlap=True):
    os.environ["BBD_NO_OVERLAP"] = "0" if overlap else "1"
    os.environ["BBD_BUCKET_BYTES"] = "16"          # several buckets even for this tiny model
    bdist = _setup(rank, world, port)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
    if rank >= 1:                      # ranks start different: attach() must broadcast rank 0's weights
        for p in net.parameters():
            p.data.add_(1.0)
    tr = _Holder()
    tr.parameters_to_train = list(net.parameters())
    tr.models = {"net": net}
    tr.grad_sync = None
    flat = bdist.attach(tr)
    assert tr.grad_sync is not None
    assert type(tr.grad_sync).__name__ == ("OverlappedGradientAverager" if overlap else "GradientAverager")
    if overlap:
        assert len(tr.grad_sync.buckets) >= 2
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4 * world, 6, generator=g)
    y = torch.randn(4 * world, 1, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    flat.zero()
    assert all(p.grad is None for p in net.parameters())
    ((net(xs) - ys) ** 2).mean().backward()
    tr.grad_sync()                             # pack into the flat buffer + one all-reduce
    lo, hi = flat.flat.data_ptr(), flat.flat.data_ptr() + flat.nbytes
    assert all(lo <= p.grad.data_ptr() < hi for p in net.parameters())   # grads are views of it
    got = flat.flat.clone()
    # single-process reference on the concatenated batch, same (rank-0) weights
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
    ref.load_state_dict(net.state_dict())
    ((ref(x) - y) ** 2).mean().backward()
    want = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    err = float((got - want).abs().max())
    # second step through the same hooks (state must reset cleanly)
    flat.zero()
    ((net(xs) - ys) ** 2).mean().backward()
    tr.grad_sync()
    err = max(err, float((flat.flat - want).abs().max()))
    q.put((rank, err))
    dist.destroy_process_group()


def _worker_flat_simple(rank, world, port, q):
    _worker_flat(rank, world, port, q, overlap=False)


def _worker_uneven_graphs(rank, world, port, q):
    """Ranks whose autograd graphs differ (rank 1 never touches the last layer) must still issue the
    same collective sequence: buckets launch strictly in order, leftovers are flushed with zeros."""
    os.environ["BBD_NO_OVERLAP"] = "0"
    os.environ["BBD_BUCKET_BYTES"] = "16"
    bdist = _setup(rank, world, port)
    torch.manual_seed(0)
    l1, l2 = torch.nn.Linear(6, 5), torch.nn.Linear(5, 1)
    tr = _Holder()
    tr.parameters_to_train = list(l1.parameters()) + list(l2.parameters())
    tr.models = {"l1": l1, "l2": l2}
    tr.grad_sync = None
    flat = bdist.attach(tr)
    x = torch.ones(4, 6) * (rank + 1)
    flat.zero()
    h = torch.tanh(l1(x))
    loss = (l2(h) ** 2).mean() if rank % 2 == 0 else (h ** 2).mean()      # odd ranks never touch the last layer
    loss.backward()
    tr.grad_sync()
    # reference: average of the two ranks' gradients computed locally
    grads = []
    for r in range(world):
        a1, a2 = torch.nn.Linear(6, 5), torch.nn.Linear(5, 1)
        a1.load_state_dict(l1.state_dict()); a2.load_state_dict(l2.state_dict())
        xr = torch.ones(4, 6) * (r + 1)
        hr = torch.tanh(a1(xr))
        ((a2(hr) ** 2).mean() if r % 2 == 0 else (hr ** 2).mean()).backward()
        grads.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                for p in list(a1.parameters()) + list(a2.parameters())]))
    want = sum(grads) / world
    q.put((rank, float((flat.flat - want).abs().max())))
    dist.destroy_process_group()


def _worker_real_networks(rank, world, port, q):
    """The four real networks: the ResNet encoders' unused `fc` layers never get a gradient, and with
    reverse-parameter-order buckets they sit in the FIRST buckets - the exchange must still start inside
    backward (it used to launch 0 of 4 buckets before the flush)."""
    os.environ["BBD_NO_OVERLAP"] = "0"
    os.environ["BBD_BUCKET_BYTES"] = str(32 << 20)
    bdist = _setup(rank, world, port)
    from baseboostdepth_amd import Trainer
    from baseboostdepth_amd.options import MonodepthOptions
    o = MonodepthOptions().parse("--no_cuda --weights_init scratch --height 64 --width 128 --batch_size 1".split())
    torch.manual_seed(3)
    tr = Trainer(o)
    tr.set_train()
    bdist.attach(tr)
    sync = tr.grad_sync
    assert len(tr.gradient_free_parameters()) == 4 and len(sync.buckets) >= 3
    g = torch.Generator().manual_seed(10 + rank)
    img = torch.rand(2, 3, 64, 128, generator=g)
    tr.flat_grads.zero()
    disp = tr.models["depth"](tr.models["encoder"](img))
    aa, tt = tr.models["pose"]([tr.models["pose_encoder"](torch.cat([img, img.flip(0)], 1))])
    loss = sum(d.mean() for d in disp.values()) + aa.sum() + tt.sum()
    loss.backward()
    sync()
    early = sync.launched_in_backward
    # result still equals the plain average of both ranks' gradients
    mine = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                      for p in tr.parameters_to_train])
    assert torch.equal(mine, tr.flat_grads.flat)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    same = all(torch.equal(g_, gathered[0]) for g_ in gathered)
    q.put((rank, (early, len(sync.buckets), same)))
    dist.destroy_process_group()


def _worker_shard(rank, world, port, q):
    _setup(rank, world, port)
    from golden_io import Case
    from fused_runner import run_direct_case
    from host_port import HostPortBackend
    case = Case("md2_b2_32x64")
    b = rank                                   # one sample per rank
    for k in list(case.inputs):
        v = case.inputs[k]
        if torch.is_tensor(v) and v.dim() >= 3 and v.shape[0] == case.B:
            case.inputs[k] = v[b:b + 1].contiguous()
    case.inputs["ordering"] = case.inputs["ordering"][b:b + 1]
    case.ms = case.ms[b:b + 1]
    case.B = 1
    case.disp = {s: d.detach()[b:b + 1].clone().requires_grad_(True) for s, d in case.disp.items()}
    case.poses = {f: T.detach()[b:b + 1].clone().requires_grad_(True) for f, T in case.poses.items()}
    case.noise = case.noise[b:b + 1].contiguous()
    tr, inputs, outputs, losses = run_direct_case(case, HostPortBackend(), materialize=False)
    losses["loss"].backward()
    loss = losses["loss"].detach().clone()
    dist.all_reduce(loss, op=dist.ReduceOp.SUM)
    loss /= world
    full = Case("md2_b2_32x64")
    err_loss = abs(float(loss) - float(full.expected("out/loss")))
    err_grad = 0.0
    for s in case.scales:
        ge = full.expected("grad/disp/%d" % s)[b:b + 1]
        err_grad = max(err_grad, float((case.disp[s].grad / world - ge).abs().max()) / float(ge.abs().max()))
    q.put((rank, err_loss, err_grad))
    dist.destroy_process_group()


def _run(worker, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    out, waited = [], 0
    while len(out) < world:
        try:
            out.append(q.get(timeout=2))
        except queue.Empty:
            waited += 2
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or waited > 180:
                for p in procs:
                    p.kill()
                raise AssertionError("worker failed (exit codes %s)" % dead)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out)


def test_flat_gradient_average_world2():
    for rank, err in _run(_worker_flat_simple):
        assert err < 1e-6, (rank, err)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_overlapped_bucketed_average(world):
    """BASELINE's metric is quoted at 1/2/4/8 GPUs: the bucketed exchange at every one of those world sizes (gloo)."""
    out = _run(_worker_flat, world)
    assert [r for r, _ in out] == list(range(world))
    for rank, err in out:
        assert err < 1e-6, (rank, err)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_overlapped_average_with_rank_specific_graphs(world):
    """Bucket order with rank-specific autograd graphs (boosted batches: every rank draws its own frame sets), at 2, 4
    and 8 ranks: odd ranks never reach the last layer, all ranks must still issue the same collective sequence."""
    out = _run(_worker_uneven_graphs, world)
    assert len(out) == world
    for rank, err in out:
        assert err < 1e-6, (rank, err)


def test_overlap_starts_inside_backward_on_the_real_networks():
    for rank, (early, n_buckets, same) in _run(_worker_real_networks):
        assert same
        assert early >= n_buckets - 1, "rank %d launched %d of %d buckets inside backward" % (rank, early, n_buckets)


@pytest.mark.parametrize("world,n,batch", [(8, 103, 5), (4, 1001, 12), (8, 39810, 12)])
def test_loader_shards_a_non_divisible_dataset(world, n, batch):
    """`order[rank::world]` of one common shuffle with a dataset length that world * batch does not divide (the KITTI
    split has 39 810 lines): disjoint shards, equal length on every rank, at most world * batch - 1 samples dropped."""
    from baseboostdepth_amd.datasets import DeviceLoader

    class _DS:
        epoch = 7
        is_train = True

        def __len__(self):
            return n

    seen, lengths = [], set()
    for rank in range(world):
        ld = DeviceLoader(_DS(), batch, collate=None, shuffle=True, drop_last=True, num_workers=0, seed=42, rank=rank, world=world)
        chunks = list(ld._batches())
        assert all(len(c) == batch for c in chunks) and len(chunks) == len(ld)
        lengths.add(len(chunks))
        seen.append({i for c in chunks for i in c})
    assert len(lengths) == 1, "ranks would run different numbers of steps: %s" % lengths      # (collectives would mis-pair)
    union = set().union(*seen)
    assert sum(len(s_) for s_ in seen) == len(union)          # disjoint
    assert n - len(union) < world * batch


def test_loader_shards_indices_by_rank():
    from baseboostdepth_amd.datasets import DeviceLoader

    class _DS:
        epoch = 3
        is_train = True

        def __len__(self):
            return 103

    seen = []
    for rank in range(4):
        ld = DeviceLoader(_DS(), 5, collate=None, shuffle=True, drop_last=True, num_workers=0, seed=42, rank=rank, world=4)
        idx = [i for chunk in ld._batches() for i in chunk]
        assert len(idx) == len(ld) * 5 == 25
        seen.append(set(idx))
    for a in range(4):
        for b in range(a + 1, 4):
            assert not (seen[a] & seen[b]), "ranks %d and %d read the same samples" % (a, b)
    assert len(set().union(*seen)) == 100


def test_hot_path_shards_by_sample_world2():
    for rank, err_loss, err_grad in _run(_worker_shard):
        assert err_loss < 1e-6, (rank, err_loss)
        assert err_grad < 1e-4, (rank, err_grad)
