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


def _worker_flat(rank, world, port, q):
    bdist = _setup(rank, world, port)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
    if rank == 1:                      # ranks start different: attach() must broadcast rank 0's weights
        for p in net.parameters():
            p.data.add_(1.0)
    tr = _Holder()
    tr.parameters_to_train = list(net.parameters())
    tr.models = {"net": net}
    tr.grad_sync = None
    flat = bdist.attach(tr)
    assert tr.grad_sync is not None
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 6, generator=g)
    y = torch.randn(8, 1, generator=g)
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
    q.put((rank, float((got - want).abs().max())))
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
    out = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out)


def test_flat_gradient_average_world2():
    for rank, err in _run(_worker_flat):
        assert err < 1e-6, (rank, err)


def test_hot_path_shards_by_sample_world2():
    for rank, err_loss, err_grad in _run(_worker_shard):
        assert err_loss < 1e-6, (rank, err_loss)
        assert err_grad < 1e-4, (rank, err_grad)
