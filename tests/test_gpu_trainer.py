"""GPU tier: Trainer.process_batch / train_step end to end with the real ResNet-18 networks, and the
hot-path part of it re-checked against the live oracle on the networks' own outputs."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_opt(H, W, B, scales, boosted):
    return types.SimpleNamespace(
        height=H, width=W, batch_size=B, scales=list(scales), frame_ids=[0, -1, 1], min_depth=0.1, max_depth=100.0,
        disparity_smoothness=1e-3, no_ssim=False, trimin=boosted, decomp=boosted, pose_error=5.5,
        incremental_skip=boosted, partial_skip=boosted, materialize_warps=False, num_layers=18,
        weights_init="scratch", learning_rate=1e-4, no_cuda=False, cuda=0, load_weights_folder="None",
        log_dir="/tmp", model_name="t")


def oracle_on_outputs(tr, inputs, outputs, opt, ms):
    """Run the CPU oracle on the disparities / poses the GPU networks produced."""
    from oracle import hotpath_ref as O
    plan = tr.plan
    cin = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in inputs.items()}
    disp = {s: outputs[("disp", s)].detach().cpu().requires_grad_(True) for s in opt.scales}
    poses, perr = {}, {}
    job_poses = tr._job_poses(inputs, outputs)
    for (kind, f), T in job_poses.items():
        if f == "s":
            continue
        (poses if kind == "T" else perr)[f] = T.detach().cpu()
    return O.hot_path(cin, disp, poses, ms, opt.scales, opt.trimin, opt.decomp, cin["noise"].cpu(), opt.height,
                      opt.width, poses_error=perr), disp


@pytest.mark.parametrize("boosted,ms,scales,cutt", [
    (False, [1, 1, 1], [0, 1, 2, 3], 0.3),
    (True, [3, 1, 2, 5], [0], 1.35),
    (True, [2, 1, 0, 2], [0, 1, 2, 3], 0.3),
])
def test_process_batch_end_to_end(boosted, ms, scales, cutt):
    from baseboostdepth_amd.trainer import Trainer
    from baseboostdepth_amd.synthetic import synthetic_batch
    H, W, B = 96, 160, len(ms)
    torch.manual_seed(0)
    # like the reference: built with the default four scales (num_scales frozen at 4, trainer.py:44),
    # the curriculum then narrows opt.scales per epoch (run_epoch, trainer.py:209-212)
    opt = make_opt(H, W, B, [0, 1, 2, 3], boosted)
    tr = Trainer(opt)
    tr.opt.scales = list(scales)
    tr.set_train()
    inputs = synthetic_batch(ms, H, W, scales, device=DEV, seed=3)
    inputs["cutt"] = torch.tensor(cutt)
    tr.opt.frame_ids = sorted(inputs["frames"], key=lambda it: float("inf") if isinstance(it, str) else abs(it))
    outputs, losses = tr.process_batch(inputs)
    assert torch.isfinite(losses["loss"])
    for s in scales:
        assert outputs[("disp", s)].shape == (B, 1, H >> s, W >> s)
        assert outputs[("depth", 0, s)].shape == (B, 1, H, W)
    ref, ref_disp = oracle_on_outputs(tr, inputs, outputs, opt, ms)
    assert abs(float(losses["loss"].detach()) - float(ref["loss"].detach())) < 1e-5
    for i, s in enumerate(scales):
        got = outputs[("bbd", "to_optimise")][i].cpu()
        assert float((got - ref["min/%d" % s]).abs().max()) < 1e-4
        mism = outputs[("bbd", "argmin")][i].cpu() != ref["argmin/%d" % s]
        assert int((mism & (ref["margin/%d" % s] > 2e-4)).sum()) == 0
    # gradients reach all four networks (the unused torchvision fc layers stay without gradient)
    losses["loss"].backward()
    for name, model in tr.models.items():
        got = [p.grad is not None and float(p.grad.abs().sum()) > 0 for n, p in model.named_parameters()
               if ".fc." not in n]
        assert sum(got) >= 0.75 * len(got), name     # unused-scale dispconvs get none when scales=[0]


@pytest.mark.parametrize("pooled", [False, True])
def test_boosted_process_batch_at_baseline_size(pooled):
    """(`pooled`: the same check through the pooled form of the step - one frame pool, static step tables, the pose pass padded
    to a measured row count with device-resident call groups - that `--rand` training runs in since round 6.)
    VERDICT r3 item 7: the boosted recipe's WHOLE step at BASELINE configs[2] size - B = 12, 192x640, trimin + decomp +
    incremental + partial pose modes (cutt 1.35: the epoch >= 10 regime, scale 0 only), per-sample offsets = bench.py's
    epoch-15 draw (mixed 8 / 14 / 18-candidate samples, i.e. the sorted slab work order is on) - against the live oracle on
    the networks' own disparities and poses: loss 1e-5, min-loss maps 1e-4, arg-min equal where the oracle's margin exceeds
    2e-4, disparity gradient with the flip-aware protocol of test_full_resolution_against_oracle (trainer.py:286-308,
    310-419, 444-570, 983-1100)."""
    import random
    from baseboostdepth_amd.trainer import Trainer
    from baseboostdepth_amd.synthetic import synthetic_batch
    H, W, B = 192, 640, 12
    ms = random.Random(1234).choices(range(1, 8), [.050, .050, .077, .094, .139, .142, .448], k=B)   # bench.py boosted15, rank 0
    assert len(set(ms)) > 2
    torch.manual_seed(0)
    opt = make_opt(H, W, B, [0, 1, 2, 3], True)
    opt.rand = pooled
    tr = Trainer(opt)
    tr.opt.scales = [0]
    tr.set_train()
    inputs = synthetic_batch(ms, H, W, [0], device=DEV, seed=42)
    inputs["cutt"] = torch.tensor(1.35)
    tr.opt.frame_ids = sorted(inputs["frames"], key=lambda it: float("inf") if isinstance(it, str) else abs(it))
    outputs, losses = tr.process_batch(inputs)
    assert (("bbd", "pose_matrices") in outputs) == pooled and (tr.last_pooled is not None) == pooled
    assert tr.plan.sample_order is not None                     # mixed candidate counts: most-candidates-first order in use
    assert sorted(len(n) for n in tr.plan.cand_names)[0] < 18 and max(len(n) for n in tr.plan.cand_names) == 18
    disp_gpu = outputs[("disp", 0)]
    disp_gpu.retain_grad()
    losses["loss"].backward()
    ref, ref_disp = oracle_on_outputs(tr, inputs, outputs, opt, ms)
    ref["loss"].backward()
    assert abs(float(losses["loss"].detach()) - float(ref["loss"].detach())) < 1e-5
    got = outputs[("bbd", "to_optimise")][0].cpu()
    assert float((got - ref["min/0"]).abs().max()) < 1e-4
    mism = outputs[("bbd", "argmin")][0].cpu() != ref["argmin/0"]
    assert int((mism & (ref["margin/0"] > 2e-4)).sum()) == 0
    flips = int(mism.sum())
    ge, gg = ref_disp[0].grad, disp_gpu.grad.cpu()
    rel = (gg - ge).abs() / float(ge.abs().max())
    n_bad = int((rel > 1e-4).sum())
    assert n_bad <= 25 * flips, (flips, n_bad, float(rel.max()))
    assert float((gg - ge).norm() / ge.norm()) < (1e-4 if flips == 0 else 5e-2)
    for name, model in tr.models.items():
        got = [p.grad is not None and float(p.grad.abs().sum()) > 0 for n, p in model.named_parameters() if ".fc." not in n]
        assert sum(got) >= 0.75 * len(got), name


def test_train_step_updates_weights_and_is_deterministic():
    from baseboostdepth_amd.trainer import Trainer
    from baseboostdepth_amd.synthetic import synthetic_batch
    H, W, ms = 96, 160, [1, 1]
    losses = []
    for rep in range(2):
        torch.manual_seed(0)
        tr = Trainer(make_opt(H, W, 2, [0, 1, 2, 3], False))
        tr.set_train()
        inputs = synthetic_batch(ms, H, W, [0, 1, 2, 3], device=DEV, seed=3)
        before = [p.detach().clone() for p in tr.models["depth"].parameters()]
        seq = []
        for _ in range(3):
            _, l = tr.train_step(dict(inputs))
            seq.append(float(l["loss"].detach()))
        after = list(tr.models["depth"].parameters())
        assert any(float((a - b).abs().max()) > 0 for a, b in zip(after, before))
        losses.append(seq)
    assert abs(losses[0][0] - losses[1][0]) < 1e-6          # same seed, same first loss
    assert all(l == l for l in losses[0])                    # finite


def test_checkpoint_roundtrip(tmp_path):
    from baseboostdepth_amd.trainer import Trainer
    opt = make_opt(96, 160, 2, [0, 1, 2, 3], False)
    opt.log_dir = str(tmp_path)
    torch.manual_seed(1)
    a = Trainer(opt)
    folder = a.save_model("unit")
    sd = torch.load(folder + "/encoder.pth")
    assert sd["height"] == 96 and sd["width"] == 160          # consumers read the resolution from here
    opt2 = make_opt(96, 160, 2, [0, 1, 2, 3], False)
    opt2.load_weights_folder = folder
    opt2.models_to_load = ["encoder", "depth", "pose_encoder", "pose"]
    torch.manual_seed(2)
    b = Trainer(opt2)
    for name in a.models:
        for (k, v), (_, w) in zip(a.models[name].state_dict().items(), b.models[name].state_dict().items()):
            assert torch.equal(v, w), (name, k)


def test_loss_goes_down_on_a_fixed_batch():
    """End-to-end sanity of the gradients through depth + pose nets: 40 Adam steps on one batch."""
    from baseboostdepth_amd.trainer import Trainer
    from baseboostdepth_amd.synthetic import synthetic_batch
    H, W, ms = 96, 160, [1, 1, 1, 1]
    torch.manual_seed(0)
    opt = make_opt(H, W, 4, [0, 1, 2, 3], False)
    opt.learning_rate = 2e-4
    tr = Trainer(opt)
    tr.set_train()
    inputs = synthetic_batch(ms, H, W, [0, 1, 2, 3], device=DEV, seed=5)
    hist = []
    for _ in range(40):
        _, l = tr.train_step(dict(inputs))
        hist.append(float(l["loss"].detach()))
    assert all(h == h for h in hist)
    assert sum(hist[-5:]) / 5 < sum(hist[:5]) / 5 - 1e-3, (hist[:5], hist[-5:])


def _deterministic_convolutions(monkeypatch):
    """MIOpen's split-K weight-gradient solvers accumulate with atomics (run-to-run different bits); with
    `cudnn.deterministic` PyTorch-ROCm sets MIOPEN_CONVOLUTION_ATTRIB_DETERMINISTIC and those solvers are not chosen.
    Everything else of the step (the HIP kernels here, fused Adam) is deterministic by construction."""
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    monkeypatch.setattr(torch.backends.cudnn, "benchmark", False)


def test_pose_stream_on_off_same_step(monkeypatch):
    """The pose network runs on a second HIP stream by default; with it disabled the step computes the
    same loss and the same gradients (deterministic convolution solvers: a missing stream dependency or a dropped
    update cannot hide under solver noise)."""
    from baseboostdepth_amd.trainer import Trainer
    from baseboostdepth_amd.synthetic import synthetic_batch
    _deterministic_convolutions(monkeypatch)
    H, W, B = 96, 320, 4
    opt = make_opt(H, W, B, [0, 1, 2, 3], False)
    torch.manual_seed(3)
    tr = Trainer(opt)
    tr.set_train()
    state = {k: {n: v.clone() for n, v in m.state_dict().items()} for k, m in tr.models.items()}
    batch = synthetic_batch([1] * B, H, W, opt.scales, device=DEV, seed=11)

    def run(flag):
        monkeypatch.setenv("BBD_POSE_STREAM", flag)
        for k, m in tr.models.items():
            m.load_state_dict(state[k])
        tr.model_optimizer.zero_grad(set_to_none=True)
        _, losses = tr.process_batch(dict(batch))
        losses["loss"].backward()
        torch.cuda.synchronize()
        g = torch.cat([p.grad.flatten() for p in tr.parameters_to_train if p.grad is not None])
        bufs = {"%s.%s" % (k, n): b.detach().clone() for k, m in tr.models.items() for n, b in m.named_buffers()}
        return float(losses["loss"].detach()), g, bufs

    l1, g1, b1 = run("1")
    l0, g0, b0 = run("0")
    assert abs(l1 - l0) < 1e-6
    assert float((g1 - g0).abs().max()) <= 1e-5 * float(g0.abs().max())
    for k in b0:                                  # BatchNorm running statistics and batch counters
        assert torch.equal(b0[k], b1[k]) if not b0[k].is_floating_point() else \
            float((b0[k] - b1[k]).abs().max()) <= 1e-6 * (1.0 + float(b0[k].abs().max())), k


def test_two_ranks_on_one_gpu_exchange_gradients():
    """Production multi-rank path (second stream + bucketed overlapped all-reduce) with both ranks on this
    GPU over gloo: tools/ddp_check.py compares the exchanged gradient with a single-process replay."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BBD_DIST_BACKEND="gloo", BBD_BUCKET_BYTES="4000000", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = str(29600 + os.getpid() % 300)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(root, "tools", "ddp_check.py")],
                         env=env, capture_output=True, text=True, timeout=900)
    assert "DDP_CHECK_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])


def test_two_ranks_split_graph_step_matches_eager_data_parallel():
    """Multi-rank step as hipGraphs (forward+backward+gradient pack | eager all-reduce | optimizer), two ranks on this
    GPU over gloo: same parameters as the eager data-parallel step, ranks identical (tools/ddp_check.py --graph)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BBD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = str(29300 + os.getpid() % 300)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(root, "tools", "ddp_check.py"),
                          "--graph"], env=env, capture_output=True, text=True, timeout=900)
    assert "DDP_GRAPH_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])


def test_rccl_collectives_captured_into_the_step_graph_one_rank():
    """Data-parallel step as ONE hipGraph with the bucketed all-reduces captured inside (`Trainer.dp_capture`,
    `bench.py --dp-mode graph-overlap`), over RCCL with the one rank a single GPU allows: the autograd hooks fire under
    capture, the RCCL nodes replay, parameters equal the eager overlapped loop's (tools/ddp_check.py --capture)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:      # a port that is free NOW (a fixed one can still be
        sock.bind(("127.0.0.1", 0))                                      # in TIME_WAIT from an earlier test of the run)
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BBD_BUCKET_BYTES="4000000", MASTER_PORT=str(port))
    env.pop("BBD_DIST_BACKEND", None)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "ddp_check.py"), "--capture"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert "DDP_CAPTURE_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])


def test_bench_two_ranks_sharing_this_gpu_print_one_split_graph_line():
    """`python bench.py --gpus 2` end to end (self-launch, rendezvous, split-graph data-parallel loop, max over ranks, ONE
    JSON line from rank 0), with both ranks on this GPU and gloo standing in for RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BBD_DIST_BACKEND="gloo", BBD_SHARE_GPU0="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "2", "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-1500:], out.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["collective"] == "gloo" and line["value"] > 0
    assert line["config"]["global_batch"] == 4 and str(line["step_graph"]).startswith("split")
    # diagnostics of the exchange: which loop, which reduction, how long the step waited for it
    assert line["dp_mode"] == "graph" and line["reduce_op"] == "SUM+div" and line["exchange_ms"] > 0
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "2", "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline", "--dp-mode", "overlap"], env=env, capture_output=True,
                         text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-1500:], out.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["dp_mode"] == "overlap" and line["step_graph"] is False and line["exchange_ms"] > 0
    assert line["buckets_launched_in_backward"] is not None
    # round 4: both ranks report their device (here the SAME GPU, which is exactly what `devices` must expose), the ranks'
    # own step times, and `--dp-mode all` measures the three loops in one launch (gloo cannot be captured: graph-overlap
    # falls back to the split-graph loop and says so)
    assert line["devices"] == 1 and line["ms_per_step_rank_min"] <= line["ms_per_step_rank_max"]
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--batch", "2", "--steps", "3",
                          "--warmup", "1", "--no-cpu-baseline", "--dp-mode", "all", "--dp-extra-timeout", "300"], env=env,
                         capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-1500:], out.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["dp_mode"] == "graph" and set(line["dp_modes"]) == {"graph", "overlap", "graph-overlap"}
    assert line["dp_modes"]["overlap"]["dp_mode"] == "overlap" and line["dp_modes"]["overlap"]["value"] > 0
    assert line["dp_modes"]["graph-overlap"]["value"] > 0


def test_step_graph_replay_matches_eager(monkeypatch):
    """Opt-in whole-step hipGraph (`opt.step_graph`): capture after an eager warm-up that must NOT train
    (parameters, BatchNorm buffers, Adam state and the step counter are restored), then every batch with
    the same signature is one graph launch.  Same parameters AND BatchNorm running statistics as the eager loop on
    the SAME batch sequence, with deterministic convolution solvers so that the bar is rounding, not solver noise."""
    import warnings
    from baseboostdepth_amd.trainer import Trainer
    from baseboostdepth_amd.synthetic import synthetic_batch
    _deterministic_convolutions(monkeypatch)
    H, W, B = 96, 320, 4
    batches = [synthetic_batch([1] * B, H, W, [0, 1, 2, 3], device=DEV, seed=20 + i) for i in range(3)]

    def run(graph):
        opt = make_opt(H, W, B, [0, 1, 2, 3], False)
        opt.step_graph = graph
        torch.manual_seed(5)
        tr = Trainer(opt)
        tr.set_train()
        seq = [0, 1, 2, 1]                  # the first batch of a signature is counted once on both paths
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for i in seq:
                _, losses = tr.train_step(dict(batches[i]))
        torch.cuda.synchronize()
        return torch.cat([p.detach().flatten() for p in tr.parameters_to_train]), float(losses["loss"]), tr

    pe, le, tre = run(False)
    pg, lg, trg = run(True)
    assert trg.use_graph and len(trg._graphs) == 1 and trg.step == 4
    assert abs(le - lg) <= 1e-5 * abs(le)
    assert float((pe - pg).abs().max()) <= 1e-5 * float(pe.abs().max())
    for k, m in tre.models.items():               # a dropped running-statistics update would show here
        bg = dict(trg.models[k].named_buffers())
        for n, be in m.named_buffers():
            if be.is_floating_point():
                assert float((be - bg[n]).abs().max()) <= 1e-5 * (1.0 + float(be.abs().max())), (k, n)
            else:
                assert torch.equal(be, bg[n]), (k, n)


@pytest.mark.parametrize("name", ["pose_plain_3105_32x64", "pose_incr_3215_32x64", "pose_incr_partial_4327_32x64",
                                  "pose_md2_b2_32x64"])
def test_predict_poses_modes_on_gpu_against_reference_vectors(name):
    """The three pose modes through the GPU path (`bbd_pose_matrix` kernels, index-select sub-batches,
    the `torch.where` partial swap, pose table -> fused launch) pinned to the SAME reference-generated
    fixtures as the CPU tier (tests/test_trainer_poses.py)."""
    from pose_checks import check_pose_case
    from baseboostdepth_amd import ops
    check_pose_case(name, ops.default_backend(), "cuda:0")


def _run_bench_inline(capsys, argv):
    import json
    import bench
    rc = bench.main(argv)
    assert rc == 0
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_line_with_step_graph_and_fallback(capsys, monkeypatch):
    """bench.py replays the step as one hipGraph by default; if capture fails it must fall back to the eager loop
    (and say so) instead of losing the benchmark; `--step-graph off` is the plain eager loop."""
    from baseboostdepth_amd.trainer import Trainer
    argv = ["--batch", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-eager-ab", "--no-secondary"]
    line = _run_bench_inline(capsys, argv)
    assert line["step_graph"] is True and line["value"] > 0 and line["n_gpus"] == 1 and line["collective"] == "none"
    assert set(line["kernels"]) == {"bbd_identity_loss_fwd", "bbd_warp_ssim_min_disp_fwd", "bbd_warp_ssim_min_disp_bwd"}
    assert 0 < line["roofline"]["frac"] < 1 and line["ms_per_step_median"] > 0
    off = _run_bench_inline(capsys, argv + ["--step-graph", "off"])
    assert off["step_graph"] is False and "inside the timed steps" in off["kernel_timing"]

    def broken(self, inputs):
        raise RuntimeError("injected capture failure")
    monkeypatch.setattr(Trainer, "_graph_step", broken)
    fb = _run_bench_inline(capsys, argv)
    assert isinstance(fb["step_graph"], str) and "capture failed" in fb["step_graph"] and fb["value"] > 0


@pytest.mark.parametrize("ms,partial", [([7, 5, 4, 3], True), ([3, 1, 2, 5], False), ([2, 1, 2, 1], True)])
def test_pose_composition_kernel_equals_the_torch_loops(ms, partial, monkeypatch):
    """the incremental chain, T_error and the partial swap as one launch each way
    (`bbd_pose_compose_fwd/bwd`) against the per-frame torch loops of `predict_poses` (reference trainer.py:359-388,
    403-405, 415-418): same key set, same matrices, same gradients into the pose networks."""
    from baseboostdepth_amd import ops
    from baseboostdepth_amd.trainer import Trainer
    from baseboostdepth_amd.synthetic import synthetic_batch
    _deterministic_convolutions(monkeypatch)    # the comparison reaches the pose networks' weight gradients
    H, W, B = 96, 160, len(ms)
    torch.manual_seed(1)
    opt = make_opt(H, W, B, [0, 1, 2, 3], True)
    opt.partial_skip = partial
    tr = Trainer(opt)
    tr.opt.scales = [0]
    tr.set_eval()                               # no BatchNorm statistics in the way of an exact comparison
    inputs = synthetic_batch(ms, H, W, [0], device=DEV, seed=5)
    inputs["cutt"] = torch.tensor(1.35)         # epoch >= 10: incremental (+ partial) pose modes
    tr.opt.frame_ids = sorted(inputs["frames"], key=lambda it: float("inf") if isinstance(it, str) else abs(it))
    tr.valid_frames_trimin(inputs)
    params = [p for k in ("pose_encoder", "pose") for p in tr.models[k].parameters()]

    def run(fused):
        monkeypatch.setattr(ops, "FUSED_POSE_COMPOSE", fused)
        for p in params:
            p.grad = None
        out = tr.predict_poses(inputs)
        gen = torch.Generator().manual_seed(9)
        loss = 0.0
        for k in sorted(out, key=str):
            loss = loss + (out[k] * torch.rand(out[k].shape, generator=gen).to(DEV)).sum()
        loss.backward()
        return out, [p.grad.clone() if p.grad is not None else None for p in params]

    a, ga = run(True)
    b, gb = run(False)
    assert set(a) == set(b)
    for k in a:
        assert a[k].shape == b[k].shape, k
        assert float((a[k] - b[k]).abs().max()) <= 2e-6, k
        assert a[k].requires_grad == b[k].requires_grad, k
    for x, y in zip(ga, gb):
        assert (x is None) == (y is None)
        if x is not None:
            assert float((x - y).abs().max()) <= 1e-5 * (float(y.abs().max()) + 1e-12)
