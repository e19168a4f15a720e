"""CPU tier, round-4 host logic: the fused launches' work-item table (a host function of the C-ABI library), the fused
loss-combine node, the multi-scale smoothness node through the host port, bench.py's staleness guard and launch timeout."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("B,S,H,W,backward,order", [
    (12, 4, 192, 640, 1, None), (12, 4, 192, 640, 0, None), (12, 1, 192, 640, 1, "rev"), (3, 2, 32, 64, 0, "rev"),
    (5, 3, 96, 160, 1, "rev"), (2, 4, 48, 80, 1, None), (1, 1, 16, 32, 0, None), (7, 2, 100, 333, 1, "rev")])
def test_fused_work_items_is_a_permutation_in_slab_order(B, S, H, W, backward, order):
    """bbd_fused_work_items: every (sample, scale, tile) exactly once; word 1 = the tile's pixel origin; the blocks an XCD
    receives (block i runs on XCD i mod 8) walk ONE contiguous range of tiles per (sample, scale) - its slab - and take the
    samples in the caller's order."""
    from baseboostdepth_amd import _lib
    lib = _lib.get_lib()
    ntiles = lib.num_tiles_bwd(H, W) if backward else lib.num_tiles_fwd(H, W)
    n = S * B * ntiles
    host = torch.empty(n, 2, dtype=torch.int32)
    seq = list(range(B))[::-1] if order else list(range(B))
    arr = (ctypes.c_int32 * B)(*seq) if order else None
    lib.call("bbd_fused_work_items", B, S, H, W, backward, arr, host.data_ptr())
    w0, w1 = host[:, 0].numpy().astype(np.int64), host[:, 1].numpy().astype(np.int64)
    b, s, t = w0 & 4095, (w0 >> 12) & 7, w0 >> 15
    assert len(set(zip(b.tolist(), s.tolist(), t.tolist()))) == n and b.max() < B and s.max() < S and t.max() < ntiles
    tw = 32 if backward else 64
    tiles_x = (W + tw - 1) // tw
    assert ((w1 & 0xffff) == (t % tiles_x) * tw).all() and ((w1 >> 16) == (t // tiles_x) * 16).all()
    if ntiles % 8 == 0:                       # slabs coincide with the XCDs' block ranges
        for x in range(8):
            tt, bb = t[x::8], b[x::8]
            assert tt.min() == x * ntiles // 8 and tt.max() == (x + 1) * ntiles // 8 - 1
            first = [int(v) for i, v in enumerate(bb) if i == 0 or bb[i - 1] != v]
            assert first == seq               # every sample once, in the caller's order


def test_fused_work_items_rejects_a_bad_order():
    from baseboostdepth_amd import _lib
    lib = _lib.get_lib()
    host = torch.empty(2 * 1 * lib.num_tiles_fwd(32, 64), 2, dtype=torch.int32)
    with pytest.raises(_lib.BbdError):
        lib.call("bbd_fused_work_items", 2, 1, 32, 64, 0, (ctypes.c_int32 * 2)(0, 5), host.data_ptr())
    # in range but not a permutation: sample 1 would be skipped, sample 0 run twice
    host3 = torch.empty(3 * 1 * lib.num_tiles_fwd(32, 64), 2, dtype=torch.int32)
    with pytest.raises(_lib.BbdError):
        lib.call("bbd_fused_work_items", 3, 1, 32, 64, 0, (ctypes.c_int32 * 3)(0, 2, 0), host3.data_ptr())


def test_plan_orders_samples_by_candidate_count():
    from baseboostdepth_amd.plan import get_plan
    uniform = get_plan([[0, 1, -1]] * 4, False, False)
    assert uniform.sample_order is None
    mixed = get_plan([[0, 7, -7], [0, 1, -1], [0, 7, -7], [0, 2, -2]], True, True)
    counts = [len(n) for n in mixed.cand_names]
    assert mixed.sample_order == sorted(range(4), key=lambda i: -counts[i]) and counts[mixed.sample_order[0]] == 18


def test_combine_losses_equals_the_reference_arithmetic():
    """ops.combine_losses vs trainer.py:557-568 written out with scalar torch ops, values and gradients."""
    from baseboostdepth_amd import ops
    torch.manual_seed(0)
    scales, n_px, sm, ns = [0, 1, 2, 3], 12 * 192 * 640, 1e-3, 4
    ls = (torch.rand(4) * 1e5).requires_grad_(True)
    sv = torch.rand(4).requires_grad_(True)
    total, per = ops.combine_losses(ls, sv, n_px, sm, scales, ns)
    ls2, sv2 = ls.detach().clone().requires_grad_(True), sv.detach().clone().requires_grad_(True)
    ref_total, ref_per = 0, []
    for i, s in enumerate(scales):
        loss = ls2[i] / n_px
        loss = loss + sm * sv2[i] / (2 ** s)
        ref_total = ref_total + loss
        ref_per.append(loss)
    ref_total = ref_total / ns
    assert abs(float(total) - float(ref_total)) <= 1e-6 * abs(float(ref_total))
    assert torch.allclose(per.detach(), torch.stack(ref_per).detach(), rtol=1e-6, atol=0)
    total.backward()
    ref_total.backward()
    assert torch.allclose(ls.grad, ls2.grad, rtol=1e-6, atol=0) and torch.allclose(sv.grad, sv2.grad, rtol=1e-6, atol=0)
    # a gradient into a per-scale loss reaches the inputs too
    ls3 = ls.detach().clone().requires_grad_(True)
    _, per3 = ops.combine_losses(ls3, sv.detach(), n_px, sm, scales, ns)
    per3[2].backward()
    assert float(ls3.grad[2]) == pytest.approx(1.0 / n_px, rel=1e-6) and float(ls3.grad[0]) == 0.0


def test_multi_scale_smoothness_equals_the_single_scale_node():
    """ops.normalised_smooth_losses (one launch pair for all scales) vs ops.normalised_smooth_loss per scale, through the
    host port of the kernels' arithmetic."""
    import host_port
    from baseboostdepth_amd import ops
    be = host_port.HostPortBackend()
    g = torch.Generator().manual_seed(3)
    B, H, W = 3, 32, 64
    disps = [torch.rand(B, 1, H >> s, W >> s, generator=g).requires_grad_(True) for s in range(4)]
    imgs = [torch.rand(B, 3, H >> s, W >> s, generator=g) for s in range(4)]
    multi = ops.normalised_smooth_losses(disps, imgs, be)
    (multi * torch.tensor([1.0, 0.5, 0.25, 0.125])).sum().backward()
    grads = [d.grad.clone() for d in disps]
    for i, (d, im) in enumerate(zip(disps, imgs)):
        d2 = d.detach().clone().requires_grad_(True)
        one = ops.normalised_smooth_loss(d2, im, be)
        assert float(one) == pytest.approx(float(multi[i]), rel=1e-6)
        (one * 0.5 ** i).backward()
        assert torch.allclose(d2.grad, grads[i], rtol=1e-6, atol=1e-12)


def test_bench_staleness_guard(tmp_path, monkeypatch):
    """bench.committed_constants: files stamped with another kernel-source hash (or none) are reported stale; the
    committed round-4 instruction mix matches the shipped source."""
    sys.path.insert(0, ROOT)
    import bench
    from baseboostdepth_amd.csrc.build import source_sha16
    cc = bench.committed_constants("md2")
    assert cc["source_sha16"] == source_sha16() and cc["isa_mix_path"].startswith("profiles/")
    mix = json.load(open(os.path.join(ROOT, cc["isa_mix_path"])))
    if mix.get("kernel_source_sha16") == source_sha16():
        assert not any(w.startswith("isa_mix") for w in cc["stale"])
    else:
        assert any(w.startswith("isa_mix") for w in cc["stale"])
    # a fake repo root whose files carry a foreign hash
    fake = tmp_path / "profiles" / "r04"
    fake.mkdir(parents=True)
    (fake / "traffic_md2.json").write_text(json.dumps({"kernel_source_sha16": "0" * 16, "bbd_warp_ssim_min_bwd": {"traffic_bytes": 1}}))
    (fake / "isa_mix.json").write_text(json.dumps({"bbd_warp_ssim_min_bwd": {"cycles_per_valu_instruction": 3.0}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    cc = bench.committed_constants("md2")
    assert len(cc["stale"]) == 2 and cc["traffic"]["bbd_warp_ssim_min_bwd"]["traffic_bytes"] == 1


def test_self_launch_times_out_and_kills_the_child_group():
    """`bench.py --gpus 2` with a launch timeout the child cannot meet: status 124, and promptly (the child session is
    terminated, not waited for)."""
    import time
    env = dict(os.environ, BBD_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch", "--launch-timeout", "1"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 124, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    assert "terminating its process group" in r.stderr and time.time() - t0 < 60
