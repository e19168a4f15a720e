"""Loader in index-table form (baseboostdepth_amd/datasets.py) on the CPU tier: recipes -> DeviceCollate
through the host port == the reference's per-item Pillow pipeline + custom_collate; frame-set selection
against the oracle restatement and the split's published statistics."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import image_checks  # noqa: E402
from host_port import HostPortBackend  # noqa: E402
from oracle import loader_ref  # noqa: E402
from baseboostdepth_amd import datasets  # noqa: E402


@pytest.mark.parametrize("epoch,trimin,scales", [(3, True, [0, 1, 2, 3]), (12, True, [0]), (0, False, [0, 1, 2, 3])])
def test_collated_batches_equal_reference_pipeline(tmp_path, epoch, trimin, scales):
    image_checks.check_loader_batches(tmp_path, "cpu", HostPortBackend(), epoch, trimin, scales)


def test_frame_selection_matches_oracle_and_draw_order():
    rng = np.random.default_rng(0)
    lines = ["d/x %d %s kt %.6f" % (100 + i, "lr"[i % 2], rng.uniform(0.01, 0.7)) for i in range(400)]
    for epoch, trimin in [(0, True), (5, True), (9, False), (10, True), (15, True), (19, False)]:
        ds = datasets.KITTIRAWDataset(lines, epoch, 192, 640, kt_path="/nowhere", rand=True, is_train=True,
                                      kt=True, naive_mix=True, trimin=trimin, seed=9)
        for i in range(len(lines)):
            present = {o for o in range(-7, 8) if (o * 7 + i) % 11 != 0 or abs(o) == 1}
            frame_index = 100 + i

            def exists(path, fi=frame_index, pr=present):
                return (int(os.path.basename(path).split(".")[0]) - fi) in pr
            got = ds.select_frames(i, ds._rngs(i)[0], exists=exists)
            want = loader_ref.select_frames_ref(lines[i], epoch, trimin, True, True, ds._rngs(i)[0], lambda o: o in present)
            assert (got[0], got[1], got[5]) == want, (epoch, i)
            m = max(f for f in got[5] if f != "s")
            assert got[5][0] == 0 and (("s" in got[5]) == (m < 3))


def test_stereo_sign_and_eval_items(tmp_path):
    lines = image_checks.make_kitti_tree(str(tmp_path), frames=18)
    ds = datasets.KITTIRAWDataset(lines, 0, 64, 128, kt_path=str(tmp_path), rand=True, is_train=True, kt=True,
                                  naive_mix=True, trimin=True, seed=1)
    for i in range(0, len(lines), 3):
        item = ds[i]
        side = lines[i].split()[2]
        assert item["stereo_T"][0, 3] == np.float32((-1 if side == "l" else 1) * (-1 if item["flip"] else 1) * 0.1)
        assert set(item["images"]) == set(item["frame_idxs"])
        assert all(f != "s" for f in item["jitter"])
    val = datasets.KITTIRAWDataset([l.rsplit(" ", 2)[0] for l in lines], 0, 64, 128, kt_path=str(tmp_path),
                                   is_train=False, kt=True, naive_mix=True)
    item = val[0]
    assert list(item["images"]) == [0] and not item["flip"] and not item["jitter"]
    batch = datasets.DeviceCollate(64, 128, [0], "cpu", HostPortBackend())([val[0], val[1]])
    assert batch[("color", 0, 0)].shape == (2, 3, 64, 128) and torch.equal(batch[("color", 0, 0)], batch[("color_aug", 0, 0)])


def test_worker_processes_with_shared_ring_equal_thread_pool(tmp_path):
    """Process workers decode into the shared-memory ring; batches equal the in-process thread pool's."""
    lines = image_checks.make_kitti_tree(str(tmp_path), frames=20)
    ds = datasets.KITTIRAWDataset(lines, 2, 64, 128, kt_path=str(tmp_path), rand=True, is_train=True, scales=[0, 1, 2, 3],
                                  kt=True, naive_mix=True, trimin=True, seed=3)
    col = datasets.DeviceCollate(64, 128, [0, 1, 2, 3], "cpu", HostPortBackend())
    a = list(datasets.DeviceLoader(ds, 4, col, num_workers=2, seed=1, workers="process"))
    b = list(datasets.DeviceLoader(ds, 4, col, num_workers=2, seed=1, workers="thread"))
    assert len(a) == len(b) == len(lines) // 4
    for x, y in zip(a, b):
        assert set(x) == set(y)
        for k in x:
            if torch.is_tensor(x[k]) and x[k].dim() > 0:
                assert torch.equal(x[k], y[k]), k
            elif not torch.is_tensor(x[k]):
                assert x[k] == y[k], k
    assert datasets._RING is None                      # ring released at the end of the epoch


@pytest.mark.parametrize("capacity_mb", [160, 16])
def test_frame_cache_batches_equal_uncached_ones_and_decode_each_frame_once(tmp_path, capacity_mb):
    """`datasets.FrameCache`: decoded frames stay resident (HBM on the GPU, host memory in this tier), a frame is decoded
    once and every later use is a table entry.  Two epochs over a small tree: every batch equals the uncached loader's bit
    for bit, the second epoch decodes (almost) nothing new when the cache holds the tree (160 MB), and a cache too small for it
    (16 MB: eleven frames) uses the rest from its scratch area - same batches."""
    from baseboostdepth_amd import datasets
    lines = image_checks.make_kitti_tree(str(tmp_path), frames=20)[:24]
    H, W, scales = 64, 128, [0, 1]

    def loader(cache, epoch):
        ds = datasets.KITTIRAWDataset(lines, epoch, H, W, kt_path=str(tmp_path), rand=True, is_train=True, scales=scales, kt=True,
                                      naive_mix=True, trimin=True, seed=3)
        col = datasets.DeviceCollate(H, W, scales, "cpu", HostPortBackend(), cache=cache)
        return datasets.DeviceLoader(ds, 4, col, shuffle=True, drop_last=True, num_workers=2, seed=1, workers="thread")

    cache = datasets.FrameCache("cpu", capacity_mb << 20, scratch_bytes=64 << 20)
    decoded = []
    for epoch in (0, 1):
        before = cache.misses
        for got, want in zip(loader(cache, epoch), loader(None, epoch)):
            assert set(got) == set(want)
            for k, v in want.items():
                if torch.is_tensor(v) and v.dim() > 0:
                    assert torch.equal(got[k], v), (epoch, k)
                else:
                    assert (float(got[k]) == float(v)) if torch.is_tensor(v) else (got[k] == v), k
        decoded.append(cache.misses - before)
    st = cache.stats()
    if capacity_mb == 160:
        assert st["passed_through"] == 0 and st["frames"] <= 80          # 20 frames x 2 cameras x 2 drives on disk
        assert decoded[0] >= st["frames"] and decoded[1] <= 0.2 * decoded[0], decoded      # (epoch 1 draws other frame sets)
        assert st["hits"] >= st["misses"] > 0
    else:
        assert 8 <= st["frames"] <= 12 and st["passed_through"] > 0 and st["hits"] > 0      # (drive 1 frames are 375 x 1242 x 3)
