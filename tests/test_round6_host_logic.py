"""CPU tier, round 6: the loader's frame cache under two collates at once and with a scratch area smaller than a batch
(ADVICE r5), `Trainer.canonicalize` refusing an inconsistent batch, and the reference's `--partial_skip` row rule
(trainer.py:331, 415-418) on the caller's order vs the canonical one."""
import threading

import pytest
import torch

import image_checks
from host_port import HostPortBackend
from baseboostdepth_amd import datasets, steptables
from baseboostdepth_amd.plan import get_plan


def _same(got, want, tag):
    assert set(got) == set(want), tag
    for k, v in want.items():
        if torch.is_tensor(v) and v.dim() > 0:
            assert torch.equal(got[k], v), (tag, k)
        else:
            assert (float(got[k]) == float(v)) if torch.is_tensor(v) else (got[k] == v), (tag, k)


def _loader(lines, root, cache, epoch, train=True, seed=3):
    H, W, scales = 64, 128, [0, 1]
    ds = datasets.KITTIRAWDataset(lines, epoch, H, W, kt_path=root, rand=train, is_train=train, scales=scales, kt=True,
                                  naive_mix=True, trimin=True, seed=seed)
    col = datasets.DeviceCollate(H, W, scales if train else [0], "cpu", HostPortBackend(), cache=cache)
    return datasets.DeviceLoader(ds, 4, col, shuffle=train, drop_last=train, num_workers=2, seed=1, workers="thread")


def test_two_loaders_share_one_frame_cache_concurrently(tmp_path):
    """A training loader and a validation loader over ONE `FrameCache`, iterated at the same time from two threads (what
    `run_epoch` does when it validates mid-epoch): each collate uploads into a scratch area of its own and slot assignment is
    locked, so every batch of both still equals the uncached loader's, and no frame got two resident homes."""
    lines = image_checks.make_kitti_tree(str(tmp_path), frames=20)[:24]
    want_train = list(_loader(lines, str(tmp_path), None, 0))
    want_val = list(_loader(lines, str(tmp_path), None, 0, train=False))
    cache = datasets.FrameCache("cpu", 160 << 20, scratch_bytes=48 << 20, regions=2)
    got, errors = {"train": [], "val": []}, []
    barrier = threading.Barrier(2)

    def drain(name, train):
        try:
            barrier.wait(timeout=60)
            for _ in range(2):                                  # twice: the second pass reads mostly resident frames
                got[name].append(list(_loader(lines, str(tmp_path), cache, 0, train=train)))
        except Exception as e:                                  # surfaced below
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=drain, args=("train", True)), threading.Thread(target=drain, args=("val", False))]
    [t.start() for t in threads]
    [t.join(timeout=300) for t in threads]
    assert not errors, errors
    for name, want in (("train", want_train), ("val", want_val)):
        assert len(got[name]) == 2
        for run in got[name]:
            assert len(run) == len(want)
            for i, (g, w) in enumerate(zip(run, want)):
                _same(g, w, (name, i))
    offsets = sorted(o for o, _, _ in cache.index.values())
    sizes = {o: h * w * 3 for o, h, w in cache.index.values()}
    assert len(set(offsets)) == len(offsets)                    # no slot handed out twice
    assert all(a + sizes[a] <= b for a, b in zip(offsets, offsets[1:])) and cache.used == offsets[-1] + sizes[offsets[-1]]
    import gc
    gc.collect()
    assert not cache._held                                      # the collates are gone: their scratch areas are free again
    with pytest.raises(RuntimeError):                           # a third live collate on a two-region cache is refused
        keep = [datasets.DeviceCollate(64, 128, [0], "cpu", HostPortBackend(), cache=cache) for _ in range(3)]
        del keep


@pytest.mark.parametrize("capacity_mb", [160, 4])
def test_a_scratch_area_smaller_than_the_batch_is_walked_in_spans(tmp_path, capacity_mb):
    """A cold (or full: 4 MB holds two frames) cache under the boosted recipe meets batches whose fresh frames exceed the
    scratch area: they go through it in several spans (upload, admit, resize per span) and give the uncached batches."""
    lines = image_checks.make_kitti_tree(str(tmp_path), frames=20)[:24]
    want = list(_loader(lines, str(tmp_path), None, 12))        # epoch 12: up to +-7 frames per sample
    frame = 375 * 1242 * 3
    cache = datasets.FrameCache("cpu", capacity_mb << 20, scratch_bytes=3 * frame + 1000)
    got = list(_loader(lines, str(tmp_path), cache, 12))
    assert len(got) == len(want) and cache.oversized > 0
    for i, (g, w) in enumerate(zip(got, want)):
        _same(g, w, i)
    with pytest.raises(RuntimeError):                           # an area smaller than ONE frame cannot work
        list(_loader(lines, str(tmp_path), datasets.FrameCache("cpu", 8 << 20, scratch_bytes=frame // 2), 12))


def test_canonicalize_refuses_a_batch_whose_frame_stacks_do_not_match_its_ordering():
    from fused_runner import bare_trainer
    import types
    from baseboostdepth_amd.synthetic import synthetic_batch
    tr = bare_trainer(types.SimpleNamespace(trimin=True, decomp=True), None, "cpu")
    batch = synthetic_batch([1, 3, 2], 16, 24, [0], device="cpu", seed=1)
    good = dict(batch)
    assert tr.canonicalize(good) == [1, 2, 0]
    bad = dict(batch)
    bad[("color", 2, 0)] = bad[("color", 2, 0)][:1]             # frame 2 has two owners (m = 3, 2)
    with pytest.raises(ValueError, match="owners"):
        tr.canonicalize(bad)
    assert bad["ordering"] == batch["ordering"]                 # nothing was permuted


def test_partial_skip_row_rule_on_the_callers_order_and_on_the_canonical_one():
    """The reference keeps the CHAINED translation of frame f for row numbers r with |f| == m_r - 2 of the non-stereo
    sample list (trainer.py:331, 417) - rows of f's stack belong to f's OWNERS, so on a shuffled batch the decision meets
    mismatched samples.  `steptables.PoseSchedule` applies the rule to whatever order it is given: the caller's order gives
    the reference's decisions on that order (pinned by the pose_incr_partial fixture, ordering 4, 3, 2, 7); the canonical
    order (owners of f = a prefix) makes the rule meet each sample's own m."""
    from baseboostdepth_amd import _lib

    def kept(ms):
        plan = get_plan([[0, m, -m] for m in ms], True, True)
        M = max(ms)
        fid = sorted(range(-M, M + 1), key=abs)
        sched = steptables.PoseSchedule(plan, fid, True, True, True, 1 << 30)
        rows, views, _ = sched.compose
        out = {}
        for okey, o0, n, _ in views:
            if okey[0] != "cam_T_cam":
                continue
            f = okey[2]
            for j, b in enumerate(plan.owners(f)):
                out[(b, f)] = not (rows[o0 + j][2] & _lib.COMPOSE_REPLACE)      # chained translation kept
        return out, plan

    caller = [4, 3, 2, 7]
    got, plan = kept(caller)
    nonstereo = [m for m in caller if m != 0]
    for (b, f), keep in got.items():
        if f in plan.valid_frames and abs(f) > 1:
            r = plan.owners(f).index(b)                         # the reference indexes its all-sample list by ROW number
            assert keep == (abs(f) == nonstereo[r] - 2), (b, f)
    canon = sorted(caller, reverse=True)
    got_c, plan_c = kept(canon)
    for (b, f), keep in got_c.items():
        if f in plan_c.valid_frames and abs(f) > 1:
            assert keep == (abs(f) == canon[b] - 2), (b, f)      # each sample's own m
    # the two orders do decide differently for some (sample, frame): the documented deviation of the canonical order
    by_m = lambda table, ms: {(ms[b], f): k for (b, f), k in table.items()}
    assert by_m(got, caller) != by_m(got_c, canon)
