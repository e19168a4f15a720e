"""CPU tier: the host side of the pooled step (`baseboostdepth_amd/pooled.py`) - the fixed-offset table buffer a batch
signature is packed into.  The tables are EMULATED here with plain torch indexing (pool gathers, the composition table walked
row by row, the pose-table gather) and must select exactly what the per-signature path (`Trainer.predict_poses` /
`_job_poses`, i.e. reference trainer.py:310-419, 444-475) selects: the same image pairs for every pose-network call in the
same order, the same matrix for every pose-table row, the same source image for every candidate and identity item."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from fused_runner import bare_trainer
from baseboostdepth_amd import _lib, ops, pooled, tuning
from baseboostdepth_amd.layers import transformation_from_parameters
from baseboostdepth_amd.plan import STEREO, frame_slot
from baseboostdepth_amd.synthetic import synthetic_batch

H, W = 16, 24


class _RowEncoder(nn.Module):
    """Row-wise stand-in for the pose encoder: a row's features depend on that row only (so call batching cannot matter)."""
    num_ch_enc = np.array([6])

    def forward(self, x):
        return [x.mean(dim=(2, 3), keepdim=True) + 0.1 * x[:, :, :1, :1]]


class _RowDecoder(nn.Module):
    def forward(self, feats):
        f = feats[0][0][:, :, 0, 0]                         # [n, 6]
        aa = 0.3 * torch.stack([f[:, :3], f[:, 3:]], 1).unsqueeze(2)
        tt = 0.5 * torch.stack([f[:, 3:], f[:, :3]], 1).unsqueeze(2)
        return aa - 0.1, tt - 0.2


def _opt(B, boosted=True):
    import types
    return types.SimpleNamespace(height=H, width=W, batch_size=B, scales=[0], frame_ids=[0, -1, 1], min_depth=0.1,
                                 max_depth=100.0, disparity_smoothness=1e-3, no_ssim=False, trimin=boosted, decomp=boosted,
                                 pose_error=5.5, incremental_skip=boosted, partial_skip=boosted, materialize_warps=False,
                                 batched_pose=False)


def _trainer(B, cutt, boosted=True):
    tr = bare_trainer(_opt(B, boosted), None, "cpu")
    tr.models = {"pose_encoder": _RowEncoder(), "pose": _RowDecoder()}
    tr.pose_pad_rows = 32
    return tr


def _emulate(tab, layout, caps, inputs, models, pose_error):
    hb = tab.host.numpy()
    sec = lambda name: layout.view(hb, name)
    pool_c = torch.zeros(caps.F, 3, H, W)
    pool_a = torch.zeros(caps.F, 3, H, W)
    for f, (at, n) in tab.frame_rows.items():
        pool_c[at:at + n] = inputs[("color", f, 0)]
        if f != STEREO:
            pool_a[at:at + n] = inputs[("color_aug", f, 0)]
    R = tab.R
    ia, ib, inv = (torch.from_numpy(sec(k)[:R].astype(np.int64)) for k in ("idx_a", "idx_b", "invert"))
    A, Bp = pool_a[ia], pool_a[ib]
    aa, tt = models["pose"]([models["pose_encoder"](torch.cat([A, Bp], 1))])
    plain = transformation_from_parameters(aa[:, 0], tt[:, 0], invert=False)
    flipped = transformation_from_parameters(aa[:, 0], tt[:, 0], invert=True)
    M = torch.where(inv.view(-1, 1, 1) > 0, flipped, plain)
    ctab = sec("compose_tab")
    out = []
    for o in range(caps.NO):
        row = ctab[o]
        T = torch.eye(4)
        for k in range(int(row[0])):
            T = T @ M[int(row[1 + k])]
        T = T.clone()
        if row[9] & _lib.COMPOSE_REPLACE:
            T[:, 3] = M[int(row[8])][:, 3]
        if row[9] & _lib.COMPOSE_ERROR:
            T[:3, 3] = T[:3, 3] / pose_error
        out.append(T)
    out = torch.stack(out)
    src = torch.cat([inputs["stereo_T"], out, M], 0)
    return pool_c, pool_a, A, Bp, M, out, src


CASES = [
    ([7, 5, 4, 3], 1.35), ([6, 6, 2, 1], 1.35), ([7, 7, 7, 1], 1.35), ([3, 1, 2, 5], 1.35), ([1, 1, 1, 1], 1.35),
    ([2, 1, 1, 0], 0.3), ([2, 2, 2, 2], 0.3), ([0, 0, 1, 0], 0.3), ([0, 0, 0, 0], 0.3), ([1, 2, 0, 2], 0.3),
]


@pytest.mark.parametrize("ms,cutt", CASES)
def test_pooled_tables_select_what_the_per_signature_path_selects(ms, cutt):
    B = len(ms)
    inputs = synthetic_batch(ms, H, W, [0], device="cpu", seed=11)
    inputs["cutt"] = torch.tensor(cutt)
    inputs["stereo_T"] = inputs["stereo_T"] + 0.01 * torch.arange(B).view(B, 1, 1)      # rows distinguishable
    tr = _trainer(B, cutt)
    tr.opt.frame_ids = sorted(inputs["frames"], key=lambda it: float("inf") if isinstance(it, str) else abs(it))
    tr.valid_frames_trimin(inputs)
    plan = tr.plan
    # ---- the per-signature path (torch loops on the CPU)
    outputs = tr.predict_poses(inputs)
    sched = tr.tables.schedule
    job = tr._job_poses(inputs, outputs)
    want_T = torch.cat([job[j] for j in plan.pose_jobs], 0) if plan.pose_jobs else torch.zeros(0, 4, 4)
    # ---- the pooled tables of the same batch
    caps = pooled.Caps(B)
    layout = pooled.Layout(caps)
    maxing = cutt > 0.5
    tab = pooled.PooledTables(plan, tuple(inputs["frames"]), tr.opt.frame_ids, maxing, maxing, True, maxing, caps, layout, 32,
                              tuning.padded_pose_rows(4 * B, 32), False)
    sec = lambda name: layout.view(tab.host.numpy(), name)
    pool_c, pool_a, A, Bp, M, out, src = _emulate(tab, layout, caps, inputs, tr.models, tr.opt.pose_error)
    # the pairs of every pose-network call, in call order; the padding pairs are zero images
    at = 0
    for _, (fa, ra), (fb, rb), invert, n in sched.requests:
        first = inputs[("color_aug", fa, 0)] if ra is None else inputs[("color_aug", fa, 0)][list(ra)]
        second = inputs[("color_aug", fb, 0)] if rb is None else inputs[("color_aug", fb, 0)][list(rb)]
        assert torch.equal(A[at:at + n], first) and torch.equal(Bp[at:at + n], second)
        assert (sec("invert")[at:at + n] == int(invert)).all()
        at += n
    assert at == tab.n_real <= tab.R and not A[at:].any() and not Bp[at:].any()
    # call groups: the real calls, then padding groups of at most `bound` rows, then empty ones; tracked = real calls
    g = sec("groups")
    sizes = np.diff(g[:ops.BN_MAX_GROUPS + 1])
    assert g[0] == 0 and g[ops.BN_MAX_GROUPS] == tab.R and (sizes >= 0).all() and sizes.max(initial=0) <= tab.bound
    assert list(sizes[:len(sched.rows)]) == list(sched.rows) and g[ops.BN_MAX_GROUPS + 1] == len(sched.rows)
    assert int((sizes > 0).sum()) <= tab.G and not sizes[tab.G:].any()
    # every pose-table row gets the matrix the per-signature path hands to it
    tsel = torch.from_numpy(sec("tsel")[:plan.NP].astype(np.int64))
    got_T = src[tsel]
    assert got_T.shape == want_T.shape
    assert torch.allclose(got_T, want_T, atol=1e-6, rtol=1e-6), float((got_T - want_T).abs().max())
    assert (sec("k_rows")[:plan.NP] == plan.k_rows).all()
    # the reference's pose keys as row ranges of the two pose buffers
    views = {k: (M if buf == "M" else out)[o0:o0 + n] for k, buf, o0, n, _ in tab.pose_views}
    assert set(views) == set(outputs)
    for k, v in outputs.items():
        assert torch.allclose(views[k], v, atol=1e-6, rtol=1e-6), k
    # candidates / identity items address the pool by row
    cand, ncand = sec("cand"), sec("ncand")
    assert (ncand == plan.ncand_np).all()
    for b in range(B):
        for k in range(int(ncand[b])):
            kind, slot, row, pose = plan.cand_np[b, k]
            if (kind & 0xff) == _lib.KIND_WARP:
                f = STEREO if slot == frame_slot(STEREO) else slot - 7
                assert cand[b, k, 1] == 0 and torch.equal(pool_c[int(cand[b, k, 2])], inputs[("color", f, 0)][row])
                assert cand[b, k, 0] == kind and cand[b, k, 3] == pose
            else:
                assert (cand[b, k] == plan.cand_np[b, k]).all()
    items = sec("items")
    for i, (b, slot, row, _) in enumerate(plan.ident_items):
        f = STEREO if slot == frame_slot(STEREO) else slot - 7
        assert items[i, 0] == b and items[i, 1] == 0 and torch.equal(pool_c[int(items[i, 2])], inputs[("color", f, 0)][row])
    assert list(sec("ident_off")) == list(plan.ident_off)
    # composition table: no-op rows are constant, the inverse table stays inside the real rows
    ctab, coff = sec("compose_tab"), sec("compose_off")
    n_out = sum(n for _, buf, _, n, _ in tab.pose_views if buf == "out")
    assert (ctab[n_out:, 0] == 0).all() and (ctab[n_out:, 9] == _lib.COMPOSE_ERROR).all()
    assert (np.diff(coff[:tab.R + 1]) >= 0).all() and (coff[tab.n_real:tab.R + 1] == coff[tab.n_real]).all()


def test_row_count_buckets():
    """Epoch >= 10: the pass is rounded up to `tuning.padded_pose_rows`; the early curriculum has ONE row count."""
    B = 12
    caps = pooled.Caps(B)
    layout = pooled.Layout(caps)
    from baseboostdepth_amd.plan import get_plan
    seen = set()
    import random
    rnd = random.Random(3)
    for _ in range(40):
        ms = sorted(rnd.choices(range(0, 3), [.062, .573, .366], k=B), reverse=True)
        frames = list(range(-max(ms), max(ms) + 1)) if max(ms) else [0]
        frames.append(STEREO)
        fid = sorted(frames, key=lambda f: float("inf") if f == STEREO else abs(f))
        plan = get_plan([[0, STEREO] if m == 0 else [0, m, -m] for m in ms], True, True)
        tab = pooled.PooledTables(plan, frames, fid, False, False, True, False, caps, layout, 32, 48, False)
        seen.add((tab.R, tab.G, tab.bound))
    assert seen == {(48, 8, 12)}
    seen = set()
    for _ in range(60):
        ms = sorted(rnd.choices(range(1, 8), [.050, .050, .077, .094, .139, .142, .448], k=B), reverse=True)
        frames = list(range(-max(ms), max(ms) + 1))
        if min(ms) < 3:
            frames.append(STEREO)
        fid = sorted(frames, key=lambda f: float("inf") if f == STEREO else abs(f))
        plan = get_plan([[0, m, -m] for m in ms], True, True)
        tab = pooled.PooledTables(plan, frames, fid, True, True, True, True, caps, layout, 32, 48, False)
        assert tab.R == tuning.padded_pose_rows(tab.n_real, 32) and tab.R in tuning.POSE_ROW_COUNTS
        seen.add((tab.R, tab.G, tab.bound))
    assert len(seen) <= len(tuning.POSE_ROW_COUNTS) and {b for _, _, b in seen} == {12}


def test_prewarm_orderings_cover_the_buckets_the_loader_draws_meet():
    """`PooledStep.bucket_orderings` (what `Trainer.prewarm` captures): the early curriculum's row counts 24 .. 48 one
    ordering each, largest first; from epoch 10 on every bucket that 300 draws from the epoch-10 / -15 / -19 offset
    distributions land in is among the prewarmed ones (a bucket it misses would be captured on first sight)."""
    import random
    import types
    from baseboostdepth_amd import steptables
    from baseboostdepth_amd.plan import ReprojectionPlan
    tr = types.SimpleNamespace(opt=types.SimpleNamespace(batch_size=12, height=192, width=640, trimin=True, decomp=True,
                                                         incremental_skip=True, partial_skip=True),
                               device=torch.device("cpu"), pose_pad_rows=32)
    ps = pooled.PooledStep(tr)
    early = ps.bucket_orderings(True)
    rows = [tuning.padded_pose_rows(sum(2 * m for m in ms), 32) for ms in early]
    assert rows == [48, 44, 40, 36, 32, 28, 24]

    def bucket(ms):
        ms = sorted(ms, reverse=True)
        plan = ReprojectionPlan([[0, m, -m] for m in ms], True, True)
        sched = steptables.PoseSchedule(plan, sorted(range(-max(ms), max(ms) + 1), key=abs), True, True, True, 1 << 30)
        R = tuning.padded_pose_rows(sched.total_rows, 32)
        groups = len(sched.rows) + -(-(R - sched.total_rows) // 12)
        return R, (8 if groups <= 8 else 32)
    late = ps.bucket_orderings(False)
    have = {bucket(ms) for ms in late}
    assert [bucket(ms)[0] for ms in late] == sorted((bucket(ms)[0] for ms in late), reverse=True)      # largest pass first
    rnd = random.Random(99)
    missed = 0
    for w in ([.108, .287, .277, .135, .068, .040, .084], [.050, .050, .077, .094, .139, .142, .448],
              [.050, .050, .050, .059, .070, .078, .644]):
        for _ in range(100):
            missed += bucket(rnd.choices(range(1, 8), w, k=12)) not in have
    assert missed <= 3, missed
