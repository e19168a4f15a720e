"""CPU tier, round 5: the host logic of the fresh-ordering regime (step tables, canonical order, pose schedule),
`Trainer.argmin_masks` against the reference's rule on the golden arg-min maps, and the loader's canonical collate."""
import numpy as np
import pytest
import torch

from golden_io import Case, DIRECT_CASES

# which arg-min ids the reference counts as 'norm' / 'guide' per sample group m (trainer.py x_min_opt): decomp branch
# :987-990 (s), :1002-1003 (m = 1), :1021-1022 (m = 2), :1044-1045 (m >= 3); non-decomp branch :1054, :1065, :1080, :1097
REFERENCE_RULE = {
    True: {0: ([0], []), 1: ([0, 1, 2], [3, 4]), 2: ([0, 1, 2, 3, 4], [5, 6, 7, 8]), 3: (list(range(6)), list(range(6, 12)))},
    False: {0: ([0], []), 1: ([0, 1, 2], []), 2: ([0, 1, 2, 3, 4], []), 3: (list(range(6)), [])},
}


@pytest.mark.parametrize("name", [n for n in DIRECT_CASES if n.startswith("tri_")])
def test_argmin_masks_follow_the_reference_rule_on_the_golden_argmin(name):
    """`Trainer.argmin_masks` (north_star's "argmin masks") on the reference's own arg-min ids (fixture `out/argmin/s`)
    vs masks written out with the reference's per-group id lists; both return shapes."""
    from fused_runner import bare_trainer, make_opt
    case = Case(name)
    tr = bare_trainer(make_opt(case), None, "cpu")
    tr.valid_frames_trimin(case.inputs)
    arg = torch.stack([case.expected("out/argmin/%d" % s) for s in case.scales])
    outputs = {("bbd", "argmin"): arg}
    for i, s in enumerate(case.scales):
        norm, guide = tr.argmin_masks(outputs, i)
        dn, dg = tr.argmin_masks(outputs, i, reference_dicts=True)
        assert norm.dtype == torch.bool and norm.shape == (case.B, case.H, case.W)
        for b, m in enumerate(case.ms):
            ids_n, ids_g = REFERENCE_RULE[case.decomp][min(m, 3)]
            want_n = sum((arg[i, b] == k) for k in ids_n).bool()
            want_g = sum((arg[i, b] == k) for k in ids_g).bool() if ids_g else torch.zeros_like(want_n)
            assert torch.equal(norm[b], want_n), (name, s, b)
            assert torch.equal(guide[b], want_g), (name, s, b)
        # the reference's dict shape: one [n_group,H,W] tensor per group key; guide only under decomp, never for 's'
        assert set(dn) == {("s" if m == 0 else m, "norm") for m in case.ms}
        for (m, _), tensors in dn.items():
            rows = [b for b, mb in enumerate(case.ms) if mb == (0 if m == "s" else m)]
            assert len(tensors) == 1 and torch.equal(tensors[0], norm[rows])
        for (m, _), tensors in dg.items():
            if case.decomp and m != "s":
                rows = [b for b, mb in enumerate(case.ms) if mb == m]
                assert len(tensors) == 1 and torch.equal(tensors[0], guide[rows])
            else:
                assert tensors == []
        # every pixel is won by exactly one kind: norm, guide or an identity map
        assert not (norm & guide).any()


def test_lru_evicts_the_least_recently_used_entry():
    from baseboostdepth_amd.steptables import LRU
    c = LRU(2)
    c.put("a", 1), c.put("b", 2)
    assert c.get("a") == 1                 # refreshes a
    c.put("c", 3)
    assert "b" not in c and "a" in c and "c" in c and len(c) == 2


def test_packer_lays_sections_out_on_256_byte_boundaries_and_views_match():
    from baseboostdepth_amd import steptables as st
    pk = st.Packer()
    a = np.arange(7, dtype=np.int32)
    b = np.arange(130, dtype=np.int32).reshape(65, 2) + 100
    pk.add("a", a, a.shape)
    pk.add("b", b, b.shape)
    pk.reserve("w", (3, 2))
    st.reset_stats()
    views = pk.upload("cpu", fill=lambda ptr: np.ctypeslib.as_array((np.ctypeslib.ctypes.c_int32 * 6).from_address(ptr("w")))
                      .__setitem__(slice(None), np.arange(6) + 900))
    assert st.STATS["packed_uploads"] == 1
    offs = {name: off for name, off, _, _ in pk.parts}
    assert all(o % st.ALIGN_WORDS == 0 for o in offs.values()) and offs["b"] == 64 and offs["w"] == 64 + 192
    assert torch.equal(views["a"], torch.from_numpy(a)) and torch.equal(views["b"], torch.from_numpy(b))
    assert views["w"].flatten().tolist() == [900, 901, 902, 903, 904, 905]
    assert views["a"].data_ptr() == views["_buffer"].data_ptr()        # views of ONE allocation


@pytest.mark.parametrize("ms,cutt", [([7, 5, 4, 3], 1.35), ([6, 6, 2, 1], 1.35), ([2, 1, 1, 0], 0.3), ([1, 1, 1, 1], 0.3)])
def test_step_tables_announce_every_row_list_the_step_asks_for(ms, cutt):
    """A whole training-mode `predict_poses` + `_job_poses` on the CPU asks `_index` only for row lists that are part of
    the step's single table pack (no single uploads), and a second batch with the same signature builds nothing."""
    import types
    from baseboostdepth_amd import steptables as st
    from baseboostdepth_amd.synthetic import synthetic_batch
    from baseboostdepth_amd.trainer import Trainer
    H, W = 32, 64
    inputs = synthetic_batch(ms, H, W, [0], device="cpu", seed=1)
    inputs["cutt"] = torch.tensor(cutt)
    tr = Trainer.__new__(Trainer)
    tr.opt = types.SimpleNamespace(height=H, width=W, scales=[0], trimin=True, decomp=True, incremental_skip=True,
                                   partial_skip=True, pose_error=5.5,
                                   frame_ids=sorted(inputs["frames"], key=lambda f: 99 if f == "s" else abs(f)))
    tr.device = torch.device("cpu")
    st._STEP_CACHE.clear()
    st.reset_stats()
    tr.valid_frames_trimin(inputs)
    tables = tr._step_tables(inputs)
    assert st.STATS["packed_uploads"] == 1 and st.STATS["builds"] == 1
    sched = tables.schedule
    for _, a, b, _, n in sched.requests:               # every announced selection resolves to a view of the pack
        for f, rows in (a, b):
            t = inputs["color_aug", f, 0]
            got = tr._rows(t, rows)
            assert got.shape[0] == n
            if rows is not None:
                assert torch.equal(got, t[list(rows)])
    outputs = {}
    for f in tr.plan.frames:
        if f != "s":
            n = len(tr.plan.owners(f)) if sched.incremental else len(tr.plan.jobs[f])
            outputs[("cam_T_cam", 0, f)] = torch.eye(4).repeat(n, 1, 1)
            outputs[("cam_T_cam_error", 0, f)] = torch.eye(4).repeat(n, 1, 1)
    tr._job_poses(inputs, outputs)
    assert st.STATS["single_uploads"] == 0, st.STATS
    # the same signature again: LRU hit
    tr.valid_frames_trimin(inputs)
    assert tr._step_tables(inputs) is tables and st.STATS["builds"] == 1
    # the invert flags of the batched pass are those of the requests, row by row
    flat = [int(inv) for _, _, _, inv, n in sched.requests for _ in range(n)]
    assert sum((c[3] for c in sched.chunks), []) == flat
    assert torch.cat(tables.invert).tolist() == flat


def test_canonical_permutation_is_stable_and_sorts_candidate_counts():
    from baseboostdepth_amd.plan import canonical_permutation, get_plan
    ms = [1, 3, 0, 2, 5, 2, 0]
    perm = canonical_permutation(ms)
    assert perm == [4, 1, 3, 5, 0, 2, 6]
    plan = get_plan([[0, "s"] if ms[p] == 0 else [0, ms[p], -ms[p]] for p in perm], True, True)
    assert plan.sample_order is None            # most candidates first already: the shared work-order table applies
    counts = plan.ncand_np.tolist()
    assert counts == sorted(counts, reverse=True)
    # early curriculum: 3^12 orderings, 91 signatures
    import itertools
    sigs = {tuple(sorted(c, reverse=True)) for c in itertools.product(range(3), repeat=6)}
    assert len(sigs) == 28                      # multisets of 6 from 3 = C(8,2); for batch 12: C(14,2) = 91


def test_canonicalize_permutes_every_tensor_by_its_own_rows():
    from baseboostdepth_amd.plan import owners_of
    from baseboostdepth_amd.synthetic import synthetic_batch
    from baseboostdepth_amd.trainer import Trainer
    ms = [1, 7, 0, 3, 2, 7]
    inp = synthetic_batch(ms, 32, 64, (0, 1), seed=1)
    ref = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in inp.items()}
    tr = Trainer.__new__(Trainer)
    perm = tr.canonicalize(inp)
    assert perm == [1, 5, 3, 4, 0, 2] and inp["batch_order"] == perm
    new_ms = [ms[p] for p in perm]
    for f in inp["frames"]:
        old, new = owners_of(ms, f), owners_of(new_ms, f)
        for i, b in enumerate(new):
            assert torch.equal(inp[("color", f, 0)][i], ref[("color", f, 0)][old.index(perm[b])]), f
    for b, p in enumerate(perm):
        assert torch.equal(inp[("color", 0, 1)][b], ref[("color", 0, 1)][p])
        assert torch.equal(inp["noise"][b], ref["noise"][p]) and torch.equal(inp["stereo_T"][b], ref["stereo_T"][p])
    assert inp["ordering"] == [ref["ordering"][p] for p in perm]
    assert float(inp["cutt"]) == float(ref["cutt"])
    # a second call composes: the batch is canonical now
    assert tr.canonicalize(inp) is None


def test_pose_multipliers_follow_the_trainer_pose_modes():
    """`synthetic.pose_multipliers` runs Trainer.predict_poses with unit-translation networks: chained steps add up,
    negative offsets beyond -1 keep the identity pose in incremental mode (trainer.py:364), the partial swap leaves the
    chain only where |f| = m - 2 (trainer.py:415-418; canonical order: row = sample)."""
    from baseboostdepth_amd.synthetic import pose_multipliers
    ms = [7, 5, 3, 1]
    frames = sorted(list(range(-7, 8)), key=abs)
    k = pose_multipliers(ms, frames)
    assert k[(0, 5)] == 5.0 and k[(0, -5)] == 0.0                 # |f| = m - 2: chain kept (empty for negative offsets)
    assert k[(0, 7)] == 1.0 and k[(0, -7)] == -1.0                # swapped for the direct call's translation
    assert k[(1, 3)] == 3.0 and k[(1, -3)] == 0.0 and k[(1, 5)] == 1.0
    assert k[(2, 1)] == 1.0 and k[(2, -1)] == -1.0 and k[(3, 1)] == 1.0
    plain = pose_multipliers([2, 1], [0, 1, -1, 2, -2, "s"], incremental=False, partial=False, cutt=0.3)
    assert plain == {(0, 1): 1.0, (1, 1): 1.0, (0, -1): -1.0, (1, -1): -1.0, (0, 2): 1.0, (0, -2): -1.0}


def test_structured_batch_registers_the_true_pose_warps():
    """On `synthetic.structured_batch` with the scene's disparity and the poses the trainer would use, the true-pose
    warps of the host port register (mean minimum loss far below the random-texture level) and the arg-min map is
    coherent: most backward tiles see a handful of live candidates, true-pose candidates win most pixels."""
    import types
    import torch.nn as nn
    from host_port import HostPortBackend
    from baseboostdepth_amd.synthetic import (STRUCT_DISP, live_candidates_per_tile, structured_batch, structured_translation,
                                              synthetic_batch)
    from baseboostdepth_amd.trainer import Trainer
    H, W, ms = 64, 192, [7, 5, 3, 2]
    tx = structured_translation(H, W)

    class _Enc(nn.Module):
        def forward(self, x):
            return [x.new_zeros(x.shape[0], 1, 1, 1)]

    class _Dec(nn.Module):
        def forward(self, feats):
            n = feats[0][0].shape[0]
            t = torch.zeros(n, 2, 1, 3)
            t[:, 0, 0, 0] = tx
            return torch.zeros(n, 2, 1, 3), t

    def run(inputs):
        tr = Trainer.__new__(Trainer)
        tr.opt = types.SimpleNamespace(height=H, width=W, scales=[0], trimin=True, decomp=True, pose_error=5.5, incremental_skip=True,
                                       partial_skip=True, batched_pose=False, min_depth=0.1, max_depth=100.0, no_ssim=False,
                                       disparity_smoothness=1e-3, materialize_warps=False,
                                       frame_ids=sorted(inputs["frames"], key=lambda f: 99 if f == "s" else abs(f)))
        tr.device, tr.backend, tr.num_scales = torch.device("cpu"), HostPortBackend(), 4
        tr.models = {"pose_encoder": _Enc(), "pose": _Dec()}
        tr.valid_frames_trimin(inputs)
        out = tr.predict_poses(inputs)
        out[("disp", 0)] = torch.full((len(ms), 1, H, W), STRUCT_DISP)
        out.update(tr.generate_images_pred(inputs, out))
        return tr, out

    tr, out = run(structured_batch(ms, H, W, (0,), seed=3))
    arg = out[("bbd", "argmin")][0]
    hist = live_candidates_per_tile(arg)
    mean_live = sum(k * v for k, v in hist.items()) / sum(hist.values())
    n_true = sum(int((arg[b] < sum(1 for kind, _ in tr.plan.cand_names[b] if kind == "T")).sum()) for b in range(len(ms)))
    assert n_true >= 0.8 * arg.numel()
    coherent_loss = float(out[("bbd", "to_optimise")][0].mean())
    rand = synthetic_batch(ms, H, W, (0,), seed=3)
    rand["cutt"] = torch.tensor(1.35)
    _, out_r = run(rand)
    assert coherent_loss < 0.5 * float(out_r[("bbd", "to_optimise")][0].mean())
    hist_r = live_candidates_per_tile(out_r[("bbd", "argmin")][0])
    mean_live_r = sum(k * v for k, v in hist_r.items()) / sum(hist_r.values())
    assert mean_live < 0.7 * mean_live_r and mean_live < 8, (hist, hist_r)


def test_pose_pass_is_padded_to_measured_row_counts():
    """`tuning.padded_pose_rows`: the next row count with shipped find results when it is no further than the next multiple of
    32, else that multiple; every epoch-15 row count of the boosted recipe (24 + 4 k) lands on a measured one with at most
    15 padding rows."""
    from baseboostdepth_amd import tuning
    for n in (1, 9, 24, 33, 48, 50, 100, 129, 180, 292, 330, 400):
        q = -(-n // 32) * 32
        want = next((r for r in tuning.POSE_ROW_COUNTS if n <= r <= q), q)      # (the table is sorted)
        assert tuning.padded_pose_rows(n) == want and want >= n and want - n < 32, (n, want)
    assert list(tuning.POSE_ROW_COUNTS) == sorted(tuning.POSE_ROW_COUNTS)
    assert [tuning.padded_pose_rows(n) for n in (48, 292, 330)] == [48, 320, 352]
    for k in range(39, 67):                      # 180 .. 288 rows: 99 % of the epoch-15 draws
        n = 24 + 4 * k
        r = tuning.padded_pose_rows(n)
        assert r in tuning.POSE_ROW_COUNTS and 0 <= r - n <= 15, (n, r)
    # every shipped row count has find results in the shipped database (forward problems of the pose encoder's stem:
    # 6 -> 64 channels, 7x7 stride 2 on 192x640, batch = the row count)
    import glob, os
    db = glob.glob(os.path.join(os.path.dirname(tuning.__file__), "miopen_db", "*.ufdb.txt"))[0]
    text = open(db).read()
    for r in tuning.POSE_ROW_COUNTS:
        assert "6-192-640-7x7-64-96-320-%d-3x3-2x2-1x1-0-NCHW-FP32-F" % r in text, r
