"""Index tables that replace the reference's boolean sub-batch masks.

The reference expresses "which source frame is warped for which target sample, and which
losses compete at a pixel" with ~10 dicts of Python bool lists rebuilt every batch
(`Trainer.valid_frames_trimin`, trainer.py:888-981) and four copy-pasted `torch.cat`/`torch.min`
branches (`x_min_opt`, trainer.py:983-1100).  Here the same semantics become one small integer
table per batch `ordering`, uploaded once and consumed by a single kernel launch:

    cand[b][k] = (kind, frame slot, source row, pose row)    k = arg-min id of the reference

so no gathered image copies and no concatenated loss stacks are ever materialised.
"""
import numpy as np
import torch

from .steptables import LRU
from ._lib import MAX_CAND, MAX_FRAME_SLOTS, KIND_WARP, KIND_IDENT, FLAG_NO_POSE_GRAD, PAIR_SHIFT

STEREO = "s"


def frame_slot(f):
    """Slot of frame id f in the frame pointer array: -7..7 -> 0..14, 's' -> 15."""
    if f == STEREO:
        return MAX_FRAME_SLOTS - 1
    if not -7 <= f <= 7:
        raise ValueError("frame offset %r outside the supported range -7..7" % (f,))
    return f + 7


def sample_max_offsets(ordering):
    """ordering (trainer.py:870): [0,'s'] -> 0 ; [0, m, -m] -> m."""
    return [0 if o[1] == STEREO else int(o[1]) for o in ordering]


def owners_of(ms, f):
    """Samples (batch positions) that own a row of inputs[("color", f, 0)] (custom_collate, trainer.py:882): the
    stereo frame is stacked for samples with m < 3, temporal frame f for samples whose own frame set reaches |f|."""
    if f == STEREO:
        return [b for b, m in enumerate(ms) if m < 3]
    return [b for b, m in enumerate(ms) if m >= abs(f)]


def canonical_permutation(ms):
    """Order in which `Trainer.train_step` takes the samples of a batch: largest frame offset first (stable).  The
    loss is a mean over all pixels of all samples and BatchNorm statistics are sums over the batch, so the step does not
    depend on the order mathematically - with ONE exception, the reference's `--partial_skip` rule, which keeps the chained
    translation of frame f for ROW NUMBERS r with |f| == m_r - 2 of the non-stereo sample list while the rows of f's stack
    belong to f's owners (trainer.py:331, 415-418): on a shuffled batch the decision meets mismatched samples, on the
    canonical order (owners of f = a prefix of the list) it always meets the sample's own m.  Every canonical batch is a
    valid reference outcome (it is what the reference computes when the loader happens to deliver that order), the
    distribution over an epoch is not the reference's; `opt.canonical_order = False` keeps the caller's order and with it
    the reference's row rule on it (tests/test_round6_host_logic.py pins both).  The loader's order is a random shuffle
    anyway - but the number of distinct batch SIGNATURES depends on it: 3^12 orderings of the early curriculum's m in {0,1,2} are 91 multisets, so step graphs and
    table uploads are reused; and samples with the most candidates come first, which is the fused launches' work order."""
    return sorted(range(len(ms)), key=lambda b: -ms[b])


def reprojection_frames(m, trimin):
    """Source frames competing for a sample whose largest usable offset is m, in the
    reference's arg-min order (trainer.py:987, :993-995, :1006-1010, :1025-1030; MD2 :549-554)."""
    if m == 0:
        return [STEREO]
    if not trimin:
        return [m, -m]
    if m <= 2:
        temporal = [f for k in range(m, 0, -1) for f in (k, -k)]
        return temporal + [STEREO]
    return [f for k in (m, m - 1, m - 2) for f in (k, -k)]


class ReprojectionPlan:
    """Everything about a batch that depends only on its `ordering` and the trimin/decomp flags."""

    def __init__(self, ordering, trimin, decomp):
        self.ms = sample_max_offsets(ordering)
        self.B = len(self.ms)
        self._own = {}
        self.trimin = bool(trimin)
        self.decomp = bool(trimin and decomp)   # error-induced warps exist only on the tri-min path
        per_sample = [reprojection_frames(m, self.trimin) for m in self.ms]

        # warp jobs: frame -> target samples (batch order), as the reference's mask dicts select them
        self.jobs = {}
        for b, frames in enumerate(per_sample):
            for f in frames:
                self.jobs.setdefault(f, []).append(b)
        self.frames = sorted(self.jobs, key=lambda f: (99, 0) if f == STEREO else (abs(f), f < 0))
        # reference `valid_frames` after the extension at trainer.py:961-981
        self.valid_frames = list(self.frames)

        # projection-table rows: all true-pose jobs, then all error-induced jobs
        self.pose_jobs = [("T", f) for f in self.frames]
        if self.decomp:
            self.pose_jobs += [("E", f) for f in self.frames if f != STEREO]
        self.pose_offset, off = {}, 0
        for job in self.pose_jobs:
            self.pose_offset[job] = off
            off += len(self.jobs[job[1]])
        self.NP = off
        # K / inv_K are sliced by COUNT in the reference (trainer.py:431-432), not by mask
        self.k_rows = np.concatenate([np.arange(len(self.jobs[f])) for _, f in self.pose_jobs]).astype(np.int64)

        # identity items: one per (sample, frame) pair
        self.ident_items, self.ident_index = [], {}
        for b, frames in enumerate(per_sample):
            for f in frames:
                self.ident_index[(b, f)] = len(self.ident_items)
                self.ident_items.append((b, frame_slot(f), self.source_row(f, b), 0))
        self.NI = len(self.ident_items)
        # items are sample-major: group g = the identity candidates of target sample g (bbd_identity_loss_grouped_fwd)
        self.ident_off = [0]
        for b in range(self.B):
            self.ident_off.append(self.ident_off[-1] + sum(1 for it in self.ident_items if it[0] == b))
        assert all(self.ident_items[i][0] == b for b in range(self.B) for i in range(self.ident_off[b], self.ident_off[b + 1]))

        cand = np.zeros((self.B, MAX_CAND, 4), dtype=np.int32)
        ncand = np.zeros(self.B, dtype=np.int32)
        self.cand_names = []
        for b, frames in enumerate(per_sample):
            entries, names = [], []
            for f in frames:
                flag = FLAG_NO_POSE_GRAD if f == STEREO else 0
                entries.append((KIND_WARP | flag, frame_slot(f), self.source_row(f, b), self.pose_row("T", f, b)))
                names.append(("T", f))
            if self.decomp:
                for f in frames:
                    if f == STEREO:
                        continue
                    entries.append((KIND_WARP | FLAG_NO_POSE_GRAD, frame_slot(f), self.source_row(f, b),
                                    self.pose_row("E", f, b)))
                    names.append(("E", f))
            for f in frames:
                entries.append((KIND_IDENT, 0, self.ident_index[(b, f)], 0))
                names.append(("I", f))
            if len(entries) > MAX_CAND:
                raise ValueError("too many candidates")
            # pairing hint for the kernels (bits 16-23 of `kind` = 1 + index of the candidate to take in the same
            # pass): the error-induced warp of a frame samples the same source image as its true-pose warp, a few
            # pixels apart, so the pair's gathers share cache lines (include/bbd_hip.h, bbd_cand_t)
            index = {nm: k for k, nm in enumerate(names)}
            entries = [list(e) for e in entries]
            for k, (kind, f) in enumerate(names):
                other = index.get(("E" if kind == "T" else "T", f)) if kind in ("T", "E") else None
                if other is not None:
                    entries[k][0] |= (other + 1) << PAIR_SHIFT
            cand[b, :len(entries)] = np.array(entries, dtype=np.int32)
            ncand[b] = len(entries)
            self.cand_names.append(names)
        self.cand_np, self.ncand_np = cand, ncand
        # the order in which the fused launches' workgroups take the samples (bbd_fused_work_items' `sample_order`): most
        # candidates first, so that a batch mixing 8-, 14- and 18-candidate samples ends on its cheap workgroups
        # (stable: equal counts keep batch order; None = batch order, nothing to upload)
        order = sorted(range(self.B), key=lambda b: -int(ncand[b]))
        self.sample_order = None if order == list(range(self.B)) else order
        self._dev = {}

    # ------------------------------------------------------------------ row bookkeeping
    def owners(self, f):
        """Samples that own a tensor row in inputs[("color", f, 0)] (custom_collate, trainer.py:882)."""
        own = self._own.get(f)
        if own is None:
            own = self._own[f] = owners_of(self.ms, f)
        return own

    def source_row(self, f, b):
        return self.owners(f).index(b)

    def pose_row(self, kind, f, b):
        return self.pose_offset[(kind, f)] + self.jobs[f].index(b)

    def job_rows_in_source(self, f):
        """Rows of inputs[("color", f, 0)] that take part in frame f's warp job
        (the reference's valid_tri_mask / valid_mask over the n_f rows)."""
        own = self.owners(f)
        return [own.index(b) for b in self.jobs[f]]

    # ------------------------------------------------------------------ device tables
    def tables(self, device):
        """cand / ncand / items / ident_off / k_rows on `device` (int32 views of ONE allocation, one asynchronous
        upload).  Inside a training step `steptables.StepTables` has already put them there together with the rest of
        the step's tables; this is the stand-alone path (ops called without a Trainer)."""
        key = str(device)
        if key not in self._dev:
            from .steptables import upload_plan
            self._dev[key] = upload_plan(self, device)
        return self._dev[key]


# plans by (per-sample offsets, flags).  Boosted `--rand` batches draw a new multiset of offsets almost every step
# (18 564 of them for batch 12 from epoch 10 on): least recently used goes first, nothing is ever cleared wholesale
_PLAN_CACHE = LRU(512)


def get_plan(ordering, trimin, decomp):
    key = (tuple(sample_max_offsets(ordering)), bool(trimin), bool(decomp))
    plan = _PLAN_CACHE.get(key)
    if plan is None:
        plan = _PLAN_CACHE.put(key, ReprojectionPlan(ordering, trimin, decomp))
    return plan
