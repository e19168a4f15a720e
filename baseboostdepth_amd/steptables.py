"""Every integer table of one training step in ONE buffer, uploaded with ONE asynchronous copy.

The boosted recipe redraws every sample's frame set per item (mono_dataset.py:87-109), restacks per batch
(trainer.py:867-886) and overwrites `frame_ids` per batch (trainer.py:250): a real epoch sees a NEW batch signature
almost every step.  Everything the step derives from the signature on the host - the candidate / identity / pose-row
tables of `plan.ReprojectionPlan`, the work order of the fused launches, the row lists the pose modes select
sub-batches with (trainer.py:348-418), the pose-composition table (`ops.ComposeTable`) and the per-row `invert` flags
of the pose matrices - used to reach the GPU as ~70 separate pageable host-to-device copies, each of which blocks the
training thread until the stream has drained.  Here they are laid out back to back in one pinned int32 buffer
(256-byte aligned sections) that goes to HBM with a single `non_blocking` copy; the device tensors the kernels and
`index_select` read are views of that one allocation.  Signatures are kept in an LRU, so a repeated signature costs
nothing and a long `--rand` run cannot grow without bound.
"""
import collections
import ctypes
import time

import numpy as np
import torch

from . import _lib

STEREO = "s"               # plan.STEREO (plan imports this module)

ALIGN_WORDS = 64            # sections start on 256-byte boundaries (bbd_cand_t rows are read as 16-byte scalar loads)
STATS = {"packed_uploads": 0, "packed_words": 0, "single_uploads": 0, "builds": 0, "build_ms": 0.0}


def reset_stats():
    for k in STATS:
        STATS[k] = 0


class LRU:
    """Bounded map; the least recently used entry goes first (replaces the clear-everything caches)."""

    def __init__(self, capacity):
        self.capacity = max(1, int(capacity))
        self.data = collections.OrderedDict()

    def get(self, key):
        hit = self.data.get(key)
        if hit is not None:
            self.data.move_to_end(key)
        return hit

    def put(self, key, value):
        self.data[key] = value
        self.data.move_to_end(key)
        while len(self.data) > self.capacity:
            self.data.popitem(last=False)
        return value

    def __len__(self):
        return len(self.data)

    def __contains__(self, key):
        return key in self.data

    def clear(self):
        self.data.clear()


class Packer:
    """Collects int32 arrays, then uploads them as one buffer; `views[name]` are the device tensors."""

    def __init__(self):
        self.parts, self.words = [], 0

    def add(self, name, array, shape=None):
        a = np.ascontiguousarray(array, dtype=np.int32).reshape(-1)
        self.parts.append((name, self.words, a, tuple(shape) if shape is not None else tuple(np.shape(array))))
        self.words += -(-max(a.size, 1) // ALIGN_WORDS) * ALIGN_WORDS
        return name

    def reserve(self, name, shape):
        """A section the caller fills itself through `host_pointer` (bbd_fused_work_items writes its table in place)."""
        n = int(np.prod(shape))
        self.parts.append((name, self.words, None, tuple(shape)))
        self.words += -(-max(n, 1) // ALIGN_WORDS) * ALIGN_WORDS
        return name

    def upload(self, device, fill=None):
        device = torch.device(device)
        if not self.parts:
            return {}
        pinned = device.type == "cuda"
        host = torch.empty(self.words, dtype=torch.int32, pin_memory=pinned)
        hv = host.numpy()
        offsets = {}
        for name, off, a, shape in self.parts:
            offsets[name] = off
            if a is not None:
                hv[off:off + a.size] = a
        if fill is not None:
            fill(lambda name: host.data_ptr() + 4 * offsets[name])
        # ONE copy; pinned source + non_blocking: the training thread does not wait for the stream (the caching host
        # allocator keeps the staging block alive until the copy has run)
        dev = host.to(device, non_blocking=True) if pinned else host
        STATS["packed_uploads"] += 1
        STATS["packed_words"] += self.words
        views = {}
        for name, off, a, shape in self.parts:
            n = int(np.prod(shape)) if len(shape) else 1
            views[name] = dev[off:off + n].view(shape) if len(shape) else dev[off:off + 1].view(())
        views["_buffer"] = dev
        return views


def upload_single(values, device, dtype=torch.int32):
    """One small table outside a pack (callers that bypass the step tables): still pinned + asynchronous."""
    device = torch.device(device)
    host = torch.as_tensor(values, dtype=dtype)
    STATS["single_uploads"] += 1
    if device.type != "cuda":
        return host
    return host.pin_memory().to(device, non_blocking=True)


# ------------------------------------------------------------------------------------------ plan tables
def add_plan_sections(pk, plan):
    pk.add("cand", plan.cand_np, plan.cand_np.shape)
    pk.add("ncand", plan.ncand_np, plan.ncand_np.shape)
    pk.add("items", np.asarray(plan.ident_items, dtype=np.int32).reshape(-1, 4), (len(plan.ident_items), 4))
    pk.add("ident_off", np.asarray(plan.ident_off, dtype=np.int32), (len(plan.ident_off),))
    pk.add("k_rows", plan.k_rows.astype(np.int32), (len(plan.k_rows),))


PLAN_KEYS = ("cand", "ncand", "items", "ident_off", "k_rows")


def upload_plan(plan, device):
    """The five tables of a plan alone (ops called without a Trainer: stand-alone tests, tools)."""
    pk = Packer()
    add_plan_sections(pk, plan)
    views = pk.upload(device)
    return {k: views[k] for k in PLAN_KEYS}


# ------------------------------------------------------------------------------------------ pose schedule
class PoseSchedule:
    """Host description of every pose-network call of a step, in the reference's call order (trainer.py:348-418):
    requests = [(key, (frame_a, rows_a | None), (frame_b, rows_b | None), invert, n_rows)] where `(frame, rows)` stands
    for `inputs["color_aug", frame, 0][rows]` (None = all rows), plus what follows from it on the host: the row count
    of every call, the per-row `invert` flags of each batched chunk, the pose-composition table."""

    def __init__(self, plan, frame_ids, incremental, partial, decomp, chunk):
        self.incremental, self.partial, self.decomp = bool(incremental), bool(partial), bool(decomp)
        self.valid_frames = list(plan.valid_frames)
        self.valid_frames_pose = [f for f in plan.frames if f != STEREO]
        self.temporal = [f for f in list(frame_ids)[1:] if f != STEREO]
        self.requests, self.slot = [], {}
        B = plan.B

        def rows_of(f):
            return B if f == 0 else len(plan.owners(f))

        def pick(f, rows):
            # the reference's masked gathers; a selection of every row in order is the tensor itself
            rows = list(rows)
            return (f, None if rows == list(range(rows_of(f))) else tuple(rows))

        def want(key, first, second, invert):
            n = rows_of(first[0]) if first[1] is None else len(first[1])
            self.slot[key] = len(self.requests)
            self.requests.append((key, first, second, bool(invert), n))

        if self.incremental:
            # one pose-net call per ADJACENT pair, chained back to frame 0 (trainer.py:348-388)
            for f in self.temporal:
                cur = (f, None)
                if abs(f) == 1:
                    ref = (0, None)
                    want(("step", f), *((cur, ref, True) if f < 0 else (ref, cur, False)))
                else:
                    nb = f + 1 if f < 0 else f - 1
                    own_f, own_nb = plan.owners(f), plan.owners(nb)
                    near = pick(nb, [own_nb.index(b) for b in own_f])
                    want(("step", f), *((cur, near, True) if f < 0 else (near, cur, False)))
        else:
            # one call per warp job on the already selected sub-batch (trainer.py:390-405)
            for f in self.valid_frames:
                if f == STEREO:
                    continue
                mid = pick(0, plan.jobs[f])
                other = pick(f, plan.job_rows_in_source(f))
                want(("job", f), *((other, mid, True) if f < 0 else (mid, other, False)))
        if self.partial:
            # direct 0->f pose supplies the translation column except where |f| == m-2 (trainer.py:407-418)
            assert self.incremental, "--partial_skip needs --incremental_skip (the reference's shapes only fit then)"
            for f in self.valid_frames:
                if f == STEREO or abs(f) <= 1:
                    continue
                mid = pick(0, plan.owners(f))
                cur = (f, None)
                want(("direct", f), *((cur, mid, True) if f < 0 else (mid, cur, False)))

        self.rows = [r[4] for r in self.requests]
        self.base, acc = [], 0
        for n in self.rows:
            self.base.append(acc)
            acc += n
        self.total_rows = acc
        # batched pose pass: chunks of at most `chunk` calls (BBD_BN_MAX_GROUPS), per-row invert flags of each chunk
        self.chunks = []
        for lo in range(0, len(self.requests), max(1, chunk)):
            part = self.requests[lo:lo + chunk]
            flags = [int(r[3]) for r in part for _ in range(r[4])]
            self.chunks.append((lo, len(part), [r[4] for r in part], flags))
        # row lists of the warp jobs' poses (Trainer._job_poses, trainer.py:468)
        self.job_rows = {}
        if STEREO in plan.jobs:
            self.job_rows[("stereo", STEREO)] = pick(0, plan.jobs[STEREO])[1]
        if self.incremental:
            for f in plan.frames:
                if f != STEREO:
                    self.job_rows[("job", f)] = pick(f, plan.job_rows_in_source(f))[1]
        self.compose = self._compose_rows(plan) if (self.incremental or self.decomp) else None

    def row_lists(self):
        """Every distinct non-trivial row list the step selects sub-batches with."""
        seen, out = set(), []
        for _, a, b, _, _ in self.requests:
            for _, rows in (a, b):
                if rows is not None and rows not in seen:
                    seen.add(rows)
                    out.append(rows)
        for rows in self.job_rows.values():
            if rows is not None and rows not in seen:
                seen.add(rows)
                out.append(rows)
        return out

    def _compose_rows(self, plan):
        """rows / views / passthrough of the step's `ops.ComposeTable` (the incremental chain back to frame 0, the
        error-induced poses and the partial swap: trainer.py:359-388, 403-405, 415-418)."""
        ERR, REP = _lib.COMPOSE_ERROR, _lib.COMPOSE_REPLACE
        rows, views, passthrough = [], [], []       # views: (output key, first row, count, constant?)
        slot, base, rows_of = self.slot, self.base, self.rows

        def emit(okey, per_row):
            # constant rows (empty chain, nothing swapped in; T_error) carry no gradient, like the reference's
            const = all((flags & ERR) or (not chain and not (flags & REP)) for chain, _, flags in per_row)
            views.append((okey, len(rows), len(per_row), const))
            rows.extend(per_row)

        pos = {}                                 # frame -> {sample: row in the frame's stack}

        def row_of(k, b):
            table = pos.get(k)
            if table is None:
                table = pos[k] = {s: i for i, s in enumerate(plan.owners(k))}
            return table[b]

        if self.incremental:
            step_rows, chains = {}, {}          # (k-1, k) -> request slot
            for f in self.temporal:
                i = slot[("step", f)]
                if abs(f) == 1:
                    passthrough.append((("cam_T_cam", 0, f), i))
                    passthrough.append((("cam_T_cam_step", 0, f), i))
                    step_rows[(0, f)] = i
                    chains[f] = [[base[i] + j] for j in range(rows_of[i])]
                else:
                    nb = f + 1 if f < 0 else f - 1
                    passthrough.append((("cam_T_cam_step", nb, f), i))
                    step_rows[(nb, f)] = i
                    if f not in self.valid_frames_pose:
                        continue
                    own_f = plan.owners(f)
                    # the reference chains with range(f, 0, -1): EMPTY for negative f (identity pose, kept)
                    chains[f] = [[base[step_rows[(k - 1, k)]] + row_of(k, b) for k in range(f, 0, -1)] for b in own_f]
                if self.decomp:
                    emit(("cam_T_cam_error", 0, f), [(c, -1, ERR) for c in chains[f]])
            nonstereo = [m for m in plan.ms if m != 0]
            for f in self.temporal:
                if abs(f) == 1 or f not in chains:
                    continue
                swap = self.partial and f in self.valid_frames and f != STEREO
                per = []
                for j, c in enumerate(chains[f]):
                    if swap and not (abs(f) == nonstereo[j] - 2):      # reference quirk: indexed by ROW number
                        per.append((c, base[slot[("direct", f)]] + j, REP))
                    else:
                        per.append((c, -1, 0))
                emit(("cam_T_cam", 0, f), per)
        else:
            for f in self.valid_frames:
                if f == STEREO:
                    continue
                i = slot[("job", f)]
                passthrough.append((("cam_T_cam", 0, f), i))
                if self.decomp:
                    emit(("cam_T_cam_error", 0, f), [([base[i] + j], -1, ERR) for j in range(rows_of[i])])
        return rows, views, passthrough


# ------------------------------------------------------------------------------------------ the step's tables
class StepTables:
    """Plan + pose schedule + composition table of one batch signature, resident on `device` after ONE upload."""

    def __init__(self, plan, schedule, device, S, H, W, lib=None):
        from . import ops                       # (ComposeTable lives with its autograd node)
        self.plan, self.schedule, self.device = plan, schedule, torch.device(device)
        STATS["builds"] += 1
        pk = Packer()
        dkey = str(self.device)
        need_plan = dkey not in plan._dev
        if need_plan:
            add_plan_sections(pk, plan)
        # work order of the fused launches: batch order (sorted batches: most candidates first already) is ONE shared
        # table per launch shape, uploaded once by the backend; any other order travels with the step
        work = []
        if lib is not None and plan.sample_order is not None and self.device.type == "cuda":
            for backward in (0, 1):
                wkey = ("work", S, H, W, backward)
                if need_plan or wkey not in plan._dev[dkey]:
                    nt = lib.num_tiles_bwd(H, W) if backward else lib.num_tiles_fwd(H, W)
                    work.append((wkey, pk.reserve("work%d" % backward, (S * plan.B * nt, 2)), backward))
        self.compose_table, self.compose_views, self.passthrough = None, [], []
        if schedule is not None and schedule.compose is not None:
            rows, self.compose_views, self.passthrough = schedule.compose
            self.compose_table = ops.ComposeTable(rows, schedule.total_rows)
            tab, off, flat = self.compose_table.np
            pk.add("compose_tab", tab, tab.shape)
            pk.add("compose_off", off, off.shape)
            pk.add("compose_refs", flat, flat.shape)
        if schedule is not None:
            for c, (_, _, _, flags) in enumerate(schedule.chunks):
                pk.add("invert%d" % c, np.asarray(flags, dtype=np.int32), (len(flags),))
            for i, rows in enumerate(schedule.row_lists()):
                pk.add(("rows", rows), np.asarray(rows, dtype=np.int32), (len(rows),))

        failed = []

        def fill(host_pointer):
            for wkey, name, backward in work:
                order = (ctypes.c_int32 * plan.B)(*plan.sample_order)
                try:
                    lib.call("bbd_fused_work_items", plan.B, S, H, W, backward, order, host_pointer(name))
                except _lib.BbdError:
                    failed.append(wkey)         # outside the table's packing limits: the launches decode the grid order

        views = pk.upload(self.device, fill if work else None)
        if need_plan:
            plan._dev[dkey] = {k: views[k] for k in PLAN_KEYS}
        for wkey, name, _ in work:
            plan._dev[dkey][wkey] = None if wkey in failed else views[name]
        if self.compose_table is not None:
            self.compose_table._dev[dkey] = (views["compose_tab"], views["compose_off"], views["compose_refs"])
        self.invert = [views["invert%d" % c] for c in range(len(schedule.chunks))] if schedule is not None else []
        self.rows = {k[1]: v for k, v in views.items() if isinstance(k, tuple) and k[0] == "rows"}

    def index(self, rows):
        """Device int32 view of a row list of this step, or None when the step never announced it."""
        return self.rows.get(tuple(rows))


_STEP_CACHE = LRU(256)


def step_key(plan, frame_ids, incremental, partial, decomp, S, H, W, device, chunk):
    return (tuple(plan.ms), plan.trimin, plan.decomp, tuple(str(f) for f in frame_ids), bool(incremental), bool(partial),
            bool(decomp), S, H, W, str(device), chunk)


def get_step_tables(plan, frame_ids, incremental, partial, decomp, S, H, W, device, lib=None, chunk=32):
    key = step_key(plan, frame_ids, incremental, partial, decomp, S, H, W, device, chunk)
    hit = _STEP_CACHE.get(key)
    if hit is None:
        t0 = time.perf_counter()
        schedule = PoseSchedule(plan, frame_ids, incremental, partial, decomp, chunk)
        hit = _STEP_CACHE.put(key, StepTables(plan, schedule, device, S, H, W, lib))
        STATS["build_ms"] += (time.perf_counter() - t0) * 1e3
    return hit
