"""The training step in a form whose tensor SHAPES depend on the batch only through the padded row count of the pose pass.

The boosted recipe redraws every sample's frame set per item (mono_dataset.py:87-109), restacks per batch
(trainer.py:867-886) and overwrites `frame_ids` per batch (trainer.py:250): a real `--rand` epoch meets a new batch
signature almost every step (18 564 possible for batch 12 from epoch 10 on).  The per-frame batch dict
(`inputs[("color", f, 0)]` with n_f rows) makes every shape of the step follow the signature, so a step graph
keyed on it is never replayed and the recipe the project is named after ran eagerly forever (round 5: 82 ms of host
enqueue work in an 89 ms step).  Here:

* the source frames of a step live in ONE pool tensor `[F_cap, 3, H, W]` (frame 0's B rows first, then every other
  frame's stack back to back, one reserved all-zero row at the end) - the fused kernels' 16 base pointers all point at
  it and the candidate / identity tables address it by ROW;
* every integer table of the step (`steptables`: candidates, identity items, pose-pass row lists, BatchNorm call groups,
  `invert` flags, the pose-composition table and its inverse, the pose-table gather lists) is a FIXED-OFFSET section of
  one static int32 device buffer, sized by the maxima a batch of B samples can reach and padded with no-op rows; a step
  costs one pinned asynchronous copy into it;
* the batched pose pass gathers its pairs from the pool by row (`index_select`) into `[R, 6, H, W]`, R = the padded row
  count (`tuning.padded_pose_rows`; ONE value for the early curriculum), its fused BatchNorms read their call groups from
  the device table (`ops.bn_call_groups_device`, bbd_bn_act_grouped_dev_*), the pose matrices, composed poses, pose
  table, `grad_P` partials and identity maps are sized by the maxima.

The step's launches then depend on `(R, group grid, scales, lr)` only: 14-19 graphs per epoch from epoch 10 on, seven for the early
curriculum (24 .. 48 pose rows in steps of 4) (`Trainer._graph_step`), captured up front by `Trainer.prewarm()`.

What is computed does not change: the padding rows are zero images in call groups of their own that take no part in
the running statistics, nobody reads their outputs and their gradient contributions are exact zeros; no-op composition
rows yield identity matrices nobody reads; unused pose-table rows get zero partials.  The golden `pose_*` / `tri_*`
cases run through this path on the GPU tier (tests/test_gpu_pooled.py).
"""
import numpy as np
import torch

from . import _lib, ops, steptables, tuning
from .plan import STEREO, frame_slot, owners_of

MAX_REQUESTS = 26            # pose-network calls of a step the group table leaves room for (7 + 7 steps, 12 direct calls)
MAX_PAD_GROUPS = ops.BN_MAX_GROUPS - MAX_REQUESTS
SMALL_GROUPS = 8             # group grid of the launches: 8 where a step's call groups fit (early curriculum), else 32


def _slot_frame(slot):
    return STEREO if slot == _lib.MAX_FRAME_SLOTS - 1 else slot - 7


class Caps:
    """Maxima a batch of B samples can reach (frame offsets -7..7 and 's')."""

    def __init__(self, B):
        self.B = B
        self.F = 16 * B + 1                     # pool rows: 15 temporal frames + stereo per sample, + the zero row
        self.Z = 16 * B                         # the all-zero row (padding pairs of the pose pass)
        self.NI = 6 * B                         # identity maps (trainer.py:1025-1042: six per sample at most)
        self.NP = 12 * B                        # pose-table rows: six true-pose + six error-induced warps per sample
        self.R = tuning.padded_pose_rows(MAX_REQUESTS * B)      # pose-pass rows
        self.NO = MAX_REQUESTS * B              # composed matrices (T_error of every step frame + swapped / chained T)
        self.NREF = 68 * B                      # references of the inverse composition table (chains of <= 7 + direct)


class Layout:
    """Fixed section offsets (int32 words, 256-byte aligned) of the static table buffer of a batch size."""

    def __init__(self, caps):
        c, G = caps, ops.BN_MAX_GROUPS
        self.sections, self.words = {}, 0
        for name, shape in (("cand", (c.B, _lib.MAX_CAND, 4)), ("ncand", (c.B,)), ("items", (c.NI, 4)),
                            ("ident_off", (c.B + 1,)), ("k_rows", (c.NP,)), ("tsel", (c.NP,)), ("idx_a", (c.R,)),
                            ("idx_b", (c.R,)), ("invert", (c.R,)), ("groups", (G + 2,)),
                            ("compose_tab", (c.NO, _lib.COMPOSE_STRIDE)), ("compose_off", (c.R + 1,)),
                            ("compose_refs", (c.NREF, 2))):
            n = int(np.prod(shape))
            self.sections[name] = (self.words, shape)
            self.words += -(-n // steptables.ALIGN_WORDS) * steptables.ALIGN_WORDS

    def view(self, buf, name):
        off, shape = self.sections[name]
        return buf[off:off + int(np.prod(shape))].reshape(shape)       # (a view, for numpy arrays and tensors alike)


def lowest_rows(R, quantum):
    """Smallest real row count `tuning.padded_pose_rows` rounds up to R."""
    n = R
    while n > 1 and tuning.padded_pose_rows(n - 1, quantum) == R:
        n -= 1
    return n


class PooledTables:
    """Host side of one batch signature: the packed table buffer (pinned) + what the trainer needs to know on the host."""

    def __init__(self, plan, frames, frame_ids, incremental, partial, decomp, maxing, caps, layout, pad_quantum, early_rows,
                 pinned):
        B = plan.B
        assert B == caps.B
        sched = steptables.PoseSchedule(plan, frame_ids, incremental, partial, decomp, chunk=1 << 30)
        self.plan, self.schedule = plan, sched
        if len(sched.requests) > MAX_REQUESTS:
            raise ValueError("more pose-network calls than the group table holds")
        # ---- pool layout: frame 0 first (targets = rows 0..B-1), then the other frames of the batch dict
        order = [0] + sorted((f for f in frames if f not in (0, STEREO)), key=lambda f: (abs(f), f < 0))
        if STEREO in frames:
            order.append(STEREO)
        self.frame_rows, at = {}, 0
        for f in order:
            n = B if f == 0 else len(owners_of(plan.ms, f))
            self.frame_rows[f] = (at, n)
            at += n
        self.used_rows = at
        if at > caps.Z:
            raise ValueError("more frame rows than the pool holds")
        prow = lambda f, r: self.frame_rows[f][0] + r

        # ---- the pose pass: R rows, real pairs first, padding pairs (zero row) behind them
        n_real = sched.total_rows
        # early curriculum (at most 4 calls: frames +-1, +-2): the small group grid whatever the ordering, the padding in at
        # most 4 groups.  `early_rows` > 0 pads every batch of the phase to that ONE row count (one graph, but the pose pass
        # then always runs the phase's maximum: measured 0.79 of the frozen batch at 48 rows for a mean of 31 real ones);
        # the default rounds up to the next measured row count like the later epochs do (24 .. 48 in steps of 4 for batch 12)
        early = not maxing and len(sched.requests) <= SMALL_GROUPS // 2
        pad_groups = SMALL_GROUPS - SMALL_GROUPS // 2 if early else MAX_PAD_GROUPS
        if early and early_rows and n_real <= early_rows:
            R = max_pad = early_rows
        else:
            R = tuning.padded_pose_rows(max(n_real, 1), pad_quantum)
            max_pad = R - lowest_rows(R, pad_quantum)
        if R > caps.R:
            raise ValueError("more pose rows than the tables hold")
        self.R, self.n_real = R, n_real
        hb = np.zeros(layout.words, dtype=np.int32)
        sec = lambda name: layout.view(hb, name)
        ia, ib, inv = sec("idx_a"), sec("idx_b"), sec("invert")
        ia[:], ib[:] = caps.Z, caps.Z
        at = 0
        for _, (fa, ra), (fb, rb), invert, n in sched.requests:
            ia[at:at + n] = [prow(fa, r) for r in (ra if ra is not None else range(n))]
            ib[at:at + n] = [prow(fb, r) for r in (rb if rb is not None else range(n))]
            inv[at:at + n] = int(invert)
            at += n
        # call groups: one per pose-network call, then the padding in chunks of at most `bound` rows, then empty ones
        self.bound = min(R, max(B, -(-max_pad // pad_groups)))      # (the launches refuse a bound above the row count)
        rows = list(sched.rows)
        pad = R - n_real
        while pad > 0:
            rows.append(min(pad, self.bound))
            pad -= rows[-1]
        assert len(rows) <= ops.BN_MAX_GROUPS and (not sched.rows or max(sched.rows) <= self.bound)
        self.G = SMALL_GROUPS if len(rows) <= SMALL_GROUPS else ops.BN_MAX_GROUPS
        assert not early or self.G == SMALL_GROUPS
        g = sec("groups")
        g[1:len(rows) + 1] = np.cumsum(rows)
        g[len(rows) + 1:ops.BN_MAX_GROUPS + 1] = R
        g[ops.BN_MAX_GROUPS + 1] = len(sched.rows)          # tracked groups = the real calls

        # ---- composition table (chains / T_error / partial swap), padded with constant no-op rows
        self.pose_views = []                   # (output key, "M" | "out", first row, rows, constant)
        where = {}
        if sched.compose is not None:
            crow, views, passthrough = sched.compose
            table = ops.ComposeTable(crow, n_real)
            tab, off, flat = table.np
            NO = table.NO
            if NO > caps.NO or flat.shape[0] > caps.NREF:
                raise ValueError("composition table larger than the static one")
            ct, co, cr = sec("compose_tab"), sec("compose_off"), sec("compose_refs")
            ct[:, 8], ct[:, 9] = -1, _lib.COMPOSE_ERROR
            if NO:
                ct[:NO] = tab[:NO]
            co[:n_real + 1] = off
            co[n_real + 1:] = off[-1]
            if off[-1]:
                cr[:off[-1]] = flat[:off[-1]]
            for okey, o0, n, const in views:
                where[okey] = ("out", o0, n)
                self.pose_views.append((okey, "out", o0, n, const))
        else:
            sec("compose_tab")[:, 8], sec("compose_tab")[:, 9] = -1, _lib.COMPOSE_ERROR
            passthrough = [(("cam_T_cam", 0, f), sched.slot[("job", f)]) for f in sched.valid_frames if f != STEREO]
        for okey, i in passthrough:
            where[okey] = ("M", sched.base[i], sched.rows[i])
            self.pose_views.append((okey, "M", sched.base[i], sched.rows[i], False))

        # ---- pose table: row p = (K row, T row of cat(stereo_T [B], composed [NO_cap], pose matrices [R]))
        src_off = {"stereo": 0, "out": B, "M": B + caps.NO}
        tsel, krows = sec("tsel"), sec("k_rows")
        krows[:plan.NP] = plan.k_rows
        per_source_rows = bool(incremental)
        for kind, f in plan.pose_jobs:
            p0 = plan.pose_offset[(kind, f)]
            for j, b in enumerate(plan.jobs[f]):
                if f == STEREO:
                    tsel[p0 + j] = src_off["stereo"] + b
                    continue
                buf, o0, n = where[("cam_T_cam" if kind == "T" else "cam_T_cam_error", 0, f)]
                r = plan.job_rows_in_source(f)[j] if per_source_rows else j
                assert r < n
                tsel[p0 + j] = src_off[buf] + o0 + r
        if plan.NP > caps.NP or plan.NI > caps.NI:
            raise ValueError("more pose rows / identity maps than the tables hold")

        # ---- candidates and identity items, addressed by pool row (slot 0 = the pool)
        cand = plan.cand_np.copy()
        for b in range(B):
            for k in range(int(plan.ncand_np[b])):
                if (cand[b, k, 0] & 0xff) == _lib.KIND_WARP:
                    cand[b, k, 2] = prow(_slot_frame(int(cand[b, k, 1])), int(cand[b, k, 2]))
                    cand[b, k, 1] = 0
        sec("cand")[:] = cand
        sec("ncand")[:] = plan.ncand_np
        items = sec("items")
        for i, (b, slot, row, _) in enumerate(plan.ident_items):
            items[i] = (b, 0, prow(_slot_frame(slot), row), 0)
        sec("ident_off")[:] = plan.ident_off
        # frames a table row refers to (a batch dict may list frames nothing reads: they need not be there)
        self.needed_aug = {0} | {fr for _, (fa, _), (fb, _), _, _ in sched.requests for fr in (fa, fb)}
        self.needed = set(plan.frames) | self.needed_aug
        self.host = torch.from_numpy(hb)
        if pinned:
            self.host = self.host.pin_memory()


class PlanView:
    """What `ops.identity_losses` / `ops.fused_reprojection_min_disp` read from a plan, backed by the static tables."""
    sample_order = None          # the shared batch-order work table (canonical batches have most candidates first already;
    #                              the order is a speed choice only - results do not depend on it)
    zero_partials = True         # pose-table rows beyond the step's own are never written by the backward

    def __init__(self, B, NP, NI, tables):
        self.B, self.NP, self.NI, self._tables = B, NP, NI, tables

    def tables(self, device):
        return self._tables


class PooledStep:
    """Static buffers of a trainer's pooled steps + the step itself on them (`forward` is what a step graph captures)."""

    def __init__(self, trainer):
        opt = trainer.opt
        self.trainer = trainer
        self.B, self.H, self.W = opt.batch_size, opt.height, opt.width
        self.caps = Caps(self.B)
        self.layout = Layout(self.caps)
        self.device = trainer.device
        self.cache = steptables.LRU(4096)
        self.allocated = False
        self.stats = {"loads": 0, "builds": 0, "build_ms": 0.0, "fallbacks": 0}

    # ------------------------------------------------------------------ host tables per signature
    def early_rows(self):
        """`opt.early_pose_rows = "max"`: ONE pose-pass row count for the whole early curriculum - every warp job of frames
        +-1 (+-2 with tri-minimisation) on all B samples (mono_dataset.py:59-66), rounded to a row count MIOpen has find
        results for; an int: that row count; default 0: the next measured row count at or above the batch's own."""
        want = getattr(self.trainer.opt, "early_pose_rows", 0)
        if want == "max":
            per = 2 * (2 if self.trainer.opt.trimin else 1) * self.B
            return tuning.padded_pose_rows(per, max(self.trainer.pose_pad_rows, 1))
        return int(want or 0)

    def tables_for(self, plan, inputs):
        """PooledTables of this batch (None: the batch does not fit the pooled form and takes the per-signature path)."""
        tr, opt = self.trainer, self.trainer.opt
        maxing = bool(tr.maxing_valid_frames)
        incremental = bool(opt.incremental_skip and maxing)
        partial = bool(opt.partial_skip and maxing)
        frames = tuple(inputs["frames"]) if "frames" in inputs else tuple(
            sorted({k[1] for k in inputs if isinstance(k, tuple) and k[0] == "color" and k[2] == 0}, key=str))
        key = (tuple(plan.ms), plan.trimin, plan.decomp, tuple(str(f) for f in opt.frame_ids), tuple(str(f) for f in frames),
               incremental, partial, bool(opt.decomp), maxing)
        hit = self.cache.get(key)
        if hit is None:
            import time
            t0 = time.perf_counter()
            try:
                hit = PooledTables(plan, frames, opt.frame_ids, incremental, partial, bool(opt.decomp), maxing, self.caps,
                                   self.layout, max(tr.pose_pad_rows, 1), self.early_rows(), self.device.type == "cuda")
            except ValueError:
                hit = False
                self.stats["fallbacks"] += 1
            if hit and tr.pose_pad_rows > 0:
                tuning.note_pose_rows(hit.R)            # (warns once per row count MIOpen has no find results for)
            self.cache.put(key, hit)
            self.stats["builds"] += 1
            self.stats["build_ms"] += (time.perf_counter() - t0) * 1e3
            steptables.STATS["builds"] += 1
            steptables.STATS["build_ms"] += (time.perf_counter() - t0) * 1e3
        return hit or None

    def bucket_orderings(self, early, draws=600):
        """Per-sample offsets whose batches between them meet the graph keys (padded pose rows, group grid) of a curriculum
        phase: the early curriculum's row counts (2 * sum(m): 24 .. 48 for batch 12), one ordering each; from epoch 10 on, two walks from the smallest to the largest pass (all samples
        alike; one sample at the largest offset) plus seeded draws from the loader's offset distributions - the
        pass's row count depends on WHICH frames a batch's samples use, not only on how many.  A bucket none of them meets
        is captured when training first meets it."""
        import random
        from .plan import ReprojectionPlan
        opt, B = self.trainer.opt, self.B
        trimin = bool(opt.trimin)
        if early:
            top = 2 if trimin else 1
            # (all samples at the largest offset = the phase's largest pass; then smaller ones down to the smallest row count)
            cands = [[top] * B] + [[top] * k + [1] * (B - k) for k in range(B - 1, -1, -1)] + [[1] * k + [0] * (B - k) for k in range(B - 1, 0, -1)]
            seen, out = set(), []
            quantum = max(self.trainer.pose_pad_rows, 1)
            for ms in cands:
                if not trimin and min(ms) == 0:
                    continue
                n = sum(2 * min(m, top) for m in ms)        # frames +-1 (+-2): one row per sample and frame
                R = self.early_rows() if self.early_rows() and n <= self.early_rows() else tuning.padded_pose_rows(max(n, 1), quantum)
                if R not in seen:
                    seen.add(R)
                    out.append((R, ms))
            return [ms for _, ms in sorted(out, key=lambda e: -e[0])]
        top = 7 if trimin else 5
        quantum = max(self.trainer.pose_pad_rows, 1)
        seen, out = set(), []

        def visit(ms):
            ms = sorted(ms, reverse=True)
            M = max(ms)
            fid = sorted(range(-M, M + 1), key=abs)
            plan = ReprojectionPlan([[0, m, -m] for m in ms], opt.trimin, opt.decomp)
            sched = steptables.PoseSchedule(plan, fid, bool(opt.incremental_skip), bool(opt.partial_skip), bool(opt.decomp), 1 << 30)
            if len(sched.requests) > MAX_REQUESTS:
                return
            R = tuning.padded_pose_rows(max(sched.total_rows, 1), quantum)
            bound = min(R, max(B, -(-(R - lowest_rows(R, quantum)) // MAX_PAD_GROUPS)))
            groups = len(sched.rows) + -(-(R - sched.total_rows) // bound)
            key = (R, SMALL_GROUPS if groups <= SMALL_GROUPS else ops.BN_MAX_GROUPS)
            if R <= self.caps.R and key not in seen:
                seen.add(key)
                out.append((R, ms))

        for first in (1, top):
            ms = [first] + [1] * (B - 1)
            visit(ms)
            i = 0
            while min(ms) < top:
                if ms[i % B] < top:
                    ms[i % B] += 1
                    visit(ms)
                i += 1
        rnd = random.Random(2025)
        weights = ([.050, .050, .077, .094, .139, .142, .448], [.108, .287, .277, .135, .068, .040, .084], [1.0] * 7)
        for i in range(draws):
            visit(rnd.choices(range(1, top + 1), weights[i % 3][:top], k=B))
        # largest pass first: the graphs share one memory pool, and a later, smaller capture then fits into the blocks an
        # earlier one freed instead of growing the pool
        return [ms for _, ms in sorted(out, key=lambda e: -e[0])]

    # ------------------------------------------------------------------ static buffers
    def allocate(self, scales):
        if self.allocated:
            return
        dev, c, B, H, W = self.device, self.caps, self.B, self.H, self.W
        self.pool_color = torch.zeros(c.F, 3, H, W, device=dev)
        self.pool_aug = torch.zeros(c.F, 3, H, W, device=dev)
        self.tables = torch.zeros(self.layout.words, dtype=torch.int32, device=dev)
        self.v = {name: self.layout.view(self.tables, name) for name in self.layout.sections}
        self.K = torch.zeros(B, 4, 4, device=dev)
        self.inv_K = torch.zeros(B, 4, 4, device=dev)
        self.stereo_T = torch.eye(4, device=dev).repeat(B, 1, 1)
        self.noise = torch.zeros(B, H, W, device=dev)
        self.pyramid = {}
        self.planview = PlanView(B, c.NP, c.NI, {"cand": self.v["cand"], "ncand": self.v["ncand"], "items": self.v["items"],
                                                 "ident_off": self.v["ident_off"], "k_rows": self.v["k_rows"]})
        self.allocated = True

    def load(self, inputs, tab, scales):
        """The batch into the static buffers (device-to-device copies on the current stream + ONE pinned table upload)."""
        self.allocate(scales)
        for f, (at, n) in tab.frame_rows.items():
            if f not in tab.needed and ("color", f, 0) not in inputs:
                continue
            self.pool_color[at:at + n].copy_(inputs[("color", f, 0)], non_blocking=True)
            if f != STEREO and (f in tab.needed_aug or ("color_aug", f, 0) in inputs):
                self.pool_aug[at:at + n].copy_(inputs[("color_aug", f, 0)], non_blocking=True)
        for s in scales:
            if s:
                src = inputs[("color", 0, s)]
                if s not in self.pyramid:
                    self.pyramid[s] = torch.empty_like(src)
                self.pyramid[s].copy_(src, non_blocking=True)
        self.K.copy_(inputs[("K", 0)], non_blocking=True)
        self.inv_K.copy_(inputs[("inv_K", 0)], non_blocking=True)
        self.stereo_T.copy_(inputs["stereo_T"], non_blocking=True)
        if inputs.get("noise") is not None:
            self.noise.copy_(inputs["noise"], non_blocking=True)
        self.tables.copy_(tab.host, non_blocking=True)
        steptables.STATS["packed_uploads"] += 1
        steptables.STATS["packed_words"] += self.layout.words
        self.stats["loads"] += 1

    def static_inputs(self, scales):
        B = self.B
        d = {("color", 0, 0): self.pool_color[:B], ("color_aug", 0, 0): self.pool_aug[:B], ("K", 0): self.K,
             ("inv_K", 0): self.inv_K, "stereo_T": self.stereo_T}
        for s in scales:
            if s:
                d[("color", 0, s)] = self.pyramid[s]
        return d

    # ------------------------------------------------------------------ the step on the static buffers
    def pose_part(self, R, G, bound):
        """The batched pose pass + pose matrices + composition on the static buffers -> (M [R,4,4], composed [NO_cap,4,4])."""
        tr, opt, v, c = self.trainer, self.trainer.opt, self.v, self.caps
        be = tr._backend()
        enc = tr.models["pose_encoder"]
        from .networks.encoder import ResnetEncoder
        if isinstance(enc, ResnetEncoder) and ops.FUSED_NN:
            # the pairs of every pose-network call, gathered from the pool AND normalised like the encoder does, in one pass
            x = ops.gather_pairs(self.pool_aug, v["idx_a"][:R], v["idx_b"][:R], normalize=(0.45, 0.225), backend=be)
            with ops.bn_call_groups_device(v["groups"], G, bound):
                feats = [enc(x, normalized=True)]
        else:
            x = torch.cat([self.pool_aug.index_select(0, v["idx_a"][:R]), self.pool_aug.index_select(0, v["idx_b"][:R])], 1)
            with ops.bn_call_groups_device(v["groups"], G, bound):
                feats = [enc(x)]
        axisangle, translation = tr.models["pose"](feats)
        M = ops.pose_matrix(axisangle[:, 0], translation[:, 0], backend=be, invert_rows=v["invert"][:R])
        out = ops.pose_compose_static(M, v["compose_tab"], v["compose_off"][:R + 1], v["compose_refs"], c.NO,
                                      float(opt.pose_error), be)
        return M, out

    def loss_part(self, M, out, outputs, has_noise):
        """generate_images_pred + compute_losses on the static buffers: `outputs` holds the decoder's ("disp", s) maps."""
        tr, opt, v, B = self.trainer, self.trainer.opt, self.v, self.B
        be = tr._backend()
        scales = list(opt.scales)
        sin = self.static_inputs(scales)
        target = sin[("color", 0, 0)]
        src = torch.cat([self.stereo_T, out, M], 0)
        proj = ops.pose_table_rows(src.index_select(0, v["tsel"]), self.K.index_select(0, v["k_rows"]),
                                   self.inv_K.index_select(0, v["k_rows"]))
        noise = self.noise if has_noise else torch.randn(B, self.H, self.W, device=self.device) * 0.00001
        ident = ops.identity_losses(self.planview, self.pool_color, target, opt.no_ssim, be)
        disps = [outputs[("disp", s)] for s in scales]
        loss_sum, min_loss, argmin, _, depth = ops.fused_reprojection_min_disp(
            disps, proj, target, ident, noise, self.planview, self.pool_color, opt.min_depth, opt.max_depth, opt.no_ssim,
            False, bool(getattr(opt, "materialize_depth", True)), be)
        if depth is not None:
            for i, s in enumerate(scales):
                outputs[("depth", 0, s)] = depth[i].unsqueeze(1)
        outputs[("bbd", "loss_sum")] = loss_sum
        outputs[("bbd", "to_optimise")] = min_loss
        outputs[("bbd", "argmin")] = argmin
        outputs[("bbd", "identity")] = ident
        outputs[("bbd", "pose_matrices")] = M
        outputs[("bbd", "composed_poses")] = out
        losses = tr.compute_losses(sin, outputs)
        return outputs, losses

    def forward(self, R, G, bound, has_noise):
        """process_batch(is_train=True) on the static buffers: nothing in here depends on the batch signature beyond
        (R, G, bound) - this is what `Trainer._graph_step` captures.  The pose network runs on the trainer's second
        stream, like in the per-signature path."""
        tr = self.trainer
        side = tr._pose_stream()
        if side is None:
            M, out = self.pose_part(R, G, bound)
        else:
            main = tr._main_stream = torch.cuda.current_stream(self.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                M, out = self.pose_part(R, G, bound)
        outputs = {}
        feats = tr.models["encoder"](self.pool_aug[:self.B])
        outputs.update(tr.models["depth"](feats))
        if side is not None:
            main.wait_stream(side)
            if not torch.cuda.is_current_stream_capturing():
                M.record_stream(main)
                out.record_stream(main)
        return self.loss_part(M, out, outputs, has_noise)

    @staticmethod
    def with_pose_views(outputs, tab):
        """The reference's pose keys (`("cam_T_cam", 0, f)`, `("cam_T_cam_step", a, b)`, `("cam_T_cam_error", 0, f)`) as row
        ranges of the step's two pose buffers - host bookkeeping only, per signature."""
        out = dict(outputs)
        M, comp = outputs[("bbd", "pose_matrices")], outputs[("bbd", "composed_poses")]
        for okey, buf, o0, n, const in tab.pose_views:
            view = (M if buf == "M" else comp)[o0:o0 + n]
            out[okey] = view.detach() if const else view
        return out
