"""KITTI loader in index-table form.

Reference: `datasets/mono_dataset.py` (`MonoDataset.__getitem__`, `preprocess`), `datasets/
kitti_dataset.py` (`KITTIRAWDataset`), `Trainer.custom_collate` (trainer.py:867-886) and the loaders
built in trainer.py:120-131, :214-220.  Same class names, constructor arguments, split-file format
(`folder frame side [kt baseline]`), frame-set curriculum and random-draw order.

What is different by design: `__getitem__` does only host work - frame-set selection, the random
draws, JPEG decode - and returns a *recipe* (decoded uint8 frames + parameters).  Resize(LANCZOS),
the scale pyramid, ColorJitter, ToTensor and the stacking of `custom_collate` run on the GPU for the
whole batch (`DeviceCollate` -> imageops -> csrc/bbd_image.hip), bit-exact against Pillow, writing
directly into the rows of the collated tensors.  `DeviceCollate(batch)` returns exactly the dict
`Trainer.custom_collate` returns for the reference's per-item dicts.
"""
import collections
import os
import random
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
from PIL import Image, ImageFile

from . import imageops
from .plan import canonical_permutation

ImageFile.LOAD_TRUNCATED_IMAGES = True          # mono_dataset.py:3
STEREO = "s"
STEREO_ID = -50                                 # how 's' travels inside the "frames" tensor (mono_dataset.py:141-142)


def readlines(filename):
    """utils.py:readlines."""
    with open(filename, "r") as f:
        return f.read().splitlines()


def pil_loader(path):
    """mono_dataset.py:15-18; returns the decoded RGB frame as uint8 [h,w,3]."""
    with open(path, "rb") as f:
        with Image.open(f) as img:
            return np.array(img.convert("RGB"))          # writable copy (worker processes wrap it in a tensor)


def draw_color_jitter(gen, brightness, contrast, saturation, hue):
    """One call of torchvision 0.9 `ColorJitter.forward`: a random permutation of the four ops, each
    factor drawn when its op comes up (`torch.tensor(1.0).uniform_(lo, hi)`).  Returns [(op, factor)]
    in application order."""
    ranges = {imageops.BRIGHTNESS: brightness, imageops.CONTRAST: contrast, imageops.SATURATION: saturation,
              imageops.HUE: hue}
    seq = []
    for fn_id in torch.randperm(4, generator=gen).tolist():
        lo, hi = ranges[fn_id]
        seq.append((fn_id, torch.tensor(1.0).uniform_(lo, hi, generator=gen).item()))
    return seq


class MonoDataset:
    """Same constructor as the reference (mono_dataset.py:22-36)."""

    def __init__(self, filenames, epoch, height, width, kt_path=None, syns_path=None, rand=False, scales=[0],
                 trimin=False, kt=False, is_train=False, img_ext=".jpg", naive_mix=False, seed=0):
        self.filenames, self.epoch, self.height, self.width = filenames, epoch, height, width
        self.kt_path, self.syns_path, self.rand, self.scales = kt_path, syns_path, rand, list(scales)
        self.trimin, self.kt, self.is_train, self.img_ext, self.naive_mix = trimin, kt, is_train, img_ext, naive_mix
        self.seed = seed
        self._on_disk = {}                 # path -> bool: every sample probes 14 neighbouring frames (mono_dataset.py:96-99)
        self.loader = pil_loader
        self.brightness, self.contrast, self.saturation, self.hue = (0.8, 1.2), (0.8, 1.2), (0.8, 1.2), (-0.1, 0.1)
        # curriculum (mono_dataset.py:58-63)
        if self.epoch < 10:
            self.to_use = 2 if self.trimin else 1
            self.cutt_off = 0.1 + (0.04 * self.epoch)
        else:
            self.to_use = 7 if self.trimin else 5
            self.cutt_off = (0.15 * self.epoch) - 0.9

    def __len__(self):
        return len(self.filenames)

    def _rngs(self, index):
        """Per-item generators (the reference uses the worker-global `random` / torch RNG; a private
        stream per (seed, epoch, index) keeps the draws reproducible under any thread schedule)."""
        s = (self.seed * 1000003 + self.epoch) * 1000003 + index
        return random.Random(s), torch.Generator().manual_seed(s & 0x7FFFFFFFFFFFFFFF)

    def _exists(self, path):
        hit = self._on_disk.get(path)
        if hit is None:
            hit = self._on_disk[path] = os.path.isfile(path)
        return hit

    def select_frames(self, index, rng, exists=None):
        """Frame-set selection of `__getitem__` (mono_dataset.py:77-106) - host integers only.
        Returns (do_color_aug, do_flip, folder, frame_index, side, frame_idxs)."""
        exists = exists or self._exists
        do_color_aug = self.is_train and rng.random() > 0.5
        do_flip = self.is_train and rng.random() > 0.5
        line = self.filenames[index].split()
        baseline = line[-1] if (self.rand and self.is_train) else 0
        folder, frame_index, side = self.index_to_folder_and_frame_idx_kt(index)
        if self.is_train:
            if self.rand:
                frame_idxs = sorted([i for i in range(-self.to_use, self.to_use + 1)
                                     if (abs(i) * float(baseline)) <= self.cutt_off], key=abs)
                if max(frame_idxs) < 3:
                    frame_idxs.append(STEREO)
            else:
                frame_idxs = [0, 1, -1, STEREO]
            mini = rng.randint(1, 6) if rng.random() > 0.7 else 0
            pos = [i for i in range(1, 8 - mini) if exists(self.get_image_path_kt(self.kt_path, frame_index + i, side, folder))]
            neg = [abs(i) for i in range(-1, -8 + mini, -1)
                   if exists(self.get_image_path_kt(self.kt_path, frame_index + i, side, folder))]
            if not pos or not neg:
                raise ValueError("no neighbouring frame on disk for %r (the reference's max([]) raises too)"
                                 % self.filenames[index])
            limit = min(max(pos), max(neg))
        else:
            frame_idxs = [0]
            limit = 7
        frame_idxs = [x for x in frame_idxs if x != STEREO and abs(x) <= abs(limit)]
        if max(frame_idxs) < 3:
            frame_idxs.append(STEREO)
        return do_color_aug, do_flip, folder, frame_index, side, frame_idxs

    def frame_paths(self, index):
        """{frame id: file} the item `index` will read - the frame-set selection alone (its draws come from the item's own
        generator, so the loader can ask ahead of the worker that decodes the item: `FrameCache`)."""
        rng, _ = self._rngs(index)
        _, _, folder, frame_index, side, frame_idxs = self.select_frames(index, rng)
        return self._paths(folder, frame_index, side, frame_idxs)

    def _paths(self, folder, frame_index, side, frame_idxs):
        if not self.is_train:
            return {0: self.get_image_path_kt(self.kt_path, frame_index, side, folder)}
        other_side = {"r": "l", "l": "r"}[side]
        return {i: (self.get_image_path_kt(self.kt_path, frame_index, other_side, folder) if i == STEREO else
                    self.get_image_path_kt(self.kt_path, frame_index + i, side, folder)) for i in frame_idxs}

    def __getitem__(self, index):
        return self.getitem(index)

    def getitem(self, index, skip=()):
        """`skip`: frame ids whose decoded pixels the caller already holds (`FrameCache`): they are not read from disk,
        `item["images"][f]` is None for them; `item["paths"]` names every frame's file either way."""
        rng, gen = self._rngs(index)
        do_color_aug, do_flip, folder, frame_index, side, frame_idxs = self.select_frames(index, rng)
        item = {"frame_idxs": frame_idxs, "flip": bool(do_flip), "images": {}, "jitter": {}, "index": index}
        item["paths"] = self._paths(folder, frame_index, side, frame_idxs)
        for i, path in item["paths"].items():
            item["images"][i] = None if i in skip else self.loader(path)
        if self.is_train:
            item["K"], item["inv_K"] = self.load_intrinsic_kt(0)
        if do_color_aug:
            # `preprocess` (mono_dataset.py:193-205) calls the ColorJitter module once per full-resolution
            # frame (result deleted at :128-131) and then once per scale-0 frame: keep that draw order
            temporal = [f for f in item["images"] if f != STEREO]
            for f in temporal:
                draw_color_jitter(gen, self.brightness, self.contrast, self.saturation, self.hue)
            for f in temporal:
                item["jitter"][f] = draw_color_jitter(gen, self.brightness, self.contrast, self.saturation, self.hue)
        stereo_T = np.eye(4, dtype=np.float32)
        baseline_sign = -1 if do_flip else 1
        side_sign = -1 if side in ["l"] else 1
        stereo_T[0, 3] = side_sign * baseline_sign * 0.1
        item["stereo_T"] = stereo_T
        item["frames"] = torch.tensor([STEREO_ID if f == STEREO else f for f in frame_idxs])
        item["cutt_off"] = torch.tensor(self.cutt_off)
        item["to_use"] = torch.tensor(self.to_use)
        return item


class KITTIDataset(MonoDataset):
    def load_intrinsic_kt(self, scale):
        """kitti_dataset.py:14-23."""
        K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
        K[0, :] *= self.width // (2 ** scale)
        K[1, :] *= self.height // (2 ** scale)
        return K, np.linalg.pinv(K)

    def index_to_folder_and_frame_idx_kt(self, index):
        """kitti_dataset.py:25-40."""
        line = self.filenames[index].split()
        folder = line[0]
        frame_index = int(line[1]) if len(line) >= 3 else 0
        side = line[2] if len(line) >= 3 else None
        return folder, frame_index, side


class KITTIRAWDataset(KITTIDataset):
    def get_image_path_kt(self, data_path, frame_index, side, folder):
        """kitti_dataset.py:49-54 (always `.jpg`, like the reference)."""
        side_map = {"l": 2, "r": 3}
        return os.path.join(data_path, folder, "image_0{}/data".format(side_map[side]),
                            "{:010d}{}".format(frame_index, ".jpg"))


class FrameCache:
    """Decoded frames kept resident in HBM (uint8 HWC, as decoded).

    The reference decodes every JPEG a sample names, every time (mono_dataset.py:119-133): a KITTI frame is read as the
    target of its own sample, as frame -k..+k of its neighbours' and as the stereo partner of the other camera's - four
    times per epoch for the MD2 frame set, up to sixteen for the boosted one - and again every epoch.  JPEG decode is the
    host's whole job in this loader, and on a box whose cgroup grants 16 CPUs it tops out near 2 300 frames/s: 565
    images/s for MD2's four frames per sample against a 720 images/s step.  All ~45 000 distinct frames of the Eigen-Zhou
    split are 63 GB decoded; an MI355X has 288 GB.  So: a frame is decoded ONCE, its bytes stay where the collate
    kernels read them anyway, and every later use - same epoch or any later one - is a table entry.  The loader asks
    ahead (`MonoDataset.frame_paths`) which frames a batch needs and tells the decode workers to skip the resident ones.
    A batch's freshly decoded frames land in a scratch area at the end of the buffer (one DMA from the decode ring); the
    new ones among them are copied to a resident home (device to device).  No eviction: when the resident area is full,
    further frames are used from the scratch area and decoded again next time (what the reference always does)."""

    def __init__(self, device, capacity_bytes, scratch_bytes=192 << 20, regions=3):
        """`regions` scratch areas of `scratch_bytes` each: every `DeviceCollate` that uses the cache takes one of its own
        (`attach`) - two loaders iterated at the same time (training + validation, each with its own thread and stream) never
        upload into the same bytes."""
        import threading
        self.device = torch.device(device)
        self.scratch_bytes = int(scratch_bytes)
        self.regions = max(1, int(regions))
        self.capacity = max(int(capacity_bytes), 0)
        self.buf = torch.empty(self.capacity + self.regions * self.scratch_bytes, dtype=torch.uint8, device=self.device)
        self.index = {}                 # path -> (byte offset, h, w)
        self.used = 0
        self.hits = self.misses = self.passed_through = self.oversized = 0
        self._held = set()              # scratch regions in use
        self._lock = threading.Lock()   # admit / index / used are shared by the loaders' threads
        self._ready = None              # event behind the last resident copy: a reader on another stream waits for it

    def __contains__(self, path):
        return path in self.index

    def attach(self, owner):
        """Byte offset of a scratch area of this cache for the exclusive use of `owner` (a collate) for as long as it lives:
        the area returns to the cache when the owner is garbage-collected (the trainer builds a new collate every epoch)."""
        import weakref
        with self._lock:
            free = [i for i in range(self.regions) if i not in self._held]
        if not free:
            import gc
            gc.collect()                # a finished loader's collate may only be waiting for the cycle collector
        with self._lock:
            free = [i for i in range(self.regions) if i not in self._held]
            if not free:
                raise RuntimeError("FrameCache: more than %d collates alive on a cache built with that many scratch regions"
                                   % self.regions)
            self._held.add(free[0])
            weakref.finalize(owner, self._release, free[0])
            return self.capacity + free[0] * self.scratch_bytes

    def _release(self, region):
        with self._lock:
            self._held.discard(region)

    def wait_ready(self):
        """Make the current stream wait for the resident copies other streams have enqueued (a frame admitted by another
        loader's stream may be read as resident here before its copy has run otherwise)."""
        ev = self._ready
        if ev is not None and self.device.type == "cuda":
            torch.cuda.current_stream(self.device).wait_event(ev)

    def mark_ready(self):
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._ready = ev

    def admit(self, path, src_off, h, w):
        """A freshly decoded frame sits at byte `src_off` of the buffer (inside the caller's scratch area): give it a
        resident home if it is new and there is room (one device-to-device copy, stream-ordered behind the upload), and say
        where this batch's kernels read it.  A frame another in-flight batch made resident in the meantime takes no second
        slot.  Thread-safe: slot assignment and the index are guarded."""
        n = h * w * 3
        with self._lock:
            hit = self.index.get(path)
            if hit is not None:
                return hit[0]
            if self.used + n > self.capacity:
                self.passed_through += 1
                return src_off
            dst = self.used
            self.used += n
            self.buf[dst:dst + n].copy_(self.buf[src_off:src_off + n], non_blocking=True)
            self.index[path] = (dst, h, w)
            return dst

    def stats(self):
        return {"frames": len(self.index), "resident_GB": round(self.used / 1e9, 3), "hits": self.hits, "misses": self.misses,
                "passed_through": self.passed_through, "oversized_batches": self.oversized}


class DeviceCollate:
    """Recipes of one batch -> the dict `Trainer.custom_collate` returns, built on the device.

    Every decoded frame of the batch is packed into one pinned uint8 buffer and uploaded with a single
    copy; job tables then route (item, frame) -> row of `("color", f, 0)` / `("color_aug", f, 0)` /
    `("color", 0, s)`.  Rows follow the reference's stacking order: the items that have the key, in batch
    order."""

    def __init__(self, height, width, scales, device, backend=None, ring=3, pack_threads=8, canonical=True, cache=None):
        self.height, self.width, self.scales = height, width, list(scales)
        self.cache = cache                 # FrameCache: decoded frames stay in HBM, a frame is decoded once (None: off)
        self._scratch = cache.attach(self) if cache is not None else 0      # this collate's own scratch area of the cache
        # training batches are stacked in `plan.canonical_permutation` order (largest frame offset first): the loader's
        # order is a random shuffle anyway, and the trainer then meets far fewer distinct batch signatures
        self.canonical = bool(canonical)
        self.device = torch.device(device)
        self.pipe = imageops.ImagePipeline(self.device, backend)
        self._ring, self._next = [None] * ring, 0
        self._pack_pool = ThreadPoolExecutor(max_workers=pack_threads)

    def _staging(self, nbytes):
        """Next pinned staging buffer of the ring (grow-only); waits until the upload that last used it
        has finished.  Host tensors (CPU test tier) are plain memory and need no event."""
        slot = self._next
        self._next = (self._next + 1) % len(self._ring)
        entry = self._ring[slot]
        if entry is not None and entry[1] is not None:
            entry[1].synchronize()
        if entry is None or entry[0].numel() < nbytes:
            buf = torch.empty(int(nbytes * 1.25) + 1, dtype=torch.uint8)
            if self.device.type == "cuda":
                buf = buf.pin_memory()
            entry = (buf, torch.cuda.Event() if self.device.type == "cuda" else None)
            self._ring[slot] = entry
        return entry

    def __call__(self, batch):
        H, W, dev = self.height, self.width, self.device
        train = "K" in batch[0]
        # ---- host: which keys exist, and which row each (item, frame) owns (trainer.py:867-886)
        max_frames = [int(torch.max(item["frames"]).item()) for item in batch]
        if train and self.canonical:
            order = canonical_permutation(max_frames)
            batch, max_frames = [batch[i] for i in order], [max_frames[i] for i in order]
        out = {}
        if train:
            out["ordering"] = [[0, STEREO] if m == 0 else [0, m, -m] for m in max_frames]
            top = max(max_frames)
            if top == 0:
                frame_ids = [0, STEREO]
            else:
                frame_ids = list(range(-top, top + 1))
                if any(m in (0, 1, 2) for m in max_frames):
                    frame_ids.append(STEREO)
        else:
            frame_ids = [0]
        # ---- one upload of all decoded frames (packed into a recycled pinned buffer by the pack pool); with a FrameCache
        #      the upload lands IN the cache buffer, frames that are resident there already were not decoded at all
        cache = self.cache
        entries = [(b, f) for b, item in enumerate(batch) for f in item["images"] if f in frame_ids]
        entries.sort(key=lambda e: e[1] != 0)           # target frames first: the pyramid reads rows [0, B)
        fresh = [e for e in entries if batch[e[0]]["images"][e[1]] is not None]
        ring = next((item["_ring"] for item in batch if "_ring" in item), None)
        if ring is not None:
            # frames already sit in a shared, host-registered ring slot (written there by the worker process):
            # one DMA straight from it, no staging copy
            buf, used, done = ring
            fresh_off = {e: batch[e[0]]["_offsets"][e[1]] for e in fresh}
            host_src = buf[:used]
        else:
            sizes = [batch[b]["images"][f].size for b, f in fresh]
            offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
            used = int(offs[-1])
            staging, done = self._staging(max(used, 1))
            flat = staging.numpy()

            def pack(k):
                b, f = fresh[k]
                flat[offs[k]:offs[k + 1]] = batch[b]["images"][f].reshape(-1)
            list(self._pack_pool.map(pack, range(len(fresh))))
            fresh_off = {e: int(offs[k]) for k, e in enumerate(fresh)}
            host_src = staging[:used]
        if cache is None:
            assert len(fresh) == len(entries), "frames were skipped but there is no FrameCache to take them from"
            src = host_src.to(dev, non_blocking=True)
            if done is not None and used:
                done.record()
            jobs = []
            for b, f in entries:
                h, w = batch[b]["images"][f].shape[:2]
                jobs.append((int(fresh_off[(b, f)]), int(h), int(w), batch[b]["flip"]))
            level0 = self.pipe.resize(src, jobs, H, W)                       # uint8 [n_img, H, W, 3]
        else:
            src, base = cache.buf, self._scratch
            cache.wait_ready()
            # the batch's fresh frames normally fit this collate's scratch area: ONE upload.  A batch that does not (a cold or
            # full cache under the boosted recipe: 120-168 full-resolution frames, 170-235 MB) goes through it in several
            # spans of the staging buffer - upload, admit, resize, next span (stream-ordered, so the area is free again)
            spans, lo = [], 0
            for e in sorted(fresh, key=fresh_off.__getitem__):
                end = fresh_off[e] + batch[e[0]]["images"][e[1]].size
                if end - lo > cache.scratch_bytes:
                    if fresh_off[e] == lo:
                        raise RuntimeError("FrameCache scratch area smaller than one frame (%d bytes)" % (end - lo))
                    spans.append((lo, fresh_off[e]))
                    lo = fresh_off[e]
            spans.append((lo, used))
            if len(spans) > 1:
                cache.oversized += 1
            parts, order = [], []
            for k, (lo, hi) in enumerate(spans):
                if hi > lo:
                    src[base:base + hi - lo].copy_(host_src[lo:hi], non_blocking=True)
                if len(spans) == 1:
                    group = list(range(len(entries)))
                else:
                    group = [i for i, e in enumerate(entries)
                             if batch[e[0]]["images"][e[1]] is not None and lo <= fresh_off[e] < hi]
                    if k == len(spans) - 1:         # resident frames can be read at any time: with the last span
                        group += [i for i, e in enumerate(entries) if batch[e[0]]["images"][e[1]] is None]
                jobs = []
                for i in group:
                    b, f = entries[i]
                    img = batch[b]["images"][f]
                    if img is not None:
                        h, w = img.shape[:2]
                        cache.misses += 1
                        off = cache.admit(batch[b]["paths"][f], base + fresh_off[(b, f)] - lo, int(h), int(w))
                    else:
                        off, h, w = cache.index[batch[b]["paths"][f]]
                        cache.hits += 1
                    jobs.append((int(off), int(h), int(w), batch[b]["flip"]))
                parts.append(self.pipe.resize(src, jobs, H, W))
                order += group
                if len(spans) > 1:
                    self.pipe.flush()              # this span's kernels are enqueued before the next upload overwrites it
            if done is not None and used:
                done.record()
            cache.mark_ready()
            if len(parts) == 1 and order == list(range(len(entries))):
                level0 = parts[0]
            else:
                inverse = [0] * len(entries)
                for row, i in enumerate(order):
                    inverse[i] = row
                level0 = torch.cat(parts, 0).index_select(0, imageops._upload(np.asarray(inverse, dtype=np.int64), dev))
        where = {e: i for i, e in enumerate(entries)}
        # ---- ("color", f, 0) and ("color_aug", f, 0)
        for f in frame_ids:
            owners = [b for b, item in enumerate(batch) if f in item["images"]]
            if not owners:
                continue
            picks = [where[(b, f)] for b in owners]
            color = torch.empty(len(owners), 3, H, W, device=dev)
            self.pipe.to_float(level0, picks, color, list(range(len(owners))))
            out[("color", f, 0)] = color
            if f == STEREO:
                continue
            aug = torch.empty(len(owners), 3, H, W, device=dev)
            jit = [(r, b) for r, b in enumerate(owners) if f in batch[b]["jitter"]]
            plain = [(r, b) for r, b in enumerate(owners) if f not in batch[b]["jitter"]]
            self.pipe.jitter_to_float(level0, [where[(b, f)] for _, b in jit], [batch[b]["jitter"][f] for _, b in jit],
                                      aug, [r for r, _ in jit])
            self.pipe.to_float(level0, [where[(b, f)] for _, b in plain], aug, [r for r, _ in plain])
            out[("color_aug", f, 0)] = aug
        # ---- pyramid of the target frame: ("color", 0, s), s >= 1, chained like Resize[s](scale s-1)
        if max(self.scales) > 0:
            owners = [b for b, item in enumerate(batch) if 0 in item["images"]]
            assert [where[(b, 0)] for b in owners] == list(range(len(owners)))
            level = level0[:len(owners)]
            for s in range(1, max(self.scales) + 1):
                level = self.pipe.halve(level)
                if s in self.scales:
                    t = torch.empty(len(owners), 3, H >> s, W >> s, device=dev)
                    self.pipe.to_float(level, list(range(len(owners))), t, list(range(len(owners))))
                    out[("color", 0, s)] = t
        if train:
            cams = imageops._upload(np.stack([np.stack([item[k] for item in batch]) for k in ("K", "inv_K", "stereo_T")]),
                                    dev)                                # [3,B,4,4], one asynchronous upload
            out[("K", 0)], out[("inv_K", 0)], out["stereo_T"] = cams[0], cams[1], cams[2]
            out["frames"] = frame_ids
            out["cutt"] = batch[0]["cutt_off"]
            out["to_use"] = batch[0]["to_use"]
        self.pipe.flush()
        return out


_RING = None        # set in the parent right before the workers are forked; inherited by them


class _ShmRing:
    """`slots` shared-memory byte buffers, allocated once per epoch by the parent and (on a GPU) registered
    with the HIP runtime as pinned host memory.  Worker processes decode straight into a slot, the parent
    DMAs from it: no per-batch shared-memory segments (their first-touch page faults cost ~15 ms per 67 MB
    batch), no staging copy, and only metadata on the result queue."""

    def __init__(self, slots, capacity, device):
        self.capacity = int(capacity)
        self.buffers = [torch.empty(self.capacity, dtype=torch.uint8).share_memory_() for _ in range(slots)]
        self.events, self.registered = [None] * slots, []
        if torch.device(device).type == "cuda":
            rt = torch.cuda.cudart()
            for b in self.buffers:
                if int(rt.cudaHostRegister(b.data_ptr(), self.capacity, 0)) != 0:
                    break
                self.registered.append(b.data_ptr())
            if len(self.registered) == slots:
                self.events = [torch.cuda.Event() for _ in range(slots)]

    def close(self):
        if self.registered:
            torch.cuda.synchronize()
            rt = torch.cuda.cudart()
            for p in self.registered:
                rt.cudaHostUnregister(p)
            self.registered = []


class _WorkerView(torch.utils.data.Dataset):
    def __init__(self, dataset):
        self.dataset = dataset

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, key):
        index, slot, skip = key
        item = self.dataset.getitem(index, skip)
        item["_slot"] = slot
        return item


def _pack_batch(items):
    """Runs inside a worker process: every decoded frame of the batch is written into the batch's ring
    slot; what travels back on the result queue is plain Python (shapes, offsets, draws) - no tensors,
    hence no shared-memory handles (each costs the parent a socket handshake: 48 per batch were 25 ms)."""
    buf = _RING.buffers[items[0]["_slot"]].numpy()
    off = 0
    for item in items:
        shapes, offsets = {}, {}
        for f, a in item["images"].items():
            if a is None:                  # resident in the parent's FrameCache: not decoded, nothing to ship
                shapes[f], offsets[f] = None, -1
                continue
            n = a.size
            if off + n > buf.size:
                raise RuntimeError("loader ring slot too small: %d bytes needed" % (off + n))
            buf[off:off + n] = a.reshape(-1)
            shapes[f], offsets[f] = a.shape, off
            off += n
        item["images"], item["_offsets"] = shapes, offsets
        item["frames"] = item["frames"].tolist()
        item["cutt_off"] = float(item["cutt_off"])
        item["to_use"] = int(item["to_use"])
    items[0]["_used"] = off
    return items


def _unpack_batch(ring, items):
    """Parent side of `_pack_batch`: frames become views of the ring slot again."""
    slot = items[0]["_slot"]
    flat = ring.buffers[slot].numpy()
    for item in items:
        item["images"] = {f: (None if shape is None else
                              flat[item["_offsets"][f]:item["_offsets"][f] + int(np.prod(shape))].reshape(shape))
                          for f, shape in item["images"].items()}
        item["frames"] = torch.tensor(item["frames"])
        item["cutt_off"] = torch.tensor(item["cutt_off"])
        item["to_use"] = torch.tensor(item["to_use"])
    items[0]["_ring"] = (ring.buffers[slot], items[0]["_used"], ring.events[slot])
    return items


class DeviceLoader:
    """Iterates a dataset in batches: JPEG decode on `num_workers` host workers, prefetching `prefetch`
    batches ahead; collation on the device.  Stands in for `DataLoader(dataset, batch_size, shuffle,
    collate_fn=custom_collate, num_workers, drop_last)` of trainer.py:218-220.

    workers="process" (default for training): worker processes, frames returned through /dev/shm - decode
    scales with the core count (one Python thread decodes ~480 KITTI frames/s; threads stop scaling at
    ~1 700 frames/s under the GIL, a 578 images/s MD2 step consumes 2 300).  workers="thread": a thread
    pool inside this process (small runs, tests)."""

    def __init__(self, dataset, batch_size, collate, shuffle=True, drop_last=True, num_workers=8, prefetch=2, seed=0,
                 workers="thread", bytes_per_sample=None, background=True, rank=0, world=1):
        self.dataset, self.batch_size, self.collate, self.background = dataset, batch_size, collate, background
        assert 0 <= rank < world
        self.rank, self.world = rank, world      # data parallel: this loader serves shard `rank` of `world`
        if bytes_per_sample is None:                  # ring-slot sizing: the frames one sample can have, KITTI-sized
            frames = 2 * getattr(dataset, "to_use", 7) + 2 if getattr(dataset, "is_train", True) else 1
            bytes_per_sample = frames * 1300 * 400 * 3
        self.bytes_per_sample = bytes_per_sample
        self.shuffle, self.drop_last, self.num_workers, self.prefetch, self.seed = shuffle, drop_last, num_workers, prefetch, seed
        self.workers = workers if num_workers > 0 else "thread"
        # where the wall time of an epoch went, seconds: producer thread waiting for decoded items / planning + launching the
        # collate / blocked on a full hand-over queue (the consumer is the slower side); consumer waiting for a batch (the
        # producer is the slower side)
        self.stats = {"fetch_s": 0.0, "collate_s": 0.0, "producer_blocked_s": 0.0, "consumer_waited_s": 0.0, "batches": 0}

    def __len__(self):
        n = len(self.dataset) // self.world
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _batches(self):
        order = list(range(len(self.dataset)))
        if self.shuffle:
            random.Random(self.seed * 7919 + self.dataset.epoch).shuffle(order)
        if self.world > 1:
            # one common shuffle (same seed on every rank), then disjoint equal-length shards: rank r takes
            # indices r, r+world, ... of the first (n // world) * world entries
            usable = (len(order) // self.world) * self.world
            order = order[self.rank:usable:self.world]
        for i in range(0, len(order), self.batch_size):
            chunk = order[i:i + self.batch_size]
            if len(chunk) == self.batch_size or not self.drop_last:
                yield chunk

    def _skip(self, index):
        """Frame ids of item `index` that are resident in the collate's FrameCache (asked at the moment the item is handed
        to a worker; a frame that becomes resident a moment later is decoded once more and ignored)."""
        cache = getattr(self.collate, "cache", None)
        if cache is None or not cache.index:
            return ()
        return tuple(f for f, path in self.dataset.frame_paths(index).items() if path in cache.index)

    def _keyed_batches(self, slots):
        # a generator: the DataLoader pulls a batch's keys when it dispatches the batch, `prefetch` batches ahead
        for k, chunk in enumerate(self._batches()):
            yield [(i, k % slots, self._skip(i)) for i in chunk]

    def __iter__(self):
        """Batches in order.  On a GPU the parent-side work (fetching from the workers, planning and launching
        the collate kernels) runs in a background thread on its own HIP stream, `prefetch` device batches
        ahead, so it overlaps the training thread's launch work instead of adding to it (the MD2 step is
        ~16 ms of host-side launches for 21 ms of GPU time: there is no slack to spend inline)."""
        device = torch.device(getattr(self.collate, "device", "cpu"))
        if device.type != "cuda" or not self.background:
            yield from self._iterate()
            return
        import queue
        import threading
        ready = queue.Queue(maxsize=max(1, self.prefetch))
        stream = torch.cuda.Stream(device=device)
        stop = threading.Event()

        def produce():
            try:
                torch.cuda.set_device(device)
                with torch.cuda.stream(stream):
                    for batch in self._iterate():
                        ev = torch.cuda.Event()
                        ev.record(stream)
                        t0 = time.perf_counter()
                        while not stop.is_set():
                            try:
                                ready.put((batch, ev), timeout=0.1)
                                break
                            except queue.Full:
                                continue
                        self.stats["producer_blocked_s"] += time.perf_counter() - t0
                        if stop.is_set():
                            return
                ready.put((None, None))
            except BaseException as e:                      # surface worker / kernel errors in the consumer
                ready.put((e, None))

        thread = threading.Thread(target=produce, daemon=True)
        thread.start()
        try:
            while True:
                t0 = time.perf_counter()
                batch, ev = ready.get()
                self.stats["consumer_waited_s"] += time.perf_counter() - t0
                if batch is None:
                    return
                if isinstance(batch, BaseException):
                    raise batch
                cur = torch.cuda.current_stream(device)
                cur.wait_event(ev)
                for v in batch.values():                    # allocated on the loader stream, consumed on this one
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
                yield batch
        finally:
            stop.set()
            thread.join(timeout=30)

    def _iterate(self):
        if self.workers == "process":
            global _RING
            depth = max(2, self.prefetch) * self.num_workers         # batches the workers may run ahead
            slots = depth + 3
            device = getattr(self.collate, "device", "cpu")
            ring = _ShmRing(slots, self.batch_size * self.bytes_per_sample, device)
            _RING = ring                                              # inherited by the forked workers
            keyed = self._keyed_batches(slots)
            inner = torch.utils.data.DataLoader(_WorkerView(self.dataset), batch_sampler=keyed, collate_fn=_pack_batch,
                                                num_workers=self.num_workers, prefetch_factor=max(2, self.prefetch),
                                                persistent_workers=False)
            try:
                uploads = collections.deque()
                it = iter(inner)
                while True:
                    t0 = time.perf_counter()
                    items = next(it, None)
                    t1 = time.perf_counter()
                    if items is None:
                        break
                    batch = self.collate(_unpack_batch(ring, items))
                    self.stats["fetch_s"] += t1 - t0
                    self.stats["collate_s"] += time.perf_counter() - t1
                    self.stats["batches"] += 1
                    # a slot is rewritten `slots` = depth + 3 batches later and the workers run at most `depth` batches ahead
                    # of this loop: the slot of the batch the workers may start once this one is handed over belongs to the
                    # batch collated two iterations ago - its upload must have finished; nothing younger is waited for, so
                    # this thread stays ahead of the GPU (which it shares with the training step)
                    uploads.append(items[0]["_ring"][2])
                    if len(uploads) > 2:
                        old = uploads.popleft()
                        if old is not None:
                            old.synchronize()
                    yield batch
            finally:
                del inner
                _RING = None
                ring.close()
            return
        with ThreadPoolExecutor(max_workers=max(1, self.num_workers)) as pool:
            pending = []
            batches = self._batches()

            def submit():
                chunk = next(batches, None)
                if chunk is not None:
                    pending.append([pool.submit(self.dataset.getitem, i, self._skip(i)) for i in chunk])

            for _ in range(self.prefetch + 1):
                submit()
            while pending:
                futures = pending.pop(0)
                submit()
                yield self.collate([f.result() for f in futures])
