"""Device image pipeline of the loader: job tables for the byte kernels of
csrc/bbd_image.hip.

Replaces the per-item Pillow/torchvision work of the reference loader (datasets/mono_dataset.py:186-205:
`Resize(LANCZOS)` chain, `ColorJitter`, `ToTensor`) - the host only decodes JPEGs.  Results are written
straight into rows of the collated batch tensors (no per-item tensors, no `torch.stack`).

`resample_table` restates Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc` (Resample.c): it is
host-side table construction (a few thousand `sin` calls per distinct (in, out) size pair, cached); the
per-pixel arithmetic that consumes the tables runs in the kernels.
"""
import functools
import math
import struct

import numpy as np
import torch

from . import ops
from ._lib import RESAMPLE_JOB, RESAMPLE_FLIP, JITTER_JOB, CONVERT_JOB, ptr

PRECISION_BITS = 22
BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    """Resample.c lanczos_filter, support 3."""
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


@functools.lru_cache(maxsize=64)
def resample_table(in_size, out_size):
    """Fixed-point LANCZOS taps for resizing an axis of `in_size` to `out_size`:
    (coef int32 [out, ksize], bounds int32 [out, 2] = (first source index, tap count), ksize)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    coef = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    inv = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_lanczos((x + xmin - center + 0.5) * inv) for x in range(xmax)]
        total = 0.0
        for v in w:
            total += v
        if total != 0.0:
            w = [v / total for v in w]
        for x, v in enumerate(w):
            coef[xx, x] = int(v * one - 0.5) if v < 0 else int(v * one + 0.5)
        bounds[xx] = (xmin, xmax)
    return coef, bounds, ksize


def _split64(v):
    lo = v & 0xFFFFFFFF
    return (lo - (1 << 32) if lo >= (1 << 31) else lo), v >> 32


def float_bits(f):
    return struct.unpack("<i", struct.pack("<f", float(f)))[0]


def hue_offset(hue_factor):
    """np.uint8(hue_factor * 255) as torchvision's adjust_hue adds it (C cast: truncate, wrap)."""
    return int(hue_factor * 255) & 0xFF


class TableBank:
    """Coefficient tables of every (in, out) size pair seen so far, concatenated; lives for the life of
    the pipeline and is re-uploaded only when a new pair appears (a handful of KITTI image sizes)."""

    def __init__(self):
        self.coef, self.bounds, self.index = [], [], {}
        self.coef_len = self.bounds_len = 0
        self.device_tensors = None

    def get(self, in_size, out_size):
        key = (in_size, out_size)
        if key not in self.index:
            c, b, k = resample_table(in_size, out_size)
            self.index[key] = (self.coef_len, self.bounds_len, k)
            self.coef.append(c.ravel())
            self.bounds.append(b.ravel())
            self.coef_len += c.size
            self.bounds_len += b.size
            self.device_tensors = None
        return self.index[key]

    def tensors(self, device):
        if self.device_tensors is None:
            self.device_tensors = (_upload(np.concatenate(self.coef), device), _upload(np.concatenate(self.bounds), device))
        return self.device_tensors


def _upload(array, device):
    """Host array -> device tensor without blocking the calling thread on the stream (pinned staging + asynchronous copy;
    the caching host allocator keeps the staging block until the copy has run).  The loader's producer thread plans and
    launches batches ahead of the training step: a pageable copy would make it wait for its stream - which shares the GPU
    with a replaying step graph - three times per batch."""
    t = torch.from_numpy(np.ascontiguousarray(array))
    if torch.device(device).type != "cuda":
        return t
    return t.pin_memory().to(device, non_blocking=True)


class _Slot:
    """Placeholder for an address that exists only after `flush()` uploaded the tables."""

    def __init__(self, kind, offset=0):
        self.kind, self.offset = kind, offset


class ImagePipeline:
    """Plans the image kernels of a batch, then launches them: every call below only *records* work
    (job tables go into one host arena); `flush()` uploads the arena with a single copy and enqueues the
    launches in order on the current stream of `device` (or runs them through the CPU test port passed
    as `backend`).  Output tensors are allocated at planning time and are valid after `flush()`."""

    def __init__(self, device, backend=None):
        self.device = torch.device(device)
        self.backend = backend or ops.default_backend()
        self.bank = TableBank()
        self._arena, self._arena_len, self._pending, self._keep = [], 0, [], []

    def _table(self, rows, width):
        arr = np.ascontiguousarray(np.array(rows, np.int32).reshape(-1, width))
        off = self._arena_len
        self._arena.append(arr.ravel())
        self._arena_len += arr.size
        return _Slot("arena", off)

    def _launch(self, name, anchor, *args):
        self._pending.append((name, anchor, args))

    def flush(self):
        if not self._pending:
            return
        arena = _upload(np.concatenate(self._arena), self.device)
        coef, bounds = self.bank.tensors(self.device) if self.bank.coef else (None, None)
        base = {"arena": arena, "coef": coef, "bounds": bounds}
        import ctypes
        for name, anchor, args in self._pending:
            real = [ctypes.c_void_p(base[a.kind].data_ptr() + 4 * a.offset) if isinstance(a, _Slot) else a for a in args]
            self.backend.run(name, anchor, *real)
        self._keep = [arena]                       # stays alive until the next flush (kernels are async)
        self._arena, self._arena_len, self._pending = [], 0, []

    # ---------------------------------------------------------------- resize
    def resize(self, src, images, out_h, out_w):
        """`src`: uint8 buffer holding HWC images; `images`: list of (byte offset, h, w, flip).
        Returns uint8 [n, out_h, out_w, 3] = PIL `img.transpose(FLIP_LEFT_RIGHT)?.resize((out_w, out_h),
        LANCZOS)` of each (horizontal pass first, like ImagingResample)."""
        n = len(images)
        out = torch.empty(n, out_h, out_w, 3, dtype=torch.uint8, device=self.device)
        hjobs, vjobs, tmp_off = [], [], 0
        for i, (off, h, w, flip) in enumerate(images):
            dst_off = i * out_h * out_w * 3
            need_h = (w != out_w) or flip
            need_v = h != out_h
            v_src = None
            if need_h:
                co, bo, k = self.bank.get(w, out_w)
                h_dst = (tmp_off if need_v else dst_off)
                hjobs.append((need_v, _split64(off) + _split64(h_dst) + (h, w, out_w, k, co, bo,
                                                                        RESAMPLE_FLIP if flip else 0, 0)))
                if need_v:
                    v_src = ("tmp", tmp_off)
                    tmp_off += h * out_w * 3
            else:
                v_src = ("src", off)
            if need_v:
                co, bo, k = self.bank.get(h, out_h)
                vjobs.append((v_src[0], _split64(v_src[1]) + _split64(dst_off) + (h, out_w, out_h, k, co, bo, 0, 0)))
            elif not need_h:                       # same size, no flip: plain copy
                nbytes = h * w * 3
                out.view(-1)[dst_off:dst_off + nbytes] = src.view(-1)[off:off + nbytes]
        tmp = torch.empty(max(tmp_off, 1), dtype=torch.uint8, device=self.device)
        self._keep.append(tmp)
        for to_tmp in (True, False):
            jobs = [j for t, j in hjobs if t == to_tmp]
            if jobs:
                self._launch("bbd_resample_h_u8", out, ptr(src), ptr(tmp if to_tmp else out),
                             self._table(jobs, RESAMPLE_JOB), len(jobs), max(j[4] for j in jobs), _Slot("coef"),
                             _Slot("bounds"), 3)
        for origin in ("tmp", "src"):
            jobs = [j for o, j in vjobs if o == origin]
            if jobs:
                self._launch("bbd_resample_v_u8", out, ptr(tmp if origin == "tmp" else src), ptr(out),
                             self._table(jobs, RESAMPLE_JOB), len(jobs), out_h, out_w * 3, _Slot("coef"),
                             _Slot("bounds"), 3)
        return out

    def halve(self, level):
        """Next pyramid level: uint8 [n,h,w,3] -> [n,h//2,w//2,3], `Resize((h//2, w//2), LANCZOS)` of the
        previous level (mono_dataset.py:190-191 chains scale s from scale s-1)."""
        n, h, w, _ = level.shape
        images = [(i * h * w * 3, h, w, False) for i in range(n)]
        return self.resize(level, images, h // 2, w // 2)

    # ---------------------------------------------------------------- ToTensor / ColorJitter
    def to_float(self, images_u8, picks, dst, rows):
        """dst[rows[k]] ([3,H,W] fp32) = images_u8[picks[k]] / 255, HWC -> CHW (torchvision ToTensor)."""
        if not picks:
            return
        _, H, W, _ = images_u8.shape
        assert dst.is_contiguous() and dst.shape[1:] == (3, H, W)
        jobs = [_split64(p * H * W * 3) + _split64(r * 3 * H * W) for p, r in zip(picks, rows)]
        self._launch("bbd_u8_to_float_chw", dst, ptr(images_u8), ptr(dst), self._table(jobs, CONVERT_JOB), len(jobs),
                     H, W)

    def jitter_to_float(self, images_u8, picks, params, dst, rows):
        """dst[rows[k]] = ToTensor(ColorJitter with params[k] applied to images_u8[picks[k]]).
        params[k] = list of (op, factor) in application order (op: BRIGHTNESS..HUE)."""
        if not picks:
            return
        _, H, W, _ = images_u8.shape
        assert dst.is_contiguous() and dst.shape[1:] == (3, H, W)
        jobs = []
        for p, r, seq in zip(picks, rows, params):
            assert len(seq) <= 4
            opcodes = [op for op, _ in seq] + [-1] * (4 - len(seq))
            bits = [hue_offset(f) if op == HUE else float_bits(f) for op, f in seq] + [0] * (4 - len(seq))
            jobs.append(_split64(p * H * W * 3) + _split64(r * 3 * H * W) + tuple(opcodes) + tuple(bits))
        scratch = torch.empty(len(jobs), dtype=torch.int32, device=self.device)
        self._keep.append(scratch)
        self._launch("bbd_color_jitter_u8", dst, ptr(images_u8), ptr(dst), self._table(jobs, JITTER_JOB), len(jobs), H,
                     W, ptr(scratch))
