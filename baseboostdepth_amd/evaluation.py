"""Validation metrics on the device.

`GroundTruthSet` keeps a whole split's ground-truth depth maps (ragged sizes) resident in HBM as
one buffer plus an int descriptor table, so scoring an image is one kernel launch with no host
round trip - the reference re-uploads the map, builds the crop mask in numpy, gathers with it and
sorts twice per image (trainer.py:594-617, evaluate_depth.py:244-297).

`depth_metrics` is the single entry: one `bbd_depth_metrics` launch for a batch of predictions.
"""
import numpy as np
import torch

from . import ops
from ._lib import (EVAL_DESC, EVAL_OUT, EVAL_PRED_IS_DISP, EVAL_MEDIAN_MIDPOINT, EVAL_NO_MEDIAN_SCALING, ptr)

METRIC_NAMES = ["de/abs_rel", "de/sq_rel", "de/rms", "de/log_rms", "da/a1", "da/a2", "da/a3"]   # trainer.py:156
GARG_CROP = (0.40810811, 0.99189189, 0.03594771, 0.96405229)     # trainer.py:603-604


def garg_window(gh, gw):
    """[r0,r1) x [c0,c1) exactly as the reference computes it (float64 products truncated to int32)."""
    c = np.array([GARG_CROP[0] * gh, GARG_CROP[1] * gh, GARG_CROP[2] * gw, GARG_CROP[3] * gw]).astype(np.int32)
    return int(c[0]), int(c[1]), int(c[2]), int(c[3])


class GroundTruthSet:
    """Ragged ground-truth depth maps of a split, packed once into device memory."""

    def __init__(self, gt_depths, device, crop=True):
        maps = [np.ascontiguousarray(np.asarray(g, dtype=np.float32)) for g in gt_depths]
        assert all(m.ndim == 2 for m in maps)
        desc = np.zeros((len(maps), EVAL_DESC), dtype=np.int32)
        off = 0
        for i, m in enumerate(maps):
            gh, gw = m.shape
            win = garg_window(gh, gw) if crop else (0, gh, 0, gw)
            desc[i] = (off & 0xFFFFFFFF if off < 2 ** 31 else (off & 0xFFFFFFFF) - 2 ** 32, off >> 32, gh, gw) + win
            off += m.size
        flat = np.concatenate([m.ravel() for m in maps]) if maps else np.zeros(0, np.float32)
        self.shapes = [m.shape for m in maps]
        self.buffer = torch.from_numpy(flat).to(device)
        self.desc = torch.from_numpy(desc).to(device)

    def __len__(self):
        return len(self.shapes)


def depth_metrics(pred, gts, indices, min_depth=1e-3, max_depth=80.0, clamp=(1e-3, 80.0), pred_is_disp=False,
                  median="torch", median_scaling=True, scale_factor=1.0, backend=None):
    """Scores pred[i] ([n,1,h,w] or [n,h,w]) against gts[indices[i]]; returns a [n, 12] device tensor
    (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3, ratio, median_gt, median_pred, count, 0)."""
    backend = backend or ops.default_backend()
    pred = pred.detach()
    if pred.dim() == 4:
        assert pred.shape[1] == 1
        pred = pred[:, 0]
    pred = pred.contiguous().float()
    backend._check(pred, gts.buffer)
    n, h, w = pred.shape
    idx = torch.as_tensor(indices, dtype=torch.long, device=gts.desc.device).view(-1)
    assert idx.numel() == n
    desc = gts.desc.index_select(0, idx).contiguous()
    out = torch.empty(n, EVAL_OUT, device=pred.device, dtype=torch.float32)
    flags = (EVAL_PRED_IS_DISP if pred_is_disp else 0) | (EVAL_MEDIAN_MIDPOINT if median == "numpy" else 0) | \
            (0 if median_scaling else EVAL_NO_MEDIAN_SCALING)
    backend.run("bbd_depth_metrics", pred, ptr(pred), ptr(gts.buffer), ptr(desc), ptr(out), n, h, w,
                float(min_depth), float(max_depth), float(clamp[0]), float(clamp[1]), float(scale_factor), flags)
    return out


# ---------------------------------------------------------------------------- evaluate_depth.py
STEREO_SCALE_FACTOR = 5.4          # evaluate_depth.py:45


def evaluate(opt, dataloader=None, gt_depths=None, models=None, batch_size=16):
    """KITTI branch of the reference's `evaluate(opt)` (evaluate_depth.py:104-317) for the ResNet and MonoViT (`--ViT`) models:
    predicts disparities for a split, scores them against `gt_depths.npz` with median (mono) or 5.4x
    (stereo) scaling, returns (mean_errors[7], ratios).  Differences by design: images are prepared by the
    device loader and each batch is scored by one `bbd_depth_metrics` launch while it is still in HBM
    (the reference copies every disparity map to the host, resizes with cv2 and runs numpy per image).

    `dataloader` / `gt_depths` / `models` may be injected (tests, synthetic splits); by default they are
    built from `opt.splits_dir/<eval_split>/{test_files.txt, gt_depths.npz}`, `opt.kt_path` and
    `opt.load_weights_folder`."""
    from . import tuning
    tuning.use_shipped_db()      # (no Trainer is built here: the tuned MIOpen database is wired explicitly)
    import os
    from . import datasets, networks
    from .layers import disp_to_depth

    assert sum((opt.eval_mono, opt.eval_stereo)) == 1, \
        "Please choose mono or stereo evaluation by setting either --eval_mono or --eval_stereo"
    device = torch.device("cuda:%d" % getattr(opt, "cuda", 0))
    if models is None:
        folder = os.path.expanduser(opt.load_weights_folder)
        assert os.path.isdir(folder), "Cannot find a folder at {}".format(folder)
        enc_dict = torch.load(os.path.join(folder, "encoder.pth"), map_location=device)
        height, width = enc_dict["height"], enc_dict["width"]
        if getattr(opt, "ViT", False):                  # MonoViT checkpoints (evaluate_depth.py:141-149)
            from . import networksvit
            encoder = networksvit.mpvit_small(checkpoint=None)
            encoder.num_ch_enc = [64, 128, 216, 288, 288]
            decoder = networksvit.DepthDecoder()
        else:
            encoder = networks.ResnetEncoder(opt.num_layers, False)
            decoder = networks.DepthDecoder(encoder.num_ch_enc)
        own = encoder.state_dict()
        encoder.load_state_dict({k: v for k, v in enc_dict.items() if k in own})
        decoder.load_state_dict(torch.load(os.path.join(folder, "depth.pth"), map_location=device), strict=False)
    else:
        encoder, decoder = models
        height, width = opt.height, opt.width
    encoder.to(device).eval()
    decoder.to(device).eval()
    split_dir = os.path.join(getattr(opt, "splits_dir", "splits"), opt.eval_split)
    if dataloader is None:
        filenames = datasets.readlines(os.path.join(split_dir, "test_files.txt"))
        ds = datasets.KITTIRAWDataset(filenames, 0, height, width, kt_path=opt.kt_path, is_train=False, kt=True,
                                      naive_mix=True)
        dataloader = datasets.DeviceLoader(ds, batch_size, datasets.DeviceCollate(height, width, [0], device),
                                           shuffle=False, drop_last=False, num_workers=getattr(opt, "num_workers", 8))
    if gt_depths is None:
        gt_depths = np.load(os.path.join(split_dir, "gt_depths.npz"), fix_imports=True, encoding="latin1",
                            allow_pickle=True)["data"]
    gts = gt_depths if isinstance(gt_depths, GroundTruthSet) else GroundTruthSet(gt_depths, device)

    median_scaling = not opt.disable_median_scaling
    scale = opt.pred_depth_scale_factor
    if opt.eval_stereo:                                   # evaluate_depth.py:233-237
        median_scaling, scale = False, STEREO_SCALE_FACTOR
    rows, first = [], 0
    with torch.no_grad():
        for data in dataloader:
            out = decoder(encoder(data[("color", 0, 0)]))
            pred_disp, _ = disp_to_depth(out[("disp", 0)], opt.min_depth, opt.max_depth)
            n = pred_disp.shape[0]
            rows.append(depth_metrics(pred_disp, gts, list(range(first, first + n)), min_depth=1e-3, max_depth=80.0,
                                      pred_is_disp=True, median="numpy", median_scaling=median_scaling,
                                      scale_factor=scale))
            first += n
    rows = torch.cat(rows).cpu().numpy().astype(np.float64)          # the only host synchronisation
    assert first == len(gts), "split has %d images, ground truth %d" % (first, len(gts))
    mean_errors = rows[:, :7].mean(0)
    ratios = rows[:, 7] if median_scaling else None
    if median_scaling:
        med = np.median(ratios)
        print(" Scaling ratios | med: {:0.3f} | std: {:0.3f}".format(med, np.std(ratios / med)))
    print("\n  " + ("{:>8} | " * 7).format("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3"))
    print(("&{: 8.3f}  " * 7).format(*mean_errors.tolist()) + "\\\\")
    return mean_errors, ratios
