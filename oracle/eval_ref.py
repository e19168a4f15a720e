"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference's validation metrics:
  * `compute_depth_losses_ref`  - Trainer.compute_depth_losses, KITTI branch (trainer.py:594-617) +
    layers.compute_depth_errors (layers.py:271-286).  Pinned by tests/golden/eval_*.npz, generated
    by running the imported reference (tools/make_golden_eval.py).
  * `evaluate_image_ref`        - the per-image loop of evaluate_depth.py:244-297 + compute_errors
    (:57-72, KITTI branch).  Its resize is cv2.resize, a third-party dependency (opencv-python, the
    reference's environment.yml pins opencv 4.5.x) that is absent here: restated from OpenCV's published
    INTER_LINEAR float path (half-pixel centres, edge taps collapsed, horizontal then vertical pass).
    PARITY UNPINNED for that resize; the metric arithmetic after it is pinned by numpy itself.
"""
import numpy as np
import torch
import torch.nn.functional as F

GARG = (0.40810811, 0.99189189, 0.03594771, 0.96405229)


def garg_mask(gt, min_depth, max_depth):
    gh, gw = gt.shape[:2]
    mask = np.logical_and(gt > min_depth, gt < max_depth)
    crop = np.array([GARG[0] * gh, GARG[1] * gh, GARG[2] * gw, GARG[3] * gw]).astype(np.int32)
    crop_mask = np.zeros(mask.shape)
    crop_mask[crop[0]:crop[1], crop[2]:crop[3]] = 1
    return np.logical_and(mask, crop_mask)


def compute_depth_errors_ref(gt, pred):
    """layers.py:271-286."""
    thresh = torch.max((gt / pred), (pred / gt))
    a1 = (thresh < 1.25).float().mean()
    a2 = (thresh < 1.25 ** 2).float().mean()
    a3 = (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean((gt - pred) ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def compute_depth_losses_ref(depth_pred, gt_depth):
    """trainer.py:594-617 for one image: depth_pred [1,1,h,w] tensor, gt_depth [GH,GW] float32 array.
    Returns dict(metrics[7], ratio, median_gt, median_pred, count)."""
    min_depth, max_depth = 1e-3, 80
    gh, gw = gt_depth.shape[:2]
    pred = torch.clamp(F.interpolate(depth_pred, [gh, gw], mode="bilinear", align_corners=False), 1e-3, 80)
    pred = pred.detach().squeeze()
    mask = torch.from_numpy(garg_mask(gt_depth, min_depth, max_depth))
    gt = torch.from_numpy(gt_depth)
    mg, mp = torch.median(gt[mask]), torch.median(pred[mask])
    ratio = mg / mp
    pred = pred * ratio
    pred = torch.clamp(pred, min=min_depth, max=max_depth)
    errs = compute_depth_errors_ref(gt[mask], pred[mask])
    return {"metrics": np.array([float(e) for e in errs], dtype=np.float64), "ratio": float(ratio),
            "median_gt": float(mg), "median_pred": float(mp), "count": int(mask.sum())}


def cv2_resize_linear_ref(img, out_w, out_h):
    """OpenCV INTER_LINEAR for a float32 single-channel image (published algorithm, see header)."""
    h, w = img.shape

    def taps(n_out, n_in):
        scale = float(n_in) / float(n_out)
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = f - i.astype(np.float32)
        lo = i < 0
        i[lo], f[lo] = 0, 0.0
        hi = i >= n_in - 1
        i[hi], f[hi] = n_in - 1, 0.0
        return i, np.minimum(i + 1, n_in - 1), f.astype(np.float32)

    x0, x1, fx = taps(out_w, w)
    y0, y1, fy = taps(out_h, h)
    one = np.float32(1.0)
    rows = img[:, x0] * (one - fx)[None, :] + img[:, x1] * fx[None, :]
    return (rows[y0] * (one - fy)[:, None] + rows[y1] * fy[:, None]).astype(np.float32)


def compute_errors_ref(gt, pred):
    """evaluate_depth.py:57-72 (KITTI branch), numpy."""
    thresh = np.maximum((gt / pred), (pred / gt))
    a1 = (thresh < 1.25).mean()
    a2 = (thresh < 1.25 ** 2).mean()
    a3 = (thresh < 1.25 ** 3).mean()
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def evaluate_image_ref(pred_disp, gt_depth, median_scaling=True, scale_factor=1.0, min_depth=1e-3, max_depth=80):
    """evaluate_depth.py:244-297 for one image: pred_disp [h,w] float32, gt_depth [GH,GW] float32."""
    gh, gw = gt_depth.shape[:2]
    pred_depth = 1 / cv2_resize_linear_ref(pred_disp, gw, gh)
    mask = garg_mask(gt_depth, min_depth, max_depth)
    pred, gt = pred_depth[mask], gt_depth[mask]
    pred = pred * np.float32(scale_factor)
    ratio = np.float32(1.0)
    if median_scaling:
        ratio = np.median(gt) / np.median(pred)
        pred = pred * ratio
    pred[pred < min_depth] = min_depth
    pred[pred > max_depth] = max_depth
    return {"metrics": np.array(compute_errors_ref(gt, pred), dtype=np.float64), "ratio": float(ratio),
            "median_gt": float(np.median(gt)), "median_pred": float(np.median(pred_depth[mask] * np.float32(scale_factor))),
            "count": int(mask.sum())}
