"""CPU reference of one full MD2 training step (TEST INFRASTRUCTURE / cpu_baseline only).

`process_batch` + `zero_grad` + `backward` + `optimizer.step` as the reference's `run_epoch`
body does it (trainer.py:260-263) for the MD2 configuration (frames [0,-1,1], no trimin):
pose nets per frame (trainer.py:390-405), encoder/decoder (:295-296), then the oracle hot path
(hotpath_ref.hot_path = generate_images_pred + compute_losses).  The networks are ordinary
PyTorch modules passed in by the caller and run on the CPU here.
"""
import torch

from . import hotpath_ref as O


def md2_step(models, optimizer, inputs, ms, scales, H, W, noise):
    """One optimisation step; returns the loss tensor."""
    poses = {}
    for f in (1, -1):
        rows = [b for b, m in enumerate(ms) if m == abs(f)]
        mid = inputs[("color_aug", 0, 0)][rows]
        other = inputs[("color_aug", f, 0)][[O.source_row(ms, f, b) for b in rows]]
        pair = [other, mid] if f < 0 else [mid, other]
        feats = [models["pose_encoder"](torch.cat(pair, 1))]
        axisangle, translation = models["pose"](feats)
        poses[f] = O.pose_matrix(axisangle[:, 0], translation[:, 0], invert=(f < 0))
    feats = models["encoder"](inputs[("color_aug", 0, 0)])
    disp_out = models["depth"](feats)
    disp = {s: disp_out[("disp", s)] for s in scales}
    out = O.hot_path(inputs, disp, poses, ms, scales, False, False, noise, H, W)
    optimizer.zero_grad()
    out["loss"].backward()
    optimizer.step()
    return out["loss"].detach()
