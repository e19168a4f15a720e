#!/usr/bin/env python3
"""Phase timeline of the fused forward kernel from in-kernel s_memtime stamps (diagnostic build
-DBBD_STAMPS, loaded through BBD_HIP_LIB).  Prints mean cycles per phase for wave 0 of each workgroup."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = "/tmp/bbdvar/libbbd_stamps.so"
os.makedirs("/tmp/bbdvar", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-std=c++17",
                "-fPIC", "-shared", "-DBBD_STAMPS"] + os.environ.get("BBD_STAMPS_FLAGS", "").split() + ["-o", out] +
               [os.path.join(ROOT, "baseboostdepth_amd/csrc", f) for f in ("bbd_kernels.hip", "bbd_eval.hip", "bbd_image.hip", "bbd_nn.hip", "bbd_vit.hip", "bbd_pose.hip", "bbd_tokens.hip")],
               check=True)
os.environ["BBD_HIP_LIB"] = out
import types, torch
from baseboostdepth_amd import ops, _lib
from baseboostdepth_amd.synthetic import synthetic_batch, synthetic_disp, synthetic_poses
from baseboostdepth_amd.trainer import Trainer
dev, H, W, B, scales = "cuda:0", 192, 640, 12, [0, 1, 2, 3]
inputs = synthetic_batch([1] * B, H, W, scales, device=dev, seed=42)
opt = types.SimpleNamespace(height=H, width=W, batch_size=B, scales=scales, frame_ids=[0], min_depth=0.1, max_depth=100.0,
                            disparity_smoothness=1e-3, no_ssim=False, trimin=False, decomp=False, pose_error=5.5,
                            incremental_skip=False, partial_skip=False, materialize_warps=False)
tr = Trainer.__new__(Trainer)
tr.opt, tr.device, tr.num_scales, tr.backend, tr.maxing_valid_frames = opt, torch.device(dev), 4, None, False
be = tr._backend()
plan = tr.valid_frames_trimin(inputs)
disp = synthetic_disp(B, H, W, scales, device=dev, seed=1)
if os.environ.get("SMOOTH_DISP"):      # spatially smooth disparities, like the training step's network outputs
    disp = {s: torch.nn.functional.avg_pool2d(torch.nn.functional.pad(d, (8, 8, 8, 8), mode="replicate"), 17, 1) for s, d in disp.items()}
outputs = {("disp", s): disp[s] for s in scales}
outputs.update(synthetic_poses(plan, device=dev, seed=2))
nblocks_f = 4 * B * be.num_tiles_fwd(H, W)
nblocks = 4 * B * max(be.num_tiles_fwd(H, W), be.num_tiles_bwd(H, W))     # the backward tile is narrower
stamps = torch.zeros(nblocks * 32, dtype=torch.int64, device=dev)
dll = be.lib._dll
dll.bbd_debug_set_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    tr.generate_images_pred(inputs, dict(outputs))
dll.bbd_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
tr.generate_images_pred(inputs, dict(outputs))
torch.cuda.synchronize()
st = stamps.view(nblocks, 32)[:nblocks_f].cpu().double()
names = {(0, 1): "cells+stage target+depth issue", (1, 2): "wait loads + barrier", (2, 3): "ystats",
         (4, 5): "cand0 project+gather+blend", (5, 6): "cand0 barrier wait", (6, 7): "cand0 SSIM",
         (8, 9): "cand1 project+gather+blend", (9, 10): "cand1 barrier wait", (10, 11): "cand1 SSIM",
         (11, 20): "identity cands + stores + reduce", (0, 20): "TOTAL workgroup"}
print("== forward")
for (a, b), n in names.items():
    d = st[:, b] - st[:, a]
    print("%-36s mean %8.0f  median %8.0f  p90 %8.0f ticks" % (n, d.mean(), d.median(), d.quantile(0.9)))

# ---- backward (MD2: candidates 0 and 1 are the two warps)
stamps.zero_()
o = tr.generate_images_pred(inputs, {k: (v.clone().requires_grad_(True) if k[0] == "disp" else v) for k, v in outputs.items()})
dll.bbd_debug_set_stamps(ctypes.c_void_p(0))
ls = o[("bbd", "loss_sum")]
dll.bbd_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
ls.sum().backward()
torch.cuda.synchronize()
st = stamps.view(nblocks, 32)[:4 * B * be.num_tiles_bwd(H, W)].cpu().double()
bn = {(0, 21): "  setup: loss-pixel arg-min ids (issue)", (21, 22): "  setup: cells, target + depth loads (issue)",
      (22, 23): "  setup: own arg ids, own depth (issue)", (23, 24): "  setup: clear coefficient planes", (24, 25): "  setup: barrier",
      (25, 26): "  setup: wait arg ids, present mask", (26, 1): "  setup: wait target, stage",
      (0, 1): "setup: clear planes, arg ids, cells, stage", (1, 2): "barrier", (2, 4): "cand0 descriptor + winners' list",
      (4, 5): "cand0 W: warp recompute",
      (5, 6): "cand0 barrier", (6, 7): "cand0 C: winners' SSIM partials", (7, 8): "cand0 barrier",
      (8, 9): "cand0 G: adjoint gather", (9, 10): "cand0 sample-grad + dP reduce", (10, 11): "cand0 barrier",
      (11, 12): "cand0 dP store, cand1 descriptor + list", (12, 13): "cand1 W", (13, 14): "cand1 barrier",
      (14, 15): "cand1 C", (15, 16): "cand1 barrier", (16, 17): "cand1 G", (17, 18): "cand1 sample-grad + reduce",
      (18, 19): "cand1 barrier", (19, 20): "dP store, depth-gradient store",
      (0, 20): "TOTAL workgroup"}
print("== backward")
for (a, b), n in bn.items():
    d = st[:, b] - st[:, a]
    print("%-44s mean %8.0f  median %8.0f  p90 %8.0f ticks" % (n, d.mean(), d.median(), d.quantile(0.9)))
