#!/usr/bin/env python3
"""Phase timeline of the sparse-item backward (warp_ssim_min_bwd3_kernel) from in-kernel s_memtime stamps
(diagnostic build -DBBD_STAMPS loaded through BBD_HIP_LIB; wave 0 of every workgroup).   usage: stamps_bwd3.py [md2|boost7]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
cfg = sys.argv[1] if len(sys.argv) > 1 else "md2"
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
dev, H, W, B = "cuda:0", 192, 640, 12
if cfg == "md2":
    ms, trimin, decomp, scales = [1] * B, False, False, [0, 1, 2, 3]
else:
    ms, trimin, decomp, scales = [7] * B, True, True, [0]
inputs = synthetic_batch(ms, H, W, scales, device=dev, seed=42)
opt = types.SimpleNamespace(height=H, width=W, batch_size=B, scales=scales, frame_ids=[0], min_depth=0.1, max_depth=100.0,
                            disparity_smoothness=1e-3, no_ssim=False, trimin=trimin, decomp=decomp, pose_error=5.5,
                            incremental_skip=False, partial_skip=False, materialize_warps=False)
tr = Trainer.__new__(Trainer)
tr.opt, tr.device, tr.num_scales, tr.backend, tr.maxing_valid_frames = opt, torch.device(dev), 4, None, False
be = tr._backend()
plan = tr.valid_frames_trimin(inputs)
disp = synthetic_disp(B, H, W, scales, device=dev, seed=1)
if os.environ.get("SMOOTH_DISP"):        # smooth disparities, like the training step's (default: per-pixel random)
    disp = {s: torch.nn.functional.avg_pool2d(torch.nn.functional.pad(d, (8, 8, 8, 8), mode="replicate"), 17, 1) for s, d in disp.items()}
outputs = {("disp", s): disp[s] for s in scales}
outputs.update(synthetic_poses(plan, device=dev, seed=2, pose_error=5.5))
S = len(scales)
nblocks = S * B * be.num_tiles_bwd(H, W)
stamps = torch.zeros(nblocks * 32, dtype=torch.int64, device=dev)
dll = be.lib._dll
dll.bbd_debug_set_stamps.argtypes = [ctypes.c_void_p]
for _ in range(2):
    o = tr.generate_images_pred(inputs, {k: (v.clone().requires_grad_(True) if k[0] == "disp" else v) for k, v in outputs.items()})
    o[("bbd", "loss_sum")].sum().backward()
o = tr.generate_images_pred(inputs, {k: (v.clone().requires_grad_(True) if k[0] == "disp" else v) for k, v in outputs.items()})
ls = o[("bbd", "loss_sum")]
torch.cuda.synchronize()
dll.bbd_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
ls.sum().backward()
torch.cuda.synchronize()
st = stamps.view(nblocks, 32).cpu().double()
bn = {(0, 1): "setup: ids, target cells, clears + barrier", (1, 2): "cell masks, band ids, counts + barrier"}
for p in range(2):
    b0 = 3 + 6 * p
    prev = 2 if p == 0 else b0 - 1
    bn[(prev, b0)] = "pass %d: (end of previous pass,) lists + barrier" % p
    bn[(b0, b0 + 1)] = "pass %d: W warp the items" % p
    bn[(b0 + 1, b0 + 2)] = "pass %d: barrier" % p
    bn[(b0 + 2, b0 + 3)] = "pass %d: C winners' partials + scatter" % p
    bn[(b0 + 3, b0 + 4)] = "pass %d: barrier" % p
    bn[(b0 + 4, b0 + 5)] = "pass %d: G item gradients + reduce" % p
bn[(0, 20)] = "TOTAL workgroup"
print("== backward (sparse-item form), config %s%s" % (cfg, ", smooth disparities" if os.environ.get("SMOOTH_DISP") else ""))
for (a, b), n in bn.items():
    ok = (st[:, a] > 0) & (st[:, b] > 0)
    d = (st[:, b] - st[:, a])[ok]
    if d.numel():
        print("%-52s mean %8.0f  median %8.0f  p90 %8.0f ticks  (%d workgroups)" % (n, d.mean(), d.median(), d.quantile(0.9), d.numel()))
