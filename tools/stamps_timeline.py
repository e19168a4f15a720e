#!/usr/bin/env python3
"""Where inside a launch do the fused kernels' workgroups run?  (diagnostic build -DBBD_STAMPS loaded through BBD_HIP_LIB)
Every workgroup records the chip-wide 100 MHz wall clock at its start and end, plus (backward) how many candidates it
processed.  Prints the launch span, the sum of workgroup lifetimes / resident slots (= the span a perfectly balanced launch
would need), lifetime statistics by live-candidate count, and how many workgroups are resident over time.
usage: stamps_timeline.py [md2|boost7|boost_e15]     (SMOOTH_DISP=1: spatially smooth disparities, like the training step's)"""
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
from baseboostdepth_amd.synthetic import synthetic_batch, synthetic_disp, synthetic_poses
from baseboostdepth_amd.trainer import Trainer
dev, H, W, B = "cuda:0", 192, 640, 12
if cfg == "md2":
    ms, trimin, decomp, scales = [1] * B, False, False, [0, 1, 2, 3]
elif cfg == "boost_e15":        # the epoch-15 offset distribution (tools/kernel_bench.py's draw): mixed 8- and 18-candidate samples
    import random
    rnd = random.Random(15)
    ms = [rnd.choices(range(1, 8), [.050, .050, .077, .094, .139, .142, .448])[0] for _ in range(B)]
    trimin, decomp, scales = True, True, [0]
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
if os.environ.get("SMOOTH_DISP"):
    disp = {s: torch.nn.functional.avg_pool2d(torch.nn.functional.pad(d, (8, 8, 8, 8), mode="replicate"), 17, 1) for s, d in disp.items()}
outputs = {("disp", s): disp[s] for s in scales}
outputs.update(synthetic_poses(plan, device=dev, seed=2, pose_error=5.5))
S = len(scales)
nf, nb = S * B * be.num_tiles_fwd(H, W), S * B * be.num_tiles_bwd(H, W)
stamps = torch.zeros(max(nf, nb) * 32, dtype=torch.int64, device=dev)
dll = be.lib._dll
dll.bbd_debug_set_stamps.argtypes = [ctypes.c_void_p]


def run(grad):
    o = tr.generate_images_pred(inputs, {k: (v.clone().requires_grad_(True) if (grad and k[0] == "disp") else v) for k, v in outputs.items()})
    return o[("bbd", "loss_sum")]


def report(name, n, slots, by_count):
    st = stamps.view(-1, 32)[:n].cpu().double()
    t0, t1 = st[:, 30], st[:, 31]
    ok = (t0 > 0) & (t1 > 0)
    t0, t1 = t0[ok], t1[ok]
    base = t0.min()
    span = (t1.max() - base) / 100.0            # us
    life = (t1 - t0) / 100.0
    print("== %s (%s%s): %d workgroups, %d resident slots" % (name, cfg, ", smooth disparities" if os.environ.get("SMOOTH_DISP") else "", t0.numel(), slots))
    print("   launch span (first start -> last end) %.1f us; sum of lifetimes / slots = %.1f us (%.2f of the span)" %
          (span, life.sum() / slots, life.sum() / slots / span))
    print("   workgroup lifetime: mean %.1f us, median %.1f, p90 %.1f, max %.1f; last start at %.1f us" %
          (life.mean(), life.median(), life.quantile(0.9), life.max(), (t0.max() - base) / 100.0))
    if by_count:
        cnt = st[:, 29][ok]
        for c in sorted(set(cnt.tolist())):
            m = cnt == c
            print("   live candidates %2d: %5d workgroups, mean lifetime %.1f us" % (c, int(m.sum()), life[m].mean()))
    edges = torch.linspace(0, float(span), 11)
    res = []
    for i in range(10):
        mid = base + (edges[i] + edges[i + 1]) / 2 * 100.0
        res.append(int(((t0 <= mid) & (t1 >= mid)).sum()))
    print("   resident workgroups at the middle of each tenth of the span:", res)


for _ in range(2):
    run(True).sum().backward()
torch.cuda.synchronize()
stamps.zero_()
dll.bbd_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
run(False)
torch.cuda.synchronize()
dll.bbd_debug_set_stamps(ctypes.c_void_p(0))
report("forward", nf, 256 * (4 if cfg == "md2" else 3), False)      # (the forward's form per launch shape: fused_fwd_form)
ls = run(True)
torch.cuda.synchronize()
stamps.zero_()
dll.bbd_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
ls.sum().backward()
torch.cuda.synchronize()
dll.bbd_debug_set_stamps(ctypes.c_void_p(0))
report("backward", nb, 256 * 4, True)
