#!/bin/bash
# HISTORICAL (round 3): patches the round-3 source of the fused kernels (`git show 6e78be1:baseboostdepth_amd/csrc/bbd_kernels.hip`);
# the patterns it replaces no longer exist in the shipped source (round 4: nine-plane backward, forward forms).
# Persistent workgroups for the fused backward, as an experiment on a patched COPY of the source:
# the launch has 4 workgroups per CU (1 024) and every workgroup walks work items blockIdx.x, + gridDim.x, ... instead of
# one workgroup per (sample, scale, tile).  No prefetch of the next item's set-up (that needs ~15 more VGPRs than the
# 128 the kernel has at 4 waves per SIMD).  Timed with tools/kernel_bench.py against the shipped build.
#   usage: tools/experiments/persistent_bwd_ab.sh        (run from the repo root on the GPU box)
set -u
CS=baseboostdepth_amd/csrc
mkdir -p /tmp/bbdvar/src_persist
python3 - <<'PY'
src = open("baseboostdepth_amd/csrc/bbd_kernels.hip").read()
def sub(text, a, b):
    assert text.count(a) == 1, (text.count(a), a[:60])
    return text.replace(a, b)
k0 = src.index("__global__ __launch_bounds__(NT2, BBD_BWD2_WGS) void warp_ssim_min_bwd2_kernel(BwdArgs a) {")
k1 = src.index("// Fused backward, sparse-item form")
body = src[k0:k1]
body = sub(body, "  int bid = a.remap ? xcd_work_item(blockIdx.x, gridDim.x) : (int)blockIdx.x;\n",
           "  const int n_items = a.S * a.B * a.ntiles;\n  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {\n"
           "  int bid = a.remap ? xcd_work_item(item, n_items) : item;\n")
body = sub(body, "  BBD_STAMP(20);\n  BBD_STAMP_RT(31);\n  BBD_STAMP_VAL(29, __builtin_popcount(present));\n#undef BBD_PARG\n}",
           "  __syncthreads();\n  }\n#undef BBD_PARG\n}")
src = src[:k0] + body + src[k1:]
src = sub(src, "  hipLaunchKernelGGL(warp_ssim_min_bwd2_kernel, dim3((unsigned)(S * B * a.ntiles)), dim3(NT2), 0,",
          "  hipLaunchKernelGGL(warp_ssim_min_bwd2_kernel, dim3((unsigned)((S * B * a.ntiles) < PERSIST_WGS ? (S * B * a.ntiles) : PERSIST_WGS)), dim3(NT2), 0,")
src = src.replace('"../../include/bbd_hip.h"', '"bbd_hip.h"')
open("/tmp/bbdvar/src_persist/bbd_kernels.hip", "w").write(src)
PY
for wgs in 1024 2048; do
  lib=/tmp/bbdvar/libbbd_persist$wgs.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -std=c++17 -fPIC -shared -DPERSIST_WGS=$wgs \
      -Rpass-analysis=kernel-resource-usage -I $PWD/include -I $PWD/$CS -o $lib /tmp/bbdvar/src_persist/bbd_kernels.hip $CS/bbd_eval.hip $CS/bbd_image.hip $CS/bbd_nn.hip \
      $CS/bbd_vit.hip $CS/bbd_pose.hip $CS/bbd_tokens.hip 2>&1 | grep -A8 "warp_ssim_min_bwd2_kernel" | grep -E "VGPRs:|Spill|Occupancy|LDS Size" | head -5
  for cfg in md2 boost7; do
    for rep in 1 2; do
      echo -n "persistent $wgs workgroups, $cfg: "; BBD_HIP_LIB=$lib python tools/kernel_bench.py --config $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bwd %.4f ms'%d['bwd']['ms'])"
    done
  done
done
for cfg in md2 boost7; do
  for rep in 1 2; do
    echo -n "shipped (one workgroup per item), $cfg: "; python tools/kernel_bench.py --config $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bwd %.4f ms'%d['bwd']['ms'])"
  done
done
