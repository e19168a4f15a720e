#!/bin/bash
# HISTORICAL (round 3): patches the round-3 source of the fused kernels (`git show 6e78be1:baseboostdepth_amd/csrc/bbd_kernels.hip`);
# the patterns it replaces no longer exist in the shipped source (round 4: nine-plane backward, forward forms).
# Forward kernel at 4 waves per SIMD (experiment on a patched copy): the warped tile single-buffered (29 KB of LDS per
# workgroup instead of 44, a second barrier per candidate) and the 128-VGPR budget.
#   usage: tools/experiments/forward_four_waves_ab.sh      (repo root, GPU box)
set -u
CS=baseboostdepth_amd/csrc
mkdir -p /tmp/bbdvar/src_f4
python3 - <<'PY'
src = open("baseboostdepth_amd/csrc/bbd_kernels.hip").read()
def sub(t, a, b):
    assert t.count(a) == 1, (t.count(a), a[:50])
    return t.replace(a, b)
src = sub(src, "  __shared__ __attribute__((aligned(16))) float s_xx[2][3][FPLANE];", "  __shared__ __attribute__((aligned(16))) float s_xx[1][3][FPLANE];")
src = sub(src, "      strip_loss(s_xx[buf], s_y, ly, lx0, mu_y, sg_y, a.no_ssim, loss);\n      BBD_STAMP(7 + 4 * (vs & 3));\n      buf ^= 1;",
          "      strip_loss(s_xx[buf], s_y, ly, lx0, mu_y, sg_y, a.no_ssim, loss);\n      __syncthreads();")
open("/tmp/bbdvar/src_f4/bbd_kernels.hip", "w").write(src.replace('"../../include/bbd_hip.h"', '"bbd_hip.h"'))
PY
for spec in "single_buf_3waves:-DBBD_FWD_WAVES=3" "single_buf_4waves:-DBBD_FWD_WAVES=4"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  lib=/tmp/bbdvar/libbbd_$name.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -std=c++17 -fPIC -shared $flags -I $PWD/include -I $PWD/$CS \
      -o $lib /tmp/bbdvar/src_f4/bbd_kernels.hip $CS/bbd_eval.hip $CS/bbd_image.hip $CS/bbd_nn.hip $CS/bbd_vit.hip $CS/bbd_pose.hip $CS/bbd_tokens.hip 2>&1 | grep error
  for cfg in md2 boost7; do
    for rep in 1 2; do
      echo -n "$name $cfg: "; BBD_HIP_LIB=$lib python tools/kernel_bench.py --config $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd %.4f ms'%d['fwd']['ms'])"
    done
  done
done
for cfg in md2 boost7; do for rep in 1 2; do echo -n "shipped (double buffer, 3 waves) $cfg: "; python tools/kernel_bench.py --config $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd %.4f ms'%d['fwd']['ms'])"; done; done
