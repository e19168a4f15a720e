#!/bin/bash
# Build tuning variants of the kernel library into build_variants/ (git-ignored, travels to the GPU box):
#   tools/mkvariants.sh "name:-DFLAG ..." ...        then on the GPU box: tools/runvariants.sh [configs]
set -u
cd "$(dirname "$0")/.."
C=baseboostdepth_amd/csrc
SRC="$C/bbd_kernels.hip $C/bbd_eval.hip $C/bbd_image.hip $C/bbd_nn.hip $C/bbd_vit.hip $C/bbd_pose.hip $C/bbd_tokens.hip"
mkdir -p build_variants
# the other translation units do not change between variants: compile them once
for f in bbd_eval bbd_image bbd_nn bbd_vit bbd_pose bbd_tokens bbd_util; do
  o=build_variants/$f.o
  if [ ! -f $o ] || [ $C/$f.hip -nt $o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -std=c++17 -fPIC -c -o $o $C/$f.hip || exit 1
  fi
done
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"; [ "$flags" = "$spec" ] && flags=""
  src=${BBD_VARIANT_SRC:-$C/bbd_kernels.hip}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -std=c++17 -fPIC -I$C -c $flags \
      -Rpass-analysis=kernel-resource-usage -o build_variants/k_$name.o $src 2> build_variants/k_$name.log || { grep error build_variants/k_$name.log; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/libbbd_$name.so build_variants/k_$name.o build_variants/bbd_{eval,image,nn,vit,pose,tokens,util}.o || exit 1
  echo -n "$name [$flags]: "
  python3 - "$name" <<'PY'
import re, sys
t = open("build_variants/k_%s.log" % sys.argv[1]).read()
out = []
for kern in ("warp_ssim_min_fwd_kernelILb0ELi1", "warp_ssim_min_fwd_kernelILb0ELi0", "warp_ssim_min_fwd_kernelILb0ELi2", "warp_ssim_min_bwd9_kernelILb1"):
    m = re.search(r"Function Name: \S*%s\S*.*?VGPRs: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)" % kern, t, re.S)
    if m:
        out.append((kern, "%s: %s VGPR, spills s%s v%s, LDS %s" % (kern.replace("warp_ssim_min_", ""), *m.groups())))
print("; ".join(o[1] for o in out))
PY
done
