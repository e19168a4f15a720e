#!/bin/bash
# Build tuning variants of the kernel library from the same sources with different -D flags and
# time each with tools/kernel_bench.py.   usage: tools/variants.sh "name:-DFLAG ..." ...
set -u
cd "$(dirname "$0")/.."
SRC="baseboostdepth_amd/csrc/bbd_kernels.hip baseboostdepth_amd/csrc/bbd_eval.hip baseboostdepth_amd/csrc/bbd_image.hip baseboostdepth_amd/csrc/bbd_nn.hip baseboostdepth_amd/csrc/bbd_vit.hip baseboostdepth_amd/csrc/bbd_pose.hip baseboostdepth_amd/csrc/bbd_tokens.hip"
mkdir -p /tmp/bbdvar
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  out=/tmp/bbdvar/libbbd_$name.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math ${BBD_BASE_FLAGS--fno-slp-vectorize} -std=c++17 -fPIC -shared $flags -o $out $SRC 2>&1 | grep -E "error" 
  for cfg in ${BBD_CONFIGS:-md2}; do
    echo -n "$name [$flags] $cfg: "
    BBD_HIP_LIB=$out python tools/kernel_bench.py --config $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ident %.4f  fwd %.4f  bwd %.4f ms'%(d['identity']['ms'],d['fwd']['ms'],d['bwd']['ms']))"
  done
done
