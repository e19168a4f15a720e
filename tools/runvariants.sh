#!/bin/bash
# On the GPU box: time every build_variants/libbbd_*.so with tools/kernel_bench.py --smooth (the in-step-like inputs).
#   tools/runvariants.sh "md2 boost7" [name ...]
set -u
cd "$(dirname "$0")/.."
cfgs=${1:-md2}; shift || true
names="$@"; [ -z "$names" ] && names=$(ls build_variants/libbbd_*.so | sed 's/.*libbbd_\(.*\)\.so/\1/')
for rep in ${REPS:-1 2 3}; do
for name in $names; do
  for cfg in $cfgs; do
    echo -n "$name $cfg: "
    BBD_HIP_LIB=$PWD/build_variants/libbbd_$name.so timeout 200 python tools/kernel_bench.py --smooth --iters ${ITERS:-200} --config $cfg 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ident %.4f  fwd %.4f  bwd %.4f ms'%(d['identity']['ms'],d['fwd']['ms'],d['bwd']['ms']))"
  done
done
done
