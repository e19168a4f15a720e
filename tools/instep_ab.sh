#!/bin/bash
# In-step A/B of kernel-library variants (build_variants/libbbd_<name>.so): bench.py's HIP-event means of the fused kernels
# inside the real training step (network-produced disparities and poses).   usage: tools/instep_ab.sh "md2 boosted" name...
set -u
cd "$(dirname "$0")/.."
cfgs=$1; shift
for rep in ${REPS:-1 2}; do for name in "$@"; do for cfg in $cfgs; do
  echo -n "$name $cfg: "
  BBD_HIP_LIB=$PWD/build_variants/libbbd_$name.so timeout 300 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-eager-ab --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], {k.replace('bbd_','').replace('warp_ssim_min_disp_',''):v['mean_ms'] for k,v in d['kernels'].items()}, d.get('live_candidates_per_backward_tile',''))"
done; done; done
