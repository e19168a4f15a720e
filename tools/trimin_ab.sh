set -u
SRC="baseboostdepth_amd/csrc/bbd_kernels.hip baseboostdepth_amd/csrc/bbd_eval.hip baseboostdepth_amd/csrc/bbd_image.hip baseboostdepth_amd/csrc/bbd_nn.hip baseboostdepth_amd/csrc/bbd_vit.hip baseboostdepth_amd/csrc/bbd_pose.hip baseboostdepth_amd/csrc/bbd_tokens.hip"
mkdir -p /tmp/bbdvar
for spec in "default:" "w3:-DBBD_BWD2_WGS=3" "guarded:-DBBD_BWD_GUARDED" "w3guarded:-DBBD_BWD2_WGS=3 -DBBD_BWD_GUARDED"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  out=/tmp/bbdvar/libbbd_$name.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -std=c++17 -fPIC -shared $flags -o $out $SRC 2>&1 | grep -E "error"
  for cfg in trimin5 md2; do
    BBD_HIP_LIB=$out python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-eager-ab 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name $cfg', d['value'], {k:v['mean_ms'] for k,v in d['kernels'].items()})"
  done
done
