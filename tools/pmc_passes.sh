#!/bin/bash
# Collect rocprofv3 PMC counters for the hot-path kernels, one counter group per pass
# (gfx950: 8 SQ slots, FETCH_SIZE and WRITE_SIZE in separate passes - MI355X_MICROARCH.md).
# usage: tools/pmc_passes.sh <out_dir> [kernel_bench args...]
#        PMC_TARGET=bench tools/pmc_passes.sh <out_dir> --config md2     (the kernels inside the real training step:
#        network-produced disparities and poses; tools/kernel_bench.py feeds per-pixel random disparities, whose
#        scattered gathers overstate the traffic)
set -u
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  if [ "${PMC_TARGET:-kernel_bench}" = "bench" ]; then
    rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT" -o pass$i -- \
        python3 bench.py --step-graph off --steps 3 --warmup 2 --no-cpu-baseline --no-eager-ab --no-secondary "$@" > "$OUT/pass$i.log" 2>&1
  else
    rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT" -o pass$i -- \
        python3 tools/kernel_bench.py --iters 3 --warmup 1 "$@" > "$OUT/pass$i.log" 2>&1
  fi
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
GRBM_GUI_ACTIVE
GROUPS
ls "$OUT"
