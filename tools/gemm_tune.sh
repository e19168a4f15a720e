#!/bin/bash
# (Re)record the TunableOp table shipped in baseboostdepth_amd/gemm_db/ on an MI355X: one eager pass of the MonoViT
# training step with tuning ON measures every hipBLASLt / rocBLAS solution for every GEMM shape of the step (forward,
# data gradient, weight gradient of the qkv / proj / MLP Linear layers and the pose decoder's small ones) and writes the
# winners.  About 90 s.  The other configurations have no GEMMs (convolutions only: tools/miopen_tune.sh).
#   usage: tools/gemm_tune.sh [config ...]        (default: vit)
set -u
cd "$(dirname "$0")/.."
OUT=$PWD/baseboostdepth_amd/gemm_db
mkdir -p $OUT /tmp/bbd_gemm_tune
rm -f /tmp/bbd_gemm_tune/*.csv
export PYTORCH_TUNABLEOP_FILENAME=/tmp/bbd_gemm_tune/tunableop_gfx950.csv
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=40 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
for cfg in ${@:-vit}; do
  python bench.py --config $cfg --step-graph off --steps 2 --warmup 2 --no-cpu-baseline --no-eager-ab | tail -c 300
done
cat /tmp/bbd_gemm_tune/tunableop_gfx950*.csv > $OUT/tunableop_gfx950.csv
wc -l $OUT/tunableop_gfx950.csv
