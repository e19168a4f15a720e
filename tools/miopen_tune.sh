#!/bin/bash
# (Re)generate the MIOpen performance database shipped in baseboostdepth_amd/miopen_db/ on an MI355X:
# one benchmark-mode (miopenFind*) pass of the training step per configuration records the measured time of every
# applicable solver for every convolution shape of the step into the user find-db / perf-db, and the compiled
# kernels of the winners into cache/*.ukdb.  About 9 minutes for the MD2 shapes on a box with an empty database.
#   usage: tools/miopen_tune.sh [config ...]        (default: md2 vit; the boosted configurations share MD2's shapes)
set -u
cd "$(dirname "$0")/.."
export MIOPEN_USER_DB_PATH=$PWD/baseboostdepth_amd/miopen_db MIOPEN_CUSTOM_CACHE_DIR=$PWD/baseboostdepth_amd/miopen_db/cache
mkdir -p "$MIOPEN_CUSTOM_CACHE_DIR"
for cfg in ${@:-md2 vit}; do
  python bench.py --config $cfg --miopen-benchmark --steps 3 --warmup 2 --no-cpu-baseline --no-eager-ab | tail -c 400
done
rm -f "$MIOPEN_USER_DB_PATH"/*.time "$MIOPEN_USER_DB_PATH"/*.lock
ls -la "$MIOPEN_USER_DB_PATH" "$MIOPEN_CUSTOM_CACHE_DIR"
