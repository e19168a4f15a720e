#!/bin/bash
# Extend the shipped MIOpen database with the padded pose-pass row counts (tools/miopen_tune_pose.py).  On a gpurun box
# only gpurun_out/ travels back: the database is worked on there and copied into the tree by the caller afterwards:
#   cp gpurun_out/miopen_db/*.txt baseboostdepth_amd/miopen_db/; cp gpurun_out/miopen_db/cache/*.ukdb baseboostdepth_amd/miopen_db/cache/
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/miopen_db
cp -r baseboostdepth_amd/miopen_db/. gpurun_out/miopen_db/
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db MIOPEN_CUSTOM_CACHE_DIR=$PWD/gpurun_out/miopen_db/cache
python tools/miopen_tune_pose.py "$@"
rm -f "$MIOPEN_USER_DB_PATH"/*.time "$MIOPEN_USER_DB_PATH"/*.lock
ls -la "$MIOPEN_USER_DB_PATH" "$MIOPEN_CUSTOM_CACHE_DIR"
