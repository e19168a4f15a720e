set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/miopen_db/cache
cp -r baseboostdepth_amd/miopen_db/. gpurun_out/miopen_db/
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db MIOPEN_CUSTOM_CACHE_DIR=$PWD/gpurun_out/miopen_db/cache
python tools/miopen_tune_pose.py --rows 288 > gpurun_out/tune_pose_288.log 2>&1
python tools/miopen_tune_pose.py --no-find --rows 32 64 192 224 256 288 > gpurun_out/tune_pose_cache.log 2>&1
rm -f gpurun_out/miopen_db/*.time gpurun_out/miopen_db/*.lock
ls -la gpurun_out/miopen_db gpurun_out/miopen_db/cache
tail -3 gpurun_out/tune_pose_288.log gpurun_out/tune_pose_cache.log
unset MIOPEN_USER_DB_PATH MIOPEN_CUSTOM_CACHE_DIR
# the freshly tuned database in place for the rest of this call
cp gpurun_out/miopen_db/*.txt baseboostdepth_amd/miopen_db/; cp gpurun_out/miopen_db/cache/*.ukdb baseboostdepth_amd/miopen_db/cache/
python -m pytest tests/test_gpu_fresh_orderings.py tests/test_gpu_nn.py tests/test_gpu_parity.py -x -q 2>&1 | tail -8 > gpurun_out/t_fresh.log; tail -4 gpurun_out/t_fresh.log
for v in 0 1; do BBD_EXPERIMENT=1 BBD_IDENT_GROUPED=$v python tools/kernel_bench.py --smooth --iters 200 --config md2 2>/dev/null | tail -1 > gpurun_out/ident_md2_grouped$v.json; BBD_EXPERIMENT=1 BBD_IDENT_GROUPED=$v python tools/kernel_bench.py --smooth --iters 200 --config boost7 2>/dev/null | tail -1 > gpurun_out/ident_boost7_grouped$v.json; done
python bench.py --steps 20 --warmup 10 > gpurun_out/bench_r05_b.json 2> gpurun_out/bench_r05_b.err
