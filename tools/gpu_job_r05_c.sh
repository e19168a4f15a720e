# backward skip paths in the coherent and the random-initialised regime (variants built by tools/mkvariants.sh)
set -u
cd $GRAFT_REPO_ROOT
BBD_HIP_LIB=$PWD/build_variants/libbbd_both.so python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 > gpurun_out/t_both.log; tail -2 gpurun_out/t_both.log
REPS="1 2" bash tools/instep_ab.sh "boosted15_coherent boosted15" base present near both > gpurun_out/bwd_skip_ab.txt 2>&1
REPS="1" bash tools/instep_ab.sh "md2 boosted" base both >> gpurun_out/bwd_skip_ab.txt 2>&1
cat gpurun_out/bwd_skip_ab.txt
