set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/f; mkdir -p $O
BBD_HIP_LIB=$PWD/build_variants/libbbd_both.so python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 > $O/t_both.log; tail -2 $O/t_both.log
REPS="1 2" bash tools/instep_ab.sh "boosted15_coherent boosted15" base present near both > $O/bwd_skip_ab.txt 2>&1
REPS="1" bash tools/instep_ab.sh "md2 boosted trimin5" base both >> $O/bwd_skip_ab.txt 2>&1
cat $O/bwd_skip_ab.txt
for cfg in md2_loader boosted15_fresh; do
  timeout 600 python -X faulthandler bench.py --config $cfg --no-cpu-baseline --no-secondary > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "$cfg rc=$?"; tail -c 200 $O/bench_$cfg.err
done
python -m pytest tests/test_gpu_loader.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
