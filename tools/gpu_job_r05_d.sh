set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/d; mkdir -p $O
python -X faulthandler -m pytest tests/test_gpu_fresh_orderings.py -x -q > $O/t_fresh.log 2>&1; tail -3 $O/t_fresh.log
python -X faulthandler -m pytest tests/test_gpu_nn.py -x -q > $O/t_nn.log 2>&1; tail -3 $O/t_nn.log
python -X faulthandler -m pytest tests/test_gpu_parity.py -x -q > $O/t_parity.log 2>&1; tail -3 $O/t_parity.log
python -X faulthandler bench.py --steps 20 --warmup 10 --no-secondary --no-cpu-baseline > $O/bench_head.json 2> $O/bench_head.err; tail -c 300 $O/bench_head.err
for cfg in boosted15 boosted15_coherent trimin5 boosted15_fresh trimin5_fresh md2_loader; do
  timeout 600 python -X faulthandler bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "$cfg rc=$?"; tail -c 200 $O/bench_$cfg.err
done
for v in 0 1; do for cfg in md2 boost7; do BBD_EXPERIMENT=1 BBD_IDENT_GROUPED=$v python tools/kernel_bench.py --smooth --iters 200 --config $cfg 2>/dev/null | tail -1 > $O/ident_${cfg}_grouped$v.json; done; done
grep -h -o '"identity": {[^}]*}' $O/ident_*.json
