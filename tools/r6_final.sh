#!/bin/bash
# round 6, last GPU job: the examples as a smoke run, in-step PMC passes of the (unchanged) fused kernels on the final tree,
# the GPU test tier as the driver runs it, the driver's bench command once more on the final tree
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06; mkdir -p $O
timeout 600 python examples/train_synthetic.py --steps 4 2>&1 | tail -2
timeout 600 python examples/train_synthetic.py --steps 4 --boosted 2>&1 | tail -2
timeout 600 python train.py --synthetic --num_epochs 1 --rand --trimin --decomp --incremental_skip --partial_skip --pose_error 5.5 --weights_init scratch --log_dir /tmp/bbd_logs 2>&1 | tail -3
for cfg in md2 boosted15_coherent; do
  PMC_TARGET=bench timeout 1200 bash tools/pmc_passes.sh /tmp/pmc_${cfg}_step --config $cfg > /dev/null 2>&1
  python tools/pmc_summary.py /tmp/pmc_${cfg}_step $O/traffic_$cfg.json "bench.py --config $cfg --step-graph off (the kernels inside the training step)" > $O/pmc_summary_${cfg}_in_step.txt
done
python -m pytest tests/ -x -q -m gpu > $O/gputests_full.log 2>&1; tail -3 $O/gputests_full.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_final_rerun.json 2> $O/bench_default_final_rerun.err
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r06/bench_default_final_rerun.json'))
print('md2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('peak_measured'), d['roofline'].get('traffic'), d['roofline'].get('traffic_source'), d.get('kernels_constants_stale'))
for s in d.get('secondary', []):
    print(s.get('config'), s.get('value'), s.get('vs_frozen_batch'), s.get('vs_frozen_batch_cold_start'), s.get('vs_frozen_batch_per_pose_row_asked'), s.get('vs_frozen_batch_per_pose_row_run'), s.get('error'), s.get('skipped'))
    for p in s.get('passes', []): print('    ', p['pass'], p['ms_per_step'], p['host_enqueue_ms_per_step'], p.get('host_cpu_ms_per_step'), p['host_call_ms_median'], p['eager_steps'], p['captures'], p['replays'], p.get('pose_rows_mean'), p.get('pose_rows_run_mean'))
PY
