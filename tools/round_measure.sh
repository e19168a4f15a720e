#!/bin/bash
# One consolidated measurement pass on the GPU box (run from the repo root): tests, smoke, bench lines of every
# config, rocprofv3 kernel stats + one steady step for md2 / boosted / boosted15 / vit, in-step PMC traffic of every
# config (eager loop: a replayed graph's kernels cannot be attributed).  Outputs -> gpurun_out/final.
set -u
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O profiles/r04
export BBD_TEST_REPORT=$PWD/$O/gradient_error_levels.txt; rm -f $BBD_TEST_REPORT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -6 > $O/gputests.log
unset BBD_TEST_REPORT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for cfg in md2 boosted boosted15 vit; do
  extra="--steps 10 --warmup 5"; [ $cfg = vit ] && extra="--steps 4 --warmup 3"
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o $cfg -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg $extra --no-cpu-baseline --no-eager-ab --no-secondary > /dev/null 2>&1 )
  cp $(find /tmp/prof_$cfg -name "*kernel_stats.csv" | head -1) $O/bench_${cfg}_kernel_stats.csv
  python tools/step_profile.py $(find /tmp/prof_$cfg -name "*kernel_trace.csv" | head -1) > $O/bench_${cfg}_one_steady_step.csv
done
PMC_TARGET=bench timeout 900 bash tools/pmc_passes.sh /tmp/pmc_md2_step --config md2 > /dev/null 2>&1
python tools/pmc_summary.py /tmp/pmc_md2_step $O/traffic_md2.json "bench.py --config md2 --step-graph off (the kernels inside the training step, batch 12, 4 scales, 640x192)" > $O/pmc_summary_md2_in_step.txt
PMC_TARGET=bench timeout 1200 bash tools/pmc_passes.sh /tmp/pmc_boosted_step --config boosted > /dev/null 2>&1
python tools/pmc_summary.py /tmp/pmc_boosted_step $O/traffic_boosted.json "bench.py --config boosted --step-graph off (the kernels inside the training step: m = 7, 18 candidates per pixel, batch 12, scale 0, network-produced disparities and poses)" > $O/pmc_summary_boosted_in_step.txt
for cfg in boosted15 trimin5 vit; do
  PMC_TARGET=bench timeout 1200 bash tools/pmc_passes.sh /tmp/pmc_${cfg}_step --config $cfg > /dev/null 2>&1
  python tools/pmc_summary.py /tmp/pmc_${cfg}_step $O/traffic_$cfg.json "bench.py --config $cfg --step-graph off (the kernels inside the training step)" > $O/pmc_summary_${cfg}_in_step.txt
done
cp $O/traffic_*.json profiles/r04/        # (this box's copy of the tree: the bench lines below read the fresh counters)
timeout 900 python bench.py > $O/bench_md2.json 2> $O/bench_md2.err
for cfg in boosted boosted15 trimin5 vit; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$cfg.json 2> $O/bench_$cfg.err
done
for cfg in md2 boost7 boost_e15; do SMOOTH_DISP=1 timeout 300 python tools/stamps_timeline.py $cfg 2>&1 | grep -v amdgpu.ids; done > $O/workgroup_timeline.txt
SMOOTH_DISP=1 timeout 300 python tools/stamps_fwd.py 2>&1 | grep -v amdgpu.ids > $O/phase_stamps_md2.txt
python tools/hot_path_trace.py 2 > /dev/null 2>&1   # (warm)
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/hp -o hp -- python3 $GRAFT_REPO_ROOT/tools/hot_path_trace.py > /dev/null 2>&1 ); python tools/step_sequence.py $(find /tmp/hp -name "*kernel_trace.csv" | head -1) > $O/hot_path_launches.txt
cat $O/gputests.log
for f in md2 boosted boosted15 trimin5 vit; do python3 -c "
import json,sys
d=json.load(open('$O/bench_$f.json'))
print('$f', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline'], {k:(v['mean_ms'],v['frac']) for k,v in d['kernels'].items()}, d.get('eager_step_images_per_sec'), d.get('cpu_baseline',{}).get('value'))
" 2>&1 | cut -c1-700; done
grep -A3 "warp_ssim_min_bwd\|warp_ssim_min_fwd" $O/pmc_summary_boosted_in_step.txt | grep "==\|FETCH\|WRITE" | head
