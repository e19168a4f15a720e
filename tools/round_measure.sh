#!/bin/bash
# One consolidated measurement pass on the GPU box (run from the repo root): tests, smoke, bench lines of every
# config, rocprofv3 kernel stats + one steady step of the headline command, PMC traffic.  Outputs -> gpurun_out/.
set -u
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
export BBD_TEST_REPORT=$PWD/$O/gradient_error_levels.txt; rm -f $BBD_TEST_REPORT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -6 > $O/gputests.log
unset BBD_TEST_REPORT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_md2.json 2> $O/bench_md2.err
for cfg in boosted boosted15 trimin5 vit; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$cfg.json 2> $O/bench_$cfg.err
done
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_md2 -o md2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-eager-ab > /dev/null 2>&1 )
cp /tmp/prof_md2/md2_kernel_stats.csv $O/bench_md2_kernel_stats.csv
python tools/step_profile.py /tmp/prof_md2/md2_kernel_trace.csv > $O/bench_md2_one_steady_step.csv
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vit -o vit -- python3 $GRAFT_REPO_ROOT/bench.py --config vit --steps 4 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 )
cp /tmp/prof_vit/vit_kernel_stats.csv $O/bench_vit_kernel_stats.csv
python tools/step_profile.py /tmp/prof_vit/vit_kernel_trace.csv > $O/bench_vit_one_steady_step.csv
PMC_TARGET=bench timeout 600 bash tools/pmc_passes.sh /tmp/pmc_md2_step --config md2 > /dev/null 2>&1
python tools/pmc_summary.py /tmp/pmc_md2_step $O/traffic_md2.json "bench.py --config md2 (the kernels inside the training step, batch 12, 4 scales, 640x192)" > $O/pmc_summary_md2_in_step.txt
timeout 600 bash tools/pmc_passes.sh /tmp/pmc_boost7_kb --config boost7 > /dev/null 2>&1
python tools/pmc_summary.py /tmp/pmc_boost7_kb $O/traffic_boosted.json "tools/kernel_bench.py --config boost7 (m = 7, 18 candidates per pixel, batch 12, scale 0; per-pixel random disparities = scattered gathers: an upper bound for the training step, whose in-step PMC run does not finish under rocprofv3)" > $O/pmc_summary_boosted_kernel_bench.txt
cat $O/gputests.log
for f in md2 boosted boosted15 trimin5 vit; do python3 -c "
import json,sys
d=json.load(open('$O/bench_$f.json'))
print('$f', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['roofline'], {k:(v['mean_ms'],v['frac']) for k,v in d['kernels'].items()}, d.get('eager_step_images_per_sec'), d.get('cpu_baseline',{}).get('value'))
" 2>&1 | cut -c1-700; done
