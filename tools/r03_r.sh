#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03r; mkdir -p $O
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vit -o vit -- python3 $GRAFT_REPO_ROOT/bench.py --config vit --steps 4 --warmup 3 --no-cpu-baseline --no-eager-ab > /dev/null 2>&1 )
cp /tmp/prof_vit/vit_kernel_stats.csv $O/bench_vit_kernel_stats.csv
python tools/step_profile.py /tmp/prof_vit/vit_kernel_trace.csv > $O/bench_vit_one_steady_step.csv
python tools/step_sequence.py /tmp/prof_vit/vit_kernel_trace.csv > $O/sequence.txt
wc -l $O/sequence.txt
