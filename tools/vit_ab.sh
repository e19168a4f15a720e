#!/bin/bash
# MonoViT tests + bench line (+ one steady step under rocprofv3): the quick loop used while fusing the encoder's glue.
set -u
export TMPDIR=/tmp
O=gpurun_out/vit_ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vit.py -q -x 2>&1 | tail -3
timeout 600 python bench.py --config vit --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_vit.json 2> $O/bench_vit.err
python3 -c "
import json; d=json.load(open('$O/bench_vit.json')); print('vit', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vit -o vit -- python3 $GRAFT_REPO_ROOT/bench.py --config vit --steps 4 --warmup 3 --no-cpu-baseline --no-eager-ab > /dev/null 2>&1 )
python tools/step_profile.py /tmp/prof_vit/vit_kernel_trace.csv > $O/bench_vit_one_steady_step.csv
python tools/step_sequence.py /tmp/prof_vit/vit_kernel_trace.csv > $O/sequence.txt
tail -1 $O/bench_vit_one_steady_step.csv
