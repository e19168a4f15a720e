#!/bin/bash
set -u
O=gpurun_out/r03q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vit.py -q -x --durations=5 2>&1 | tail -12
timeout 600 python bench.py --config vit --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_vit.json 2> $O/bench_vit.err
python3 -c "
import json; d=json.load(open('$O/bench_vit.json')); print('vit', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
bash tools/r03_r.sh > $O/steady.txt 2>&1
