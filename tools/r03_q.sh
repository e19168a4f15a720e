#!/bin/bash
set -u
O=gpurun_out/r03q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vit.py -q -x 2>&1 | tail -3
for v in 1 0; do
  BBD_FUSED_TOKEN_GLUE=$v timeout 600 python bench.py --config vit --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_vit_glue$v.json 2> $O/bench_vit_glue$v.err
  python3 -c "
import json; d=json.load(open('$O/bench_vit_glue$v.json')); print('glue=$v', d['value'], d['ms_per_step'], d['ms_per_step_median'])"
done
bash tools/r03_r.sh > $O/steady.txt 2>&1
