#!/bin/bash
set -u
O=gpurun_out/r03s; mkdir -p $O
export PYTORCH_TUNABLEOP_FILENAME=$PWD/$O/tunableop_gfx950.csv
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=40 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 \
  timeout 1500 python bench.py --config vit --step-graph off --steps 2 --warmup 2 --no-cpu-baseline > $O/tune.json 2> $O/tune.err
ls -la $O; wc -l $O/tunableop_gfx950*.csv; head -8 $O/tunableop_gfx950*.csv
for en in 0 1; do
  PYTORCH_TUNABLEOP_ENABLED=$en PYTORCH_TUNABLEOP_TUNING=0 timeout 600 python bench.py --config vit --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_vit_tunable$en.json 2> $O/bench_vit_tunable$en.err
  python3 -c "
import json; d=json.load(open('$O/bench_vit_tunable$en.json')); print('tunable=$en', d['value'], d['ms_per_step'], d['ms_per_step_median'], d['step_graph'])"
done
