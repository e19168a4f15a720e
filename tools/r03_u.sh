#!/bin/bash
set -u
O=gpurun_out/r03u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vit.py -q -x 2>&1 | tail -3
bash tools/gemm_tune.sh > $O/tune.log 2>&1; tail -2 $O/tune.log
cp baseboostdepth_amd/gemm_db/tunableop_gfx950.csv $O/tunableop_gfx950.csv
for v in 1 0; do
  BBD_GEMM_DB=$v timeout 600 python bench.py --config vit --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_vit_gemmdb$v.json 2> $O/bench_vit_gemmdb$v.err
  python3 -c "
import json; d=json.load(open('$O/bench_vit_gemmdb$v.json')); print('BBD_GEMM_DB=$v images/s', d['value'], 'ms/step', d['ms_per_step'], 'median', d['ms_per_step_median'])"
done
