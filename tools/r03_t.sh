#!/bin/bash
set -u
O=gpurun_out/r03t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_vit.py -q -x 2>&1 | tail -3
for v in 1 0; do
  BBD_GEMM_DB=$v timeout 600 python bench.py --config vit --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_vit_gemmdb$v.json 2> $O/bench_vit_gemmdb$v.err
  python3 -c "
import json; d=json.load(open('$O/bench_vit_gemmdb$v.json')); print('BBD_GEMM_DB=$v images/s', d['value'], 'ms/step', d['ms_per_step'], 'median', d['ms_per_step_median'], d['config']['gemm'])" | tee -a $O/bench_vit_gemm_db_ab.txt
done
BBD_FUSED_TOKEN_GLUE=0 BBD_GEMM_DB=0 timeout 600 python bench.py --config vit --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_vit_base.json 2> $O/bench_vit_base.err
python3 -c "
import json; d=json.load(open('$O/bench_vit_base.json')); print('BBD_GEMM_DB=0 BBD_FUSED_TOKEN_GLUE=0 (round-2 state) images/s', d['value'], 'ms/step', d['ms_per_step'])" | tee -a $O/bench_vit_gemm_db_ab.txt
timeout 600 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-eager-ab > $O/bench_md2.json 2> $O/bench_md2.err
python3 -c "
import json; d=json.load(open('$O/bench_md2.json')); print('md2', d['value'], d['ms_per_step'])"
