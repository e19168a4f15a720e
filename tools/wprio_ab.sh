#!/bin/bash
# A/B of hipcc's -mllvm --amdgpu-set-wave-priority (s_setprio around the kernels' vector-memory clauses) on the fused
# kernels: micro-benchmark three times each, then the training step with either library.
set -u
O=gpurun_out/wprio; mkdir -p $O
for rep in 1 2 3; do
  BBD_CONFIGS="md2 boost7" bash tools/variants.sh "base:" "wprio:-mllvm --amdgpu-set-wave-priority"
done 2>&1 | tee $O/variants.txt
for lib in base wprio; do
  for cfg in md2 boosted; do
    BBD_HIP_LIB=/tmp/bbdvar/libbbd_$lib.so timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-eager-ab > $O/bench_${cfg}_$lib.json 2> $O/bench_${cfg}_$lib.err
    python3 -c "
import json; d=json.load(open('$O/bench_${cfg}_$lib.json')); print('$lib $cfg images/s', d['value'], 'ms/step', d['ms_per_step'], {k:v['mean_ms'] for k,v in d['kernels'].items()})" | tee -a $O/in_step.txt
  done
done
