#!/bin/bash
# quick loop for forward-kernel changes: bit-exact parity tests, micro-benchmark, the training step
set -u
O=gpurun_out/fwd_ab; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -q -x 2>&1 | tail -3
for cfg in md2 boost7; do for rep in 1 2; do python tools/kernel_bench.py --config $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg ident %.4f  fwd %.4f  bwd %.4f ms'%(d['identity']['ms'],d['fwd']['ms'],d['bwd']['ms']))"; done; done
for cfg in md2 boosted; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-eager-ab > $O/bench_$cfg.json 2> $O/bench_$cfg.err
  python3 -c "
import json; d=json.load(open('$O/bench_$cfg.json')); print('$cfg images/s', d['value'], 'ms/step', d['ms_per_step'], {k:v['mean_ms'] for k,v in d['kernels'].items()})"
done
