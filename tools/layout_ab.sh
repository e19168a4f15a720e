#!/bin/bash
# VERDICT r4 item 7, "one honest layout experiment": would channels-last tensors remove the NCHW<->NHWC transposes around
# MIOpen's NHWC implicit-GEMM solvers (9.7 % of the MD2 step's kernel time)?  Both layouts with the SAME glue (stock ATen
# ops: the fused NCHW glue kernels call .contiguous() and would convert every activation back) and each with its OWN
# find-mode database recorded in this run (empty at start), so that each layout runs its measured-best solvers.
#   usage: tools/layout_ab.sh            (about 20 minutes on one MI355X; prints two bench lines + kernel-time shares)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp BBD_FUSED_BN=0 BBD_FUSED_NN=0
O=gpurun_out/layout; mkdir -p $O
for layout in nchw channels_last; do
  export MIOPEN_USER_DB_PATH=/tmp/miopen_layout_$layout MIOPEN_CUSTOM_CACHE_DIR=/tmp/miopen_layout_$layout/cache
  mkdir -p $MIOPEN_CUSTOM_CACHE_DIR
  flag=""; [ $layout = channels_last ] && flag="--channels-last"
  # pass 1 records the find results, pass 2 is the measurement (immediate mode reading them)
  python bench.py $flag --miopen-benchmark --steps 3 --warmup 2 --no-cpu-baseline --no-eager-ab --no-secondary > /dev/null 2>&1
  python bench.py $flag --steps 20 --warmup 5 --no-cpu-baseline --no-eager-ab --no-secondary > $O/bench_$layout.json 2> $O/bench_$layout.err
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_layout_$layout -o $layout -- python3 $OLDPWD/bench.py $flag --steps 5 --warmup 3 --no-cpu-baseline --no-eager-ab --no-secondary > /dev/null 2>&1 )
  cp $(find /tmp/prof_layout_$layout -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$layout.csv
  python3 - "$layout" <<'PY'
import csv, json, sys
layout = sys.argv[1]
d = json.load(open('gpurun_out/layout/bench_%s.json' % layout))
rows = list(csv.DictReader(open('gpurun_out/layout/kernel_stats_%s.csv' % layout)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
share = lambda pred: 100 * sum(float(r['TotalDurationNs']) for r in rows if pred(r['Name'])) / tot
print('%-14s %7.2f images/s  %6.3f ms/step | kernel time: transposes %.1f %%, batchnorm %.1f %%, igemm/conv (MIOpen) %.1f %%' % (
    layout, d['value'], d['ms_per_step'], share(lambda n: 'transpose' in n.lower()), share(lambda n: 'batch_norm' in n.lower() or 'batchnorm' in n.lower()),
    share(lambda n: any(k in n.lower() for k in ('igemm', 'conv', 'winograd', 'gemm')))))
PY
done
