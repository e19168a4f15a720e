#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03m
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v Warning | tail -3
for rep in 1 2; do
  for cfg in md2 boost7; do
    echo -n "$cfg: " >> $O/ab.txt
    timeout 300 python tools/kernel_bench.py --config $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ident %.4f  fwd %.4f  bwd %.4f ms'%(d['identity']['ms'],d['fwd']['ms'],d['bwd']['ms']))" >> $O/ab.txt
  done
done
cat $O/ab.txt
timeout 600 python bench.py --config boosted --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_boosted.json 2> $O/err.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eager-ab > $O/bench_md2.json 2>> $O/err.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03m/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d["value"], d["ms_per_step"], {k:(v["mean_ms"], v["frac"]) for k,v in d["kernels"].items()})
    except Exception as e: print(f, "failed", e)
PY
