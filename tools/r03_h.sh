#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03h
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v Warning | tail -12
for cfg in boosted boosted15; do
  BBD_FUSED_POSE_COMPOSE=0 timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --step-graph off > $O/bench_${cfg}_eager_loops.json 2> $O/err.txt
  BBD_FUSED_POSE_COMPOSE=1 timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --step-graph off > $O/bench_${cfg}_eager_compose.json 2>> $O/err.txt
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03h/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d["value"], d["ms_per_step"], d["ms_per_step_median"])
    except Exception as e: print(f, "failed", e)
PY
