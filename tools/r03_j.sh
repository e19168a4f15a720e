#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03j
mkdir -p $O
BBD_CONFIGS="md2 boost7" bash tools/variants.sh "default:" "bwd_scalar_proj:-DBBD_PAIR_PROJECT_BWD=0" "fwd_pair_proj:-DBBD_PAIR_PROJECT_FWD=1" "default2:" 2>&1 | tee $O/pair_project_ab.txt
timeout 600 python bench.py --config boosted --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_boosted.json 2> $O/err.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eager-ab > $O/bench_md2.json 2>> $O/err.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03j/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d["value"], d["ms_per_step"], {k:v["mean_ms"] for k,v in d["kernels"].items()})
    except Exception as e: print(f, "failed", e)
PY
