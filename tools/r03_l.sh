#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03l
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -m gpu -q -x 2>&1 | grep -v Warning | tail -3
for v in "BBD_XCD_REMAP=0" "BBD_XCD_REMAP=1"; do
  tag=$(echo $v | tr -d ' =')
  env $v timeout 600 python bench.py --config boosted --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_boosted_$tag.json 2> $O/err.txt
  i=0
  for grp in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    env $v rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_$tag -o pass$i -- python3 bench.py --config boosted --step-graph off --steps 3 --warmup 2 --no-cpu-baseline --no-eager-ab > /tmp/pmc_$tag.log 2>&1
  done
  python tools/pmc_summary.py /tmp/pmc_$tag $O/traffic_boosted_$tag.json "bench.py --config boosted --step-graph off, $v" > $O/pmc_summary_boosted_$tag.txt
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eager-ab > $O/bench_md2.json 2>> $O/err.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03l/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d["value"], d["ms_per_step"], {k:v["mean_ms"] for k,v in d["kernels"].items()})
    except Exception as e: print(f, "failed", e)
for f in sorted(glob.glob("gpurun_out/r03l/traffic_*.json")):
    d=json.load(open(f)); print(f.split('/')[-1], {k:round(v["traffic_bytes"]/1e6,1) for k,v in d.items() if isinstance(v,dict)})
PY
