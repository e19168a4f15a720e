#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03g
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v Warning | tail -5
for rep in 1 2; do
for v in "BBD_BWD=2 BBD_FWD_SCALE_LOOP=0" "BBD_BWD=3 BBD_FWD_SCALE_LOOP=1"; do
  for cfg in md2 boost7 boost_e15; do
    echo -n "$v $cfg: " >> $O/ab.txt
    env $v timeout 300 python tools/kernel_bench.py --config $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ident %.4f  fwd %.4f  bwd %.4f ms'%(d['identity']['ms'],d['fwd']['ms'],d['bwd']['ms']))" >> $O/ab.txt
  done
done
done
cat $O/ab.txt
for v in "BBD_BWD=2 BBD_FWD_SCALE_LOOP=0" "BBD_BWD=3 BBD_FWD_SCALE_LOOP=1"; do
  tag=$(echo $v | tr -d ' =' )
  env $v timeout 600 python bench.py --config boosted --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_boosted_$tag.json 2> $O/bench_boosted_$tag.err
  env $v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eager-ab > $O/bench_md2_$tag.json 2> $O/bench_md2_$tag.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03g/bench_*.json")):
    try:
        d=json.load(open(f))
        print(f.split('/')[-1], d["value"], d["ms_per_step"], {k:v["mean_ms"] for k,v in d["kernels"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
