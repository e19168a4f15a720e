#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03b
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -w -o /tmp/valu_rate tools/microbench/valu_rate.hip && timeout 300 /tmp/valu_rate > $O/valu_rate.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v Warning | tail -15 > $O/gputests.log
tail -15 $O/gputests.log
for rep in 1 2; do
for f in 2 3; do
  for cfg in md2 boost7 boost_e15; do
    echo -n "bwd form $f $cfg: " >> $O/bwd3_ab.txt
    BBD_BWD=$f timeout 300 python tools/kernel_bench.py --config $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ident %.4f  fwd %.4f  bwd %.4f ms'%(d['identity']['ms'],d['fwd']['ms'],d['bwd']['ms']))" >> $O/bwd3_ab.txt
  done
done
done
cat $O/bwd3_ab.txt
for f in 2 3; do
  BBD_BWD=$f timeout 600 python bench.py --config boosted --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_boosted_bwd$f.json 2> $O/bench_boosted_bwd$f.err
  BBD_BWD=$f timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eager-ab > $O/bench_md2_bwd$f.json 2> $O/bench_md2_bwd$f.err
done
python3 - <<'PY'
import json
for f in ("boosted_bwd2","boosted_bwd3","md2_bwd2","md2_bwd3"):
    try:
        d=json.load(open("gpurun_out/r03b/bench_%s.json"%f))
        print(f, d["value"], d["ms_per_step"], {k:v["mean_ms"] for k,v in d["kernels"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
