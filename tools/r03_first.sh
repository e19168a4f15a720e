#!/bin/bash
# Round-3 first measurement pass: VALU issue rates, GPU tests on the cleaned source, XCD-remap A/B, boosted in-step profile.
set -u
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -w -o /tmp/valu_rate tools/microbench/valu_rate.hip && timeout 300 /tmp/valu_rate > $O/valu_rate.txt 2>&1
tail -3 $O/valu_rate.txt
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -v Warning | tail -8 > $O/gputests.log
tail -3 $O/gputests.log
for rep in 1 2; do
for r in 0 1; do
  for cfg in md2 boost7; do
    echo -n "remap=$r $cfg: " >> $O/xcd_remap_ab.txt
    BBD_XCD_REMAP=$r timeout 300 python tools/kernel_bench.py --config $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ident %.4f  fwd %.4f  bwd %.4f ms'%(d['identity']['ms'],d['fwd']['ms'],d['bwd']['ms']))" >> $O/xcd_remap_ab.txt
  done
done
done
cat $O/xcd_remap_ab.txt
for r in 0 1; do
  BBD_XCD_REMAP=$r timeout 600 python bench.py --config boosted --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_boosted_remap$r.json 2> $O/bench_boosted_remap$r.err
  BBD_XCD_REMAP=$r timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eager-ab > $O/bench_md2_remap$r.json 2> $O/bench_md2_remap$r.err
done
python3 - <<'PY'
import json
for f in ("boosted_remap0","boosted_remap1","md2_remap0","md2_remap1"):
    try:
        d=json.load(open("gpurun_out/r03a/bench_%s.json"%f))
        print(f, d["value"], d["ms_per_step"], {k:v["mean_ms"] for k,v in d["kernels"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_boost -o boost -- python3 $GRAFT_REPO_ROOT/bench.py --config boosted --steps 5 --warmup 3 --no-cpu-baseline --no-eager-ab > /tmp/prof_boost.log 2>&1 )
cp /tmp/prof_boost/boost_kernel_stats.csv $O/bench_boosted_kernel_stats.csv 2>/dev/null
python tools/step_profile.py /tmp/prof_boost/boost_kernel_trace.csv > $O/bench_boosted_one_steady_step.csv 2>/dev/null
head -12 $O/bench_boosted_kernel_stats.csv
# in-step PMC of the boosted step: eager loop (no graph), FETCH_SIZE pass only, bounded
( cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_boost -o fetch -- python3 bench.py --config boosted --step-graph off --steps 3 --warmup 2 --no-cpu-baseline --no-eager-ab > /tmp/pmc_boost_fetch.log 2>&1 ; echo "pmc boosted fetch rc=$?" )
ls -la /tmp/pmc_boost 2>/dev/null | head
tail -3 /tmp/pmc_boost_fetch.log
