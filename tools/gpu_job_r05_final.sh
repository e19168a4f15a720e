set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -4 > $O/gputests.log; cat $O/gputests.log
( cd /tmp && timeout 900 rocprofv3 --memory-copy-trace --stats --output-format csv -d /tmp/prof_fresh -o fresh -- python3 $GRAFT_REPO_ROOT/bench.py --config boosted15_fresh --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$O/bench_boosted15_fresh_under_trace.json 2> /dev/null )
sed -n "/^python - <<'PY' > gpurun_out\/r05\/memory_copies/,/^PY$/p" tools/round5_measure.sh | sed '1d;$d' > /tmp/memsum.py; python /tmp/memsum.py > $O/memory_copies_boosted15_fresh.txt; cat $O/memory_copies_boosted15_fresh.txt
for cfg in boosted trimin5; do
  PMC_TARGET=bench timeout 1200 bash tools/pmc_passes.sh /tmp/pmc_${cfg}_step --config $cfg > /dev/null 2>&1
  python tools/pmc_summary.py /tmp/pmc_${cfg}_step $O/traffic_$cfg.json "bench.py --config $cfg --step-graph off (the kernels inside the training step)" > $O/pmc_summary_${cfg}_in_step.txt
done
cp $O/traffic_*.json profiles/r05/
timeout 1200 python bench.py > $O/bench_default_final.json 2> $O/bench_default_final.err
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r05/bench_default_final.json'))
print('md2', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: (v['mean_ms'], v['frac']) for k, v in d['kernels'].items()}, d.get('secondary_seconds'))
for s in d.get('secondary', []):
    print(s.get('config'), s.get('value'), s.get('ms_per_step'), s.get('vs_frozen_batch'), s.get('vs_frozen_batch_seen_signatures'), s.get('error'), s.get('skipped'), s.get('decode_every_use_images_per_sec'), s.get('loop_ms_per_batch'))
    for p in s.get('passes', []): print('    ', p['ms_per_step'], p['host_call_ms_median'], p['host_call_ms_slowest3'], p['eager_steps'], p['captures'], p['replays'])
PY
