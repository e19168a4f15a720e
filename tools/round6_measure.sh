#!/bin/bash
# Round 6's consolidated measurement pass on the GPU box (run from the repo root): GPU tests, smoke, rocprofv3 kernel stats
# of the headline command and of the fresh-ordering regime (every step a graph replay), its memory-copy trace (one table
# upload per step), then the driver's ONE bench command.  The fused kernels' source is unchanged since round 5 (hash
# 2642ad43d1d578c9): their in-step PMC passes under profiles/r05/ still describe the shipped code and are not repeated.
# Outputs -> gpurun_out/r06 (copy what is to be judged into profiles/r06/).   usage: tools/round6_measure.sh [quick]
set -u
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06
mkdir -p $O
if [ "${1:-}" != "quick" ]; then
  export BBD_TEST_REPORT=$PWD/$O/gradient_error_levels.txt; rm -f $BBD_TEST_REPORT
  timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -8 > $O/gputests.log
  unset BBD_TEST_REPORT
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
fi
for cfg in md2 boosted15 boosted15_fresh; do
  extra="--steps 10 --warmup 5 --no-cpu-baseline --no-eager-ab --no-secondary"
  [ $cfg = boosted15_fresh ] && extra=""
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o $cfg -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg $extra > /dev/null 2>&1 )
  cp $(find /tmp/prof_$cfg -name "*kernel_stats.csv" | head -1) $O/bench_${cfg}_kernel_stats.csv
  [ $cfg != boosted15_fresh ] && python tools/step_profile.py $(find /tmp/prof_$cfg -name "*kernel_trace.csv" | head -1) > $O/bench_${cfg}_one_steady_step.csv
done
( cd /tmp && timeout 900 rocprofv3 --memory-copy-trace --stats --output-format csv -d /tmp/prof_fresh -o fresh -- python3 $GRAFT_REPO_ROOT/bench.py --config boosted15_fresh > $GRAFT_REPO_ROOT/$O/bench_boosted15_fresh_under_trace.json 2> /dev/null )
python - <<'PY' > gpurun_out/r06/memory_copies_boosted15_fresh.txt
import csv, glob, collections, json
files = glob.glob('/tmp/prof_fresh/**/*memory_copy_trace.csv', recursive=True)
rows = [r for f in files for r in csv.DictReader(open(f))]
print('rocprofv3 --memory-copy-trace of `bench.py --config boosted15_fresh` (pooled form): Trainer.prewarm() (one capture per pose-row')
print('bucket), then 90 + 30 + 30 steps with a new ordering each, every step a graph replay.  Columns of the trace:', list(rows[0].keys()) if rows else None)
by = collections.Counter(); dur = collections.Counter(); size = collections.Counter()
for r in rows:
    d = r.get('Direction') or r.get('direction') or '?'
    by[d] += 1
    try:
        dur[d] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    except Exception:
        pass
for k in sorted(by):
    print('%-34s %7d copies   %10.3f ms in total' % (k, by[k], dur[k] / 1e6))
try:
    line = json.load(open('gpurun_out/r06/bench_boosted15_fresh_under_trace.json'))
    print('prewarm:', line.get('prewarm'), ' graphs in use:', line.get('step_graphs_in_use'))
    for p in line['passes']:
        print({k: p.get(k) for k in ('pass', 'steps', 'ms_per_step', 'table_uploads_per_step', 'table_bytes_per_step', 'eager_steps', 'captures', 'replays',
                                     'synchronising_calls_per_step', 'synchronising_calls_at')})
except Exception as e:
    print('bench line not parsed:', e)
PY
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_final.json 2> $O/bench_default_final.err
cat $O/gputests.log 2>/dev/null
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r06/bench_default_final.json'))
print('md2', d['value'], d['ms_per_step'], d['roofline'], {k: (v['mean_ms'], v['frac']) for k, v in d['kernels'].items()}, d.get('secondary_seconds'))
print('cpu', d.get('cpu_baseline'))
print('miopen', d['config']['miopen'])
for s in d.get('secondary', []):
    print(s.get('config'), s.get('value'), s.get('ms_per_step'), s.get('vs_frozen_batch'), s.get('vs_frozen_batch_seen_signatures'), s.get('vs_frozen_batch_cold_start'), s.get('frozen_batch_pose_rows'), s.get('error'), s.get('skipped'), s.get('decode_every_use_images_per_sec'), s.get('prewarm'))
    for p in s.get('passes', []): print('    ', p['pass'], p['steps'], p['ms_per_step'], p['host_enqueue_ms_per_step'], p['host_call_ms_median'], p['host_call_ms_slowest3'], p['eager_steps'], p['captures'], p['replays'], p.get('pose_rows_mean'), p.get('pose_rows_run_mean'))
PY
