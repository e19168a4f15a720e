#!/bin/bash
# Round 5's consolidated measurement pass on the GPU box (run from the repo root): GPU tests, smoke, rocprofv3 kernel
# stats + one steady step for md2 / boosted / boosted15 / boosted15_coherent, in-step PMC passes (traffic, LDS conflicts)
# for md2 / boosted15 / boosted15_coherent, the memory-copy trace of the fresh-ordering regime, then the bench lines.
# Outputs -> gpurun_out/r05 (copy what is to be judged into profiles/r05/).   usage: tools/round5_measure.sh [quick]
set -u
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
if [ "${1:-}" != "quick" ]; then
  export BBD_TEST_REPORT=$PWD/$O/gradient_error_levels.txt; rm -f $BBD_TEST_REPORT
  timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -6 > $O/gputests.log
  unset BBD_TEST_REPORT
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
fi
for cfg in md2 boosted boosted15 boosted15_coherent; do
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o $cfg -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 10 --warmup 5 --no-cpu-baseline --no-eager-ab --no-secondary > /dev/null 2>&1 )
  cp $(find /tmp/prof_$cfg -name "*kernel_stats.csv" | head -1) $O/bench_${cfg}_kernel_stats.csv
  python tools/step_profile.py $(find /tmp/prof_$cfg -name "*kernel_trace.csv" | head -1) > $O/bench_${cfg}_one_steady_step.csv
done
for cfg in md2 boosted15 boosted15_coherent; do
  PMC_TARGET=bench timeout 1200 bash tools/pmc_passes.sh /tmp/pmc_${cfg}_step --config $cfg > /dev/null 2>&1
  python tools/pmc_summary.py /tmp/pmc_${cfg}_step $O/traffic_$cfg.json "bench.py --config $cfg --step-graph off (the kernels inside the training step)" > $O/pmc_summary_${cfg}_in_step.txt
done
# the fresh-ordering regime: every host->device copy of 30 steps with 30 new orderings (no counters with this trace)
( cd /tmp && timeout 900 rocprofv3 --memory-copy-trace --stats --output-format csv -d /tmp/prof_fresh -o fresh -- python3 $GRAFT_REPO_ROOT/bench.py --config boosted15_fresh > $GRAFT_REPO_ROOT/$O/bench_boosted15_fresh_under_trace.json 2> /dev/null )
python - <<'PY' > gpurun_out/r05/memory_copies_boosted15_fresh.txt
import csv, glob, collections, json
files = glob.glob('/tmp/prof_fresh/**/*memory_copy_trace.csv', recursive=True)
rows = [r for f in files for r in csv.DictReader(open(f))]
print('rocprofv3 --memory-copy-trace of `bench.py --config boosted15_fresh`: 7 warm-up signatures x 2 steps, then 3 passes x 30 steps')
print('over 30 distinct orderings (pass 1: every per-signature cache cold).  Columns of the trace:', list(rows[0].keys()) if rows else None)
by = collections.Counter(); dur = collections.Counter()
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
    line = json.load(open('gpurun_out/r05/bench_boosted15_fresh_under_trace.json'))
    up = sum(p['table_uploads_per_step'] * p['steps'] for p in line['passes'])
    print('table uploads counted by steptables.STATS in the three timed passes: %d (pass 1: %.2f per step, passes 2-3: %.2f / %.2f)' % (
        up, *[p['table_uploads_per_step'] for p in line['passes']]))
    print('synchronising calls flagged by torch.cuda.set_sync_debug_mode in the timed passes:', [p['synchronising_calls_at'] for p in line['passes']],
          '(the one entry is the debug mode announcing itself)')
except Exception as e:
    print('bench line not parsed:', e)
PY
cp gpurun_out/r05/traffic_*.json profiles/r05/ 2>/dev/null     # (this box's copy of the tree: the bench lines below read the fresh counters)
timeout 1200 python bench.py > $O/bench_md2.json 2> $O/bench_md2.err
for cfg in boosted boosted15 trimin5 boosted15_coherent; do
  timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$cfg.json 2> $O/bench_$cfg.err
done
cat $O/gputests.log 2>/dev/null
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r05/bench_md2.json'))
print('md2', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: (v['mean_ms'], v['frac']) for k, v in d['kernels'].items()})
for s in d.get('secondary', []):
    print(s.get('config'), s.get('value'), s.get('ms_per_step'), s.get('vs_frozen_batch'), s.get('vs_frozen_batch_seen_signatures'), s.get('error'), s.get('skipped'))
PY
