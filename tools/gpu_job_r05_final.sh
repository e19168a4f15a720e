# last GPU call of round 5: find results for 48 pose-pass rows, then the GPU test tier and the driver's bench command on the
# final tree (outputs under gpurun_out/r05; the database is copied back into baseboostdepth_amd/miopen_db/ by the caller)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
bash tools/miopen_tune_pose.sh --rows 48 > $O/tune_pose_48.log 2>&1; grep rows $O/tune_pose_48.log
cp gpurun_out/miopen_db/*.txt baseboostdepth_amd/miopen_db/
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v Warning | tail -4 > $O/gputests.log; cat $O/gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default_final.json 2> $O/bench_default_final.err
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r05/bench_default_final.json'))
print('md2', d['value'], d['ms_per_step'], d['roofline']['frac'], {k: (v['mean_ms'], v['frac']) for k, v in d['kernels'].items()}, d.get('secondary_seconds'))
for s in d.get('secondary', []):
    print(s.get('config'), s.get('value'), s.get('ms_per_step'), s.get('vs_frozen_batch'), s.get('vs_frozen_batch_seen_signatures'), s.get('vs_frozen_batch_cold_start'), s.get('error'), s.get('skipped'), s.get('decode_every_use_images_per_sec'))
    for p in s.get('passes', []): print('    ', p['pass'], p['steps'], p['ms_per_step'], p['host_call_ms_median'], p['host_call_ms_slowest3'], p['eager_steps'], p['captures'], p['replays'])
PY
