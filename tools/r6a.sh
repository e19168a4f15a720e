set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_pooled.py -q -m gpu -k "frame_pool" 2>&1 | tail -5
python bench.py --config boosted15 --no-secondary --no-cpu-baseline --no-eager-ab --steps 20 --warmup 5 > gpurun_out/r06/boosted15_frozen.json 2> gpurun_out/r06/boosted15_frozen.err
python bench.py --config boosted15_fresh > gpurun_out/r06/boosted15_fresh.json 2> gpurun_out/r06/boosted15_fresh.err
python bench.py --config trimin5 --no-secondary --no-cpu-baseline --no-eager-ab --steps 20 --warmup 5 > gpurun_out/r06/trimin5_frozen.json 2> gpurun_out/r06/trimin5_frozen.err
python bench.py --config trimin5_fresh > gpurun_out/r06/trimin5_fresh.json 2> gpurun_out/r06/trimin5_fresh.err
tail -3 gpurun_out/r06/*.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e); continue
    print(f, j.get("value"), j.get("ms_per_step"))
    for p in j.get("passes", []):
        print("   ", {k: p[k] for k in ("pass","steps","ms_per_step","images_per_sec","host_enqueue_ms_per_step","host_call_ms_median","host_call_ms_slowest3","eager_steps","captures","replays","table_uploads_per_step")})
    if "prewarm" in j: print("   prewarm", j["prewarm"], "graphs", j.get("step_graphs_in_use"), "fallbacks", j.get("pooled_fallbacks"))
PY
