#!/bin/bash
# round 6, GPU job A: the fresh-ordering lines against their frozen batches (stand-alone forms of the secondary lines)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for cfg in boosted15 trimin5; do
  python bench.py --config $cfg --no-secondary --no-cpu-baseline --no-eager-ab --steps 20 --warmup 5 > gpurun_out/r06/${cfg}_frozen.json 2> gpurun_out/r06/${cfg}_frozen.err
  python bench.py --config ${cfg}_fresh > gpurun_out/r06/${cfg}_fresh.json 2> gpurun_out/r06/${cfg}_fresh.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06/*_f*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-1500:]); continue
    print(f, j.get("value"), j.get("ms_per_step"))
    for p in j.get("passes", []):
        print("   ", {k: p.get(k) for k in ("pass","steps","ms_per_step","images_per_sec","host_enqueue_ms_per_step","host_call_ms_median","host_call_ms_slowest3","eager_steps","captures","replays","pose_rows_mean","pose_rows_run_mean")})
    if "prewarm" in j: print("   prewarm", j["prewarm"], "graphs", j.get("step_graphs_in_use"), "fallbacks", j.get("pooled_fallbacks"))
PY
