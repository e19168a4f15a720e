#!/bin/bash
# round 6, GPU job B: pooled tests incl. the data-parallel one; backward list variants (timing + parity); early-curriculum buckets
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_pooled.py -q -m gpu -x 2>&1 | tail -8
REPS="1 2" ITERS=200 bash tools/runvariants.sh "md2 boost7 boost_e15" base wavelists winner3 both 2>&1 | tee gpurun_out/r06/bwd_list_variants_raw.txt
for v in wavelists both; do
  echo "== parity under variant $v"
  BBD_HIP_LIB=$PWD/build_variants/libbbd_$v.so python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
done
python bench.py --config trimin5 --no-secondary --no-cpu-baseline --no-eager-ab --steps 20 --warmup 5 > gpurun_out/r06/trimin5_frozen.json 2> gpurun_out/r06/trimin5_frozen.err
python bench.py --config trimin5_fresh > gpurun_out/r06/trimin5_fresh.json 2> gpurun_out/r06/trimin5_fresh.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06/trimin5*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-1500:]); continue
    print(f, j.get("value"), j.get("ms_per_step"))
    for p in j.get("passes", []):
        print("   ", {k: p[k] for k in ("pass","steps","ms_per_step","images_per_sec","host_enqueue_ms_per_step","host_call_ms_median","host_call_ms_slowest3","eager_steps","captures","replays")})
    if "prewarm" in j: print("   prewarm", j["prewarm"], "graphs", j.get("step_graphs_in_use"), "fallbacks", j.get("pooled_fallbacks"))
PY
