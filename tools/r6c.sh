#!/bin/bash
# round 6, GPU job C: parity of the backward list variants + their LDS conflict counters (kernel_bench, PMC pass)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/r06/bwd_lists_pmc
mkdir -p $OUT
for v in wavelists both; do
  echo "== parity under variant $v"
  BBD_HIP_LIB=$PWD/build_variants/libbbd_$v.so python -m pytest tests/test_gpu_parity.py -q -m gpu -k "not native_library" 2>&1 | tail -3
done
for v in base wavelists winner3; do
  BBD_HIP_LIB=$PWD/build_variants/libbbd_$v.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
      --kernel-trace --output-format csv -d $OUT -o $v -- python3 tools/kernel_bench.py --smooth --iters 3 --warmup 1 > $OUT/$v.log 2>&1
done
python3 - $OUT <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
for path in sorted(glob.glob(out + "/**/*_counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "warp_ssim_min_bwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {n: sum(v) / len(v) for n, v in agg.items()}
    conf, act = m.get("SQ_LDS_BANK_CONFLICT", 0), m.get("SQ_LDS_IDX_ACTIVE", 1)
    print("%-40s backward: LDS conflict cycles %.3e of %.3e active = %4.1f %%; LDS insts %.3e; VALU insts %.3e; WAIT_INST_ANY/WAVE_CYCLES %.2f"
          % (path.split("/")[-1].replace("_counter_collection.csv", ""), conf, act, 100 * conf / max(act, 1), m.get("SQ_INSTS_LDS", 0), m.get("SQ_INSTS_VALU", 0),
             m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
PY
