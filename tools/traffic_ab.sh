#!/bin/bash
# In-step HBM traffic of the fused kernels for a work-order mode: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes of
# bench.py's eager loop (tools/pmc_passes.sh does the full set).   usage: tools/traffic_ab.sh <config> <mode> <outdir>
set -u
cfg=$1; mode=$2; out=$3
export TMPDIR=/tmp BBD_EXPERIMENT=1 BBD_XCD_REMAP=$mode
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -o ${cfg}_m${mode}_$c -- \
      python3 bench.py --config $cfg --step-graph off --steps 3 --warmup 2 --no-cpu-baseline --no-eager-ab --no-secondary > $out/${cfg}_m${mode}_$c.log 2>&1
done
python3 - $out $cfg $mode <<'PY'
import csv, glob, sys, collections
out, cfg, mode = sys.argv[1:4]
v = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("%s/**/%s_m%s_*counter_collection.csv" % (out, cfg, mode), recursive=True):
    for r in csv.DictReader(open(f)):
        for key in ("warp_ssim_min_fwd", "warp_ssim_min_bwd", "identity_loss"):
            if key in r["Kernel_Name"]:
                v[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(v.items()):
    f = 2.0 * sum(cs["FETCH_SIZE"]) / max(1, len(cs["FETCH_SIZE"])) * 1024
    w = sum(cs["WRITE_SIZE"]) / max(1, len(cs["WRITE_SIZE"])) * 1024
    print("%s mode %s %-20s fetch %.1f MB  write %.1f MB  traffic %.1f MB" % (cfg, mode, k, f / 1e6, w / 1e6, (f + w) / 1e6))
PY
