#!/bin/bash
# A/B of the LDS window reads: the same kernels built with whole 16-byte window reads
# (default) and with the loads left to hipcc (-DBBD_WINDOW_PLAIN: narrowed and re-paired as ds_read2_b32 /
# ds_read_b64), under rocprofv3 PMC passes of SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (+ wave cycles).
# usage: tools/lds_conflict_ab.sh <out_dir>        (run from the repo root on the GPU box)
set -u
OUT=$1
export TMPDIR=/tmp
mkdir -p "$OUT" /tmp/bbdvar
SRC="baseboostdepth_amd/csrc/bbd_kernels.hip baseboostdepth_amd/csrc/bbd_eval.hip baseboostdepth_amd/csrc/bbd_image.hip baseboostdepth_amd/csrc/bbd_nn.hip baseboostdepth_amd/csrc/bbd_vit.hip baseboostdepth_amd/csrc/bbd_pose.hip"
for spec in "whole:" "plain:-DBBD_WINDOW_PLAIN" ${BBD_EXTRA_SPECS:-}; do
  name="${spec%%:*}"; flags="${spec#*:}"
  lib=/tmp/bbdvar/libbbd_lds_$name.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math ${BBD_BASE_FLAGS--fno-slp-vectorize} -std=c++17 -fPIC -shared $flags -o $lib $SRC 2>&1 | grep -E "error"
  BBD_HIP_LIB=$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS \
      --kernel-trace --output-format csv -d "$OUT" -o $name -- python3 tools/kernel_bench.py --iters 3 --warmup 1 > "$OUT/$name.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
for path in sorted(glob.glob(out + "/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        for short in ("warp_ssim_min_fwd", "warp_ssim_min_bwd", "identity_loss"):
            if short in k:
                agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", path.split("/")[-1])
    for k, c in agg.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        conf, act = m.get("SQ_LDS_BANK_CONFLICT", 0), m.get("SQ_LDS_IDX_ACTIVE", 1)
        print("  %-18s LDS conflict cycles %.3e of %.3e active = %.1f %%; WAVE_CYCLES %.3e WAIT_ANY %.3e (%.0f %%) VALU %.3e LDS insts %.3e"
              % (k, conf, act, 100 * conf / act, m.get("SQ_WAVE_CYCLES", 0), m.get("SQ_WAIT_ANY", 0),
                 100 * m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_LDS", 0)))
PY
