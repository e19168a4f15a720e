#!/bin/bash
# HISTORICAL (round 3): patches the round-3 source of the fused kernels (`git show 6e78be1:baseboostdepth_amd/csrc/bbd_kernels.hip`);
# the patterns it replaces no longer exist in the shipped source (round 4: nine-plane backward, forward forms).
# Which LDS instruction of the fused backward owns its bank-conflict cycles?  
# The shipped source carries no ablation switches: this script patches COPIES of bbd_kernels.hip (one phase's LDS
# traffic removed or re-shaped per variant; results are numerically meaningless, only the counters matter), builds
# each into /tmp and reads SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel from a rocprofv3 PMC pass of
# tools/kernel_bench.py.     usage: tools/lds_conflict_ab.sh <out_dir>      (run from the repo root on the GPU box)
set -u
OUT=$1
export TMPDIR=/tmp
mkdir -p "$OUT" /tmp/bbdvar
CS=baseboostdepth_amd/csrc
python3 - <<'PY'
import os, re
src = open("baseboostdepth_amd/csrc/bbd_kernels.hip").read()
def sub(text, a, b, count=1):
    assert text.count(a) >= 1, a
    return text.replace(a, b) if count == 0 else text.replace(a, b, count)
variants = {
    "base": src,
    # W phase: the warped value is not stored (stride-2 dword stores into the (x, y) pair planes)
    "no_w_store": sub(src, "s[ch][XS * cl.lds[k]] = val[ch];",
                      "if (XS == 1) s[ch][XS * cl.lds[k]] = val[ch]; else asm volatile(\"\" :: \"v\"(val[ch]));"),
    # W phase: the store becomes a whole 8-byte (x, y) pair (y read back first)
    "w_store_pair": sub(src, "s[ch][XS * cl.lds[k]] = val[ch];",
                        "if (XS == 1) s[ch][XS * cl.lds[k]] = val[ch]; else { float2* p2 = reinterpret_cast<float2*>(&s[ch][XS * cl.lds[k]]); "
                        "float2 t2 = *p2; t2.x = val[ch]; *p2 = t2; }"),
    # C phase: no winner evaluates its window (9 x 8-byte reads at compacted, scattered cells + 3 coefficient stores)
    "no_c": sub(src, "const int nwin = a.no_ssim ? 0 : s_count;", "const int nwin = 0 * s_count;"),
    # G phase: no 3x3 gather of the coefficient planes
    "no_g": sub(src, "      if (!a.no_ssim) {\n        if (interior) {", "      if (a.no_ssim == 12345) {\n        if (interior) {"),
    # d warped / d (ix, iy) planes neither written nor read
    "no_dv": sub(sub(src, "if (dv != nullptr && cl.own(k)) {", "if (dv != nullptr && cl.own(k) && !BWD) {"),
                 "const float2 q = *reinterpret_cast<const float2*>(&s_dv[pl][ly * TW2 + lx0]);", "const float2 q = make_float2(gx[0][0], gx[1][1]);"),
}
for name, text in variants.items():
    d = "/tmp/bbdvar/src_" + name
    os.makedirs(d, exist_ok=True)
    open(d + "/bbd_kernels.hip", "w").write(text.replace('"../../include/bbd_hip.h"', '"bbd_hip.h"'))
PY
for d in /tmp/bbdvar/src_*; do
  name=${d##*/src_}
  lib=/tmp/bbdvar/libbbd_lds_$name.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -std=c++17 -fPIC -shared -I $PWD/include -I $PWD/$CS -o $lib \
      $d/bbd_kernels.hip $CS/bbd_eval.hip $CS/bbd_image.hip $CS/bbd_nn.hip $CS/bbd_vit.hip $CS/bbd_pose.hip $CS/bbd_tokens.hip 2>&1 | grep -E "error"
  [ -f $lib ] || { echo "build of $name failed"; continue; }
  BBD_HIP_LIB=$lib rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES \
      --kernel-trace --output-format csv -d "$OUT" -o $name -- python3 tools/kernel_bench.py --iters 3 --warmup 1 > "$OUT/$name.log" 2>&1
  echo -n "$name: "; BBD_HIP_LIB=$lib python tools/kernel_bench.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd %.4f  bwd %.4f ms'%(d['fwd']['ms'],d['bwd']['ms']))"
done
python3 - "$OUT" <<'PY'
import csv, collections, glob, sys
out = sys.argv[1]
for path in sorted(glob.glob(out + "/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "warp_ssim_min_bwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {n: sum(v) / len(v) for n, v in agg.items()}
    conf, act = m.get("SQ_LDS_BANK_CONFLICT", 0), m.get("SQ_LDS_IDX_ACTIVE", 1)
    print("%-28s backward: LDS conflict cycles %.3e of %.3e active = %4.1f %%; LDS insts %.3e; WAIT_ANY/WAVE_CYCLES %.2f"
          % (path.split("/")[-1].replace("_counter_collection.csv", ""), conf, act, 100 * conf / act, m.get("SQ_INSTS_LDS", 0),
             m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
PY
