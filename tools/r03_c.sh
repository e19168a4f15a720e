#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03c
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -w -o /tmp/valu_rate tools/microbench/valu_rate.hip && timeout 300 /tmp/valu_rate > $O/valu_rate.txt 2>&1
grep "waves/SIMD 4" $O/valu_rate.txt | cut -c1-150
timeout 600 python tools/stamps_bwd3.py md2 > $O/stamps_bwd3_md2.txt 2>&1; cat $O/stamps_bwd3_md2.txt | tail -20
timeout 600 python tools/stamps_bwd3.py boost7 > $O/stamps_bwd3_boost7.txt 2>&1; cat $O/stamps_bwd3_boost7.txt | tail -20
SMOOTH_DISP=1 timeout 600 python tools/stamps_bwd3.py boost7 > $O/stamps_bwd3_boost7_smooth.txt 2>&1; cat $O/stamps_bwd3_boost7_smooth.txt | tail -20
