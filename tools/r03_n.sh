#!/bin/bash
set -u
O=gpurun_out/r03n; mkdir -p $O
for c in md2 boost7; do
  timeout 300 python tools/stamps_timeline.py $c > $O/timeline_$c.txt 2>&1
  SMOOTH_DISP=1 timeout 300 python tools/stamps_timeline.py $c > $O/timeline_${c}_smooth.txt 2>&1
done
tail -30 $O/timeline_*.txt
