#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03f
mkdir -p $O
BBD_FWD=2 timeout 600 python tools/stamps_fwd.py 2>&1 | head -14 > $O/stamps_fwdp.txt; cat $O/stamps_fwdp.txt
BBD_FWD=1 timeout 600 python tools/stamps_fwd.py 2>&1 | head -14 > $O/stamps_fwd1.txt; cat $O/stamps_fwd1.txt
