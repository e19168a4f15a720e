#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03d
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/lds_rate tools/microbench/lds_rate.hip && timeout 300 /tmp/lds_rate > $O/lds_rate.txt 2>&1
cat $O/lds_rate.txt | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -m gpu -q -x 2>&1 | grep -v Warning | tail -4
BBD_CONFIGS="md2 boost7 boost_e15" bash tools/variants.sh "v2:" "v2serial:-DBBD_BWD3_SERIAL_SCATTER" "v2w2:-DBBD_BWD3_WAVES=2" 2>&1 | tee $O/bwd3_v2_variants.txt
echo "bwd2 reference:" | tee -a $O/bwd3_v2_variants.txt
for cfg in md2 boost7 boost_e15; do BBD_BWD=2 python tools/kernel_bench.py --config $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg ident %.4f  fwd %.4f  bwd %.4f ms'%(d['identity']['ms'],d['fwd']['ms'],d['bwd']['ms']))" | tee -a $O/bwd3_v2_variants.txt; done
timeout 600 python tools/stamps_bwd3.py md2 > $O/stamps_bwd3_md2.txt 2>&1; tail -12 $O/stamps_bwd3_md2.txt
timeout 600 python tools/stamps_bwd3.py boost7 > $O/stamps_bwd3_boost7.txt 2>&1; tail -18 $O/stamps_bwd3_boost7.txt
