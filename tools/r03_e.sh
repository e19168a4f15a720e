#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r03e
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trainer.py -m gpu -q -x 2>&1 | grep -v Warning | tail -6
for rep in 1 2; do
for f in 1 2; do
  for cfg in md2 boost7 boost_e15; do
    echo -n "fwd form $f $cfg: " >> $O/fwdp_ab.txt
    BBD_FWD=$f timeout 300 python tools/kernel_bench.py --config $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ident %.4f  fwd %.4f  bwd %.4f ms'%(d['identity']['ms'],d['fwd']['ms'],d['bwd']['ms']))" >> $O/fwdp_ab.txt
  done
done
done
cat $O/fwdp_ab.txt
BBD_CONFIGS="md2 boost7" bash tools/variants.sh "b1:-DBBD_FWDP_BATCH=1" "b3:-DBBD_FWDP_BATCH=3" "w2:-DBBD_FWDP_WAVES=2" "w2b3:-DBBD_FWDP_WAVES=2 -DBBD_FWDP_BATCH=3" 2>&1 | tee $O/fwdp_variants.txt
